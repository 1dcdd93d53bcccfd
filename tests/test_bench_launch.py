"""`python bench.py --gpus N` with no WORLD_SIZE in the environment must start N ranks itself (the driver's
command ran ONE rank and printed n_gpus = 1).  --dry-run exercises exactly that path -- re-exec under torch.distributed.run,
rendezvous on 127.0.0.1, barrier-bracketed timed region, max over ranks, one JSON line from rank 0 -- without a GPU."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), capture_output=True, text=True, env=env,
                       cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-1500:])          # ONE line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_reports_them():
    d = _run("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["dry_run"] is True and d["steps"] == 3
    assert d["scaling"] == "strong"


def test_gpus_1_runs_in_process():
    d = _run("--gpus", "1", "--dry-run", "--steps", "2", "--scaling", "weak")
    # one rank scales nothing: the label says so; the mode the flags select is reported next to it
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["scaling"] == "none" and d["scaling_when_sharded"] == "weak"
