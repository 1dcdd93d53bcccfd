"""CPU: the hand-over kit for a CUDA box cannot rot.  scripts/export_parity_scenes.py writes the parity scenes as .npz (upstream's
argument order, oracle outputs and gradients); scripts/compare_with_upstream_cuda.py replays them through `diff_gaussian_rasterization`
on an NVIDIA machine.  Here the replay script runs UNCHANGED against a stand-in module of that name backed by the oracle (and
`.cuda()` made a no-op): every file must load, every key the script reads must exist, and -- the stand-in being the oracle -- every
relative error it prints must be ~0.  The rasterizer itself stays parity-unpinned against the real CUDA build: this test
only keeps the tool that would pin it in working order."""
import contextlib
import io
import os
import re
import runpy
import subprocess
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _standin_module():
    from oracle import raster_oracle as O
    from diff_gaussian_rasterization import GaussianRasterizationSettings       # (the NamedTuple: no GPU needed)

    class GaussianRasterizer:
        def __init__(self, raster_settings):
            self.rs = raster_settings

        def __call__(self, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, theta, rho):
            rs, d = self.rs, torch.float64
            s = O.OracleSettings(rs.image_height, rs.image_width, rs.tanfovx, rs.tanfovy, rs.bg.to(d), rs.scale_modifier, rs.viewmatrix.to(d),
                                 rs.projmatrix.to(d), rs.projmatrix_raw.to(d), rs.sh_degree, rs.campos.to(d), False, False)
            return O.rasterize(means3D.to(d), means2D.to(d), opacities.to(d), shs=shs.to(d), scales=scales.to(d), rotations=rotations.to(d),
                               theta=theta.to(d), rho=rho.to(d), settings=s)
    m = types.ModuleType("diff_gaussian_rasterization")
    m.GaussianRasterizationSettings, m.GaussianRasterizer = GaussianRasterizationSettings, GaussianRasterizer
    return m


def test_parity_kit_exports_and_replays(tmp_path, monkeypatch):
    out = str(tmp_path / "kit")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "export_parity_scenes.py"), out, "--only=tiny,two,narrow"],
                       capture_output=True, text=True, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    assert r.returncode == 0, r.stderr[-2000:]
    files = sorted(os.path.join(out, f) for f in os.listdir(out) if f.endswith(".npz"))
    assert [os.path.basename(f) for f in files] == ["narrow.npz", "tiny.npz", "two.npz"]
    # the replay script, unchanged, with `diff_gaussian_rasterization` = the oracle and the CUDA placement calls neutralised
    monkeypatch.setitem(sys.modules, "diff_gaussian_rasterization", _standin_module())
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    real_zeros = torch.zeros
    monkeypatch.setattr(torch, "zeros", lambda *a, **k: real_zeros(*a, **{kk: v for kk, v in k.items() if kk != "device"}))
    monkeypatch.setattr(sys, "argv", ["compare_with_upstream_cuda.py"] + files)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        runpy.run_path(os.path.join(ROOT, "scripts", "compare_with_upstream_cuda.py"), run_name="__main__")
    text = buf.getvalue()
    assert text.count("== ") == 3, text
    rows = [l for l in text.splitlines() if l.strip() and not l.startswith("==")]
    names = {l.split()[0] for l in rows}
    assert {"color", "radii", "depth", "opacity", "n_touched", "grad_means3D", "grad_means2D", "grad_opacities", "grad_shs", "grad_scales",
            "grad_rotations", "grad_theta", "grad_rho"} <= names, names
    errs = [float(x) for x in re.findall(r"oracle_(\d\.\d+e[+-]\d+)", text)]
    assert len(errs) >= 13 * 3
    assert max(errs) < 1e-6, max(errs)             # the stand-in IS the oracle (the files keep the pixel weights in fp32: ~6e-8)
    upj = [float(x) for x in re.findall(r"oracle_upj_(\d\.\d+e[+-]\d+)", text)]
    assert len(upj) == 2 * 3                        # the upstream-pose-Jacobian column is there for theta and rho
