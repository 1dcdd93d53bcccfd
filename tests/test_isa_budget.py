"""The tile kernels are VALU-issue bound: their time IS their instruction count, and a good part of round 4 was
removing instructions the compiler had added on its own (a canonicalising v_max behind fminf, copies around DPP operands, compares
issued twice, 64-bit address arithmetic).  This test compiles csrc/sgr_blend.hip to gfx950 assembly (no GPU needed) and holds the hot
loops of the fused tile kernel to their measured budgets, so that a source change or a compiler update that undoes that work fails
HERE and not as a slower bench line."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FUSED = "_ZN3sgr16blend_fwd_kernelILi512ELb1EEE"


@pytest.fixture(scope="module")
def fused_asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("isa") / "blend.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-o", out,
                    os.path.join(ROOT, "splat_slam_amd", "csrc", "sgr_blend.hip")], check=True, capture_output=True)
    text = open(out).read()
    lines = text.split("\n")
    a = next(i for i, l in enumerate(lines) if l.startswith(FUSED) and l.rstrip().endswith(":") or l.startswith(FUSED) and ":" in l[:len(FUSED) + 80])
    b = next(i for i in range(a, len(lines)) if "s_endpgm" in lines[i])
    meta = text[text.index(".name:           " + FUSED):]
    meta = meta[:meta.index("\n  - ") if "\n  - " in meta else len(meta)]
    return lines[a:b + 1], text, meta


def _valu(block):
    return sum(1 for l in block if re.match(r"\s+v_", l))


def _loops(lines):
    """Every innermost loop of the kernel as a dict of counts: all blocks the assembler tags with the loop's header label."""
    out = []
    for h, l in enumerate(lines):
        if "Inner Loop Header" not in l:
            continue
        label = None
        for k in range(h, max(h - 3, 0), -1):
            m = re.match(r"(\.LBB\d+_\d+):", lines[k])
            if m:
                label = m.group(1)
                break
        if label is None:
            continue
        tag = re.compile("Header=" + re.escape(label.replace(".L", "")) + r"(?!\d)")
        body, cur = [], False
        for x in lines:
            m = re.match(r"(\.LBB\d+_\d+):", x)
            if m or x.startswith("; %bb."):
                cur = bool(tag.search(x)) or (m is not None and m.group(1) == label)
            if cur:
                body.append(x)
        cnt = lambda pat: sum(1 for x in body if pat in x)
        out.append(dict(valu=_valu(body), rcp=cnt("v_rcp_f32"), exp=cnt("v_exp_f32"), bcast31=cnt("row_bcast:31"), bcast15=cnt("row_bcast:15"),
                        shr1=cnt("row_shr:1 "), shr2=cnt("row_shr:2 "), ds_write=cnt("ds_write"), atomic=cnt("global_atomic")))
    return out


def test_fused_tile_kernel_keeps_its_registers_and_has_no_scratch(fused_asm):
    _, _, meta = fused_asm
    vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1))
    spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1))
    scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1))
    assert vgpr <= 96, f"{vgpr} VGPRs: fewer than 5 waves per SIMD"
    assert spill == 0 and scratch == 0, (spill, scratch)


def test_backward_loops_stay_within_their_instruction_budget(fused_asm):
    """Per (pixel pair x group) iteration.  Round 6 runs the loop row by row: a 64-lane chunk's loop body holds the FOUR pairs of a row
    (eight v_rcp), a 32-lane chunk's two, the narrower ones one."""
    lines, _, _ = fused_asm
    loops = [l for l in _loops(lines) if l["rcp"]]
    per_pair = lambda l: l["valu"] / (l["rcp"] / 2)
    wide = [per_pair(l) for l in loops if l["bcast31"]]                              # 64 lanes (two sources)
    half = [per_pair(l) for l in loops if l["bcast15"] and not l["bcast31"]]         # 32 lanes
    narrow = [per_pair(l) for l in loops if not l["bcast15"] and l["exp"]]           # 16 / 8 / 4 lanes, footprint re-evaluated
    stash = [per_pair(l) for l in loops if not l["exp"]]                             # 16 / 8 / 4 lanes reading the G stash
    assert len(wide) >= 2 and len(half) >= 2 and len(narrow) >= 6 and len(stash) >= 3, loops
    assert all(l["rcp"] == 8 for l in loops if l["bcast31"]) and all(l["rcp"] == 4 for l in loops if l["bcast15"] and not l["bcast31"]), loops
    assert max(wide) <= 70 and min(wide) <= 67.5, wide         # (round 5: 81-83 at 64 lanes; round 6: 67 -- 269 per row of four pairs, the dy sums folded per half row)
    assert max(half) <= 73, half                               # (71)
    assert max(narrow) <= 67 and min(narrow) <= 59, narrow     # (16 lanes: 65 (round 5: 75), 8: 61, 4: 57-58)
    assert max(stash) <= 59 and min(stash) <= 52, stash        # (16 lanes: 57 (round 5: 62), 8: 53, 4: 50)


def test_forward_walk_stays_within_its_instruction_budget(fused_asm):
    lines, _, _ = fused_asm
    # the walks: innermost loops with two v_exp_f32 and no v_rcp_f32 (two splats per trip)
    walks = [l for l in _loops(lines) if l["exp"] == 2 and not l["rcp"]]
    plain = [l["valu"] for l in walks if not l["ds_write"] and not l["atomic"]]
    stashing = [l["valu"] for l in walks if l["ds_write"] and not l["atomic"]]
    assert plain and stashing, walks
    # (counts include the five instructions of the rarely taken "a pixel's walk ended in this trip" block)
    assert min(plain) <= 34, walks                             # (two splats per trip: 28 + 5; round 5: 35; 45 in round 3)
    assert min(stashing) <= 37, walks                          # (+3: the masked exp2(-npow) of both splats goes to the stash)


def test_no_v_readlane_in_the_rank_sort(fused_asm):
    """A one-chunk tile ranks its keys through LDS broadcasts: no loop of the kernel may read a lane per iteration through SGPRs (a
    v_readlane costs the VALU-saturated SIMD more than its issue slot: -2.5 % of the kernel when they went)."""
    lines, _, _ = fused_asm
    for h, l in enumerate(lines):
        if "Inner Loop Header" in l:
            body = lines[h:h + 12]
            assert not any("v_readlane_b32" in x for x in body), body
