"""Builds libsplat_hip.so (gfx950) in-tree with hipcc.  No torch involved: the library is a plain C ABI."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsplat_hip.so")
ARCH = "gfx950"


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdr = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdr.append(os.path.join(os.path.dirname(HERE), "include", "splat_hip.h"))
    return _sources() + hdr


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in _deps())


def build_native(force=False, verbose=True):
    """hipcc --offload-arch=gfx950 -> splat_slam_amd/lib/libsplat_hip.so (object files cached next to it)."""
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    hdr_t = max(os.path.getmtime(p) for p in _deps() if p.endswith(".h"))
    procs = []
    for src in _sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and os.path.getmtime(obj) > hdr_t):
            continue
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + \
            os.environ.get("SGR_CXXFLAGS", "").split() + ["-c", src, "-o", obj]       # (SGR_CXXFLAGS: build-time experiments, e.g. -DSGR_TILE_WAVES=4)
        if verbose:
            print("[build]", " ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print("[build]", " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv))
