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


# ---- the host-only C++ half of the drop-in package (autograd nodes + workspace state over the C ABI; no kernels) ----------------------
DROPIN_DIR = os.path.join(os.path.dirname(HERE), "diff_gaussian_rasterization")
DROPIN_SRC = os.path.join(DROPIN_DIR, "csrc", "dgr_native.cpp")
DROPIN_LIB = os.path.join(DROPIN_DIR, "_dgr.so")


def dropin_needs_build():
    if not os.path.exists(DROPIN_LIB):
        return True
    t = os.path.getmtime(DROPIN_LIB)
    return any(os.path.getmtime(p) > t for p in (DROPIN_SRC, os.path.join(os.path.dirname(HERE), "include", "splat_hip.h")))


def build_dropin_ext(force=False, verbose=True):
    """g++ -> diff_gaussian_rasterization/_dgr.so: a pybind11 / libtorch extension that links libsplat_hip.so (rpath relative to the
    package).  Host code only, so plain g++ with torch's include / library paths (what torch.utils.cpp_extension would pass)."""
    if not force and not dropin_needs_build():
        return DROPIN_LIB
    build_native(verbose=verbose)
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cxx = os.environ.get("CXX", "g++")
    # (torch ships the pybind11 headers it was built with under its own include path: no separate pybind11 package needed)
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], os.path.join(rocm, "include")]
    tlib = ce.library_paths()[0]
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-sign-compare",
           "-DTORCH_EXTENSION_NAME=_dgr", "-DTORCH_API_INCLUDE_EXTENSION_H", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ["-I" + p for p in inc]
    cmd += [DROPIN_SRC, "-o", DROPIN_LIB, "-L" + tlib, "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10", "-lc10_hip", "-ltorch_hip",
            "-L" + LIBDIR, "-lsplat_hip", "-Wl,-rpath," + tlib, "-Wl,-rpath,$ORIGIN/../splat_slam_amd/lib"]
    if verbose:
        print("[build]", " ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return DROPIN_LIB


if __name__ == "__main__":
    print(build_native(force="--force" in sys.argv))
    print(build_dropin_ext(force="--force" in sys.argv))
