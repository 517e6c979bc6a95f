"""Build the C-ABI kernel library (and, for the CPU test-suite only, its SIMT-emulated twin).

    libfsv2v_hip.so   hipcc --offload-arch=gfx950, every csrc/*.hip          <- the product
    libfsv2v_emu.so   clang++ -DFSV_EMU against tests/emu/hip_emu.h          <- `not gpu` logic tests only

Both are written next to this file (git-ignored, but they travel to the GPU box with the snapshot).
"""
import glob
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
HIP_LIB = os.path.join(HERE, "libfsv2v_hip.so")
EMU_LIB = os.path.join(HERE, "libfsv2v_emu.so")
OBJ_DIR = os.path.join(HERE, "build")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CLANGXX = os.environ.get("FSV_HOST_CXX", "/opt/rocm/lib/llvm/bin/clang++")
# -ffp-contract=off: the warp kernel replays ATen's un-normalise sequence op by op,
# and the parity story for every other kernel is simpler when a*b+c never silently becomes an fma.
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I", CSRC, "-I", os.path.join(ROOT, "include")]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _digest(paths, extra=""):
    h = hashlib.sha1(extra.encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")))


def source_digest():
    """digest of every kernel source and header: names the build a profile / counter file under profiles/ describes"""
    return _digest(_sources() + _headers(), "src")[:16]


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    return r


def build_hip(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = _sources()
    hdr = _headers()
    objs, jobs = [], []
    for s in srcs:
        tag = _digest([s] + hdr, "hip")[:16]
        o = os.path.join(OBJ_DIR, os.path.basename(s) + "." + tag + ".o")
        objs.append(o)
        if force or not os.path.exists(o):
            jobs.append([HIPCC, "--offload-arch=gfx950"] + COMMON + ["-c", s, "-o", o])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for r in ex.map(_run, jobs):
                if verbose and r.stderr:
                    print(r.stderr)
    # one "current" record per library (not one stamp per digest: after switching sources back and forth an old stamp
    # would claim that the library on disk still matches)
    if force or jobs or not os.path.exists(HIP_LIB) or _current("hip") != _digest(objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs)
        _set_current("hip", _digest(objs))
    # objects of older source revisions are dead weight (the directory travels with every gpurun snapshot)
    keep = set(objs)
    for old in glob.glob(os.path.join(OBJ_DIR, "*.hip.*.o")):
        if old not in keep:
            try:
                os.remove(old)
            except OSError:
                pass
    return HIP_LIB


def build_hip_diag():
    """tools/_diag/libfsv2v_hip_diag.so: the library with -DFSV_DIAG (knock-out forms of the dominant kernel behind force_tile ids
    30+, tools/knockout.py).  Never loaded by the product path: lib.get_lib() takes it only through FSV2V_LIB."""
    out_dir = os.path.join(ROOT, "tools", "_diag")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    jobs = []
    for s in _sources():
        o = os.path.join(out_dir, os.path.basename(s) + ".o")
        objs.append(o)
        jobs.append([HIPCC, "--offload-arch=gfx950"] + COMMON + ["-DFSV_DIAG", "-c", s, "-o", o])
    with ThreadPoolExecutor(max_workers=8) as ex:
        list(ex.map(_run, jobs))
    out = os.path.join(out_dir, "libfsv2v_hip_diag.so")
    _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    for o in objs:
        os.remove(o)
    return out


def _current(which):
    try:
        with open(os.path.join(OBJ_DIR, which + ".current")) as f:
            return f.read().strip()
    except OSError:
        return None


def _set_current(which, digest):
    with open(os.path.join(OBJ_DIR, which + ".current"), "w") as f:
        f.write(digest)


def build_emu(force=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = _sources()
    emu_h = os.path.join(ROOT, "tests", "emu", "hip_emu.h")
    tag = _digest(srcs + _headers() + [emu_h], "emu")
    if not force and os.path.exists(EMU_LIB) and _current("emu") == tag:
        return EMU_LIB
    unity = os.path.join(OBJ_DIR, "fsv_emu_unity.cpp")
    with open(unity, "w") as f:
        for s in srcs:
            f.write('#include "%s"\n' % s)
    _run([CLANGXX, "-x", "c++", "-DFSV_EMU", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mfma",
          "-Wno-unused-value", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "emu"),
          "-shared", "-o", EMU_LIB, unity])
    _set_current("emu", tag)
    return EMU_LIB


if __name__ == "__main__":
    which = sys.argv[1:] or ["hip"]
    if "hip" in which:
        print(build_hip(verbose=True))
    if "emu" in which:
        print(build_emu())
