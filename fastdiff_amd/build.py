"""Build libfastdiff_hip.so (gfx950) in-tree:  python -m fastdiff_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the resulting .so is git-ignored but travels to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
SOURCES = ["fd_api.cpp", "fd_api_ext.cpp", "fd_api_train.cpp", "fd_kernels_naive.hip", "fd_generic.hip", "fd_kernels_first_final.hip", "fd_kernels_dblock.hip", "fd_kernels_kp.hip", "fd_kernels_convt.hip", "fd_kernels_lvc.hip", "fd_kernels_mel.hip", "fd_kernels_train.hip", "fd_kernels_kconv.hip", "fd_kernels_cconv.hip"]
LIB = os.path.join(LIBDIR, "libfastdiff_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-honor-nans: lets fmaxf() be one v_max_f32 (no canonicalising v_max(v,v) first); fp32 VALU work is not hidden under
# fp32 MFMA on gfx950, so every VALU instruction in the inner loops counts.  No kernel tests for NaN.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-fno-honor-nans", "-x", "hip", "-Wall",
         "-Wno-unused-function"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(not os.path.exists(d) or os.path.getmtime(d) > t for d in deps)


def _deps(obj, src):
    """Everything `obj` was compiled from, as the compiler itself listed it (-MMD writes <obj>.d next to the object): the source and
    every project header it included, directly or not.  No .d file yet = stale."""
    d = os.path.splitext(obj)[0] + ".d"
    if not os.path.exists(d):
        return [src, d]
    txt = open(d).read().replace("\\\n", " ")
    return [src] + [p for p in txt.split(":", 1)[-1].split() if os.sep + "opt" + os.sep not in p and not p.startswith("/usr/")]


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    jobs = []
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, _deps(obj, sp)):
            jobs.append([HIPCC] + FLAGS + (["-Rpass-analysis=kernel-resource-usage"] if verbose else []) + ["-MMD", "-c", sp, "-o", obj])
    if jobs and not os.path.exists(HIPCC):
        raise RuntimeError(f"{HIPCC} not found and {LIB} is stale: cannot build the HIP extension")

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0 or verbose:
            sys.stderr.write(r.stdout + r.stderr)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
