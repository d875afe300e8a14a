"""Build libplonk_hip.so (the C-ABI HIP library) in-tree for gfx950.

    python -m distributed_plonk_amd.build [--force]

hipcc cross-compiles without a GPU.  One object per translation unit, compiled in parallel, linked
into distributed_plonk_amd/lib/libplonk_hip.so (git-ignored, travels to the GPU box with gpurun).
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
OUT = os.path.join(LIBDIR, "libplonk_hip.so")
UNITS = ["plonk_api.hip", "ntt_engine.hip", "msm_engine.hip", "synth.hip", "quotient.hip", "poly_ops.hip"]
HEADERS = ["fp.cuh", "fp29.cuh", "flimb.cuh", "ec.cuh", "ec_lazy.cuh", "constants.h", "ntt_kernels.cuh", "plonk_internal.hpp",
           "../../include/plonk_hip.h"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-Wno-pass-failed"]


def _newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)


def _compile(unit, force):
    src = os.path.join(CSRC, unit)
    obj = os.path.join(OBJDIR, unit.replace(".hip", ".o"))
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), _newest_header()):
        return obj, False
    subprocess.check_call([HIPCC, *FLAGS, "-c", src, "-o", obj])
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        res = list(ex.map(lambda u: _compile(u, force), UNITS))
    objs = [r[0] for r in res]
    if force or any(r[1] for r in res) or not os.path.exists(OUT):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", OUT])
        if verbose:
            print("built", OUT)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
