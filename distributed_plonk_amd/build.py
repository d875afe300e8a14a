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
UNITS = ["plonk_api.hip", "ntt_engine.hip", "msm_engine.hip", "synth.hip", "quotient.hip", "poly_ops.hip", "comm_rccl.hip"]
HEADERS = ["fp.hpp", "fp29.hpp", "flimb.hpp", "ec.hpp", "ec_lazy.hpp", "constants.h", "ntt_kernels.hpp", "plonk_internal.hpp",
           "../../include/plonk_hip.h"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-Wno-pass-failed"]
# The NTT pass kernel unrolls nests of fully unrolled 9x9-limb products; with the pinned accumulation chains of
# fp29.hpp (one empty asm per mad) the default `#pragma unroll` budget is exceeded, the butterfly loops stay loops and the per-lane
# element arrays land in SCRATCH memory (160 B per lane, measured 1.6-2.8x slower in round 1).  With the budget raised the pinned
# kernel needs 111 VGPRs, no scratch, no spills: 18.4 -> 16.3 ms per 8n coset FFT (profiles/r02_ntt_pins_experiment.txt).
UNROLL = ["-mllvm", "-pragma-unroll-threshold=131072", "-mllvm", "-unroll-threshold=131072"]
# The column accumulators of fp29.hpp / flimb.hpp must stay ONE dependent chain of v_mad_u64_u32; what splits them is the SLP vectoriser's
# horizontal-reduction matching.  Every unit keeps the round 1-3 remedy — the "+v" accumulator pin (fp29.hpp) — although it costs one
# `s_nop 0` per mad.  Round 4 measured the alternatives (profiles/r04_pin_nop_experiment.txt): NOHOR + -DPLONK_PIN_NONE (no pins, the
# matching switched off instead) makes the MSM kernels 5-25 % faster on one class of gpurun boxes and the bucket accumulation 46 % SLOWER on
# another (16.3 -> 23.8 ms at 2^24 points; in the two-context step 276 -> 295 ms), -DPLONK_PIN_USE costs the accumulate kernel its SROA, and
# the NTT pass kernel is indifferent.  The forms stay selectable per unit (build_variant) for whoever measures on a known machine.
NOHOR = ["-mllvm", "-slp-vectorize-hor=false"]
UNIT_FLAGS = {"ntt_engine.hip": UNROLL}
# Not for poly_ops.hip (measured worse: perm product 5.1 -> 6.8 ms, division 1.5 -> 2.9 ms).  Not for quotient.hip either: standalone the
# kernel gains 5 % (55.1 -> 52.1 ms) and the fused variants stop using scratch, but inside bench.py (and in a process started right
# after it) the same binary ran at 112 ms twice out of twice — an unexplained slow mode, so the default budget stays there.


def source_hash() -> str:
    """sha256 over the kernel sources (csrc/* and the C header): PMC-derived numbers committed under profiles/ carry it, and
    bench.py refuses to quote them for a library built from different sources."""
    import hashlib
    h = hashlib.sha256()
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".hpp")))
    for f in names:
        h.update(f.encode())
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(HERE, "..", "include", "plonk_hip.h"), "rb") as fh:
        h.update(fh.read())
    h.update(repr((FLAGS, sorted(UNIT_FLAGS.items()))).encode())       # the compile flags shape the kernels too
    return h.hexdigest()[:16]


def _newest_header():
    # this file counts as a dependency of every object: the compile flags live here
    return max([os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS] + [os.path.getmtime(os.path.abspath(__file__))])


def _compile(unit, force, objdir=OBJDIR, unit_flags=None):
    src = os.path.join(CSRC, unit)
    obj = os.path.join(objdir, unit.replace(".hip", ".o"))
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), _newest_header()):
        return obj, False
    subprocess.check_call([HIPCC, *FLAGS, *(UNIT_FLAGS if unit_flags is None else unit_flags).get(unit, []), "-c", src, "-o", obj])
    return obj, True


def build_variant(name, unit_flags, force=False):
    """An A/B build: the same sources with other per-unit flags (e.g. {"quotient.hip": UNROLL, **UNIT_FLAGS}) as
    lib/variants/<name>/libplonk_hip.so — selected at run time with PLONK_HIP_LIB=<that path> (distributed_plonk_amd/_ffi.py).
    Units whose flags equal the product build's reuse its objects."""
    objdir = os.path.join(OBJDIR, "variants", name)
    outdir = os.path.join(LIBDIR, "variants", name)
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(outdir, exist_ok=True)
    build(verbose=False)
    objs = []
    for u in UNITS:
        if unit_flags.get(u, []) == UNIT_FLAGS.get(u, []):
            objs.append(os.path.join(OBJDIR, u.replace(".hip", ".o")))
        else:
            objs.append(_compile(u, force, objdir, unit_flags)[0])
    out = os.path.join(outdir, "libplonk_hip.so")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", out])
    return out


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        res = list(ex.map(lambda u: _compile(u, force), UNITS))
    objs = [r[0] for r in res]
    if force or any(r[1] for r in res) or not os.path.exists(OUT):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", OUT])      # librccl is dlopen'ed (comm_rccl.hip)
        if verbose:
            print("built", OUT)
    code_hashes(refresh=True)
    return OUT


CODE_HASHES = os.path.join(LIBDIR, "kernel_code_hashes.json")


def code_hashes(refresh=False) -> dict:
    """{kernel base name: hash of its gfx950 machine code, all instantiations} of the objects this library was linked from
    (codehash.py); written beside the .so at build time so that it travels with it.  {} when neither the file nor the objects exist."""
    import json
    from . import codehash
    try:
        objs = [os.path.join(OBJDIR, f) for f in os.listdir(OBJDIR) if f.endswith(".o")] if os.path.isdir(OBJDIR) else []
        stale = not os.path.exists(CODE_HASHES) or any(os.path.getmtime(o) > os.path.getmtime(CODE_HASHES) for o in objs)
        if objs and (refresh or stale):
            return codehash.write_hashes(OBJDIR, CODE_HASHES)
        with open(CODE_HASHES) as f:       # on the GPU box the objects stay behind (.gpurunignore): the file written at build time speaks for them
            return json.load(f)
    except Exception:       # noqa: BLE001 - a diagnostic: bench.py then falls back to the all-sources hash
        return {}


if __name__ == "__main__":
    build(force="--force" in sys.argv)
