"""Build the HOST EMULATION of libplonk_hip.so (test infrastructure; see hip/hip_runtime.h in this directory).

    python -m tests.hostemu.build [--asan | --ubsan] [--force]

The kernel sources are taken as they are from distributed_plonk_amd/csrc; the one construct a header cannot emulate —
`extern __shared__ T name[];`, the dynamic LDS window — is rewritten on a COPY under _build/src/ into a pointer to the emulated
workgroup's LDS buffer.  comm_rccl.hip (dlopen of librccl, GPU collectives) is replaced by comm_local.cpp, a communicator of world
size 1.  Output: tests/hostemu/_build/<variant>/libplonk_hostemu.so (git-ignored).
"""
import concurrent.futures as cf
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "distributed_plonk_amd", "csrc")
UNITS = ["plonk_api.hip", "ntt_engine.hip", "msm_engine.hip", "synth.hip", "quotient.hip", "poly_ops.hip"]
EXTERN_SHARED = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][A-Za-z0-9_ ]*?)\s+([A-Za-z_][A-Za-z0-9_]*)\[\];")


def _stage_sources(src_dir):
    """csrc/* -> src_dir with the dynamic-LDS declarations rewritten; returns the newest source mtime."""
    os.makedirs(src_dir, exist_ok=True)
    newest = 0.0
    for name in sorted(os.listdir(CSRC)):
        if not name.endswith((".hip", ".hpp", ".h")):
            continue
        path = os.path.join(CSRC, name)
        newest = max(newest, os.path.getmtime(path))
        with open(path) as fh:
            text = fh.read()
        text = EXTERN_SHARED.sub(lambda m: f"{m.group(1)}* {m.group(2)} = reinterpret_cast<{m.group(1)}*>(hipemu::dyn_smem());", text)
        assert "extern __shared__" not in text, f"{name}: a dynamic-LDS declaration the rewrite does not understand"
        out = os.path.join(src_dir, name.replace(".hip", ".cpp"))
        if not os.path.exists(out) or open(out).read() != text:
            with open(out, "w") as fh:
                fh.write(text)
    for extra in ("hipemu_runtime.cpp", "comm_local.cpp", os.path.join("hip", "hip_runtime.h"), "build.py"):
        newest = max(newest, os.path.getmtime(os.path.join(HERE, extra)))
    return newest


def build(asan=False, force=False, verbose=True, ubsan=False):
    variant = "ubsan" if ubsan else ("asan" if asan else "plain")
    bdir = os.path.join(HERE, "_build", variant)
    src_dir = os.path.join(HERE, "_build", "src")
    os.makedirs(bdir, exist_ok=True)
    newest = _stage_sources(src_dir)
    out = os.path.join(bdir, "libplonk_hostemu.so")
    if not force and os.path.exists(out) and os.path.getmtime(out) > newest:
        return out
    flags = ["-std=c++17", "-fPIC", "-pthread", "-I", HERE, "-I", src_dir, "-I", os.path.join(ROOT, "include"), "-include", "hip/hip_runtime.h",
             "-Wno-unknown-pragmas", "-Wno-attributes", "-fno-strict-aliasing"]
    san = ["-fsanitize=undefined", "-fno-sanitize-recover=undefined"] if ubsan else (["-fsanitize=address"] if asan else [])
    flags += ["-O1", "-g", "-fno-omit-frame-pointer", *san] if san else ["-O2"]
    jobs = [(os.path.join(src_dir, u.replace(".hip", ".cpp")), os.path.join(bdir, u.replace(".hip", ".o"))) for u in UNITS]
    jobs += [(os.path.join(HERE, f), os.path.join(bdir, f.replace(".cpp", ".o"))) for f in ("hipemu_runtime.cpp", "comm_local.cpp")]

    def compile_one(job):
        src, obj = job
        subprocess.check_call(["g++", *flags, "-c", src, "-o", obj])
        return obj

    with cf.ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        objs = list(ex.map(compile_one, jobs))
    subprocess.check_call(["g++", "-shared", "-pthread", *san, *objs, "-ldl", "-lrt", "-o", out])
    if verbose:
        print("built", out)
    return out


if __name__ == "__main__":
    build(asan="--asan" in sys.argv, ubsan="--ubsan" in sys.argv, force="--force" in sys.argv)
