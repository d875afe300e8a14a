"""Build recipe for the CPU oracle (TEST INFRASTRUCTURE ONLY).

    python oracle/build.py        ->  oracle/libplonk_oracle.so

The Rust reference cannot be compiled here (no cargo/rustc/capnp, git deps, nightly features —
SURVEY.md fact 5), so there is no oracle/_ref/ build: DESIGN.md records it as unbuildable.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = ["plonk_oracle.c", "field_impl.h", "curve_impl.h"]
OUT = os.path.join(HERE, "libplonk_oracle.so")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(HERE, s)) > t for s in SRC)


def build(force: bool = False) -> str:
    if force or needs_build():
        cmd = ["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", "-Wall",
               "-Wno-unused-function", os.path.join(HERE, "plonk_oracle.c"), "-o", OUT]
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
