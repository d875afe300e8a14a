"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol that
include/plonk_hip.h declares (and nothing is bound by ctypes that the header does not declare), fails
loudly without a GPU, and the host-side mirror of the reference's partitioning / layout helpers is right.
No compute is invoked here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from distributed_plonk_amd import _ffi
from distributed_plonk_amd.dispatcher import (decimate_rows, make_fft_workloads, make_msm_workloads, split_rc,
                                               undecimate_cols)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "plonk_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(plonk_[a-z0-9_]+)\s*\(", text)) - {"plonk_exchange_fn"})


def test_library_exports_every_header_symbol():
    lib = _ffi.lib()
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/plonk_hip.h but not exported by libplonk_hip.so"
    assert sorted(_ffi.SIGNATURES) == syms, "ctypes binding and header disagree"


def _c_arity(text, name):
    m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", text, flags=re.S)
    args = m.group(1).strip()
    return 0 if args in ("", "void") else args.count(",") + 1


def test_rust_binding_declares_every_header_symbol_with_the_same_arity():
    """ffi/plonk_hip.rs is source, not compiled here (no Rust toolchain): keep it in lock-step with the header mechanically —
    every function of include/plonk_hip.h is declared in the `extern "C"` block with the same number of parameters, and the
    ctypes binding has that arity too."""
    htext = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "plonk_hip.h")).read(), flags=re.S)
    rtext = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "ffi", "plonk_hip.rs")).read())
    for s in _header_symbols():
        m = re.search(r"pub fn " + s + r"\s*\(([^;]*?)\)\s*(->\s*[^;]+)?;", rtext, flags=re.S)
        assert m, f"{s} is not declared in ffi/plonk_hip.rs"
        rargs = m.group(1).strip()
        r_arity = 0 if not rargs else len([a for a in rargs.split(",") if a.strip()])
        assert r_arity == _c_arity(htext, s), (s, r_arity, _c_arity(htext, s))
        assert len(_ffi.SIGNATURES[s][1]) == r_arity, s
    assert "PLONK_COMM_ID_BYTES: usize = 128" in rtext and "#[repr(C)]" in rtext


def test_header_cites_reference_interfaces():
    text = open(os.path.join(ROOT, "include", "plonk_hip.h")).read()
    for cite in ["worker.rs:126-157", "worker.rs:159-185", "worker.rs:187-233", "worker.rs:235-278", "worker.rs:280-345",
                 "worker.rs:347-381", "worker.rs:383-408", "hello_world.capnp", "utils.rs:27-43", "transpose.rs:413"]:
        assert cite in text, cite


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _ffi.lib()
    ctx = C.c_void_p()
    rc = lib.plonk_create(C.byref(ctx), 0, 0)
    assert rc == -3 and b"hipGetDeviceCount" in lib.plonk_last_error()
    from distributed_plonk_amd.worker import PlonkWorker
    with pytest.raises(_ffi.PlonkError):
        PlonkWorker(0, 0, "bn254")


def test_host_only_entry_points_reject_nulls():
    lib = _ffi.lib()
    assert lib.plonk_g1_add(0, None, None, None) == -1
    assert lib.plonk_sync(None) == -1
    assert lib.plonk_set_option(None, b"msm_window", 0) == -1


def test_product_package_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under distributed_plonk_amd/ may import or load it."""
    pkg = os.path.join(ROOT, "distributed_plonk_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "import oracle" not in src and "from oracle" not in src and "plonk_oracle" not in src, fn


# ---- reference partition / layout helpers
def test_split_rc_matches_reference():
    # worker.rs:143-144: r = 1 << (log >> 1), c = N / r ; odd log N => c = 2r (2^11, 2^13, 2^27)
    assert split_rc(1 << 11) == (32, 64)
    assert split_rc(1 << 13) == (64, 128)
    assert split_rc(1 << 20) == (1024, 1024)
    assert split_rc(1 << 27) == (1 << 13, 1 << 14)
    with pytest.raises(AssertionError):
        split_rc(1000)


def test_workloads_match_reference_formulas():
    wl = make_fft_workloads(1 << 13, 4)        # dispatcher2.rs:272-291
    assert [(w.row_start, w.row_end, w.col_start, w.col_end) for w in wl] == [(0, 16, 0, 32), (16, 32, 32, 64), (32, 48, 64, 96), (48, 64, 96, 128)]
    assert wl[1].num_rows() == 16 and wl[1].num_cols() == 32
    ms = make_msm_workloads(1 << 20, 8)        # dispatcher.rs:223-226
    assert (ms[0].start, ms[0].end, ms[7].start, ms[7].end) == (0, 1 << 17, 7 << 17, 1 << 20)
    assert C.sizeof(_ffi.FftWorkload) == 32 and C.sizeof(_ffi.MsmWorkload) == 16     # capnp structs: 4 / 2 x u64


def test_decimate_undecimate_are_the_reference_transposes():
    N, (r, c) = 1 << 7, split_rc(1 << 7)
    v = np.arange(N * 4, dtype=np.uint64).reshape(N, 4)
    t = decimate_rows(v, r)                    # dispatcher2.rs:754: t[b][a] = coeffs[a*r + b]
    assert t.shape == (r, c, 4)
    for b in (0, 3, r - 1):
        for a in (0, 5, c - 1):
            assert np.array_equal(t[b, a], v[a * r + b])
    u = np.arange(c * r * 4, dtype=np.uint64).reshape(c, r, 4)
    out = undecimate_cols(u)                   # dispatcher2.rs:786: out[j*c + i] = u[i][j]
    for i in (0, 2, c - 1):
        for j in (0, 1, r - 1):
            assert np.array_equal(out[j * c + i], u[i, j])


def test_bench_and_entry_contract_files_exist():
    for f in ["bench.py", "__graft_entry__.py", "DESIGN.md", "INTEGRATION.md", "include/plonk_hip.h", "oracle/plonk_oracle.c", "ffi/plonk_hip.rs",
              "tools/preflight_multi.sh"]:
        assert os.path.exists(os.path.join(ROOT, f)), f


def test_kernel_machine_code_hashes():
    """distributed_plonk_amd/codehash.py: the per-kernel hashes of the gfx950 machine code that let bench.py keep quoting PMC-derived
    numbers (profiles/pmc_current.json) for a kernel whose code did not change while other sources did.  Needs the objects of the
    in-tree build (`__graft_entry__.build()`); no GPU."""
    import hashlib
    import json
    import re
    from distributed_plonk_amd import build, codehash
    objdir = build.OBJDIR
    if not (os.path.isdir(objdir) and os.path.exists(os.path.join(objdir, "ntt_engine.o"))):
        pytest.skip("the HIP library has not been built in this tree")
    hashes = codehash.kernel_code_hashes(objdir)
    # every kernel the sources declare is a FUNC symbol of some code object
    declared = set()
    for f in os.listdir(build.CSRC):
        if f.endswith((".hip", ".hpp")):
            txt = open(os.path.join(build.CSRC, f)).read()
            declared |= set(re.findall(r"__global__\s+void\s+(?:__launch_bounds__\([^)]*\)\s+)?(\w+)\s*\(", txt))
    declared -= {"ntt_steps0123_xlane", "__attribute__"}      # (a kernel declared with an attribute list before its name: its name is still found at another site or is checked by the GPU tests)
    missing = sorted(k for k in declared if k not in hashes)
    assert len(declared) > 40 and not missing, missing
    # the hash is over the code bytes: one flipped byte in one instantiation changes that kernel's hash and no other
    co = bytearray(codehash.device_code_object(os.path.join(objdir, "ntt_engine.o")))
    syms = codehash.kernel_symbols(bytes(co))
    name = next(k for k in sorted(syms) if codehash.base_name(k) == "ntt_pass_kernel")
    at, _size = codehash.kernel_symbol_ranges(bytes(co))[name]
    co[at + 8] ^= 1
    syms2 = codehash.kernel_symbols(bytes(co))
    changed = [k for k in syms if hashlib.sha256(syms[k]).digest() != hashlib.sha256(syms2[k]).digest()]
    assert changed == [name]
    # what build() wrote beside the library is what the objects say, and the committed PMC file names kernels that exist
    if os.path.exists(build.CODE_HASHES) and os.path.getmtime(build.CODE_HASHES) >= os.path.getmtime(os.path.join(objdir, "ntt_engine.o")):
        written = json.load(open(build.CODE_HASHES))
        assert {k_: v_ for k_, v_ in written.items() if ":" not in k_} == hashes
        # ... plus, since round 4, the HOST code hash of every unit and the unit each kernel is launched from (a kernel's counters depend on its
        # launch shape too: benchlib/pmc.py)
        assert {k_: v_ for k_, v_ in written.items() if ":" in k_} == codehash.host_code_hashes(objdir)
        assert written["unit_of:ntt_pass_kernel"] == "ntt_engine" and written["unit_of:msm_accumulate_kernel"] == "msm_engine" and "host:plonk_api" in written
    db = json.load(open(os.path.join(ROOT, "profiles", "pmc_current.json")))
    assert {k_ for k_ in db.get("code_hashes", {}) if ":" not in k_} <= set(hashes)


@pytest.mark.parametrize("curve", [0, 1])
def test_host_side_point_functions_match_the_oracle(curve):
    """plonk_g1_add / plonk_g1_to_affine (the dispatcher's reduce and `Commitment(commitment.into())`, dispatcher.rs:236-238, dispatcher2.rs:892) are
    host code of the REAL library — since round 6 on the 64-bit-limb host statement of the Montgomery product (csrc/fp.hpp, hipcc host pass only) that
    also runs the MSM's host fold.  No GPU needed: random sums, P + P, P - P, the identity on either side, against the oracle bit for bit."""
    import ctypes as C
    import numpy as np
    from distributed_plonk_amd import _ffi
    from oracle import oracle as O
    lib = _ffi.lib()
    Q = O.FQ_LIMBS[curve]
    one = O.field_const(curve, 1, 1)[:Q]
    bases = O.gen_bases(curve, 99, 24, 24)
    jac = lambda xy: np.concatenate([xy, one]).astype(np.uint64)
    zero = np.concatenate([one, one, np.zeros(Q, dtype=np.uint64)])
    p = lambda a: a.ctypes.data_as(C.c_void_p)

    def add(a, b):
        out = np.zeros(3 * Q, dtype=np.uint64)
        assert lib.plonk_g1_add(curve, p(np.ascontiguousarray(a)), p(np.ascontiguousarray(b)), p(out)) == 0
        return out

    def affine(j):
        xy, inf = np.zeros(2 * Q, dtype=np.uint64), C.c_int(0)
        assert lib.plonk_g1_to_affine(curve, p(np.ascontiguousarray(j)), p(xy), C.byref(inf)) == 0
        return xy, bool(inf.value)

    def same(j, want_jac):
        (a, ai), (b, bi) = affine(j), O.jac_to_affine(curve, want_jac)
        return ai == bi and np.array_equal(a, b)

    acc_l, acc_o = zero.copy(), zero.copy()
    for i in range(24):
        acc_l, acc_o = add(acc_l, jac(bases[i])), O.jac_add(curve, acc_o, jac(bases[i]))
        assert same(acc_l, acc_o), i
    dbl = add(jac(bases[3]), jac(bases[3]))
    assert same(dbl, O.jac_add(curve, jac(bases[3]), jac(bases[3])))
    neg = bases[5].copy()
    qmod = int.from_bytes(O.field_const(curve, 1, 0)[:Q].tobytes(), "little")
    y = int.from_bytes(neg[Q:].tobytes(), "little")
    neg[Q:] = np.frombuffer(((qmod - y) % qmod).to_bytes(8 * Q, "little"), dtype=np.uint64)
    assert affine(add(jac(bases[5]), jac(neg)))[1] is True                       # P - P
    assert same(add(zero, jac(bases[7])), jac(bases[7])) and same(add(jac(bases[7]), zero), jac(bases[7]))
    assert same(add(acc_l, acc_l), O.jac_add(curve, acc_o, acc_o))               # doubling of a non-normalised point
