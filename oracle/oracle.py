"""ctypes front-end of the C CPU oracle (TEST INFRASTRUCTURE ONLY — see oracle/plonk_oracle.c).

PARITY UNPINNED (no reference golden vectors exist; see plonk_oracle.c header) — pinned instead to third-party sympy transforms and group
law (tests/test_oracle_thirdparty.py, tests/golden/sympy_*.json) beside the repository's own big-integer statement.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
All arrays are numpy uint64, little-endian limbs, layouts of /root/reference/src/utils.rs:27-43:
Fr (n,4) Montgomery; scalars (n,4) canonical; affine bases (n, 2*Q) x||y Montgomery + inf flags;
Jacobian (3*Q,) X||Y||Z Montgomery, with Q = 4 (BN254) or 6 (BLS12-381).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

BN254, BLS12_381 = 0, 1
CURVE_IDS = {"bn254": BN254, "bls12_381": BLS12_381}
FQ_LIMBS = {BN254: 4, BLS12_381: 6}

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = _build.OUT
        if not os.path.exists(path):
            _build.build()
        _lib = C.CDLL(path)
        _lib.orc_field_inv64.restype = C.c_uint64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def max_threads() -> int:
    return lib().orc_max_threads()


def ntt(curve, v, is_inv=False, is_coset=False, threads=1):
    v = _u64(v).copy()
    n = v.shape[0]
    log_n = n.bit_length() - 1
    assert 1 << log_n == n and v.shape[1] == 4
    rc = lib().orc_ntt(curve, _p(v), log_n, int(is_inv), int(is_coset), threads)
    if rc:
        raise ValueError("DomainCreationError")
    return v


def naive_dft(curve, v, is_inv=False):
    v = _u64(v)
    out = np.empty_like(v)
    log_n = v.shape[0].bit_length() - 1
    assert lib().orc_naive_dft(curve, _p(v), _p(out), log_n, int(is_inv)) == 0
    return out


def fourstep(curve, v, is_inv=False, is_coset=False):
    v = _u64(v).copy()
    log_n = v.shape[0].bit_length() - 1
    assert lib().orc_fourstep(curve, _p(v), log_n, int(is_inv), int(is_coset)) == 0
    return v


def distributed_fft(curve, v, S, is_inv=False, is_coset=False):
    v = _u64(v).copy()
    log_n = v.shape[0].bit_length() - 1
    rc = lib().orc_distributed_fft(curve, _p(v), log_n, S, int(is_inv), int(is_coset))
    assert rc == 0, rc
    return v


def fft1_helper(curve, row, i, log_n, is_inv, is_coset):
    row = _u64(row).copy()
    assert lib().orc_fft1_helper(curve, _p(row), C.c_uint64(i), log_n, int(is_inv), int(is_coset)) == 0
    return row


def fft2_helper(curve, col, i, log_n, is_inv, is_coset):
    col = _u64(col).copy()
    assert lib().orc_fft2_helper(curve, _p(col), C.c_uint64(i), log_n, int(is_inv), int(is_coset)) == 0
    return col


def exchange_pack(rows, col_start, col_end):
    rows = _u64(rows)
    n_rows, c = rows.shape[0], rows.shape[1]
    out = np.empty((n_rows * (col_end - col_start), 4), dtype=np.uint64)
    lib().orc_exchange_pack(_p(rows), C.c_size_t(n_rows), C.c_size_t(c), C.c_size_t(col_start), C.c_size_t(col_end), _p(out))
    return out


def exchange_scatter(cols, from_row_start, v):
    """cols: (nc, r, 4) modified in place."""
    v = _u64(v)
    nc, r = cols.shape[0], cols.shape[1]
    lib().orc_exchange_scatter(_p(cols), C.c_size_t(nc), C.c_size_t(r), C.c_size_t(from_row_start), _p(v), C.c_size_t(v.shape[0]))


OPS = {"mul": 0, "add": 1, "sub": 2, "to_mont": 3, "from_mont": 4, "inv": 5, "sqr": 6}


def field_op(curve, field, op, a, b=None):
    a = _u64(a)
    b = _u64(b) if b is not None else a
    out = np.empty_like(a)
    assert lib().orc_field_op(curve, field, OPS[op], _p(a), _p(b), _p(out), C.c_size_t(a.shape[0])) == 0
    return out


def field_const(curve, field, which, arg=0):
    out = np.zeros(6, dtype=np.uint64)
    n = lib().orc_field_const(curve, field, which, arg, _p(out))
    assert n > 0
    return out[:n].copy()


def field_inv64(curve, field):
    return int(lib().orc_field_inv64(curve, field))


def rand_fr(curve, seed, n):
    out = np.empty((n, 4), dtype=np.uint64)
    lib().orc_rand_fr(curve, C.c_uint64(seed), C.c_size_t(n), _p(out))
    return out


def from_mont(curve, a):
    return field_op(curve, 0, "from_mont", a)


def gen_bases(curve, seed, unique, n):
    out = np.empty((n, 2 * FQ_LIMBS[curve]), dtype=np.uint64)
    lib().orc_gen_bases(curve, C.c_uint64(seed), C.c_size_t(unique), C.c_size_t(n), _p(out))
    return out


def generator(curve):
    out = np.empty(2 * FQ_LIMBS[curve], dtype=np.uint64)
    lib().orc_generator(curve, _p(out))
    return out


def _inf(inf):
    return np.ascontiguousarray(inf, dtype=np.uint8) if inf is not None else None


def msm(curve, bases, scalars, inf=None, threads=1):
    bases, scalars, inf = _u64(bases), _u64(scalars), _inf(inf)
    n = min(bases.shape[0], scalars.shape[0])
    out = np.empty(3 * FQ_LIMBS[curve], dtype=np.uint64)
    lib().orc_msm(curve, _p(bases), _p(inf), _p(scalars), C.c_size_t(n), _p(out), threads)
    return out


def msm_naive(curve, bases, scalars, inf=None):
    bases, scalars, inf = _u64(bases), _u64(scalars), _inf(inf)
    n = min(bases.shape[0], scalars.shape[0])
    out = np.empty(3 * FQ_LIMBS[curve], dtype=np.uint64)
    lib().orc_msm_naive(curve, _p(bases), _p(inf), _p(scalars), C.c_size_t(n), _p(out))
    return out


def sharded_msm(curve, bases, scalars, S, inf=None, threads=1):
    bases, scalars, inf = _u64(bases), _u64(scalars), _inf(inf)
    out = np.empty(3 * FQ_LIMBS[curve], dtype=np.uint64)
    lib().orc_sharded_msm(curve, _p(bases), _p(inf), _p(scalars), C.c_size_t(scalars.shape[0]), S, _p(out), threads)
    return out


def jac_add(curve, a, b):
    out = np.empty(3 * FQ_LIMBS[curve], dtype=np.uint64)
    lib().orc_jac_add(curve, _p(_u64(a)), _p(_u64(b)), _p(out))
    return out


def jac_to_affine(curve, jac):
    """-> (xy Montgomery limbs, is_infinity)."""
    out = np.zeros(2 * FQ_LIMBS[curve], dtype=np.uint64)
    inf = lib().orc_jac_to_affine(curve, _p(_u64(jac)), _p(out))
    return out, bool(inf)


def on_curve(curve, xy):
    return bool(lib().orc_on_curve(curve, _p(_u64(xy))))


def scalar_mul(curve, xy, k):
    out = np.empty(3 * FQ_LIMBS[curve], dtype=np.uint64)
    lib().orc_scalar_mul(curve, _p(_u64(xy)), _p(_u64(k)), _p(out))
    return out


def commit_polynomial(curve, bases, coeffs_mont, inf=None, threads=1):
    bases, coeffs, inf = _u64(bases), _u64(coeffs_mont), _inf(inf)
    out = np.empty(3 * FQ_LIMBS[curve], dtype=np.uint64)
    lib().orc_commit_polynomial(curve, _p(bases), _p(inf), C.c_size_t(bases.shape[0]), _p(coeffs),
                                C.c_size_t(coeffs.shape[0]), _p(out), threads)
    return out


def round1(curve, bases, evals, blinders, inf=None, threads=1):
    """worker.rs:383-408 with explicit blinders -> (poly coeffs (n+2,4), commitment Jacobian)."""
    bases, evals, bl, inf = _u64(bases), _u64(evals), _u64(blinders), _inf(inf)
    n = evals.shape[0]
    log_n = n.bit_length() - 1
    poly = np.zeros((n + 2, 4), dtype=np.uint64)
    out = np.empty(3 * FQ_LIMBS[curve], dtype=np.uint64)
    rc = lib().orc_round1(curve, _p(bases), _p(inf), C.c_size_t(bases.shape[0]), _p(evals), log_n, _p(bl),
                          _p(poly), _p(out), threads)
    assert rc == 0
    return poly, out


def quotient_evals(curve, log_n, sel, sig, wire, z, pi, alpha, beta, gamma, k, threads=1):
    """dispatcher2.rs:362-504.  sel (13,m,4), sig (5,m,4), wire (5,m,4), z (m,4), pi (m,4); alpha/beta/gamma (4,), k (5,4)."""
    m = 8 << log_n
    sel, sig, wire, z, pi = _u64(sel), _u64(sig), _u64(wire), _u64(z), _u64(pi)
    assert sel.shape == (13, m, 4) and sig.shape == (5, m, 4) and wire.shape == (5, m, 4) and z.shape == (m, 4) and pi.shape == (m, 4)
    out = np.empty((m, 4), dtype=np.uint64)
    rc = lib().orc_quotient_evals(curve, log_n, _p(sel), _p(sig), _p(wire), _p(z), _p(pi), _p(_u64(alpha)), _p(_u64(beta)),
                                  _p(_u64(gamma)), _p(_u64(k)), _p(out), threads)
    if rc:
        raise ValueError("DomainCreationError")
    return out


def perm_product(curve, wires, id_perm, perm_idx, beta, gamma):
    """dispatcher2.rs:329-344.  wires (5,n,4), id_perm (5n,4), perm_idx (5n,) u64 flattened perm_i*n+perm_j -> (n,4)."""
    wires, id_perm, perm_idx = _u64(wires), _u64(id_perm), _u64(perm_idx)
    n = wires.shape[1]
    assert wires.shape == (5, n, 4) and id_perm.shape == (5 * n, 4) and perm_idx.shape == (5 * n,)
    out = np.empty((n, 4), dtype=np.uint64)
    rc = lib().orc_perm_product(curve, C.c_size_t(n), _p(wires), _p(id_perm), _p(perm_idx), _p(_u64(beta)), _p(_u64(gamma)), _p(out))
    if rc:
        raise ZeroDivisionError("permutation product: zero denominator (the reference panics)")
    return out


def poly_eval(curve, coeffs, z):
    """DensePolynomial::evaluate, dispatcher2.rs:545-555 -> (4,)"""
    coeffs = _u64(coeffs)
    out = np.empty(4, dtype=np.uint64)
    lib().orc_poly_eval(curve, _p(coeffs), C.c_size_t(coeffs.shape[0]), _p(_u64(z)), _p(out))
    return out


def poly_lincomb(curve, polys, coeffs):
    """sum_k coeffs[k] * polys[k]; polys: list of (len_k,4); coeffs (k,4) -> (max len, 4)"""
    polys = [_u64(q) for q in polys]
    coeffs = _u64(coeffs)
    k = len(polys)
    out_len = max(q.shape[0] for q in polys)
    ptrs = (C.c_void_p * k)(*[q.ctypes.data for q in polys])
    lens = (C.c_size_t * k)(*[q.shape[0] for q in polys])
    out = np.empty((out_len, 4), dtype=np.uint64)
    lib().orc_poly_lincomb(curve, C.c_size_t(k), ptrs, lens, _p(coeffs), _p(out), C.c_size_t(out_len))
    return out


def poly_div_linear(curve, coeffs, z):
    """quotient of poly / (X - z), dispatcher2.rs:651-666 -> (len-1, 4)"""
    coeffs = _u64(coeffs)
    n = coeffs.shape[0]
    out = np.empty((max(n - 1, 0), 4), dtype=np.uint64)
    lib().orc_poly_div_linear(curve, _p(coeffs), C.c_size_t(n), _p(_u64(z)), _p(out))
    return out


def blind(curve, coeffs, n, blinders):
    """(sum b_i X^i)(X^n - 1) + poly -> (n + k, 4)"""
    bl = _u64(blinders)
    k = bl.shape[0]
    out = np.zeros((n + k, 4), dtype=np.uint64)
    c = _u64(coeffs)
    out[:c.shape[0]] = c
    lib().orc_blind(curve, _p(out), C.c_size_t(n), _p(bl), C.c_size_t(k))
    return out
