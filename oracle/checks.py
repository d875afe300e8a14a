"""Exact expected values for MSMs over the synthetic base distributions, at ANY size, from small oracle MSMs.
Test infrastructure (like everything under oracle/): used by tests/ and by bench.py's post-run verification leg, never by the product.

Both synthetic SRS distributions of `plonk_synth_bases` have structure the check exploits:
  tiled    (unique = u):  P_i = T[i % u]                      =>  sum_i s_i P_i = sum_j (sum_{i = j mod u} s_i) T_j
  distinct (unique = 0):  P_i = A[i % 4096] + B[i / 4096]     =>  sum_i s_i P_i = sum_a (sum_{i % 4096 = a} s_i) A_a + sum_b (sum_{i / 4096 = b} s_i) B_b
with T = gen_bases(seed, u), A = gen_bases(seed, 4096), B = gen_bases(seed + 1, ceil(n / 4096)) (csrc/synth.hip, oracle orc_gen_bases:
k_j * G with k_j the raw limbs of rand_fr(seed)[j]).  The aggregated scalars are exact integers reduced mod the group order."""
import numpy as np

from . import oracle as O

NA = 4096


def _limbs_from_halves(halves: np.ndarray, p: int) -> np.ndarray:
    out = np.zeros((halves.shape[0], 4), dtype=np.uint64)
    for j in range(halves.shape[0]):
        v = sum(int(halves[j, k]) << (32 * k) for k in range(8)) % p
        out[j] = [(v >> (64 * k)) & (2**64 - 1) for k in range(4)]
    return out


def _order(cid: int) -> int:
    return int.from_bytes(O.field_const(cid, 0, 0).tobytes(), "little")


def msm_expected_tiled(cid: int, seed: int, unique: int, sc: np.ndarray, threads: int = 8) -> np.ndarray:
    """sc: canonical scalars (n, 4) u64, n a multiple of `unique`.  -> Jacobian point (oracle layout)."""
    n = sc.shape[0]
    assert n % unique == 0
    halves = np.ascontiguousarray(sc).view(np.uint32).reshape(n // unique, unique, 8).astype(np.uint64).sum(axis=0)
    return O.msm(cid, O.gen_bases(cid, seed, unique, unique), _limbs_from_halves(halves, _order(cid)), threads=threads)


def msm_expected_distinct(cid: int, seed: int, sc: np.ndarray, threads: int = 8) -> np.ndarray:
    """sc: canonical scalars (n, 4) u64 for bases plonk_synth_bases(seed, unique = 0, n).  -> Jacobian point (oracle layout)."""
    n = sc.shape[0]
    nbb = (n + NA - 1) // NA
    h = np.ascontiguousarray(sc).view(np.uint32).reshape(n, 8)
    if nbb * NA != n:
        h = np.vstack([h, np.zeros((nbb * NA - n, 8), dtype=np.uint32)])
    h = h.reshape(nbb, NA, 8)
    p = _order(cid)
    agg_a = _limbs_from_halves(h.sum(axis=0, dtype=np.uint64), p)      # sums of 32-bit halves: no overflow below 2^32 rows
    agg_b = _limbs_from_halves(h.sum(axis=1, dtype=np.uint64), p)
    A = O.gen_bases(cid, seed, NA, NA)
    B = O.gen_bases(cid, seed + 1, nbb, nbb)
    return O.jac_add(cid, O.msm(cid, A, agg_a, threads=threads), O.msm(cid, B, agg_b, threads=threads))
