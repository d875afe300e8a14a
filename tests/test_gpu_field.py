"""Device field arithmetic (fp.hpp) vs the oracle: bit-exact limbs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CURVES = [("bn254", 0), ("bls12_381", 1)]
OPS = {"mul": 0, "add": 1, "sub": 2, "to_mont": 3, "from_mont": 4, "inv": 5, "sqr": 6}


def _rand_elems(oracle, cid, field, n, rng):
    p = int.from_bytes(oracle.field_const(cid, field, 0).tobytes(), "little")
    nl = 6 if (field == 1 and cid == 1) else 4
    vals = [int(rng.integers(0, 2**63)) ** 7 % p for _ in range(n - 4)] + [0, 1, p - 1, p - 2]
    return np.array([[(v >> (64 * i)) & (2**64 - 1) for i in range(nl)] for v in vals], dtype=np.uint64)


@pytest.mark.parametrize("curve,cid", CURVES)
@pytest.mark.parametrize("field", [0, 1])
def test_field_ops_bit_exact(gpu_workers, oracle, curve, cid, field):
    w = gpu_workers(curve)
    rng = np.random.default_rng(11 + cid + 2 * field)
    a = _rand_elems(oracle, cid, field, 1000, rng)
    b = _rand_elems(oracle, cid, field, 1000, rng)[::-1].copy()
    for name, op in OPS.items():
        n = 64 if name == "inv" else len(a)
        got = w.field_op(field, op, a[:n], b[:n])
        want = oracle.field_op(cid, field, name, a[:n], b[:n])
        assert np.array_equal(got, want), f"{curve} field{field} {name}"
