"""Evaluation on / interpolation from arbitrary cosets (plonk_coset_eval_dev / plonk_coset_interp_dev) against the oracle's
whole-domain coset FFT: the class s of G is exactly the stride-G slice of the reference's 8n-point coset evaluations
(dispatcher2.rs:387-424), and the G per-class contributions sum to quot_domain.coset_ifft (dispatcher2.rs:507)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _consts(oracle, cid, log_m):
    from oracle import prover_ref as P
    f = P.CURVE_OBJ[cid].fr
    g = f.generator
    w_m = P.fr_from_limbs(f, oracle.field_const(cid, 0, 4, log_m))
    return P, f, g, w_m


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_n,G", [(4, 1), (6, 2), (10, 4), (12, 8)])
def test_class_evaluations_are_slices_of_the_coset_fft(gpu_workers, oracle, curve, cid, log_n, G):
    w = gpu_workers(curve)
    n, m = 1 << log_n, 8 << log_n
    P, f, g, w_m = _consts(oracle, cid, log_n + 3)
    length = n + 3                                              # the permutation polynomial's length
    poly = oracle.rand_fr(cid, 70 + log_n, length)
    padded = np.zeros((m, 4), dtype=np.uint64)
    padded[:length] = poly
    want = oracle.ntt(cid, padded, False, True, threads=8)      # quot_domain.coset_fft
    dp = w.alloc(length * 32).upload(poly)
    out = w.alloc((m // G) * 32)
    for s in range(G):
        shift = P.fr_to_limbs(f, g * pow(w_m, s, f.p))
        w.coset_eval_dev(dp.ptr, length, m // G, shift, out.ptr)
        assert np.array_equal(out.download((m // G, 4)), want[s::G]), (s, G)
    dp.free(); out.free()


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_m,G", [(5, 1), (9, 2), (13, 8)])
def test_class_contributions_sum_to_the_coset_ifft(gpu_workers, oracle, curve, cid, log_m, G):
    w = gpu_workers(curve)
    m = 1 << log_m
    P, f, g, w_m = _consts(oracle, cid, log_m)
    evals = oracle.rand_fr(cid, 90 + log_m, m)
    want = oracle.ntt(cid, evals, True, True, threads=8)        # quot_domain.coset_ifft
    scale = P.fr_to_limbs(f, pow(G, -1, f.p))
    i0, count = m // 4 + 3, m // 2                              # an arbitrary coefficient range
    acc = np.zeros((count, 4), dtype=np.uint64)
    de = w.alloc((m // G) * 32)
    out = w.alloc(count * 32)
    for s in range(G):
        de.upload(np.ascontiguousarray(evals[s::G]))
        shift = P.fr_to_limbs(f, g * pow(w_m, s, f.p))
        w.coset_interp_dev(de.ptr, m // G, shift, scale, i0, count, out.ptr)
        acc = oracle.field_op(cid, 0, "add", acc, out.download((count, 4)))
    assert np.array_equal(acc, want[i0:i0 + count])
    de.free(); out.free()


def test_coset_eval_argument_checks(gpu_workers, oracle):
    from distributed_plonk_amd._ffi import PlonkError
    w = gpu_workers("bn254")
    P, f, g, _ = _consts(oracle, 0, 5)
    d = w.alloc(64 * 32)
    o = w.alloc(64 * 32)
    one = P.fr_to_limbs(f, 1)
    with pytest.raises(PlonkError):
        w.coset_eval_dev(d.ptr, 8, 24, one, o.ptr)              # not a power of two
    with pytest.raises(PlonkError):
        w.coset_eval_dev(d.ptr, 64, 8, one, o.ptr)              # more than 4x folding
    with pytest.raises(PlonkError):
        w.coset_interp_dev(d.ptr, 16, np.zeros(4, dtype=np.uint64), one, 0, 16, o.ptr)      # zero shift
    d.free(); o.free()


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_n,G", [(5, 2), (9, 4), (11, 8)])
def test_quotient_kernel_on_a_class_equals_the_slice(gpu_workers, oracle, curve, cid, log_n, G):
    """plonk_quotient_evals_class_dev on the stride-G slices of the 25 input vectors == the stride-G slice of the whole-domain
    quotient evaluations (dispatcher2.rs:435-504); z(w x) stays inside the class."""
    w = gpu_workers(curve)
    n, m = 1 << log_n, 8 << log_n
    w.init(None, n, m)
    vecs = oracle.rand_fr(cid, 500 + log_n, 25 * m).reshape(25, m, 4)
    ch = oracle.rand_fr(cid, 77, 8)
    want = oracle.quotient_evals(cid, log_n, vecs[0:13], vecs[13:18], vecs[18:23], vecs[23], vecs[24], ch[0], ch[1], ch[2], ch[3:8], threads=8)
    mL = m // G
    buf = w.alloc(25 * mL * 32)
    out = w.alloc(mL * 32)
    ptr = [buf.ptr + j * mL * 32 for j in range(25)]
    for s in (0, G - 1, G // 2):
        buf.upload(np.ascontiguousarray(vecs[:, s::G]))
        w.quotient_evals_dev(ptr[0:13], ptr[13:18], ptr[18:23], ptr[23], ptr[24], ch[0], ch[1], ch[2], ch[3:8], out.ptr, class_stride=G, class_offset=s)
        assert np.array_equal(out.download((mL, 4)), want[s::G]), (s, G)
    buf.free(); out.free()
