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
    d = w.alloc(80 * 32)
    o = w.alloc(64 * 32)
    one = P.fr_to_limbs(f, 1)
    with pytest.raises(PlonkError):
        w.coset_eval_dev(d.ptr, 8, 24, one, o.ptr)              # not a power of two
    with pytest.raises(PlonkError):
        w.coset_eval_dev(d.ptr, 65, 8, one, o.ptr)              # more than 8x folding (NTT_MAX_FOLD)
    # exactly 8x is what the residue-class iFFT of an 8-rank prover needs: 64 coefficients on an 8-point coset = the oracle's value of the folded polynomial
    h = 3 * g % f.p
    coeffs = oracle.rand_fr(0, 91, 64)
    d.upload(coeffs)
    w.coset_eval_dev(d.ptr, 64, 8, P.fr_to_limbs(f, h), o.ptr)
    w8 = f.root_of_unity(8)
    for k in (0, 3, 7):
        want = oracle.poly_eval(0, coeffs, P.fr_to_limbs(f, h * pow(w8, k, f.p) % f.p))
        assert np.array_equal(o.download((8, 4))[k], want), k
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


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_size,length", [
    (3, 1), (4, 3), (6, 64), (6, 65), (6, 100), (6, 256), (6, 0), (9, 70), (10, 1024),
    (12, (1 << 9) + 3),        # 8 classes of 2^9 points: one pass per class, interleaved stores
    (13, (1 << 10) + 3),       # 8 classes of 2^10 points: two passes (5, 5)
    (16, (1 << 12) + 1),       # 16 classes (the cap) of 2^12 points, one folded coefficient
    (19, (1 << 16) + 3),       # 8 classes of 2^16 points: passes (8, 8) — the shape of n = 2^16 gates
    (20, 5),                   # almost everything zero: 16 classes, inputs beyond `length` read as zero
    (20, 1 << 20),             # dense: one class, three passes (7, 7, 6)
    (20, (1 << 20) + 3),       # dense with fold (the six-coset / class-prover call shape)
    (21, (1 << 18) + 2),       # 8 classes of 2^18 points: passes (9, 9)
])
def test_coset_eval_zero_padding_classes_match_oracle(gpu_workers, oracle, curve, cid, log_size, length):
    """plonk_coset_eval_dev(shift = g, size) == quot_domain.coset_fft of the zero-padded coefficient vector (dispatcher2.rs:387-424,746)
    for every class count the planner can choose (1, 2, 4, 8, 16), with and without folding, single- and multi-pass classes."""
    w = gpu_workers(curve)
    size = 1 << log_size
    P, f, g, w_s = _consts(oracle, cid, log_size)
    poly = oracle.rand_fr(cid, 4000 + log_size, max(length, 1))[:length]
    folded = np.zeros((size, 4), dtype=np.uint64)
    if length <= size:
        folded[:length] = poly
    else:                                                     # X^size = g^size on the coset: fold on the host with exact integers
        c = pow(g, size, f.p)
        acc = [0] * size
        ints = [P.fr_from_limbs(f, x) for x in poly]
        for i, v in enumerate(ints):
            acc[i % size] = (acc[i % size] + v * pow(c, i // size, f.p)) % f.p
        folded = np.stack([P.fr_to_limbs(f, v) for v in acc])
    want = oracle.ntt(cid, folded, False, True, threads=32)
    dp = w.alloc(max(length, 1) * 32)
    if length:
        dp.upload(poly)
    out = w.alloc(size * 32)
    w.coset_eval_dev(dp.ptr, length, size, P.fr_to_limbs(f, g), out.ptr)
    got = out.download((size, 4))
    assert np.array_equal(got, want)
    if length:
        assert np.array_equal(dp.download((length, 4)), poly)          # the input is not modified
    dp.free(); out.free()


def test_coset_eval_full_size_8n_against_independent_kernels(gpu_workers, oracle):
    """BASELINE's size: n + 3 coefficients on the 8n = 2^27-point quotient coset (8 classes of 2^24 points, three passes each).
    (a) every output equals the dense transform of the explicitly zero-padded vector (plonk_ntt_dev: different planes, different
    tile mapping, 9-stage passes); (b) sampled outputs equal Horner evaluations by plonk_poly_eval_dev at g * w^k (an unrelated
    kernel, itself checked against the oracle in test_gpu_polyops); (c) the coset iNTT returns the padded coefficients."""
    w = gpu_workers("bn254")
    log_n = 24
    n, m = 1 << log_n, 8 << log_n
    P, f, g, w_m = _consts(oracle, 0, log_n + 3)
    length = n + 3
    dp = w.alloc(length * 32)
    w.synth_fr(0xC05E7, dp.ptr, length)
    a, b, c = w.alloc(m * 32), w.alloc(m * 32), w.alloc(m * 32)
    w.coset_eval_dev(dp.ptr, length, m, P.fr_to_limbs(f, g), a.ptr)
    w.memset_dev(b.ptr, 0, m * 32)
    w.memcpy_d2d(b.ptr, dp.ptr, length * 32)
    w.ntt_dev(b.ptr, c.ptr, m, False, True)                   # dense route (b destroyed)
    CH = 1 << 22
    for off in range(0, m, CH):
        assert np.array_equal(a.download((CH, 4), byte_offset=off * 32), c.download((CH, 4), byte_offset=off * 32)), off
    for k in (0, 1, 7, 8, 9, 12345, (5 << 24) + 3, m - 1):
        x = P.fr_to_limbs(f, g * pow(w_m, k, f.p) % f.p)
        assert np.array_equal(a.download((1, 4), byte_offset=k * 32)[0], w.poly_eval_dev(dp.ptr, length, x)), k
    w.ntt_dev(a.ptr, b.ptr, m, True, True)                    # coset iNTT of the class result (a destroyed)
    assert np.array_equal(b.download((length, 4)), dp.download((length, 4)))
    for off in range(length * 32, m * 32, CH * 32):
        nb = min(CH * 32, m * 32 - off)
        assert not b.download((nb // 8,), byte_offset=off).any(), off
    for x in (dp, a, b, c):
        x.free()
