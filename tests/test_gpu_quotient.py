"""Next row: coset evaluations of the quotient polynomial (dispatcher2.rs:362-504) on the
device vs the oracle restatement, bit-exact Montgomery limbs; then the reference's next step (:507, coset_ifft) on
the same buffer."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_n", [2, 5, 10, 13])
def test_quotient_evals_match_oracle(gpu_workers, oracle, curve, cid, log_n):
    w = gpu_workers(curve)
    n, m = 1 << log_n, 8 << log_n
    w.init(None, n, m)
    vecs = oracle.rand_fr(cid, 4000 + log_n, 25 * m).reshape(25, m, 4)
    ch = oracle.rand_fr(cid, 77, 8)                         # alpha, beta, gamma, k[5]
    alpha, beta, gamma, k = ch[0], ch[1], ch[2], ch[3:8]
    buf = w.alloc(25 * m * 32).upload(vecs)
    out = w.alloc(m * 32)
    ptr = [buf.ptr + j * m * 32 for j in range(25)]
    w.quotient_evals_dev(ptr[0:13], ptr[13:18], ptr[18:23], ptr[23], ptr[24], alpha, beta, gamma, k, out.ptr)
    want = oracle.quotient_evals(cid, log_n, vecs[0:13], vecs[13:18], vecs[18:23], vecs[23], vecs[24], alpha, beta, gamma, k, threads=16)
    got = out.download((m, 4))
    assert np.array_equal(got, want)
    # dispatcher2.rs:507: quot_domain.coset_ifft_in_place(&mut quot_poly_coset_evals)
    coeffs = w.alloc(m * 32)
    w.ntt_dev(out.ptr, coeffs.ptr, m, True, True)
    assert np.array_equal(coeffs.download((m, 4)), oracle.ntt(cid, want, True, True, threads=16))
    for b in (buf, out, coeffs):
        b.free()


def test_quotient_evals_structured_inputs(gpu_workers, oracle):
    """Zeros / ones / p-1 in the inputs (real selector vectors are sparse) and a second call reusing the cached tables."""
    w = gpu_workers("bn254")
    log_n = 6
    n, m = 1 << log_n, 8 << log_n
    w.init(None, n, m)
    vecs = oracle.rand_fr(0, 9, 25 * m).reshape(25, m, 4)
    pm1 = oracle.field_const(0, 0, 0) - np.array([1, 0, 0, 0], dtype=np.uint64)
    vecs[0:13, ::3] = 0
    vecs[18, ::5] = oracle.field_const(0, 0, 1)            # wire a = 1 (Montgomery one)
    vecs[23, ::7] = pm1
    ch = oracle.rand_fr(0, 78, 8)
    buf = w.alloc(25 * m * 32).upload(vecs)
    out = w.alloc(m * 32)
    ptr = [buf.ptr + j * m * 32 for j in range(25)]
    for _ in range(2):
        w.quotient_evals_dev(ptr[0:13], ptr[13:18], ptr[18:23], ptr[23], ptr[24], ch[0], ch[1], ch[2], ch[3:8], out.ptr)
        want = oracle.quotient_evals(0, log_n, vecs[0:13], vecs[13:18], vecs[18:23], vecs[23], vecs[24], ch[0], ch[1], ch[2], ch[3:8])
        assert np.array_equal(out.download((m, 4)), want)
    buf.free(); out.free()


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 7, 8])
def test_quotient_kernel_variants_agree_with_oracle(gpu_workers, oracle, curve, cid, variant):
    """The experimental formulations kept behind the `quotient_fuse` option (lifted wires with 1 / 2 / 3 products per Montgomery
    reduction, and the unlifted kernel at an uncapped register budget) compute the same values as the default kernel, bit for bit,
    on random and on extreme inputs (all p-1: the largest lazy sums the bound bookkeeping allows)."""
    w = gpu_workers(curve)
    log_n = 7
    n, m = 1 << log_n, 8 << log_n
    w.init(None, n, m)
    vecs = oracle.rand_fr(cid, 5000, 25 * m).reshape(25, m, 4)
    pm1 = oracle.field_const(cid, 0, 0) - np.array([1, 0, 0, 0], dtype=np.uint64)
    vecs[:, : m // 4] = pm1
    vecs[:, m // 4: m // 2: 2] = 0
    ch = oracle.rand_fr(cid, 79, 8)
    ch[0] = pm1                                              # alpha = p - 1 as well
    buf = w.alloc(25 * m * 32).upload(vecs)
    out = w.alloc(m * 32)
    ptr = [buf.ptr + j * m * 32 for j in range(25)]
    want = oracle.quotient_evals(cid, log_n, vecs[0:13], vecs[13:18], vecs[18:23], vecs[23], vecs[24], ch[0], ch[1], ch[2], ch[3:8], threads=8)
    try:
        for v in (6, variant):
            w.set_option("quotient_fuse", v)
            w.memset_dev(out.ptr, 0, m * 32)
            w.quotient_evals_dev(ptr[0:13], ptr[13:18], ptr[18:23], ptr[23], ptr[24], ch[0], ch[1], ch[2], ch[3:8], out.ptr)
            assert np.array_equal(out.download((m, 4)), want), v
    finally:
        w.set_option("quotient_fuse", 6)                  # back to the shipped default
    buf.free(); out.free()


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
@pytest.mark.parametrize("log_n,G", [(1, 8), (1, 4), (2, 8), (3, 8), (2, 2), (4, 8)])
def test_quotient_coset_class_of_a_tiny_domain(gpu_workers, oracle, curve, cid, log_n, G):
    """One coset class (points s + G*k) of domains so small that the class holds fewer points (2 ... 16) than one lane of the
    1/(x - 1) table generator handles (16): found by tools/fuzz_abi.py's `quotient` operation under AddressSanitizer — the generator
    stored its whole chunk, past the end of the m/G-entry table (regression test; on the host emulation's ASan build this is an error,
    on the GPU the values must match the oracle for every class)."""
    w = gpu_workers(curve)
    n, m = 1 << log_n, 8 << log_n
    w.init(None, n, m)
    vecs = oracle.rand_fr(cid, 6100 + 8 * log_n + G, 25 * m).reshape(25, m, 4)
    ch = oracle.rand_fr(cid, 80, 8)
    want = oracle.quotient_evals(cid, log_n, vecs[0:13], vecs[13:18], vecs[18:23], vecs[23], vecs[24], ch[0], ch[1], ch[2], ch[3:8])
    mL = m // G
    out = w.alloc(mL * 32)
    for s in range(G):
        buf = w.alloc(25 * mL * 32).upload(np.ascontiguousarray(vecs[:, s::G]))
        ptr = [buf.ptr + j * mL * 32 for j in range(25)]
        w.quotient_evals_dev(ptr[0:13], ptr[13:18], ptr[18:23], ptr[23], ptr[24], ch[0], ch[1], ch[2], ch[3:8], out.ptr, class_stride=G, class_offset=s)
        assert np.array_equal(out.download((mL, 4)), want[s::G]), s
        buf.free()
    out.free()


def test_output_overlapping_an_input_is_refused(gpu_workers):
    """d_out aliasing an input would be silently wrong (z is read at a shifted index; the split form writes d_out before it reads
    wires / sigmas / z) — plonk_hip.h states the no-alias rule and the entry point enforces it for every formulation."""
    from distributed_plonk_amd._ffi import PlonkError
    w = gpu_workers("bn254")
    log_n = 5
    n, m = 1 << log_n, 8 << log_n
    w.init(None, n, m)
    vecs = [w.alloc(m * 32) for _ in range(25)]
    for j, buf in enumerate(vecs):
        w.synth_fr(0x77 + j, buf.ptr, m)
    ptr = [b.ptr for b in vecs]
    ch = np.arange(32, dtype=np.uint64).reshape(8, 4) + 3
    out = w.alloc(m * 32)
    try:
        for fuse in (6, 8):
            w.set_option("quotient_fuse", fuse)
            for bad in (ptr[23], ptr[18], ptr[13], ptr[0], ptr[24], ptr[23] + 32 * (m - 1)):          # z, a wire, a sigma, a selector, pi, one element of overlap
                with pytest.raises(PlonkError) as e:
                    w.quotient_evals_dev(ptr[0:13], ptr[13:18], ptr[18:23], ptr[23], ptr[24], ch[0], ch[1], ch[2], ch[3:8], bad)
                assert e.value.code == -1 and "overlaps an input" in str(e.value)
            w.quotient_evals_dev(ptr[0:13], ptr[13:18], ptr[18:23], ptr[23], ptr[24], ch[0], ch[1], ch[2], ch[3:8], out.ptr)
    finally:
        w.set_option("quotient_fuse", 6)
        for b in vecs + [out]:
            b.free()
