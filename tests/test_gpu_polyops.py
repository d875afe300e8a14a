"""Next rows on the device vs the oracle restatement, bit-exact Montgomery limbs:
permutation grand product (dispatcher2.rs:329-344), DensePolynomial::evaluate (:545-555), scalar*poly sums (:566-633,
646-649), synthetic division by X - z (:651-666) and blinding (:311-312,347-348)."""
import numpy as np
import pytest

from distributed_plonk_amd._ffi import PlonkError

pytestmark = pytest.mark.gpu

CURVES = [("bn254", 0), ("bls12_381", 1)]


def _perm_inputs(oracle, cid, n, seed):
    rs = np.random.RandomState(seed)
    wires = oracle.rand_fr(cid, seed, 5 * n).reshape(5, n, 4)
    id_perm = oracle.rand_fr(cid, seed + 1, 5 * n)
    perm_idx = rs.permutation(5 * n).astype(np.uint64)
    ch = oracle.rand_fr(cid, seed + 2, 2)
    return wires, id_perm, perm_idx, ch[0], ch[1]


@pytest.mark.parametrize("curve,cid", CURVES)
@pytest.mark.parametrize("n", [2, 5, 64, 2048, 2049, 10000, 1 << 16])
def test_perm_product_matches_oracle(gpu_workers, oracle, curve, cid, n):
    w = gpu_workers(curve)
    wires, id_perm, perm_idx, beta, gamma = _perm_inputs(oracle, cid, n, 100 + n)
    dw = w.alloc(5 * n * 32).upload(wires)
    di = w.alloc(5 * n * 32).upload(id_perm)
    dp = w.alloc(5 * n * 8).upload(perm_idx)
    out = w.alloc(n * 32)
    w.perm_product_dev([dw.ptr + i * n * 32 for i in range(5)], di.ptr, dp.ptr, beta, gamma, n, out.ptr)
    want = oracle.perm_product(cid, wires, id_perm, perm_idx, beta, gamma)
    assert np.array_equal(out.download((n, 4)), want)
    for b in (dw, di, dp, out):
        b.free()


@pytest.mark.parametrize("curve,cid", CURVES)
@pytest.mark.parametrize("n,G", [(8, 2), (64, 4), (4096, 8), (5000, 8), (1 << 14, 3)])
def test_perm_product_by_gate_range_matches_oracle(gpu_workers, oracle, curve, cid, n, G):
    """plonk_perm_product_range_dev as G workers would call it (class_prover.py): slice s with one extra value = the slice's total; the
    slices multiplied by the totals before them are the oracle's product vector (dispatcher2.rs:329-344), bit for bit — uneven slices and
    slices that cross a scan tile included."""
    w = gpu_workers(curve)
    wires, id_perm, perm_idx, beta, gamma = _perm_inputs(oracle, cid, n, 300 + n)
    dw = w.alloc(5 * n * 32).upload(wires)
    di = w.alloc(5 * n * 32).upload(id_perm)
    dp = w.alloc(5 * n * 8).upload(perm_idx)
    want = oracle.perm_product(cid, wires, id_perm, perm_idx, beta, gamma)
    mul = lambda a, b: oracle.field_op(cid, 0, "mul", np.ascontiguousarray(a).reshape(-1, 4), np.ascontiguousarray(b).reshape(-1, 4))
    pre = oracle.field_const(cid, 0, 1)[:4]
    for s in range(G):
        lo, hi = s * n // G, (s + 1) * n // G
        cnt, extra = hi - lo, (0 if hi == n else 1)
        out = w.alloc((cnt + 1) * 32)
        w.perm_product_range_dev([dw.ptr + i * n * 32 for i in range(5)], di.ptr, dp.ptr, beta, gamma, n, lo, cnt + extra, out.ptr)
        loc = out.download((cnt + extra, 4))
        assert np.array_equal(mul(loc[:cnt], np.tile(pre, (cnt, 1))), want[lo:hi]), (s, lo, hi)
        if extra:
            pre = mul(pre, loc[cnt])[0]
        out.free()
    with pytest.raises(PlonkError):
        w.perm_product_range_dev([dw.ptr + i * n * 32 for i in range(5)], di.ptr, dp.ptr, beta, gamma, n, n - 2, 3, dw.ptr)
    with pytest.raises(PlonkError):
        w.perm_product_range_dev([dw.ptr + i * n * 32 for i in range(5)], di.ptr, dp.ptr, beta, gamma, n, 0, 0, dw.ptr)
    for b in (dw, di, dp):
        b.free()


def test_perm_product_valid_permutation_closes(gpu_workers, oracle):
    """With id_perm = k_i w^j and a real copy-constraint permutation the full-cycle product is 1: z[n-1] * ratio[n-1] = 1
    (size-independent property, checked at 2^20 against the oracle's last ratio only)."""
    from oracle import prover_ref as P
    w = gpu_workers("bn254")
    log_n = 20
    n = 1 << log_n
    circ = P.make_circuit(0, log_n, seed=5)
    ch = oracle.rand_fr(0, 31, 2)
    dw = w.alloc(5 * n * 32).upload(circ["wires"])
    di = w.alloc(5 * n * 32).upload(circ["id_perm"])
    dp = w.alloc(5 * n * 8).upload(circ["perm_idx"])
    out = w.alloc(n * 32)
    w.perm_product_dev([dw.ptr + i * n * 32 for i in range(5)], di.ptr, dp.ptr, ch[0], ch[1], n, out.ptr)
    got = out.download((n, 4))
    want = oracle.perm_product(0, circ["wires"], circ["id_perm"], circ["perm_idx"], ch[0], ch[1])
    assert np.array_equal(got, want)
    # last gate: num/den of j = n-1 times z[n-1] == 1
    j = n - 1
    mul = lambda a, b: oracle.field_op(0, 0, "mul", a.reshape(1, 4), b.reshape(1, 4))[0]
    add = lambda a, b: oracle.field_op(0, 0, "add", a.reshape(1, 4), b.reshape(1, 4))[0]
    num = den = oracle.field_const(0, 0, 1)[:4]
    for i in range(5):
        t = add(circ["wires"][i, j], ch[1])
        num = mul(num, add(t, mul(ch[0], circ["id_perm"][i * n + j])))
        den = mul(den, add(t, mul(ch[0], circ["id_perm"][int(circ["perm_idx"][i * n + j])])))
    assert np.array_equal(mul(got[j], num), den)
    for b in (dw, di, dp, out):
        b.free()


def test_perm_product_errors(gpu_workers, oracle):
    w = gpu_workers("bn254")
    n = 64
    wires, id_perm, perm_idx, beta, gamma = _perm_inputs(oracle, 0, n, 7)
    dw = w.alloc(5 * n * 32).upload(wires)
    di = w.alloc(5 * n * 32).upload(id_perm)
    out = w.alloc(n * 32)
    bad = perm_idx.copy()
    bad[17] = 5 * n                                        # out of range: the reference would index out of bounds
    dp = w.alloc(5 * n * 8).upload(bad)
    with pytest.raises(PlonkError) as e:
        w.perm_product_dev([dw.ptr + i * n * 32 for i in range(5)], di.ptr, dp.ptr, beta, gamma, n, out.ptr)
    assert e.value.code == -1
    # zero denominator: w + gamma + beta*id = 0 at (wire 2, gate 9): id := -(w + gamma)/beta
    sub = lambda a, b: oracle.field_op(0, 0, "sub", a.reshape(1, 4), b.reshape(1, 4))[0]
    mul = lambda a, b: oracle.field_op(0, 0, "mul", a.reshape(1, 4), b.reshape(1, 4))[0]
    inv = lambda a: oracle.field_op(0, 0, "inv", a.reshape(1, 4))[0]
    zero = np.zeros(4, dtype=np.uint64)
    t = oracle.field_op(0, 0, "add", wires[2, 9].reshape(1, 4), gamma.reshape(1, 4))[0]
    idp = id_perm.copy()
    idp[int(perm_idx[2 * n + 9])] = mul(sub(zero, t), inv(beta))
    di.upload(idp)
    dp.upload(perm_idx)
    with pytest.raises(PlonkError) as e:
        w.perm_product_dev([dw.ptr + i * n * 32 for i in range(5)], di.ptr, dp.ptr, beta, gamma, n, out.ptr)
    assert e.value.code == -1 and "zero denominator" in str(e.value)
    with pytest.raises(ZeroDivisionError):
        oracle.perm_product(0, wires, idp, perm_idx, beta, gamma)
    for b in (dw, di, dp, out):
        b.free()


@pytest.mark.parametrize("curve,cid", CURVES)
@pytest.mark.parametrize("length", [1, 2, 7, 1024, 1025, 2048, 70000, (1 << 20) + 3])
def test_poly_eval_and_div_match_oracle(gpu_workers, oracle, curve, cid, length):
    w = gpu_workers(curve)
    poly = oracle.rand_fr(cid, 900 + length % 1000, length)
    z = oracle.rand_fr(cid, 55, 1)[0]
    dpoly = w.alloc(length * 32).upload(poly)
    got = w.poly_eval_dev(dpoly.ptr, length, z)
    assert np.array_equal(got, oracle.poly_eval(cid, poly, z))
    dq = w.alloc(max(length - 1, 1) * 32)
    w.poly_div_linear_dev(dpoly.ptr, length, z, dq.ptr)
    if length > 1:
        assert np.array_equal(dq.download((length - 1, 4)), oracle.poly_div_linear(cid, poly, z))
    dpoly.free(); dq.free()


def test_poly_eval_div_special_points(gpu_workers, oracle):
    """z = 0, z = 1, z = p-1 and a polynomial with zero top coefficients (DensePolynomial would have trimmed them)."""
    w = gpu_workers("bn254")
    length = 5000
    poly = oracle.rand_fr(0, 12, length)
    poly[-37:] = 0
    one = oracle.field_const(0, 0, 1)[:4]
    zero = np.zeros(4, dtype=np.uint64)
    pm1 = oracle.field_op(0, 0, "sub", zero.reshape(1, 4), one.reshape(1, 4))[0]
    dpoly = w.alloc(length * 32).upload(poly)
    dq = w.alloc(length * 32)
    for z in (zero, one, pm1):
        assert np.array_equal(w.poly_eval_dev(dpoly.ptr, length, z), oracle.poly_eval(0, poly, z))
        w.poly_div_linear_dev(dpoly.ptr, length, z, dq.ptr)
        assert np.array_equal(dq.download((length - 1, 4)), oracle.poly_div_linear(0, poly, z))
    dpoly.free(); dq.free()


def test_power_table_cache_eviction_keeps_tables_of_the_current_call(gpu_workers, oracle):
    """the power-table cache holds 64 tables.  poly_div_linear(z) fetches the z table and then the 1/z table; with z
    the OLDEST cached entry and 1/z a miss on a full cache, a FIFO eviction freed the z table before the kernel read it.
    Fill the cache exactly, then divide by the oldest point (and keep dividing so freed memory gets reused)."""
    w = gpu_workers("bn254")
    length = 3000
    poly = oracle.rand_fr(0, 4242, length)
    pts = oracle.rand_fr(0, 4243, 70)
    dpoly = w.alloc(length * 32).upload(poly)
    dq = w.alloc(length * 32)
    # a fresh context would start empty; this one is shared, so first flush whatever is cached with 64 throw-away points
    for z in oracle.rand_fr(0, 4244, 64):
        w.poly_eval_dev(dpoly.ptr, length, z)
    w.poly_eval_dev(dpoly.ptr, length, pts[0])                 # oldest entry from here on
    for z in pts[1:64]:
        w.poly_eval_dev(dpoly.ptr, length, z)                  # cache full: pts[0] .. pts[63]
    for z in (pts[0], pts[1], pts[2]):                         # hit on z (oldest), miss on 1/z -> eviction inside the call
        w.poly_div_linear_dev(dpoly.ptr, length, z, dq.ptr)
        for filler in pts[64:]:                                # churn: the freed block is handed out again
            w.poly_eval_dev(dpoly.ptr, length, filler)
        w.poly_div_linear_dev(dpoly.ptr, length, z, dq.ptr)
        assert np.array_equal(dq.download((length - 1, 4)), oracle.poly_div_linear(0, poly, z))
    dpoly.free(); dq.free()


def test_poly_div_identity_full_size(gpu_workers, oracle):
    """Size-independent property at 2^24 + 3 coefficients (the batch polynomial's size at BASELINE's n):
    poly(r) == q(r) * (r - z) + poly(z) at a random r, all evaluated on the device."""
    w = gpu_workers("bn254")
    length = (1 << 24) + 3
    dpoly = w.alloc(length * 32)
    w.synth_fr(77, dpoly.ptr, length)
    dq = w.alloc(length * 32)
    z, r = oracle.rand_fr(0, 56, 2)
    w.poly_div_linear_dev(dpoly.ptr, length, z, dq.ptr)
    pz, pr = w.poly_eval_dev(dpoly.ptr, length, z), w.poly_eval_dev(dpoly.ptr, length, r)
    qr = w.poly_eval_dev(dq.ptr, length - 1, r)
    f = lambda op, a, b: oracle.field_op(0, 0, op, a.reshape(1, 4), b.reshape(1, 4))[0]
    assert np.array_equal(pr, f("add", f("mul", qr, f("sub", r, z)), pz))
    # spot-check the device evaluation itself against the oracle on the first 2^20 coefficients
    head = dpoly.download((1 << 20, 4))
    assert np.array_equal(w.poly_eval_dev(dpoly.ptr, 1 << 20, z), oracle.poly_eval(0, head, z))
    dpoly.free(); dq.free()


@pytest.mark.parametrize("curve,cid", CURVES)
def test_poly_lincomb_and_blind(gpu_workers, oracle, curve, cid):
    w = gpu_workers(curve)
    lens = [4096, 4098, 4099, 1, 0, 4096, 37] + [4096] * 25          # 32 terms, ragged
    polys = [oracle.rand_fr(cid, 300 + i, max(L, 1))[:L] for i, L in enumerate(lens)]
    coeffs = oracle.rand_fr(cid, 299, len(lens))
    coeffs[5] = 0
    coeffs[6] = oracle.field_const(cid, 0, 1)[:4]
    bufs = [w.alloc(max(L, 1) * 32).upload(q) if L else w.alloc(32) for q, L in zip(polys, lens)]
    out_len = max(lens)
    out = w.alloc(out_len * 32)
    w.poly_lincomb_dev([(b.ptr, L) for b, L in zip(bufs, lens)], coeffs, out.ptr, out_len)
    want = oracle.poly_lincomb(cid, [q.reshape(-1, 4) for q in polys], coeffs)
    assert np.array_equal(out.download((out_len, 4)), want)
    with pytest.raises(PlonkError):
        w.poly_lincomb_dev([(out.ptr, out_len)], coeffs[:1], out.ptr, out_len)          # aliasing
    with pytest.raises(PlonkError):
        w.poly_lincomb_dev([(b.ptr, L) for b, L in zip(bufs, lens)] + [(bufs[0].ptr, 1)], np.concatenate([coeffs, coeffs[:1]]), out.ptr, out_len)
    # blinding, k = 2 (wire polynomials) and k = 3 (permutation polynomial)
    n = 4096
    for k in (2, 3):
        base = np.zeros((n + k, 4), dtype=np.uint64)
        base[:n] = polys[0]
        bl = oracle.rand_fr(cid, 640 + k, k)
        d = w.alloc((n + k) * 32).upload(base)
        w.blind_dev(d.ptr, n, bl)
        assert np.array_equal(d.download((n + k, 4)), oracle.blind(cid, polys[0], n, bl))
        d.free()
    # a domain smaller than the mask (n = 1, 2 with three blinders): the two index ranges overlap (found by tools/fuzz_abi.py's prove_verify)
    for n_small, k in ((2, 3), (1, 3), (1, 2), (2, 2), (3, 3)):
        base = np.zeros((n_small + k, 4), dtype=np.uint64)
        base[:n_small] = polys[0][:n_small]
        bl = oracle.rand_fr(cid, 650 + 8 * n_small + k, k)
        d = w.alloc((n_small + k) * 32).upload(base)
        w.blind_dev(d.ptr, n_small, bl)
        assert np.array_equal(d.download((n_small + k, 4)), oracle.blind(cid, polys[0][:n_small], n_small, bl)), (n_small, k)
        d.free()
    for b in bufs + [out]:
        b.free()


@pytest.mark.parametrize("curve,cid", [("bn254", 0), ("bls12_381", 1)])
def test_poly_degree(gpu_workers, oracle, curve, cid):
    """plonk_poly_degree_dev = DensePolynomial::degree() after trimming (the WrongQuotientPolyDegree check, dispatcher2.rs:511-518): the index of the
    highest non-zero coefficient, -1 for the zero polynomial — for lengths around the kernel's 2048-coefficient workgroup chunk, a single non-zero
    at either end, trailing zeros, and a 2^20 + 5 vector whose top sits in the last partial chunk."""
    w = gpu_workers(curve)
    rs = np.random.RandomState(11 + cid)
    for length in (1, 2, 255, 256, 257, 2047, 2048, 2049, 5000, (1 << 20) + 5):
        v = oracle.rand_fr(cid, 300 + length % 97, length)
        cases = []
        for top in sorted({0, length - 1, int(rs.randint(0, length)), max(0, length - 2048), max(0, length - 2049)}):
            a = v.copy()
            a[top + 1:] = 0
            if not a[top].any():
                a[top, 0] = 1
            cases.append((a, top))
        z = np.zeros_like(v)
        cases.append((z, -1))
        one = z.copy()
        one[0, 3] = 1                      # only the top limb of coefficient 0 is set
        cases.append((one, 0))
        buf = w.alloc(length * 32)
        for a, want in cases:
            buf.upload(a)
            assert w.poly_degree_dev(buf.ptr, length) == want, (length, want)
        buf.free()
