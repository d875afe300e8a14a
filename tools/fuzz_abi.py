#!/usr/bin/env python3
"""Differential fuzzing of the C ABI against the CPU oracle: random operations with random shapes, flags and options — sizes that
are not in any fixed test list (odd lengths, single elements, lengths around the class / pass / slice boundaries), special field
values, repeated and infinite bases, forced Pippenger windows — every result compared bit-for-bit with the oracle.

    python tools/fuzz_abi.py --seconds 600 [--seed 1] [--curve bn254|bls12_381|both] [--max-log 13]

It drives whatever library `distributed_plonk_amd._ffi` loads: libplonk_hip.so on an MI355X, or — in the GPU-less build container —
the host emulation (tests/hostemu; PLONK_HIP_LIB=tests/hostemu/_build/plain/libplonk_hostemu.so PLONK_ALLOW_HOSTEMU=1), where it
is also worth running against the AddressSanitizer build.  The first mismatch prints the operation and its parameters (enough to
replay it with --seed / --only) and exits 1.  tests/test_hostemu.py runs a short fixed-seed slice of it in the CPU suite.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class Fuzz:
    def __init__(self, curve, cid, seed, max_log):
        from distributed_plonk_amd import fr as _fr
        from distributed_plonk_amd.worker import PlonkWorker
        from oracle import oracle as O
        self.O, self.cid, self.curve = O, cid, curve
        self.w = PlonkWorker(me=0, device=0, curve=curve)
        self.f = _fr.FIELDS[curve]
        self.rs = np.random.RandomState(seed)
        self.max_log = max_log
        self.counter = 0
        self.n_bases = 0

    # ---------------------------------------------------------------- inputs
    def seed(self):
        self.counter += 1
        return int(self.rs.randint(1, 1 << 30)) + self.counter

    def fr(self, n):
        """n field elements (Montgomery limbs): random, with a sprinkling of 0, 1, p - 1 and small values"""
        v = self.O.rand_fr(self.cid, self.seed(), max(n, 1))[:n].copy()
        if n and self.rs.rand() < 0.5:
            special = [self.f.to_limbs(x) for x in (0, 1, self.f.p - 1, 2, self.f.p - 2)]
            for _ in range(int(self.rs.randint(1, 4))):
                v[int(self.rs.randint(0, n))] = special[int(self.rs.randint(0, len(special)))]
        if n and self.rs.rand() < 0.05:
            v[:] = 0
        return v

    def one(self):
        return self.fr(3)[int(self.rs.randint(0, 3))]

    def up(self, arr):
        b = self.w.alloc(max(arr.nbytes, 32))
        if arr.nbytes:
            b.upload(np.ascontiguousarray(arr))
        return b

    # ---------------------------------------------------------------- operations
    def op_ntt(self):
        log_n = int(self.rs.randint(0, self.max_log + 1))
        n = 1 << log_n
        inv, coset = bool(self.rs.randint(0, 2)), bool(self.rs.randint(0, 2))
        v = self.fr(n)
        d_in, d_out = self.up(v), self.w.alloc(n * 32)
        self.w.ntt_dev(d_in.ptr, d_out.ptr, n, inv, coset)
        got = d_out.download((n, 4))
        d_in.free(); d_out.free()
        return np.array_equal(got, self.O.ntt(self.cid, v, inv, coset, threads=4)), dict(log_n=log_n, inv=inv, coset=coset)

    def op_coset_eval_interp(self):
        log_s = int(self.rs.randint(1, self.max_log + 1))
        size = 1 << log_s
        pick = self.rs.rand()
        if pick < 0.3:
            length = int(self.rs.randint(1, 4 * size + 1))
        elif pick < 0.6:                                         # around the class boundaries size / 2^k (+ the 3 folded coefficients)
            length = max(1, (size >> int(self.rs.randint(0, min(log_s, 5) + 1))) + int(self.rs.randint(-2, 5)))
        else:
            length = int(self.rs.randint(1, size + 4))
        length = min(length, 4 * size)
        shift_i = int.from_bytes(self.rs.bytes(31), "little") % self.f.p or 5
        shift = self.f.to_limbs(shift_i)
        poly = self.fr(length)
        d_p, d_o = self.up(poly), self.w.alloc(size * 32)
        self.w.coset_eval_dev(d_p.ptr, length, size, shift, d_o.ptr)
        got = d_o.download((size, 4))
        # expected: fold the coefficients beyond `size` (X^size = shift^size on the coset), scale by shift^i, plain NTT
        p = self.f.p
        folded = [0] * size
        ssz = pow(shift_i, size, p)
        for i in range(length):
            folded[i % size] = (folded[i % size] + self.f.from_limbs(poly[i]) * pow(ssz, i // size, p)) % p
        sp = 1
        for i in range(size):
            folded[i] = folded[i] * sp % p
            sp = sp * shift_i % p
        want = self.O.ntt(self.cid, self.f.vec_to_limbs(folded), False, False, threads=4)
        ok = np.array_equal(got, want)
        info = dict(log_size=log_s, length=length)
        if ok and self.rs.rand() < 0.5:                          # and back: interpolation of a window of coefficients
            i0 = int(self.rs.randint(0, size))
            count = int(self.rs.randint(1, size - i0 + 1))
            scale_i = int.from_bytes(self.rs.bytes(31), "little") % p or 1
            d_c = self.w.alloc(count * 32)
            self.w.coset_interp_dev(d_o.ptr, size, shift, self.f.to_limbs(scale_i), i0, count, d_c.ptr)
            back = d_c.download((count, 4))
            d_c.free()
            # the interpolant of the evaluations is the FOLDED polynomial (before the shift scaling)
            sinv = pow(shift_i, -1, p)
            coeff = [0] * size
            for i in range(length):
                coeff[i % size] = (coeff[i % size] + self.f.from_limbs(poly[i]) * pow(ssz, i // size, p)) % p
            want_c = self.f.vec_to_limbs([scale_i * coeff[i0 + t] % p for t in range(count)])
            ok = np.array_equal(back, want_c)
            info.update(interp=(i0, count))
            del sinv
        d_p.free(); d_o.free()
        return ok, info

    def ensure_bases(self, n):
        """a fresh SRS now and then: distinct points, tiled points (equal bases in one bucket), a few points at infinity"""
        if self.n_bases >= n and self.rs.rand() < 0.8:
            return
        n_new = max(n, int(self.rs.randint(1, 1 << min(self.max_log, 12)) + 1))
        unique = n_new if self.rs.rand() < 0.5 else int(self.rs.randint(1, min(n_new, 64) + 1))
        bases = self.O.gen_bases(self.cid, self.seed(), unique, n_new)
        inf = np.zeros(n_new, dtype=np.uint8)
        for _ in range(int(self.rs.randint(0, 3))):
            i = int(self.rs.randint(0, n_new))
            bases[i] = 0                                              # (0, 0) is infinity in the XY layout
            inf[i] = 1
        self.bases, self.inf, self.n_bases = bases, inf, n_new
        self.w.init(bases, 1 << self.max_log, 8 << self.max_log)

    def scalars(self, n):
        s = self.O.from_mont(self.cid, self.fr(n))
        if n and self.rs.rand() < 0.3:                                # skew: many equal digits
            s[self.rs.rand(n) < 0.7] = s[0]
        return s

    def affine_eq(self, jac, want_jac):
        a, ai = self.w.g1_to_affine(jac)
        b, bi = self.O.jac_to_affine(self.cid, want_jac)
        return ai == bi and np.array_equal(a, b)

    def op_msm(self):
        from distributed_plonk_amd._ffi import MsmWorkload
        n = int(self.rs.randint(1, 1 << min(self.max_log, 12)) + 1)
        self.ensure_bases(n)
        start = int(self.rs.randint(0, self.n_bases - n + 1))
        window = int(self.rs.choice([0, 0, 0, 2, 3, 5, 8, 11, 13]))
        persist = int(self.rs.choice([4, 4, 0, -1, -3]))
        grid = int(self.rs.choice([0, 1]))
        self.w.set_option("msm_window", window)
        self.w.set_option("msm_acc_persist", persist)
        self.w.set_option("msm_reduce_grid", grid)
        sc = self.scalars(n)
        try:
            jac = self.w.var_msm(MsmWorkload(start, start + n), sc)
        finally:
            self.w.set_option("msm_window", 0)
            self.w.set_option("msm_acc_persist", 4)
            self.w.set_option("msm_reduce_grid", 0)
        return (self.affine_eq(jac, self.O.msm(self.cid, self.bases[start:start + n], sc, inf=self.inf[start:start + n], threads=4)),
                dict(n=n, start=start, window=window, persist=persist, grid=grid))

    def op_msm_table(self):
        """The fixed-base window table forced on (msm_precompute = 2) with a random window width and bucket sets per scalar, a fresh SRS with
        repeated / infinite bases: a sub-range MSM and a batched round of ragged commitments against the oracle; the table is switched off again."""
        from distributed_plonk_amd._ffi import MsmWorkload
        n_b = int(self.rs.randint(2, 1 << min(self.max_log, 11)) + 1)
        unique = n_b if self.rs.rand() < 0.5 else int(self.rs.randint(1, min(n_b, 64) + 1))
        bases = self.O.gen_bases(self.cid, self.seed(), unique, n_b)
        inf = np.zeros(n_b, dtype=np.uint8)
        if self.rs.rand() < 0.5:
            i = int(self.rs.randint(0, n_b))
            bases[i] = 0
            inf[i] = 1
        tc = int(self.rs.choice([0, 0, 4, 7, 9, 12]))
        ts = int(self.rs.choice([0, 1, 2, 3])) if tc else 0
        info = dict(n_bases=n_b, unique=unique, table_c=tc, table_sets=ts)
        try:
            self.w.set_option("msm_precompute", 2)
            self.w.set_option("msm_table_c", tc)
            self.w.set_option("msm_table_sets", ts)
            self.w.init(bases, 0, 0)
            lo = int(self.rs.randint(0, n_b))
            hi = int(self.rs.randint(lo, n_b)) + 1
            sc = self.scalars(hi - lo)
            ok = self.affine_eq(self.w.var_msm(MsmWorkload(lo, hi), sc), self.O.msm(self.cid, bases[lo:hi], sc, inf=inf[lo:hi], threads=4))
            info.update(range=(lo, hi))
            if ok:
                K = int(self.rs.randint(1, 5))
                lens = [int(self.rs.randint(0, n_b + 1)) for _ in range(K)]
                polys = [self.fr(ln) for ln in lens]
                bufs = [self.up(p_) for p_ in polys]
                jacs = self.w.commit_many_dev([(b.ptr, ln) for b, ln in zip(bufs, lens)])
                for j, (p_, ln) in enumerate(zip(polys, lens)):
                    want = self.O.commit_polynomial(self.cid, bases[:max(ln, 1)], p_ if ln else self.f.vec_to_limbs([0]), inf=inf[:max(ln, 1)], threads=4)
                    ok = ok and self.affine_eq(jacs[j], want)
                for b in bufs:
                    b.free()
                info.update(lens=lens)
        finally:
            self.w.set_option("msm_precompute", 0)
            self.w.set_option("msm_table_c", 0)
            self.w.set_option("msm_table_sets", 0)
            self.n_bases = 0                                          # the next MSM operation installs a plain SRS
        return ok, info

    def op_commit_many(self):
        K = int(self.rs.randint(1, 7))
        n = int(self.rs.randint(1, 1 << min(self.max_log, 11)) + 1)
        self.ensure_bases(n)
        start = int(self.rs.randint(0, self.n_bases - n + 1))
        lens = [int(self.rs.randint(0, n + 1)) for _ in range(K)]
        lens[int(self.rs.randint(0, K))] = n
        polys = [self.fr(ln) for ln in lens]
        bufs = [self.up(p_) for p_ in polys]
        grid = int(self.rs.choice([0, 1]))
        self.w.set_option("msm_reduce_grid", grid)
        try:
            jacs = self.w.commit_many_dev([(b.ptr, ln) for b, ln in zip(bufs, lens)], start=start)
        finally:
            self.w.set_option("msm_reduce_grid", 0)
        ok = True
        for j, (p_, ln) in enumerate(zip(polys, lens)):
            want = self.O.commit_polynomial(self.cid, self.bases[start:start + max(ln, 1)], p_ if ln else self.f.vec_to_limbs([0]),
                                            inf=self.inf[start:start + max(ln, 1)], threads=4)
            ok = ok and self.affine_eq(jacs[j], want)
        for b in bufs:
            b.free()
        return ok, dict(K=K, n=n, start=start, lens=lens, grid=grid)

    def op_poly(self):
        n = int(self.rs.randint(1, 1 << self.max_log) + 1)
        poly = self.fr(n)
        z = self.one()
        d_p = self.up(poly)
        ok = np.array_equal(self.w.poly_eval_dev(d_p.ptr, n, z), self.O.poly_eval(self.cid, poly, z))
        info = dict(n=n, what="eval")
        if ok and n >= 2:
            d_q = self.w.alloc(n * 32)
            self.w.poly_div_linear_dev(d_p.ptr, n, z, d_q.ptr)
            ok = np.array_equal(d_q.download((n - 1, 4)), self.O.poly_div_linear(self.cid, poly, z))
            d_q.free()
            info["what"] = "div_linear"
        if ok:
            top = int(self.rs.randint(0, n))
            poly2 = poly.copy()
            poly2[top + 1:] = 0
            d2 = self.up(poly2)
            want = -1
            for i in range(n - 1, -1, -1):
                if poly2[i].any():
                    want = i
                    break
            ok = self.w.poly_degree_dev(d2.ptr, n) == want
            d2.free()
            info["what"] = "degree"
        d_p.free()
        return ok, info

    def op_lincomb(self):
        T = int(self.rs.randint(1, 25))
        out_len = int(self.rs.randint(1, 1 << min(self.max_log, 12)) + 1)
        lens = [int(self.rs.randint(1, out_len + 1)) for _ in range(T)]
        polys = [self.fr(ln) for ln in lens]
        coeffs = self.fr(T)
        bufs = [self.up(p_) for p_ in polys]
        d_o = self.w.alloc(out_len * 32)
        self.w.poly_lincomb_dev([(b.ptr, ln) for b, ln in zip(bufs, lens)], coeffs, d_o.ptr, out_len)
        got = d_o.download((out_len, 4))
        padded = []
        for p_ in polys:
            q = np.zeros((out_len, 4), dtype=np.uint64)
            q[:len(p_)] = p_
            padded.append(q)
        want = self.O.poly_lincomb(self.cid, padded, coeffs)
        for b in bufs:
            b.free()
        d_o.free()
        return np.array_equal(got, want[:out_len]), dict(T=T, out_len=out_len, lens=lens)

    def op_perm_product(self):
        n = int(self.rs.randint(2, 1 << min(self.max_log, 12)) + 1)
        # plain random values: with the special ones a denominator can vanish, which is an error here as it is a panic in the reference
        wires = self.O.rand_fr(self.cid, self.seed(), 5 * n).reshape(5, n, 4)
        id_perm = self.O.rand_fr(self.cid, self.seed(), 5 * n)
        perm_idx = self.rs.permutation(5 * n).astype(np.uint64)
        beta, gamma = self.O.rand_fr(self.cid, self.seed(), 2)
        dw, di, dp = self.up(wires), self.up(id_perm), self.up(perm_idx)
        out = self.w.alloc(n * 32)
        self.w.perm_product_dev([dw.ptr + i * n * 32 for i in range(5)], di.ptr, dp.ptr, beta, gamma, n, out.ptr)
        got = out.download((n, 4))
        try:
            want = self.O.perm_product(self.cid, wires, id_perm, perm_idx, beta, gamma)
        finally:
            for b in (dw, di, dp, out):
                b.free()
        return np.array_equal(got, want), dict(n=n)

    def op_perm_product_ranges(self):
        """the product vector assembled from G gate ranges (plonk_perm_product_range_dev; class_prover.py): uneven slices, slices of one gate,
        every slice multiplied by the totals before it == the oracle's vector"""
        n = int(self.rs.randint(2, 1 << min(self.max_log, 12)) + 1)
        G = int(self.rs.randint(1, min(n, 9)))
        wires = self.O.rand_fr(self.cid, self.seed(), 5 * n).reshape(5, n, 4)
        id_perm = self.O.rand_fr(self.cid, self.seed(), 5 * n)
        perm_idx = self.rs.permutation(5 * n).astype(np.uint64)
        beta, gamma = self.O.rand_fr(self.cid, self.seed(), 2)
        dw, di, dp = self.up(wires), self.up(id_perm), self.up(perm_idx)
        want = self.O.perm_product(self.cid, wires, id_perm, perm_idx, beta, gamma)
        cuts = sorted(set([0, n] + [int(x) for x in self.rs.randint(1, n, size=G - 1)])) if G > 1 else [0, n]
        mul = lambda a, b: self.O.field_op(self.cid, 0, "mul", np.ascontiguousarray(a).reshape(-1, 4), np.ascontiguousarray(b).reshape(-1, 4))
        pre = self.f.to_limbs(1)
        ok = True
        try:
            for lo, hi in zip(cuts, cuts[1:]):
                cnt, extra = hi - lo, (0 if hi == n else 1)
                out = self.w.alloc((cnt + 1) * 32)
                self.w.perm_product_range_dev([dw.ptr + i * n * 32 for i in range(5)], di.ptr, dp.ptr, beta, gamma, n, lo, cnt + extra, out.ptr)
                loc = out.download((cnt + extra, 4))
                out.free()
                ok = ok and np.array_equal(mul(loc[:cnt], np.tile(pre, (cnt, 1))), want[lo:hi])
                if extra:
                    pre = mul(pre, loc[cnt])[0]
        finally:
            for b in (dw, di, dp):
                b.free()
        return ok, dict(n=n, cuts=cuts)

    def op_class_ifft(self):
        """a size-n iFFT by residue class on one context: G class evaluations (plonk_coset_eval_dev with a G-fold) + plonk_class_interleave_dev
        (reverse, 1/n) == the oracle's domain.ifft; random class stride (several polynomials side by side)"""
        G = int(2 ** self.rs.randint(0, 4))
        log_n = int(self.rs.randint(max(1, int(np.log2(G)) + 1), min(self.max_log, 13) + 1))
        n = 1 << log_n
        L = n // G
        K, k_sel = int(self.rs.randint(1, 4)), 0
        k_sel = int(self.rs.randint(0, K))
        ev = self.fr(n)
        d_ev = self.up(ev)
        d_all, d_out = self.w.alloc(G * K * L * 32), self.w.alloc(n * 32)
        self.w.memset_dev(d_all.ptr, 0xA5, G * K * L * 32)
        try:
            for s_ in range(G):
                shift = self.f.to_limbs(pow(self.f.root_of_unity(n), (n - s_) % n, self.f.p))
                self.w.coset_eval_dev(d_ev.ptr, n, L, shift, d_all.ptr + ((s_ * K + k_sel) * L) * 32)
            self.w.class_interleave_dev(d_all.ptr + k_sel * L * 32, G, L, True, self.f.to_limbs(self.f.inv(n)), d_out.ptr, in_stride=K * L)
            got = d_out.download((n, 4))
        finally:
            for b in (d_ev, d_all, d_out):
                b.free()
        return np.array_equal(got, self.O.ntt(self.cid, ev, True, False)), dict(log_n=log_n, G=G, K=K, k=k_sel)

    def op_init_refuses_bad_srs(self):
        """plonk_init: a random corruption of a valid SRS (a flipped bit, swapped coordinates of one point, an unreduced limb) is refused with the
        offender's index; the untouched SRS installs and commits like the oracle"""
        from distributed_plonk_amd._ffi import MsmWorkload, PlonkError
        n = int(self.rs.randint(1, 300))
        bases = self.O.gen_bases(self.cid, self.seed(), min(n, 16), n)
        half = bases.shape[1] // 2
        bad = bases.copy()
        i = int(self.rs.randint(0, n))
        kind = int(self.rs.randint(0, 3))
        if kind == 0:
            bad[i, int(self.rs.randint(0, bases.shape[1]))] ^= np.uint64(1 << int(self.rs.randint(0, 60)))
        elif kind == 1:
            bad[i] = np.concatenate([bases[i, half:], bases[i, :half]])
        else:
            bad[i, half - 1] = np.uint64(0xFFFFFFFFFFFFFFFF)
        refused = False
        try:
            self.w.init(bad, 0, 0)
        except PlonkError as ex:
            refused = ex.code == -1 and f"base {i} of" in str(ex)
        self.w.init(bases, 0, 0)
        self.n_bases = 0                                   # the shared SRS of the other operations is gone: they re-initialise
        sc = self.O.from_mont(self.cid, self.O.rand_fr(self.cid, self.seed(), n))
        got = self.w.g1_to_affine(self.w.var_msm(MsmWorkload(0, n), sc))
        exp = self.O.jac_to_affine(self.cid, self.O.msm(self.cid, bases, sc, threads=4))
        return refused and got[1] == exp[1] and np.array_equal(got[0], exp[0]), dict(n=n, i=i, kind=kind)

    def op_trim(self):
        """plonk_trim between two operations: every rebuildable cache of the context (pooled exchange buffers, factor planes and class tables, MSM
        workspace, scratch) goes back to the device; the transform right after it and everything the fuzzer draws next must rebuild what it needs."""
        self.w.trim()
        log_n = int(self.rs.randint(1, min(self.max_log, 11) + 1))
        v = self.fr(1 << log_n)
        inv, coset = bool(self.rs.randint(0, 2)), bool(self.rs.randint(0, 2))
        return np.array_equal(self.w.ntt(v, inv, coset), self.O.ntt(self.cid, v, inv, coset, threads=4)), dict(log_n=log_n, inv=inv, coset=coset)

    def op_transpose(self):
        rows, cols = int(self.rs.randint(1, 200)), int(self.rs.randint(1, 200))
        v = self.fr(rows * cols)
        got = self.w.transpose(v, rows, cols)
        return np.array_equal(got, np.ascontiguousarray(v.reshape(rows, cols, 4).transpose(1, 0, 2)).reshape(-1, 4)), dict(rows=rows, cols=cols)

    def op_distributed_fft(self):
        from distributed_plonk_amd.dispatcher import Dispatcher
        from distributed_plonk_amd.worker import PlonkWorker
        log_n = int(self.rs.randint(2, min(self.max_log, 12) + 1))
        n = 1 << log_n
        S = int(self.rs.choice([1, 2, 4]))
        r = 1 << (log_n >> 1)
        if r % S or (n // r) % S:
            S = 1
        inv, coset = bool(self.rs.randint(0, 2)), bool(self.rs.randint(0, 2))
        ws = [PlonkWorker(me=i, device=0, curve=self.curve) for i in range(S)]
        try:
            d = Dispatcher(ws)
            d.init(None, n, 8 * n)
            v = self.fr(n)
            got = d.fft(v, is_quot=False, is_inv=inv, is_coset=coset)
        finally:
            for x in ws:
                x.close()
        return np.array_equal(got, self.O.ntt(self.cid, v, inv, coset, threads=4)), dict(log_n=log_n, S=S, inv=inv, coset=coset)

    def op_round1(self):
        """worker.rs:383-408: evaluations -> ifft -> + blinders * Z_H -> commitment over the whole SRS; the polynomial stays in the context."""
        log_n = int(self.rs.randint(1, min(self.max_log, 11) + 1))
        n = 1 << log_n
        n_b = n + 2 + int(self.rs.randint(0, 40))
        unique = n_b if self.rs.rand() < 0.5 else int(self.rs.randint(1, min(n_b, 64) + 1))
        bases = self.O.gen_bases(self.cid, self.seed(), unique, n_b)
        inf = np.zeros(n_b, dtype=np.uint8)
        if self.rs.rand() < 0.5:
            i = int(self.rs.randint(0, n_b))
            bases[i] = 0
            inf[i] = 1
        self.w.init(bases, n, 8 * n)
        self.n_bases = 0                                              # the next MSM operation installs its own SRS
        evals, bl = self.fr(n), self.fr(2)
        poly, cm = self.O.round1(self.cid, bases, evals, bl, inf, threads=4)
        got = self.w.round1(evals, bl)
        ok = np.array_equal(self.w.get_wire(n + 2), poly)
        g, gi = self.w.g1_to_affine(got)
        o, oi = self.O.jac_to_affine(self.cid, cm)
        return ok and gi == oi and np.array_equal(g, o), dict(log_n=log_n, n_bases=n_b, unique=unique)

    def op_prove_verify(self):
        """The reference's end-to-end test (dispatcher2.rs:1273-1295: prove, then verify) on a random instance: a satisfied circuit and a
        trapdoor SRS generated on the device, the five rounds with the merlin transcript and the quotient-degree check on, both quotient
        routes, key cosets cached or not; the proof must be accepted by oracle/verifier_ref.py, which must also have drawn the prover's
        challenges, and a flipped evaluation must be rejected."""
        from distributed_plonk_amd.prover import Prover
        from distributed_plonk_amd.synthetic import SyntheticInstance
        from distributed_plonk_amd.transcript import PlonkTranscript
        from oracle import bigint_ref as B, verifier_ref as V
        # n >= 4: the quotient's numerator has degree 6n + 7, which the reference's 8n-point domain only holds from n = 4 on
        log_n = int(self.rs.randint(2, min(self.max_log, 7) + 1))
        mode = "classes6" if (log_n >= 4 and self.rs.rand() < 0.5) else "coset8n"
        cache = bool(self.rs.randint(0, 2))
        n_in = int(self.rs.randint(0, min(1 << log_n, 6) + 1))
        tau = int.from_bytes(self.rs.bytes(31), "little") % self.f.p or 7
        seed = self.seed()
        inst = SyntheticInstance(self.w, log_n, seed=seed, num_inputs=n_in, tau=tau)
        self.n_bases = 0                                              # the instance installed its own commit key
        pv = Prover(self.w, log_n, quotient_mode=mode, cache_key_cosets=cache)
        info = dict(log_n=log_n, mode=mode, cache=cache, num_inputs=n_in, seed=seed)
        try:
            pv.load_key_dev(inst.sel_ptrs, inst.sig_ptrs, inst.k)
            pub = inst.public_inputs()
            fs = pv.fiat_shamir(pub)
            # blinders: plain random values (a zero blinder lowers the quotient's degree and fails the reference's degree check too)
            bl = dict(wires=self.O.rand_fr(self.cid, self.seed(), 10).reshape(5, 2, 4), perm=self.O.rand_fr(self.cid, self.seed(), 3))
            proof = pv.prove_dev(inst.wev, inst.d_id.ptr, inst.d_idx.ptr, inst.d_pi.ptr, bl, fs, check_degree=True)
            vk = pv.verifying_key()
            cv = B.CURVES[self.curve]
            out = V.verify(cv, vk, pub, proof, tau, transcript=PlonkTranscript(self.curve))
            ok = all(np.array_equal(out["challenges"][k_], fs.drawn[k_]) for k_ in ("beta", "gamma", "alpha", "zeta", "v"))
            bad = [x.copy() for x in proof["wires_evals"]]
            bad[int(self.rs.randint(0, 5))][0] ^= np.uint64(1)
            try:
                V.verify(cv, vk, pub, dict(proof, wires_evals=bad), tau, transcript=PlonkTranscript(self.curve))
                ok = False
            except V.VerificationError:
                pass
        except V.VerificationError as ex:
            ok, info["rejected"] = False, str(ex)[:200]
        finally:
            pv.close()
            inst.close()
        return ok, info

    def op_class_prove(self):
        """The multi-rank prover by coset classes (class_prover.py) as G threads sharing the device: random size, rank count, whole or
        sharded commit key; every rank must produce the oracle prover's proof (commitments, evaluations, quotient / linearisation / batch
        polynomials) for the same circuit, blinders and challenges."""
        from distributed_plonk_amd.class_prover import ClassProver, key_shard_range, run_local_ranks
        from oracle import prover_ref as P
        log_n = int(self.rs.randint(3, min(self.max_log, 7) + 1))
        n = 1 << log_n
        G = int(self.rs.choice([2, 4, 8]))
        sharded = bool(self.rs.randint(0, 2))
        seed = self.seed()
        circ = P.make_circuit(self.cid, log_n, seed=seed)
        ck, inf = P.make_ck(self.cid, n, seed=seed + 1, unique=min(64, n))
        bl = dict(wires=self.O.rand_fr(self.cid, seed + 2, 10).reshape(5, 2, 4), perm=self.O.rand_fr(self.cid, seed + 3, 3))
        ch = {k: self.O.rand_fr(self.cid, seed + 10 + i, 1)[0] for i, k in enumerate(("beta", "gamma", "alpha", "zeta", "v"))}
        K = len(ck)

        def rank_main(comm, w):
            klo, khi = key_shard_range(K, comm.rank, comm.size) if sharded else (0, K)
            w.init(ck[klo:khi], n, 8 * n)
            pv = ClassProver(w, log_n, comm, key_range=(klo, khi) if sharded else None)
            try:
                pv.load_key(circ["selectors"], circ["sigmas"], circ["k"])
                return pv.prove(circ["wires"], circ["id_perm"], circ["perm_idx"], circ["pub_input"], bl, lambda label, _: ch[label], keep=True)
            finally:
                pv.close()

        results = run_local_ranks(G, rank_main, curve=self.curve)
        want = P.prove_rounds(self.cid, log_n, ck, inf, circ, bl, ch, threads=4)
        same = lambda a, b: a[1] == b[1] and np.array_equal(a[0], b[0])
        ok = True
        for got in results:
            for key in ("wires_poly_comms", "split_quot_poly_comms"):
                ok &= len(got[key]) == 5 and all(same(g, x) for g, x in zip(got[key], want[key]))
            for key in ("prod_perm_poly_comm", "opening_proof", "shifted_opening_proof"):
                ok &= same(got[key], want[key])
            for key in ("wires_evals", "wire_sigma_evals"):
                ok &= bool(np.array_equal(np.stack(got[key]), np.stack(want[key])))
            ok &= bool(np.array_equal(got["perm_next_eval"], want["perm_next_eval"]))
            for key in ("quot_poly", "lin_poly", "batch_poly"):
                ok &= bool(np.array_equal(got["_debug"][key], want[key]))
        return bool(ok), dict(log_n=log_n, G=G, sharded=sharded, seed=seed)

    def op_compact_rows_fft(self):
        """plonk_fft1_dev_compact: the distributed forward transform of a ZERO-PADDED vector from the leading coefficients of every decimated row
        (dispatcher2.rs:746-766), random domain, rank count, polynomial length (from a handful of coefficients to nearly dense), plain / coset."""
        from distributed_plonk_amd.dispatcher import Dispatcher, make_fft_workloads, split_rc
        from distributed_plonk_amd.worker import PlonkWorker
        log_n = int(self.rs.randint(2, min(self.max_log, 13) + 1))
        N = 1 << log_n
        r, c = split_rc(N)
        S = int(self.rs.choice([1, 2, 4, 8]))
        while r % S or c % S:
            S //= 2
        pick = self.rs.rand()
        if pick < 0.4:
            length = max(1, N // 8 + int(self.rs.randint(0, 4)))                # the prover's n, n + 2, n + 3 coefficients on the 8n domain
        elif pick < 0.7:
            length = int(self.rs.randint(1, N + 1))
        else:                                                                    # around the class boundaries N / 2^k
            length = max(1, min(N, (N >> int(self.rs.randint(0, min(log_n, 5) + 1))) + int(self.rs.randint(-3, 4))))
        row_len = min(c, (length + r - 1) // r)
        coset = bool(self.rs.randint(0, 2))
        coeffs = self.fr(length)
        v = np.zeros((N, 4), dtype=np.uint64)
        v[:length] = coeffs
        t = np.ascontiguousarray(v.reshape(c, r, 4).transpose(1, 0, 2))          # t[b][a] = v[a*r + b]
        ws = [PlonkWorker(me=i, device=0, curve=self.curve) for i in range(S)]
        bufs = []
        try:
            d = Dispatcher(ws)
            d.init(None, N, 0)
            wl = make_fft_workloads(N, S)
            for i, x in enumerate(ws):
                x.fft_init(77, wl, False, False, coset)
                rows = np.ascontiguousarray(t[wl[i].row_start:wl[i].row_end, :row_len])
                bufs.append(x.alloc(max(rows.nbytes, 32)).upload(rows))
                x.fft1_dev_compact(77, bufs[-1].ptr, row_len)
            d._fft2_prepare_all(77)
            u = np.empty((c, r, 4), dtype=np.uint64)
            for i, x in enumerate(ws):
                u[wl[i].col_start:wl[i].col_end] = x.fft2(77, r)
            got = np.ascontiguousarray(u.transpose(1, 0, 2)).reshape(-1, 4)
        finally:
            for b in bufs:
                b.free()
            for x in ws:
                x.close()
        return np.array_equal(got, self.O.ntt(self.cid, v, False, coset, threads=4)), dict(log_n=log_n, S=S, length=length, coset=coset)

    def op_quotient(self):
        """dispatcher2.rs:362-504: the quotient's coset evaluations — every kernel formulation behind `quotient_fuse`, whole domain or one
        coset class (class_stride G | 8: the vectors hold the points class_offset + G*k), sparse / extreme inputs."""
        log_n = int(self.rs.randint(1, max(2, min(self.max_log - 3, 8) + 1)))
        n, m = 1 << log_n, 8 << log_n
        self.w.init(None, n, m)
        self.n_bases = 0                                              # the SRS is gone: the next MSM re-installs one
        vecs = self.fr(25 * m).reshape(25, m, 4)
        pick = self.rs.rand()
        if pick < 0.3:                                                # real selector vectors are sparse
            vecs[0:13, ::int(self.rs.randint(2, 6))] = 0
        elif pick < 0.45:                                             # the largest lazy sums the bound bookkeeping allows
            vecs[:, : m // 2] = self.f.to_limbs(self.f.p - 1)
        ch = self.fr(8)
        variant = int(self.rs.randint(0, 8))
        G = int(self.rs.choice([1, 1, 2, 4, 8]))
        off = int(self.rs.randint(0, G))
        want = self.O.quotient_evals(self.cid, log_n, vecs[0:13], vecs[13:18], vecs[18:23], vecs[23], vecs[24], ch[0], ch[1], ch[2], ch[3:8], threads=4)
        mine = np.ascontiguousarray(vecs[:, off::G]) if G > 1 else vecs
        mL = m // G
        buf, out = self.up(mine), self.w.alloc(mL * 32)
        ptr = [buf.ptr + j * mL * 32 for j in range(25)]
        try:
            self.w.set_option("quotient_fuse", variant)
            self.w.quotient_evals_dev(ptr[0:13], ptr[13:18], ptr[18:23], ptr[23], ptr[24], ch[0], ch[1], ch[2], ch[3:8], out.ptr, class_stride=G, class_offset=off)
            got = out.download((mL, 4))
        finally:
            self.w.set_option("quotient_fuse", 6)             # back to the shipped default
            buf.free(); out.free()
        return np.array_equal(got, want[off::G] if G > 1 else want), dict(log_n=log_n, variant=variant, G=G, off=off)

    OPS = ["ntt", "coset_eval_interp", "msm", "commit_many", "poly", "lincomb", "perm_product", "transpose", "distributed_fft", "quotient", "compact_rows_fft", "round1", "prove_verify", "class_prove", "msm_table",
           "perm_product_ranges", "class_ifft", "init_refuses_bad_srs", "trim"]

    def close(self):
        self.w.close()


def run(seconds, seed, curves, max_log, only=None, max_ops=None, verbose=True):
    t_end = time.time() + seconds
    fz = [Fuzz(c, cid, seed + 1000 * cid, max_log) for c, cid in curves]
    counts = {}
    it = 0
    try:
        while time.time() < t_end and (max_ops is None or it < max_ops):
            f = fz[it % len(fz)]
            name = only or Fuzz.OPS[int(f.rs.randint(0, len(Fuzz.OPS)))]
            ok, info = getattr(f, "op_" + name)()
            counts[name] = counts.get(name, 0) + 1
            it += 1
            if not ok:
                print(f"MISMATCH curve={f.curve} op={name} iteration={it} seed={seed} params={info}", flush=True)
                return 1, counts
    finally:
        for f in fz:
            f.close()
    if verbose:
        print(f"fuzz ok: {it} operations, no mismatch; per op {counts}", flush=True)
    return 0, counts


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--curve", default="both", choices=["bn254", "bls12_381", "both"])
    ap.add_argument("--max-log", type=int, default=13)
    ap.add_argument("--only", default=None, choices=Fuzz.OPS)
    ap.add_argument("--max-ops", type=int, default=None)
    a = ap.parse_args()
    cs = [("bn254", 0), ("bls12_381", 1)]
    if a.curve != "both":
        cs = [c for c in cs if c[0] == a.curve]
    raise SystemExit(run(a.seconds, a.seed, cs, a.max_log, a.only, a.max_ops)[0])
