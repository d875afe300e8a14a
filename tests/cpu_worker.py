"""A CPU stand-in for distributed_plonk_amd.worker.PlonkWorker, for tests only: the same method surface, "device memory" is a
numpy arena, every operation is delegated to the oracle.  It lets the multi-process orchestration of the product (SPMD code,
torch.distributed collectives) run under `gloo` on a box without a GPU — the product itself has no CPU path and never imports
this file."""
import ctypes as C

import numpy as np

from oracle import oracle as O
from oracle import prover_ref as P

_BASE = 1 << 20          # "device pointers" are arena offsets + _BASE (never 0)


class _Lib:
    """The two raw C entry points host code calls directly (prover._download)."""

    def __init__(self, worker):
        self._w = worker

    def plonk_memcpy_d2h(self, ctx, dst, src, nbytes):
        dst = dst.value if hasattr(dst, "value") else int(dst)
        C.memmove(dst, self._w._addr(src), nbytes)
        return 0


class CpuBuffer:
    def __init__(self, worker, nbytes):
        self.worker, self.nbytes = worker, nbytes
        self.ptr = worker._bump(nbytes)

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        self.worker.write_bytes(self.ptr, a)
        return self

    def download(self, shape, dtype=np.uint64, byte_offset=0):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return self.worker.read_bytes(self.ptr + byte_offset, n).view(dtype).reshape(shape).copy()

    def free(self):
        self.ptr = None


class _Arena:
    def __init__(self, nbytes):
        self.mem = np.zeros(nbytes // 8, dtype=np.uint64)
        self.top = 0


class CpuWorker:
    def __init__(self, curve="bn254", arena_bytes=64 << 20, me=0, share=None):
        """share: another CpuWorker whose "device memory" this one addresses too (several contexts on one GPU)."""
        self.curve_name, self.me = curve, me
        self.curve = O.CURVE_IDS[curve]
        self.q64 = O.FQ_LIMBS[self.curve]
        self.f = P.CURVE_OBJ[self.curve].fr
        self._a = share._a if share is not None else _Arena(arena_bytes)
        self.lib = _Lib(self)
        self.ctx = None
        self.bases = self.inf = None
        self._tasks = {}

    @property
    def arena(self):
        return self._a.mem

    # ---- memory
    def _bump(self, nbytes):
        nbytes = (max(nbytes, 8) + 255) & ~255
        assert self._a.top + nbytes <= self.arena.nbytes, "arena exhausted"
        p = self._a.top + _BASE
        self._a.top += nbytes
        return p

    def _addr(self, ptr):
        return self.arena.ctypes.data + (ptr - _BASE)

    def _fr(self, ptr, count):
        off = (ptr - _BASE) // 8
        return self.arena[off:off + 4 * count].reshape(count, 4)

    def alloc(self, nbytes):
        return CpuBuffer(self, nbytes)

    def read_bytes(self, src, nbytes):
        off = (src - _BASE) // 8
        return self.arena[off:off + nbytes // 8].view(np.int64).copy()

    def write_bytes(self, dst, arr):
        a = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
        C.memmove(self._addr(dst), a.ctypes.data, a.nbytes)

    def memcpy_d2d(self, dst, src, nbytes):
        C.memmove(self._addr(dst), self._addr(src), nbytes)

    def memset_dev(self, dst, byte, nbytes):
        C.memset(self._addr(dst), byte, nbytes)

    def sync(self):
        pass

    def close(self):
        pass

    # ---- field helpers (vectorised through the oracle)
    def _op(self, op, a, b=None):
        return O.field_op(self.curve, 0, op, np.ascontiguousarray(a), None if b is None else np.ascontiguousarray(b))

    def _const(self, x, count):
        return np.broadcast_to(P.fr_to_limbs(self.f, x), (count, 4)).copy()

    def _powers(self, base, count, first=1):
        """first * base^i, i < count (Montgomery limbs), by repeated doubling"""
        out = np.zeros((max(count, 1), 4), dtype=np.uint64)
        out[0] = P.fr_to_limbs(self.f, first)
        filled = 1
        while filled < count:
            k = min(filled, count - filled)
            out[filled:filled + k] = self._op("mul", out[:k], self._const(pow(base, filled, self.f.p), k))
            filled += k
        return out[:count]

    # ---- PlonkSlave surface used by the provers
    def init(self, bases, domain_size, quot_domain_size, layout=0):
        if bases is not None and len(bases):
            b = np.ascontiguousarray(bases, dtype=np.uint64)
            self.bases = b
            self.inf = np.array([0 if row.any() else 1 for row in b], dtype=np.uint8)
        self.n, self.m = domain_size, quot_domain_size

    # ---- PlonkSlave @1..@5 as the reference worker implements them (worker.rs:159-381), on the oracle's helpers
    def var_msm(self, workload, scalars):
        lo, hi = workload.start, workload.end
        sc = np.ascontiguousarray(scalars, dtype=np.uint64)[:hi - lo]
        if hi == lo:
            return O.msm(self.curve, self.bases[:1], np.zeros((1, 4), dtype=np.uint64), np.ones(1, dtype=np.uint8))
        return O.msm(self.curve, self.bases[lo:lo + len(sc)], sc, self.inf[lo:lo + len(sc)])

    def field_op(self, field, op, a, b=None):
        names = {0: "mul", 1: "add", 2: "sub", 3: "to_mont", 4: "from_mont", 5: "inv", 6: "sqr"}
        return O.field_op(self.curve, field, names[op], a, b)

    def fft_init(self, id, workloads, is_quot, is_inv, is_coset):
        N = self.m if is_quot else self.n
        me = workloads[self.me]
        self._tasks[id] = dict(wl=list(workloads), log_n=N.bit_length() - 1, inv=is_inv, coset=is_coset, rows={}, N=N,
                               r=1 << ((N.bit_length() - 1) >> 1), me=me)

    def fft1(self, id, i, v):
        t = self._tasks[id]
        t["rows"][i] = O.fft1_helper(self.curve, np.ascontiguousarray(v, dtype=np.uint64), i + t["me"].row_start, t["log_n"], t["inv"], t["coset"])

    def fft2_prepare(self, id, exchange=None):
        t = self._tasks[id]
        rows = np.stack([t["rows"][i] for i in range(t["me"].num_rows())])
        send = np.concatenate([O.exchange_pack(rows, w.col_start, w.col_end) for w in t["wl"]])      # worker.rs:327-330
        S = len(t["wl"])
        nbytes = send.nbytes // S
        d_send, d_recv = self._bump(send.nbytes), self._bump(send.nbytes)
        self.write_bytes(d_send, send)
        if exchange is None:
            assert S == 1
            self.memcpy_d2d(d_recv, d_send, send.nbytes)
        else:
            assert exchange(d_send, d_recv, nbytes, S, 0) in (0, None)
        t["recv"], t["nbytes"] = d_recv, nbytes

    def fft2(self, id, r):
        t = self._tasks.pop(id)
        S, me = len(t["wl"]), t["me"]
        cols = np.zeros((me.num_cols(), r, 4), dtype=np.uint64)
        for src in range(S):                                                                          # worker.rs:432-435
            blk = self.read_bytes(t["recv"] + src * t["nbytes"], t["nbytes"]).view(np.uint64).reshape(-1, 4)
            O.exchange_scatter(cols, t["wl"][src].row_start, blk)
        return np.stack([O.fft2_helper(self.curve, cols[i], i + me.col_start, t["log_n"], t["inv"], t["coset"]) for i in range(me.num_cols())])

    def ntt_dev(self, d_in, d_out, n, is_inv=False, is_coset=False):
        self._fr(d_out, n)[:] = O.ntt(self.curve, self._fr(d_in, n).copy(), is_inv, is_coset)

    def blind_dev(self, d_poly, n, blinders):
        bl = np.ascontiguousarray(blinders, dtype=np.uint64).reshape(-1, 4)
        k = bl.shape[0]
        v = self._fr(d_poly, n + k)
        v[:] = O.blind(self.curve, v.copy(), n, bl)

    def commit_range_dev(self, d_coeffs, start, count):
        if start < 0 or count < 0 or start > self.bases.shape[0]:       # the C ABI takes size_t and rejects start > n_bases
            raise ValueError(f"commit_range_dev: start {start}, count {count} against {self.bases.shape[0]} bases")
        count = min(count, self.bases.shape[0] - start)
        if count <= 0:
            return O.commit_polynomial(self.curve, self.bases[:1], np.zeros((1, 4), dtype=np.uint64), inf=np.ones(1, dtype=np.uint8))
        return O.commit_polynomial(self.curve, self.bases[start:start + count], self._fr(d_coeffs, count).copy(), inf=self.inf[start:start + count])

    def commit_dev(self, d_coeffs, n_coeffs):
        return self.commit_range_dev(d_coeffs, 0, n_coeffs)

    def commit_many_dev(self, items, start=0):
        """PlonkWorker.commit_many_dev: the same points as one commit_range_dev per polynomial."""
        return np.stack([self.commit_range_dev(ptr, start, count) for ptr, count in items]) if items else np.empty((0, 0), dtype=np.uint64)

    def g1_add(self, a, b):
        return O.jac_add(self.curve, a, b)

    def g1_to_affine(self, jac):
        xy, inf = O.jac_to_affine(self.curve, jac)
        return xy, bool(inf)

    def perm_product_dev(self, wires, d_id, d_idx, beta, gamma, n, d_out):
        w5 = np.stack([self._fr(p, n).copy() for p in wires])
        idx = self.read_bytes(d_idx, 5 * n * 8).view(np.uint64)
        self._fr(d_out, n)[:] = O.perm_product(self.curve, w5, self._fr(d_id, 5 * n).copy(), idx, beta, gamma)

    def perm_product_range_dev(self, wires, d_id, d_idx, beta, gamma, n, first, count, d_out):
        """out[t] = prod over gates [first, first + t) of the oracle's ratios: the slice of the product vector divided by its first value"""
        w5 = np.stack([self._fr(p, n).copy() for p in wires])
        idx = self.read_bytes(d_idx, 5 * n * 8).view(np.uint64)
        z = O.perm_product(self.curve, w5, self._fr(d_id, 5 * n).copy(), idx, beta, gamma)
        if first + count > n:
            raise ValueError("gate range")
        ext = z[first:first + count]
        inv0 = self._op("inv", z[first:first + 1])
        self._fr(d_out, count)[:] = self._op("mul", ext, np.tile(inv0, (count, 1)))

    def class_interleave_dev(self, d_in, classes, size, reverse, scale, d_out, in_stride=0):
        stride = in_stride or size
        src = (size - np.arange(size)) % size if reverse else np.arange(size)
        cols = [self._fr(d_in + s * stride * 32, size)[src] for s in range(classes)]
        v = np.stack(cols, axis=1).reshape(size * classes, 4)
        if scale is not None:
            v = self._op("mul", v, np.tile(np.ascontiguousarray(scale, dtype=np.uint64).reshape(1, 4), (size * classes, 1)))
        self._fr(d_out, size * classes)[:] = v

    def memcpy_d2d_async(self, dst, src, nbytes):
        self.memcpy_d2d(dst, src, nbytes)

    def poly_eval_dev(self, d_poly, length, point):
        return O.poly_eval(self.curve, self._fr(d_poly, length).copy(), point)

    def poly_lincomb_dev(self, polys, coeffs, d_out, out_len):
        acc = np.zeros((out_len, 4), dtype=np.uint64)
        for (ptr, ln), c in zip(polys, np.ascontiguousarray(coeffs, dtype=np.uint64)):
            ln = min(ln, out_len)
            if ln:
                acc[:ln] = self._op("add", acc[:ln], self._op("mul", self._fr(ptr, ln), np.broadcast_to(c, (ln, 4)).copy()))
        self._fr(d_out, out_len)[:] = acc

    def poly_div_linear_dev(self, d_poly, length, point, d_out):
        if length > 1:
            self._fr(d_out, length - 1)[:] = O.poly_div_linear(self.curve, self._fr(d_poly, length).copy(), point)

    def poly_degree_dev(self, d_poly, length):
        v = self._fr(d_poly, length)
        nz = np.flatnonzero(v.any(axis=1))
        return int(nz[-1]) if nz.size else -1

    def coset_eval_dev(self, d_poly, length, size, shift, d_out):
        h = P.fr_from_limbs(self.f, shift)
        a = self._fr(d_poly, length).copy()
        c = pow(h, size, self.f.p)
        acc = np.zeros((size, 4), dtype=np.uint64)
        for u in range((length + size - 1) // size):
            part = a[u * size:(u + 1) * size]
            acc[:part.shape[0]] = self._op("add", acc[:part.shape[0]], self._op("mul", part, self._const(pow(c, u, self.f.p), part.shape[0])))
        self._fr(d_out, size)[:] = O.ntt(self.curve, self._op("mul", acc, self._powers(h, size)), False, False)

    def coset_interp_dev(self, d_evals, size, shift, scale, i0, count, d_out):
        h_inv = pow(P.fr_from_limbs(self.f, shift), -1, self.f.p)
        E = O.ntt(self.curve, self._fr(d_evals, size).copy(), True, False)
        idx = (i0 + np.arange(count)) % size
        pw = self._powers(h_inv, count, first=P.fr_from_limbs(self.f, scale) * pow(h_inv, i0, self.f.p) % self.f.p)
        self._fr(d_out, count)[:] = self._op("mul", E[idx], pw)

    def quotient_evals_dev(self, selectors, sigmas, wires, perm, pub_input, alpha, beta, gamma, k, d_out, class_stride=1, class_offset=0):
        """dispatcher2.rs:435-504 on the points j = class_offset + class_stride * i, vectorised with the oracle's field ops."""
        f, p = self.f, self.f.p
        n, m, G, s = self.n, self.m, class_stride, class_offset
        mL, ratio = m // G, m // n
        ld = lambda ptr: self._fr(ptr, mL).copy()
        mul, add, sub = (lambda a, b: self._op("mul", a, b)), (lambda a, b: self._op("add", a, b)), (lambda a, b: self._op("sub", a, b))
        cst = lambda x: self._const(x, mL)
        w_m = P.fr_from_limbs(f, O.field_const(self.curve, 0, 4, m.bit_length() - 1))
        x = self._powers(pow(w_m, G, p), mL, first=f.generator * pow(w_m, s, p) % p)          # x_j
        W = [ld(q) for q in wires]
        S = [ld(q) for q in selectors]
        a, b, c, d, e = W
        ab, cd = mul(a, b), mul(c, d)
        p5 = lambda v: mul(mul(mul(v, v), mul(v, v)), v)
        gate = add(S[11], ld(pub_input))
        for t, v in ((0, a), (1, b), (2, c), (3, d), (4, ab), (5, cd), (6, p5(a)), (7, p5(b)), (8, p5(c)), (9, p5(d))):
            gate = add(gate, mul(S[t], v))
        gate = add(gate, mul(S[12], mul(mul(ab, cd), e)))
        gate = sub(gate, mul(S[10], e))
        z = ld(perm)
        zn = np.roll(z, -(ratio // G), axis=0)                                               # z(w x): point j + m/n, same class
        al, be, ga = (P.fr_from_limbs(f, v) for v in (alpha, beta, gamma))
        kk = [P.fr_from_limbs(f, v) for v in np.ascontiguousarray(k, dtype=np.uint64)]
        acc1, acc2 = z, zn
        for j in range(5):
            t = add(W[j], cst(ga))
            acc1 = mul(acc1, add(t, mul(x, cst(kk[j] * be % p))))
            acc2 = mul(acc2, add(t, mul(ld(sigmas[j]), cst(be))))
        permt = mul(cst(al), sub(acc1, acc2))
        one = cst(1)
        l1 = mul(mul(cst(al * al % p * pow(n, -1, p) % p), sub(z, one)), self._op("inv", sub(x, one)))
        zh_inv = np.zeros((mL, 4), dtype=np.uint64)
        for i in range(mL):
            j = s + G * i
            zh_inv[i] = P.fr_to_limbs(f, pow((pow(f.generator * pow(w_m, j % ratio, p) % p, n, p) - 1) % p, -1, p))
        self._fr(d_out, mL)[:] = add(mul(zh_inv, add(gate, permt)), l1)
