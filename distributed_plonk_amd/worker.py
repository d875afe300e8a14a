"""Host-side mirror of the reference worker (`impl plonk_slave::Server for PlonkImpl`,
/root/reference/src/worker.rs:125-439) on top of the C ABI: same method names, argument meaning and
call order as the Cap'n Proto interface (src/hello_world.capnp:16-24,49-50); the work runs in
libplonk_hip.so on one MI355X.  numpy arrays are uint64 limbs in the reference's raw layouts
(src/utils.rs:27-43).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Sequence

import numpy as np

from . import _ffi
from ._ffi import FftWorkload, MsmWorkload, PlonkError, check  # noqa: F401


def _u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _ptr(a: Optional[np.ndarray]):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class DeviceBuffer:
    """A piece of HBM owned through the C ABI (plonk_dev_alloc)."""

    def __init__(self, worker: "PlonkWorker", nbytes: int):
        self.worker, self.nbytes = worker, nbytes
        p = C.c_void_p()
        check(worker.lib.plonk_dev_alloc(worker.ctx, nbytes, C.byref(p)))
        self.ptr = p.value

    def offset(self, nbytes: int) -> int:
        return self.ptr + nbytes

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        check(self.worker.lib.plonk_memcpy_h2d(self.worker.ctx, self.ptr, _ptr(arr), arr.nbytes))
        return self

    def download(self, shape, dtype=np.uint64, byte_offset: int = 0) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        check(self.worker.lib.plonk_memcpy_d2h(self.worker.ctx, _ptr(out), self.ptr + byte_offset, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            check(self.worker.lib.plonk_dev_free(self.worker.ctx, self.ptr))
            self.ptr = None

    def __cuda_array_interface_for__(self, nbytes=None):
        return {"shape": ((nbytes or self.nbytes) // 8,), "typestr": "<i8", "data": (self.ptr, False), "version": 2}


class PlonkWorker:
    """One worker = one GPU context (reference `State`, worker.rs:42-59)."""

    def __init__(self, me: int = 0, device: int = 0, curve: str = "bn254"):
        self.lib = _ffi.lib()
        self.me = me
        self.curve_name = curve
        self.curve = _ffi.CURVES[curve]
        self.q64 = _ffi.FQ_LIMBS64[self.curve]
        ctx = C.c_void_p()
        check(self.lib.plonk_create(C.byref(ctx), device, self.curve))
        self.ctx = ctx
        self._tasks = {}
        self._keepalive = []

    def close(self):
        if self.ctx:
            self.lib.plonk_destroy(self.ctx)
            self.ctx = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ in-library RCCL transport
    @staticmethod
    def comm_unique_id() -> bytes:
        """ncclGetUniqueId (rank 0 calls it and ships the 128 bytes to the other ranks out of band)."""
        buf = C.create_string_buffer(_ffi.PLONK_COMM_ID_BYTES)
        check(_ffi.lib().plonk_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        """ncclCommInitRank on this context's GPU; collective over all `world` ranks.  Afterwards fft2_prepare(id) without a callback
        runs the all-to-all through RCCL inside the library."""
        assert len(unique_id) == _ffi.PLONK_COMM_ID_BYTES
        check(self.lib.plonk_comm_init(self.ctx, unique_id, rank, world))
        self.me = rank

    def comm_destroy(self):
        check(self.lib.plonk_comm_destroy(self.ctx))

    def comm_info(self):
        """-> (rank, world, rccl_version) as RCCL reports them."""
        r, w, v = C.c_int(), C.c_int(), C.c_int()
        check(self.lib.plonk_comm_info(self.ctx, C.byref(r), C.byref(w), C.byref(v)))
        return r.value, w.value, v.value

    def comm_alltoall_dev(self, d_send: int, d_recv: int, bytes_per_peer: int):
        check(self.lib.plonk_comm_alltoall_dev(self.ctx, d_send, d_recv, bytes_per_peer))

    def comm_allgather_dev(self, d_send: int, d_recv: int, nbytes: int):
        check(self.lib.plonk_comm_allgather_dev(self.ctx, d_send, d_recv, nbytes))

    def comm_allgather_host(self, arr: np.ndarray, world: int) -> np.ndarray:
        """Every rank's `arr` (same size everywhere) -> (world, *arr.shape)."""
        a = np.ascontiguousarray(arr)
        out = np.empty((world,) + a.shape, dtype=a.dtype)
        check(self.lib.plonk_comm_allgather_host(self.ctx, _ptr(a), a.nbytes, _ptr(out)))
        return out

    # ------------------------------------------------------------------ PlonkSlave @0
    def init(self, bases: Optional[np.ndarray], domain_size: int, quot_domain_size: int, layout=_ffi.PLONK_BASES_XY):
        """worker.rs:126-157.  bases: (n, 2*Q) x||y Montgomery limbs, or raw ark bytes for PLONK_BASES_ARK."""
        if bases is None or len(bases) == 0:
            check(self.lib.plonk_init(self.ctx, None, 0, layout, domain_size, quot_domain_size))
            return
        if layout == _ffi.PLONK_BASES_XY:
            b = _u64(bases)
            n = b.shape[0]
        else:
            b = np.ascontiguousarray(bases, dtype=np.uint8)
            n = b.size // (16 * self.q64 + 8)
        check(self.lib.plonk_init(self.ctx, _ptr(b), n, layout, domain_size, quot_domain_size))

    def init_dev(self, d_bases_ptr: int, n_bases: int, domain_size: int, quot_domain_size: int):
        check(self.lib.plonk_init_dev(self.ctx, d_bases_ptr, n_bases, domain_size, quot_domain_size))

    # ------------------------------------------------------------------ PlonkSlave @1
    def var_msm(self, workload: MsmWorkload, scalars: np.ndarray) -> np.ndarray:
        """worker.rs:159-185 -> raw Jacobian (3*Q u64)."""
        s = _u64(scalars)
        out = np.empty(3 * self.q64, dtype=np.uint64)
        check(self.lib.plonk_var_msm(self.ctx, C.byref(workload), _ptr(s), s.shape[0], _ptr(out)))
        return out

    def msm_dev(self, start: int, end: int, d_scalars_ptr: int) -> np.ndarray:
        out = np.empty(3 * self.q64, dtype=np.uint64)
        check(self.lib.plonk_msm_dev(self.ctx, start, end, d_scalars_ptr, _ptr(out)))
        return out

    def commit(self, coeffs_mont: np.ndarray) -> np.ndarray:
        """commit_polynomial, worker.rs:117-123."""
        c = _u64(coeffs_mont)
        out = np.empty(3 * self.q64, dtype=np.uint64)
        check(self.lib.plonk_commit(self.ctx, _ptr(c), c.shape[0], _ptr(out)))
        return out

    def commit_dev(self, d_coeffs_ptr: int, n_coeffs: int) -> np.ndarray:
        out = np.empty(3 * self.q64, dtype=np.uint64)
        check(self.lib.plonk_commit_dev(self.ctx, d_coeffs_ptr, n_coeffs, _ptr(out)))
        return out

    def commit_many_dev(self, items, start: int = 0) -> np.ndarray:
        """[(device pointer at coefficient `start`, count), ...] -> (k, 3*Q) Jacobian points: k commitments against bases
        [start, start + count) in one set of launches (plonk_commit_many_dev)."""
        k = len(items)
        out = np.empty((k, 3 * self.q64), dtype=np.uint64)
        if k == 0:
            return out
        ptrs = (C.c_void_p * k)(*[int(p) for p, _ in items])
        lens = (C.c_size_t * k)(*[int(n) for _, n in items])
        check(self.lib.plonk_commit_many_dev(self.ctx, k, ptrs, lens, start, _ptr(out)))
        return out

    def commit_range_dev(self, d_coeffs_ptr: int, start: int, count: int) -> np.ndarray:
        """One shard of commit_polynomial: coefficients [start, start+count) against bases [start, start+count)."""
        out = np.empty(3 * self.q64, dtype=np.uint64)
        check(self.lib.plonk_commit_range_dev(self.ctx, d_coeffs_ptr, start, count, _ptr(out)))
        return out

    # ------------------------------------------------------------------ PlonkSlave @2..@5
    def fft_init(self, id: int, workloads: Sequence[FftWorkload], is_quot: bool, is_inv: bool, is_coset: bool):
        arr = (FftWorkload * len(workloads))(*workloads)
        check(self.lib.plonk_fft_init(self.ctx, id, arr, len(workloads), self.me, int(is_quot), int(is_inv), int(is_coset)))
        self._tasks[id] = (workloads[self.me].num_rows(), workloads[self.me].num_cols(), is_quot)

    def fft1(self, id: int, i: int, v: np.ndarray):
        v = _u64(v)
        check(self.lib.plonk_fft1(self.ctx, id, i, _ptr(v), v.shape[0]))

    def fft1_dev(self, id: int, d_rows_ptr: int):
        """All local rows at once; the buffer is consumed (must stay alive until fft2_prepare returns, contents destroyed)."""
        check(self.lib.plonk_fft1_dev(self.ctx, id, d_rows_ptr))

    def fft1_dev_compact(self, id: int, d_rows_ptr: int, row_len: int):
        """Rows of a zero-padded vector: [num_rows][row_len] leading coefficients, the rest of every row zero by construction
        (forward transforms; the buffer is not modified)."""
        check(self.lib.plonk_fft1_dev_compact(self.ctx, id, d_rows_ptr, row_len))

    def fft2_prepare(self, id: int, exchange: Optional[Callable] = None):
        """exchange(send_ptr, recv_ptr, bytes_per_peer, n_ranks, stream_ptr) -> int (0 = ok)."""
        if exchange is None:
            cb = C.cast(None, _ffi.EXCHANGE_FN)
        elif isinstance(exchange, _ffi.EXCHANGE_FN):          # a plonk_exchange_fn of the library itself (plonk_exchange_standin): no Python in the call
            cb = exchange
        else:
            def _tramp(user, send, recv, nbytes, n_ranks, stream):
                try:
                    return int(exchange(send, recv, nbytes, n_ranks, stream) or 0)
                except Exception as e:  # never unwind through C
                    self._last_exchange_error = e
                    return 1
            cb = _ffi.EXCHANGE_FN(_tramp)
            self._keepalive.append(cb)
        try:
            check(self.lib.plonk_fft2_prepare(self.ctx, id, cb, None))
        finally:
            self._keepalive.clear()

    def fft2(self, id: int, r: int) -> np.ndarray:
        """-> (num_cols, r, 4): the reply of worker.rs:366-376."""
        _, ncols, _ = self._tasks.pop(id)
        out = np.empty((ncols, r, 4), dtype=np.uint64)
        check(self.lib.plonk_fft2(self.ctx, id, _ptr(out)))
        return out

    def fft2_dev(self, id: int, d_out_ptr: int, layout: int = 1):
        self._tasks.pop(id, None)
        check(self.lib.plonk_fft2_dev(self.ctx, id, d_out_ptr, layout))

    # ------------------------------------------------------------------ PlonkSlave @6
    def round1(self, evals: np.ndarray, blinders: np.ndarray) -> np.ndarray:
        e, b = _u64(evals), _u64(blinders)
        out = np.empty(3 * self.q64, dtype=np.uint64)
        check(self.lib.plonk_round1(self.ctx, _ptr(e), e.shape[0], _ptr(b), _ptr(out)))
        return out

    def get_wire(self, n_coeffs: int) -> np.ndarray:
        out = np.empty((n_coeffs, 4), dtype=np.uint64)
        check(self.lib.plonk_get_wire(self.ctx, _ptr(out), n_coeffs))
        return out

    # ------------------------------------------------------------------ operator boundary
    def ntt(self, v: np.ndarray, is_inv=False, is_coset=False) -> np.ndarray:
        """Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place on a host vector."""
        v = _u64(v).copy()
        check(self.lib.plonk_ntt(self.ctx, _ptr(v), v.shape[0], int(is_inv), int(is_coset)))
        return v

    def ntt_dev(self, d_in: int, d_out: int, n: int, is_inv=False, is_coset=False):
        check(self.lib.plonk_ntt_dev(self.ctx, d_in, d_out, n, int(is_inv), int(is_coset)))

    def transpose(self, v: np.ndarray, rows: int, cols: int) -> np.ndarray:
        v = _u64(v).copy()
        check(self.lib.plonk_transpose(self.ctx, _ptr(v), rows, cols))
        return v

    def transpose_dev(self, d_in: int, d_out: int, rows: int, cols: int):
        """[rows][cols] -> [cols][rows] of Fr in HBM (ip_transpose, transpose.rs:413; out of place)."""
        check(self.lib.plonk_transpose_dev(self.ctx, d_in, d_out, rows, cols))

    def g1_add(self, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        out = np.empty(3 * self.q64, dtype=np.uint64)
        check(self.lib.plonk_g1_add(self.curve, _ptr(_u64(a)), _ptr(_u64(b)), _ptr(out)))
        return out

    def g1_to_affine(self, jac: np.ndarray):
        out = np.zeros(2 * self.q64, dtype=np.uint64)
        inf = C.c_int(0)
        check(self.lib.plonk_g1_to_affine(self.curve, _ptr(_u64(jac)), _ptr(out), C.byref(inf)))
        return out, bool(inf.value)

    # ------------------------------------------------------------------ next row: quotient evaluations (dispatcher2.rs:362-504)
    def quotient_evals_dev(self, selectors, sigmas, wires, perm, pub_input, alpha, beta, gamma, k, d_out: int, class_stride: int = 1,
                           class_offset: int = 0):
        """selectors (13), sigmas (5), wires (5): device pointers to m coset evaluations each; perm / pub_input: device
        pointers; alpha, beta, gamma (4,) and k (5,4): Montgomery limbs on the host.  class_stride G > 1: every vector holds the
        m/G evaluations of coset class `class_offset` (points class_offset + G*k)."""
        q = _ffi.QuotientInputs()
        for j in range(13):
            q.selectors[j] = selectors[j]
        for j in range(5):
            q.sigmas[j] = sigmas[j]
            q.wires[j] = wires[j]
        q.perm, q.pub_input = perm, pub_input
        al, be, ga, kk = _u64(alpha), _u64(beta), _u64(gamma), _u64(k)
        if class_stride == 1:
            check(self.lib.plonk_quotient_evals_dev(self.ctx, C.byref(q), _ptr(al), _ptr(be), _ptr(ga), _ptr(kk), d_out))
        else:
            check(self.lib.plonk_quotient_evals_class_dev(self.ctx, C.byref(q), _ptr(al), _ptr(be), _ptr(ga), _ptr(kk), class_stride, class_offset, d_out))

    # ------------------------------------------------------------------ next rows: permutation product, round 4/5 polynomial ops
    def perm_product_dev(self, wires, d_id_perm: int, d_perm_idx: int, beta, gamma, n: int, d_out: int):
        """dispatcher2.rs:329-344.  wires: 5 device pointers to n wire values; d_id_perm: 5n Fr; d_perm_idx: 5n u64."""
        arr = (C.c_void_p * 5)(*[int(w) for w in wires])
        be, ga = _u64(beta), _u64(gamma)
        check(self.lib.plonk_perm_product_dev(self.ctx, C.byref(arr), d_id_perm, d_perm_idx, _ptr(be), _ptr(ga), n, d_out))

    def perm_product_range_dev(self, wires, d_id_perm: int, d_perm_idx: int, beta, gamma, n: int, first: int, count: int, d_out: int):
        """d_out[t] = the product of the ratios of gates [first, first + t), t < count: a worker's slice of the product vector up to the
        product of the gates before it (pointers are to the whole vectors)."""
        arr = (C.c_void_p * 5)(*[int(w) for w in wires])
        be, ga = _u64(beta), _u64(gamma)
        check(self.lib.plonk_perm_product_range_dev(self.ctx, C.byref(arr), d_id_perm, d_perm_idx, _ptr(be), _ptr(ga), n, first, count, d_out))

    def class_interleave_dev(self, d_in: int, classes: int, size: int, reverse: bool, scale, d_out: int, in_stride: int = 0):
        """d_out[t * classes + s] = scale * d_in[s * in_stride + (t, or (size - t) % size when reverse)]; scale None = 1; in_stride 0 = size."""
        sc = None if scale is None else _u64(scale)
        check(self.lib.plonk_class_interleave_dev(self.ctx, d_in, classes, size, in_stride, 1 if reverse else 0, None if sc is None else _ptr(sc), d_out))

    def poly_eval_dev(self, d_poly: int, length: int, point) -> np.ndarray:
        """DensePolynomial::evaluate (dispatcher2.rs:545-555) -> Fr Montgomery limbs (4,)."""
        out = np.empty(4, dtype=np.uint64)
        z = _u64(point)
        check(self.lib.plonk_poly_eval_dev(self.ctx, d_poly, length, _ptr(z), _ptr(out)))
        return out

    def poly_lincomb_dev(self, polys, coeffs, d_out: int, out_len: int):
        """d_out = sum_t coeffs[t] * polys[t]; polys: sequence of (device pointer, length); coeffs (k,4) (dispatcher2.rs:566-633)."""
        k = len(polys)
        ptrs = (C.c_void_p * k)(*[int(q[0]) for q in polys])
        lens = (C.c_size_t * k)(*[int(q[1]) for q in polys])
        cf = _u64(coeffs)
        assert cf.shape == (k, 4)
        check(self.lib.plonk_poly_lincomb_dev(self.ctx, k, ptrs, lens, _ptr(cf), d_out, out_len))

    def poly_div_linear_dev(self, d_poly: int, length: int, point, d_out: int):
        """quotient of poly / (X - point), remainder dropped (dispatcher2.rs:651-666): length-1 coefficients."""
        z = _u64(point)
        check(self.lib.plonk_poly_div_linear_dev(self.ctx, d_poly, length, _ptr(z), d_out))

    def coset_eval_dev(self, d_poly: int, length: int, size: int, shift, d_out: int):
        """d_out[k] = poly(shift * w_size^k), k < size (any shift; length <= 8*size folds back)."""
        h = _u64(shift)
        check(self.lib.plonk_coset_eval_dev(self.ctx, d_poly, length, size, _ptr(h), d_out))

    def coset_interp_dev(self, d_evals: int, size: int, shift, scale, i0: int, count: int, d_out: int):
        """d_out[t] = scale * shift^-(i0+t) * iNTT_size(evals)[(i0+t) mod size]; d_evals is destroyed."""
        h, sc = _u64(shift), _u64(scale)
        check(self.lib.plonk_coset_interp_dev(self.ctx, d_evals, size, _ptr(h), _ptr(sc), i0, count, d_out))

    def poly_degree_dev(self, d_poly: int, length: int) -> int:
        """DensePolynomial::degree() after trimming (-1 = zero polynomial), dispatcher2.rs:511-518."""
        d = C.c_int64(0)
        check(self.lib.plonk_poly_degree_dev(self.ctx, d_poly, length, C.byref(d)))
        return d.value

    def blind_dev(self, d_poly: int, n: int, blinders):
        """d_poly[0..n+k) += (sum b_i X^i)(X^n - 1) (dispatcher2.rs:311-312,347-348)."""
        b = _u64(blinders).reshape(-1, 4)
        check(self.lib.plonk_blind_dev(self.ctx, d_poly, n, _ptr(b), b.shape[0]))

    # ------------------------------------------------------------------ device memory, synthetic inputs, debug
    def trim(self):
        """plonk_trim: give back every cache this context can rebuild on demand (exchange buffers of finished FFT tasks, NTT factor planes,
        MSM workspace, scratch) — a worker's State outlives a circuit (worker.rs:42-59)."""
        check(self.lib.plonk_trim(self.ctx))

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def read_bytes(self, src: int, nbytes: int) -> np.ndarray:
        """device -> a fresh host array of int64 words (host-staged transports)."""
        out = np.empty(nbytes // 8, dtype=np.int64)
        check(self.lib.plonk_memcpy_d2h(self.ctx, _ptr(out), src, nbytes))
        return out

    def write_bytes(self, dst: int, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        check(self.lib.plonk_memcpy_h2d(self.ctx, dst, _ptr(a), a.nbytes))

    def memset_dev(self, dst: int, byte: int, nbytes: int):
        check(self.lib.plonk_memset_dev(self.ctx, dst, byte, nbytes))

    def memcpy_d2d(self, dst: int, src: int, nbytes: int):
        check(self.lib.plonk_memcpy_d2d(self.ctx, dst, src, nbytes))

    def memcpy_d2d_async(self, dst: int, src: int, nbytes: int):
        """ordered on the context's stream, not synchronised"""
        check(self.lib.plonk_memcpy_d2d_async(self.ctx, dst, src, nbytes))

    def synth_fr(self, seed: int, d_out: int, n: int):
        check(self.lib.plonk_synth_fr(self.ctx, seed, d_out, n))

    def synth_bases(self, seed: int, unique: int, n: int, d_out: int):
        check(self.lib.plonk_synth_bases(self.ctx, seed, unique, n, d_out))

    def synth_srs(self, tau: np.ndarray, n: int, d_out: int):
        """d_out[i] = tau^i * G (x || y), i < n: a commit key with a known trapdoor (tau: Fr limbs, Montgomery)."""
        check(self.lib.plonk_synth_srs(self.ctx, _ptr(_u64(tau)), n, d_out))

    def synth_circuit(self, seed: int, n: int, num_inputs: int, k: np.ndarray, d_wires: int, d_sel_evals: int, d_sigma_evals: int, d_id_perm: int,
                      d_perm_idx: int, d_pub_input: int):
        """A random satisfied TurboPlonk instance generated in HBM (include/plonk_hip.h: plonk_synth_circuit)."""
        check(self.lib.plonk_synth_circuit(self.ctx, seed, n, num_inputs, _ptr(_u64(k)), d_wires, d_sel_evals, d_sigma_evals, d_id_perm, d_perm_idx,
                                           d_pub_input))

    def field_op(self, field: int, op: int, a: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
        a = _u64(a)
        b = _u64(b) if b is not None else None
        out = np.empty_like(a)
        check(self.lib.plonk_debug_field_op(self.ctx, field, op, _ptr(a), _ptr(b), _ptr(out), a.shape[0]))
        return out

    def set_option(self, key: str, value: int):
        check(self.lib.plonk_set_option(self.ctx, key.encode(), value))

    def sync(self):
        check(self.lib.plonk_sync(self.ctx))

    def stream_ptr(self) -> int:
        return self.lib.plonk_stream(self.ctx)

    def last_kernel_ms(self) -> float:
        ms = C.c_double(0)
        check(self.lib.plonk_last_kernel_ms(self.ctx, C.byref(ms)))
        return ms.value

    def profile_enable(self, on: bool):
        check(self.lib.plonk_profile_enable(self.ctx, int(on)))

    def profile_reset(self):
        check(self.lib.plonk_profile_reset(self.ctx))

    def profile_get(self, name: str):
        ms, n = C.c_double(0), C.c_uint64(0)
        check(self.lib.plonk_profile_get(self.ctx, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value
