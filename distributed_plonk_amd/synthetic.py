"""A whole synthetic proving instance resident in HBM: a random SATISFIED TurboPlonk circuit (`plonk_synth_circuit`), its proving
key in coefficient form, and a commit key with a known trapdoor (`plonk_synth_srs`).

Stands in for what the reference's end-to-end test builds with jellyfish (`generate_circuit`, `universal_setup`, `preprocess`:
/root/reference/src/dispatcher2.rs:1214-1282): north_star asks for throughput on synthetic random circuits, and a satisfied
instance is what lets the prover run with its `WrongQuotientPolyDegree` check (dispatcher2.rs:511-518) ON at 2^24 gates and lets
the finished proof be handed to a verifier.  Nothing here touches the host except the 5 coset representatives k_i.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import fr as _fr
from .worker import PlonkWorker

NUM_WIRE_TYPES = 5
NUM_SELECTORS = 13


def wire_subset_separators(field: _fr.FrField, seed: int) -> np.ndarray:
    """vk.k: k_0 = 1 and four more coset representatives (jf-plonk derives them from a hash; any values whose cosets k_i * H are
    pairwise disjoint serve — random 62-bit values are, except with negligible probability).  -> (5, 4) Montgomery limbs."""
    rs = np.random.RandomState(seed & 0x7FFFFFFF)
    return field.vec_to_limbs([1] + [int(x) for x in rs.randint(2, 1 << 62, size=4)])


class SyntheticInstance:
    """Device buffers of one instance; `close()` frees them.  Attribute names follow Prover.load_key_dev / prove_dev."""

    def __init__(self, worker: PlonkWorker, log_n: int, seed: int = 1, num_inputs: int = 2, tau: Optional[int] = None,
                 init_worker: bool = True, helpers=()):
        """tau: the trapdoor as an integer (None = keep whatever commit key the worker already holds).  With init_worker the
        worker (and every context in `helpers`) is `init`-ed with the n + 3 powers padded with points at infinity to a multiple of
        32, exactly the key of dispatcher2.rs:206-208."""
        self.w = w = worker
        f = self.f = _fr.FIELDS[w.curve_name]
        n = self.n = 1 << log_n
        self.log_n, self.num_inputs, self.tau = log_n, num_inputs, tau
        self.k = wire_subset_separators(f, seed)
        self._bufs = []
        alloc = lambda n_fr: self._keep(w.alloc(max(n_fr, 1) * 32))
        self.d_wires = alloc(5 * n)
        d_sel_ev = alloc(NUM_SELECTORS * n)
        d_sig_ev = alloc(NUM_WIRE_TYPES * n)
        self.d_id = alloc(5 * n)
        self.d_idx = self._keep(w.alloc(5 * n * 8))
        self.d_pi = alloc(n)
        w.synth_circuit(seed, n, num_inputs, self.k, self.d_wires.ptr, d_sel_ev.ptr, d_sig_ev.ptr, self.d_id.ptr, self.d_idx.ptr, self.d_pi.ptr)
        self.d_sel_ev, self.d_sig_ev = d_sel_ev, d_sig_ev
        # proving key: selector and sigma polynomials in coefficient form (n-point iNTTs; d_in is consumed, so transform a copy)
        self.d_sel = alloc(NUM_SELECTORS * n)
        self.d_sig = alloc(NUM_WIRE_TYPES * n)
        tmp = w.alloc(n * 32)
        try:
            for src, dst, cnt in ((d_sel_ev, self.d_sel, NUM_SELECTORS), (d_sig_ev, self.d_sig, NUM_WIRE_TYPES)):
                for t in range(cnt):
                    w.memcpy_d2d(tmp.ptr, src.ptr + t * n * 32, n * 32)
                    w.ntt_dev(tmp.ptr, dst.ptr + t * n * 32, n, True, False)
        finally:
            tmp.free()
        self.wev = [self.d_wires.ptr + i * n * 32 for i in range(5)]
        self.sel_ptrs = [self.d_sel.ptr + t * n * 32 for t in range(NUM_SELECTORS)]
        self.sig_ptrs = [self.d_sig.ptr + t * n * 32 for t in range(NUM_WIRE_TYPES)]
        self.key_size = ((n + 3 + 31) >> 5) << 5
        self.d_ck = None
        if tau is not None:
            q = 64 if w.curve_name == "bn254" else 96
            self.d_ck = self._keep(w.alloc(self.key_size * q))
            w.memset_dev(self.d_ck.ptr, 0, self.key_size * q)                 # x = y = 0: the padding points at infinity
            w.synth_srs(f.to_limbs(tau), n + 3, self.d_ck.ptr)
            if init_worker:
                for ctx in (w, *helpers):
                    ctx.init_dev(self.d_ck.ptr, self.key_size, n, 8 * n)
        w.sync()

    def _keep(self, b):
        self._bufs.append(b)
        return b

    def public_inputs(self) -> np.ndarray:
        """`circuit.public_input()`: the num_inputs values (not padded) -> (num_inputs, 4)."""
        return self.d_pi.download((self.num_inputs, 4))

    def download(self) -> dict:
        """Everything as host arrays in the layout oracle/prover_ref.py uses (small sizes only)."""
        n = self.n
        return dict(wires=self.d_wires.download((5, n, 4)), selectors=self.d_sel.download((NUM_SELECTORS, n, 4)),
                    sigmas=self.d_sig.download((NUM_WIRE_TYPES, n, 4)), id_perm=self.d_id.download((5 * n, 4)),
                    perm_idx=self.d_idx.download((5 * n,)), pub_input=self.d_pi.download((n, 4)), k=self.k.copy(),
                    selector_evals=self.d_sel_ev.download((NUM_SELECTORS, n, 4)), sigma_evals=self.d_sig_ev.download((NUM_WIRE_TYPES, n, 4)))

    def close(self):
        for b in self._bufs:
            b.free()
        self._bufs = []
