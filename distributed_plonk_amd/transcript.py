"""Fiat-Shamir transcript and canonical serialization of the reference prover, host side.

What it mirrors: `FakeStandardTranscript` (/root/reference/src/dispatcher2.rs:44-154), a wrapper of `merlin::Transcript`
(merlin 3.0.0, Cargo.toml:40: STROBE-128 over Keccak-f[1600]; an un-vendored third-party crate), and
`jf_utils::to_bytes!` = ark-serialize 0.3 `CanonicalSerialize::serialize` (compressed):
  * Fr            32 bytes little-endian of the canonical (non-Montgomery) integer
  * G1Affine      the x coordinate little-endian (32 B BN254 / 48 B BLS12-381) with SWFlags in the two top bits of the
                  last byte: bit 7 = "y is the larger of {y, -y}", bit 6 = point at infinity (x = 0 then)
  * usize         8 bytes little-endian (`to_le_bytes` on a 64-bit target)
Challenges: 64 transcript bytes reduced mod r (`from_le_bytes_mod_order`, :150).

Tiny compute (~25 Keccak permutations per proof): plain Python, with the permutation itself through the library's host-side
`plonk_keccak_f1600` when it is loadable (the interpreter needs 0.3-0.7 ms per permutation: 5 % of an 8-rank proof).  Pinned by known answers in
tests/test_transcript.py: the permutation against hashlib's SHA3-256/SHAKE128 through a sponge built on it, STROBE +
Merlin framing against the Merlin project's published "test protocol" vector.  The reference itself holds no vectors.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np

from . import fr as _fr

# --------------------------------------------------------------------------------------------- Keccak-f[1600]
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]   # [x][y]
_M64 = (1 << 64) - 1


def _rol(v: int, r: int) -> int:
    r %= 64
    return ((v << r) | (v >> (64 - r))) & _M64 if r else v


def keccak_f1600(state: bytearray) -> None:
    """In-place permutation of a 200-byte state (lane (x, y) at byte offset 8*(x + 5*y), little-endian)."""
    a = [[int.from_bytes(state[8 * (x + 5 * y):8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y] & _M64) for y in range(5)] for x in range(5)]
        a[0][0] ^= _RC[rnd]
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y):8 * (x + 5 * y) + 8] = a[x][y].to_bytes(8, "little")


_keccak_py = keccak_f1600          # the pure-Python statement above: pinned by tests/test_transcript.py, and what runs when no library is loadable
_native = None                      # None: not looked up yet; False: unavailable


def _keccak_fast(state: bytearray) -> None:
    """The same permutation through libplonk_hip.so's host-side `plonk_keccak_f1600` when the library is loadable (it is wherever a prover runs):
    a proof's transcript is ~25 permutations — 7-15 ms per proof in the interpreter (5 % of an 8-rank proof), microseconds natively.
    tests/test_transcript.py checks the two against each other and against hashlib."""
    global _native
    if _native is None:
        try:
            import ctypes as C
            from . import _ffi
            fn = _ffi.lib().plonk_keccak_f1600
            _native = (fn, C.c_uint8 * 200)
        except Exception:       # noqa: BLE001 - no library (a transcript-only use on a machine without the build): the interpreter does it
            _native = False
    if _native:
        fn, arr_t = _native
        if fn(arr_t.from_buffer(state)) != 0:
            raise RuntimeError("plonk_keccak_f1600 failed")
    else:
        _keccak_py(state)


# --------------------------------------------------------------------------------------------- STROBE-128 (the subset Merlin uses)
_STROBE_R = 166
_FLAG_I, _FLAG_A, _FLAG_C, _FLAG_T, _FLAG_M, _FLAG_K = 1, 2, 4, 8, 16, 32


class Strobe128:
    def __init__(self, protocol_label: bytes):
        st = bytearray(200)
        st[0:6] = bytes([1, _STROBE_R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        _keccak_fast(st)
        self.state, self.pos, self.pos_begin, self.cur_flags = st, 0, 0, 0
        self.meta_ad(protocol_label, False)

    def meta_ad(self, data: bytes, more: bool):
        self._begin_op(_FLAG_M | _FLAG_A, more)
        self._absorb(data)

    def ad(self, data: bytes, more: bool):
        self._begin_op(_FLAG_A, more)
        self._absorb(data)

    def prf(self, n: int, more: bool = False) -> bytes:
        self._begin_op(_FLAG_I | _FLAG_A | _FLAG_C, more)
        return self._squeeze(n)

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[_STROBE_R + 1] ^= 0x80
        _keccak_fast(self.state)
        self.pos = self.pos_begin = 0

    def _absorb(self, data: bytes):
        for byte in data:
            self.state[self.pos] ^= byte
            self.pos += 1
            if self.pos == _STROBE_R:
                self._run_f()

    def _squeeze(self, n: int) -> bytes:
        out = bytearray(n)
        for i in range(n):
            out[i] = self.state[self.pos]
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == _STROBE_R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags: int, more: bool):
        if more:
            assert self.cur_flags == flags
            return
        assert not flags & _FLAG_T
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if flags & (_FLAG_C | _FLAG_K) and self.pos != 0:
            self._run_f()


class MerlinTranscript:
    """merlin::Transcript — new / append_message / challenge_bytes."""

    def __init__(self, label: bytes):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label: bytes, message: bytes):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(len(message).to_bytes(4, "little"), True)
        self.strobe.ad(message, False)

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(n.to_bytes(4, "little"), True)
        return self.strobe.prf(n, False)


# --------------------------------------------------------------------------------------------- ark-serialize 0.3 (compressed)
FQ_MODULI = {
    "bn254": 21888242871839275222246405745257275088696311157297823662689037894645226208583,
    "bls12_381": 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
}


def _limbs_to_int(l) -> int:
    return sum(int(w) << (64 * i) for i, w in enumerate(np.asarray(l, dtype=np.uint64).reshape(-1)))


def serialize_fr(curve: str, limbs) -> bytes:
    """`to_bytes!(&Fr)`: canonical integer, 32 bytes little-endian (input: Montgomery limbs)."""
    return _fr.FIELDS[curve].from_limbs(limbs).to_bytes(32, "little")


def serialize_g1(curve: str, xy, is_inf: bool) -> bytes:
    """`to_bytes!(&Commitment)` = GroupAffine::serialize (compressed).  xy: x || y Montgomery limbs (2Q u64)."""
    q = FQ_MODULI[curve]
    nbytes = 32 if curve == "bn254" else 48
    if is_inf:
        out = bytearray(nbytes)
        out[-1] |= 1 << 6
        return bytes(out)
    xy = np.asarray(xy, dtype=np.uint64).reshape(-1)
    Q = xy.shape[0] // 2
    r_inv = pow(pow(2, 64 * Q, q), -1, q)
    x = _limbs_to_int(xy[:Q]) * r_inv % q
    y = _limbs_to_int(xy[Q:]) * r_inv % q
    out = bytearray(x.to_bytes(nbytes, "little"))
    if y > (q - y) % q:                      # SWFlags::from_y_sign(self.y > -self.y)
        out[-1] |= 1 << 7
    return bytes(out)


def serialize_proof(curve: str, proof: dict) -> bytes:
    """`CanonicalSerialize::serialize` of jf-plonk 0.1.1's `Proof<E>` as built at dispatcher2.rs:699-710 — derive order of the
    struct fields [upstream: jellyfish turbo-plonk branch, Cargo.lock:801-803, not vendored]: wires_poly_comms: Vec<Commitment>,
    prod_perm_poly_comm, split_quot_poly_comms: Vec<Commitment>, opening_proof, shifted_opening_proof, poly_evals
    { wires_evals: Vec<Fr>, wire_sigma_evals: Vec<Fr>, perm_next_eval }.  A Vec is its length as u64 little-endian followed
    by the elements (ark-serialize 0.3).

    UNVERIFIED LAYOUT: the struct's definition lives in an un-vendored git dependency; the reference only shows the field names in
    construction order (dispatcher2.rs:699-710) and holds no serialized proof.  The byte layout of this function is therefore
    outside every parity claim of this repository until one reference-generated `Proof` pins it; what IS pinned is
    each element's encoding (serialize_fr / serialize_g1, against the curve generators) and everything the transcript absorbs."""
    def vec(items, enc):
        return len(items).to_bytes(8, "little") + b"".join(enc(x) for x in items)

    pt = lambda c: serialize_g1(curve, *c)
    fe = lambda x: serialize_fr(curve, x)
    return (vec(proof["wires_poly_comms"], pt) + pt(proof["prod_perm_poly_comm"]) + vec(proof["split_quot_poly_comms"], pt)
            + pt(proof["opening_proof"]) + pt(proof["shifted_opening_proof"])
            + vec(proof["wires_evals"], fe) + vec(proof["wire_sigma_evals"], fe) + fe(proof["perm_next_eval"]))


# --------------------------------------------------------------------------------------------- the PLONK transcript
class PlonkTranscript:
    """dispatcher2.rs:44-154 (`FakeStandardTranscript`), method for method.  Points are (xy limbs, is_infinity) pairs as
    returned by PlonkWorker.g1_to_affine; field elements are Montgomery limbs."""

    def __init__(self, curve: str, label: bytes = b"PlonkProof"):          # :238
        self.curve = curve
        self.f = _fr.FIELDS[curve]
        self.t = MerlinTranscript(label)

    def append_vk_and_pub_input(self, domain_size: int, num_inputs: int, k: Sequence, selector_comms: Sequence, sigma_comms: Sequence,
                                pub_input: Sequence):                       # :56-91
        self.t.append_message(b"field size in bits", self.f.p.bit_length().to_bytes(8, "little"))
        self.t.append_message(b"domain size", int(domain_size).to_bytes(8, "little"))
        self.t.append_message(b"input size", int(num_inputs).to_bytes(8, "little"))
        for ki in k:
            self.t.append_message(b"wire subsets separators", serialize_fr(self.curve, ki))
        for c in selector_comms:
            self.t.append_message(b"selector commitments", serialize_g1(self.curve, *c))
        for c in sigma_comms:
            self.t.append_message(b"sigma commitments", serialize_g1(self.curve, *c))
        for x in pub_input:
            self.t.append_message(b"public input", serialize_fr(self.curve, x))

    def append_commitments(self, label: bytes, comms: Sequence):           # :94-107
        for c in comms:
            self.t.append_message(label, serialize_g1(self.curve, *c))

    def append_commitment(self, label: bytes, comm):                        # :110-121
        self.t.append_message(label, serialize_g1(self.curve, *comm))

    def append_proof_evaluations(self, wires_evals, wire_sigma_evals, perm_next_eval):   # :124-138
        for e in wires_evals:
            self.t.append_message(b"wire_evals", serialize_fr(self.curve, e))
        for e in wire_sigma_evals:
            self.t.append_message(b"wire_sigma_evals", serialize_fr(self.curve, e))
        self.t.append_message(b"perm_next_eval", serialize_fr(self.curve, perm_next_eval))

    def get_and_append_challenge(self, label: bytes) -> np.ndarray:         # :142-153 -> Montgomery limbs (4,)
        buf = self.t.challenge_bytes(label, 64)
        c = int.from_bytes(buf, "little") % self.f.p
        self.t.append_message(label, c.to_bytes(32, "little"))
        return self.f.to_limbs(c)
