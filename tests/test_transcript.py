"""Known-answer tests for the host-side transcript (the reference holds no vectors for it):
Keccak-f[1600] against hashlib through a sponge built on it, STROBE-128 + Merlin framing against the Merlin project's
published test vector, serialization against the curve generators' known encodings."""
import hashlib

import numpy as np

from distributed_plonk_amd import fr as FR
from distributed_plonk_amd import transcript as T


def _sponge(data: bytes, rate: int, suffix: int, outlen: int) -> bytes:
    st = bytearray(200)
    padded = bytearray(data) + bytes([suffix])
    padded += bytes((-len(padded)) % rate)
    padded[-1] |= 0x80
    for off in range(0, len(padded), rate):
        for i in range(rate):
            st[i] ^= padded[off + i]
        T.keccak_f1600(st)
    out = b""
    while len(out) < outlen:
        out += bytes(st[:rate])
        T.keccak_f1600(st)
    return out[:outlen]


def test_keccak_permutation_against_hashlib():
    for msg in (b"", b"abc", bytes(range(200)) * 3):
        assert _sponge(msg, 136, 0x06, 32) == hashlib.sha3_256(msg).digest()
        assert _sponge(msg, 168, 0x1F, 400) == hashlib.shake_128(msg).digest(400)


def test_native_keccak_equals_the_python_statement_and_is_what_the_transcript_runs():
    """libplonk_hip.so's host-side plonk_keccak_f1600 (what STROBE runs when the library is loadable: ~25 permutations per proof, 7-15 ms in the
    interpreter) against the pure-Python permutation that the hashlib test above pins, on random states; and Merlin's published vector both ways."""
    import ctypes as C
    from distributed_plonk_amd import _ffi
    fn = _ffi.lib().plonk_keccak_f1600
    rs = np.random.RandomState(7)
    for _ in range(20):
        a = bytearray(rs.randint(0, 256, 200).astype(np.uint8).tobytes())
        b = bytearray(a)
        T.keccak_f1600(a)
        assert fn((C.c_uint8 * 200).from_buffer(b)) == 0 and a == b
    assert fn(None) != 0
    want = "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    for native in (None, False):                    # None: look the library up (native); False: the interpreter's permutation
        T._native = native
        t = T.MerlinTranscript(b"test protocol")
        t.append_message(b"some label", b"some data")
        assert t.challenge_bytes(b"challenge", 32).hex() == want
        assert bool(T._native) == (native is None)
    T._native = None


def test_merlin_published_vector():
    """merlin's `equivalence_simple` inputs; the challenge value is the one the Merlin ports (Go, Python, JS) pin."""
    t = T.MerlinTranscript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_merlin_long_message_crosses_the_rate():
    """absorb/squeeze across the 166-byte STROBE block; determinism and sensitivity."""
    a, b = T.MerlinTranscript(b"x"), T.MerlinTranscript(b"x")
    a.append_message(b"big", bytes(1000))
    b.append_message(b"big", bytes(999) + b"\x01")
    ca, cb = a.challenge_bytes(b"c", 400), b.challenge_bytes(b"c", 400)
    assert len(ca) == 400 and ca != cb
    a2 = T.MerlinTranscript(b"x")
    a2.append_message(b"big", bytes(1000))
    assert a2.challenge_bytes(b"c", 400) == ca


def _mont(limbs64, x, p):
    v = x * pow(2, 64 * limbs64, p) % p
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(limbs64)]


def test_serialization_layouts():
    # Fr: canonical little-endian
    f = FR.FIELDS["bn254"]
    assert T.serialize_fr("bn254", f.to_limbs(0x0102)) == bytes([2, 1]) + bytes(30)
    # BN254 generator (1, 2): y = 2 < q - 2  -> no flag
    q = T.FQ_MODULI["bn254"]
    g = np.array(_mont(4, 1, q) + _mont(4, 2, q), dtype=np.uint64)
    assert T.serialize_g1("bn254", g, False) == (1).to_bytes(32, "little")
    neg = np.array(_mont(4, 1, q) + _mont(4, q - 2, q), dtype=np.uint64)
    enc = T.serialize_g1("bn254", neg, False)
    assert enc[:-1] == (1).to_bytes(32, "little")[:-1] and enc[-1] == 0x80
    inf = T.serialize_g1("bn254", g, True)
    assert inf == bytes(31) + b"\x40"
    # BLS12-381: 48 bytes, flags in the top bits of byte 47
    q = T.FQ_MODULI["bls12_381"]
    gx = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
    gy = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1
    g = np.array(_mont(6, gx, q) + _mont(6, gy, q), dtype=np.uint64)
    enc = T.serialize_g1("bls12_381", g, False)
    assert len(enc) == 48 and int.from_bytes(enc, "little") & ((1 << 382) - 1) == gx
    assert (enc[-1] >> 7) == (1 if gy > q - gy else 0)


def test_plonk_transcript_challenges_are_reduced_and_chained():
    t = T.PlonkTranscript("bn254")
    f = FR.FIELDS["bn254"]
    one = f.to_limbs(1)
    pt = (np.array(_mont(4, 1, T.FQ_MODULI["bn254"]) + _mont(4, 2, T.FQ_MODULI["bn254"]), dtype=np.uint64), False)
    t.append_vk_and_pub_input(8, 2, [one] * 5, [pt] * 13, [pt] * 5, [one, one])
    t.append_commitments(b"witness_poly_comms", [pt] * 5)
    beta, gamma = t.get_and_append_challenge(b"beta"), t.get_and_append_challenge(b"gamma")
    assert f.from_limbs(beta) < f.p and not np.array_equal(beta, gamma)
    t2 = T.PlonkTranscript("bn254")
    t2.append_vk_and_pub_input(8, 2, [one] * 5, [pt] * 13, [pt] * 5, [one, one])
    t2.append_commitments(b"witness_poly_comms", [pt] * 5)
    assert np.array_equal(t2.get_and_append_challenge(b"beta"), beta)


def test_serialize_proof_layout():
    """13 compressed points + 10 field elements + four u64 length prefixes (BN254: 416 + 320 + 32 bytes)."""
    f = FR.FIELDS["bn254"]
    q = T.FQ_MODULI["bn254"]
    pt = (np.array(_mont(4, 1, q) + _mont(4, 2, q), dtype=np.uint64), False)
    inf = (np.zeros(8, dtype=np.uint64), True)
    proof = dict(wires_poly_comms=[pt] * 5, prod_perm_poly_comm=inf, split_quot_poly_comms=[pt] * 5, opening_proof=pt, shifted_opening_proof=pt,
                 wires_evals=[f.to_limbs(i + 1) for i in range(5)], wire_sigma_evals=[f.to_limbs(9)] * 4, perm_next_eval=f.to_limbs(7))
    b = T.serialize_proof("bn254", proof)
    assert len(b) == 13 * 32 + 10 * 32 + 4 * 8
    assert b[:8] == (5).to_bytes(8, "little") and b[8:40] == (1).to_bytes(32, "little")
    assert b[8 + 5 * 32:8 + 6 * 32] == bytes(31) + b"\x40"                   # the infinity commitment
    assert b[-32:] == (7).to_bytes(32, "little")
