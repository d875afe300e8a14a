"""Host-side scalar arithmetic in Fr for the handful of challenge-derived coefficients the prover computes between
device calls (dispatcher2.rs:558-646 works on single `Fr` values there too).  Python ints; limbs are the reference's
wire format: 4 x u64 little-endian, Montgomery with R = 2^256 (utils.rs:27-43)."""
from __future__ import annotations

import numpy as np


class FrField:
    def __init__(self, name: str, p: int, generator: int, two_adicity: int):
        self.name, self.p, self.generator, self.two_adicity = name, p, generator, two_adicity
        self.R = pow(2, 256, p)
        self.R_inv = pow(self.R, -1, p)

    def to_limbs(self, x: int) -> np.ndarray:
        """plain residue -> Montgomery limbs (4,) u64"""
        v = x % self.p * self.R % self.p
        return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)

    def from_limbs(self, l) -> int:
        v = sum(int(w) << (64 * i) for i, w in enumerate(np.asarray(l, dtype=np.uint64).reshape(4)))
        return v * self.R_inv % self.p

    def vec_to_limbs(self, xs) -> np.ndarray:
        return np.stack([self.to_limbs(x) for x in xs]) if len(xs) else np.zeros((0, 4), dtype=np.uint64)

    def root_of_unity(self, n: int) -> int:
        """ark-ff FftField::get_root_of_unity(n): the 2^s-th root g^((p-1)/2^s) squared down to order n."""
        log = n.bit_length() - 1
        if 1 << log != n or log > self.two_adicity:
            raise ValueError("DomainCreationError")
        w = pow(self.generator, (self.p - 1) >> self.two_adicity, self.p)
        for _ in range(self.two_adicity - log):
            w = w * w % self.p
        return w

    def inv(self, x: int) -> int:
        return pow(x, -1, self.p)


FIELDS = {
    "bn254": FrField("bn254", 21888242871839275222246405745257275088548364400416034343698204186575808495617, 5, 28),
    "bls12_381": FrField("bls12_381", 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001, 7, 32),
}
