"""Byte-level codec between oracle big-ints and the C-ABI layouts of include/hekaton.h.

TEST INFRASTRUCTURE (oracle) — see params.py header.  Layouts: field element = LE limbs,
Montgomery form with R = 2^(64 N) (ark-ff `MontBackend` in-memory form); G1 affine = x||y;
G2 affine = x.c0||x.c1||y.c0||y.c1; infinity = all zero bytes.
"""

import numpy as np


class Codec:
    def __init__(self, cp):
        self.cp = cp
        self.fr_bytes = 8 * cp.fr_limbs64
        self.fq_bytes = 8 * cp.fq_limbs64
        self.g1_bytes = 2 * self.fq_bytes
        self.g2_bytes = 4 * self.fq_bytes

    # --- field elements ---------------------------------------------------------------------
    def fr_mont(self, x):
        return ((x % self.cp.r) * self.cp.fr_R % self.cp.r).to_bytes(self.fr_bytes, "little")

    def fr_canon(self, x):
        return (x % self.cp.r).to_bytes(self.fr_bytes, "little")

    def fq_mont(self, x):
        return ((x % self.cp.q) * self.cp.fq_R % self.cp.q).to_bytes(self.fq_bytes, "little")

    def fr_from_mont(self, b):
        return int.from_bytes(bytes(b), "little") * pow(self.cp.fr_R, -1, self.cp.r) % self.cp.r

    def fq_from_mont(self, b):
        return int.from_bytes(bytes(b), "little") * pow(self.cp.fq_R, -1, self.cp.q) % self.cp.q

    def fr_vec_mont(self, xs):
        return np.frombuffer(b"".join(self.fr_mont(x) for x in xs), dtype=np.uint8).copy()

    def fr_vec_canon(self, xs):
        return np.frombuffer(b"".join(self.fr_canon(x) for x in xs), dtype=np.uint8).copy()

    def fr_vec_from_mont(self, buf):
        b = bytes(buf)
        n = len(b) // self.fr_bytes
        return [self.fr_from_mont(b[i * self.fr_bytes:(i + 1) * self.fr_bytes]) for i in range(n)]

    # --- points -------------------------------------------------------------------------------
    def g1(self, P):
        if P is None:
            return bytes(self.g1_bytes)
        return self.fq_mont(P[0]) + self.fq_mont(P[1])

    def g2(self, P):
        if P is None:
            return bytes(self.g2_bytes)
        (x0, x1), (y0, y1) = P
        return self.fq_mont(x0) + self.fq_mont(x1) + self.fq_mont(y0) + self.fq_mont(y1)

    def g1_vec(self, pts):
        return np.frombuffer(b"".join(self.g1(p) for p in pts), dtype=np.uint8).copy()

    def g2_vec(self, pts):
        return np.frombuffer(b"".join(self.g2(p) for p in pts), dtype=np.uint8).copy()

    def g1_from(self, b):
        b = bytes(b)
        if not any(b):
            return None
        fb = self.fq_bytes
        return (self.fq_from_mont(b[:fb]), self.fq_from_mont(b[fb:2 * fb]))

    def g2_from(self, b):
        b = bytes(b)
        if not any(b):
            return None
        fb = self.fq_bytes
        c = [self.fq_from_mont(b[i * fb:(i + 1) * fb]) for i in range(4)]
        return ((c[0], c[1]), (c[2], c[3]))
