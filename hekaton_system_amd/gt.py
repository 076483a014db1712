"""Host-side arithmetic in GT = Fq12 (ark tower order) for the aggregation mirror: the reference adds and scales IPP
commitments on the host (`com_s0 + com_s1 * pub_inputs[0] + ...`, distributed-prover/src/aggregation.rs:171-174,
312-316 - additive notation for products and powers in GT).  A handful of operations per job, so plain Python ints;
the pairings themselves run on the GPU (capi.Context.pairing_products).  No oracle import.

Tower: Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v), xi = 9 + u (BN254) / 1 + u (BLS12-381);
an element is the 12 Fq of ark's Fp12 { c0: Fp6 { c0, c1, c2: Fp2 { c0, c1 } }, c1 }."""
from .cp_groth16 import CURVE_PARAMS

_XI = {"bn254": (9, 1), "bls12_381": (1, 1)}


class GtField:
    def __init__(self, curve):
        p = CURVE_PARAMS[curve]
        self.q = p["q"]
        self.nb = p["fq_bytes"]
        self.xi = _XI[curve]
        self.R = 1 << (8 * self.nb)
        self.Ri = pow(self.R, -1, self.q)
        self.one = tuple([1] + [0] * 11)

    # ---- bytes <-> ints -------------------------------------------------------------------------------
    def decode(self, buf):
        b = bytes(buf)
        return tuple(int.from_bytes(b[i:i + self.nb], "little") * self.Ri % self.q for i in range(0, 12 * self.nb, self.nb))

    def encode(self, x):
        return b"".join((c * self.R % self.q).to_bytes(self.nb, "little") for c in x)

    def serialize(self, x):
        """ark-serialize (uncompressed = compressed for fields): 12 canonical little-endian Fq, tower order."""
        return b"".join(c.to_bytes(self.nb, "little") for c in x)

    # ---- tower ----------------------------------------------------------------------------------------
    def _f2m(self, a, b):
        q = self.q
        return ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)

    def _f6m(self, a, b):
        q, xi = self.q, self.xi
        t = [(0, 0)] * 5
        for i in range(3):
            for j in range(3):
                m = self._f2m(a[i], b[j])
                t[i + j] = ((t[i + j][0] + m[0]) % q, (t[i + j][1] + m[1]) % q)
        h3, h4 = self._f2m(t[3], xi), self._f2m(t[4], xi)
        return (((t[0][0] + h3[0]) % q, (t[0][1] + h3[1]) % q), ((t[1][0] + h4[0]) % q, (t[1][1] + h4[1]) % q), t[2])

    def mul(self, x, y):
        q = self.q
        a = [tuple((x[6 * i + 2 * j], x[6 * i + 2 * j + 1]) for j in range(3)) for i in range(2)]
        b = [tuple((y[6 * i + 2 * j], y[6 * i + 2 * j + 1]) for j in range(3)) for i in range(2)]
        add6 = lambda u, v: tuple(((s[0] + t[0]) % q, (s[1] + t[1]) % q) for s, t in zip(u, v))
        v1 = self._f6m(a[1], b[1])
        c0 = add6(self._f6m(a[0], b[0]), (self._f2m(v1[2], self.xi), v1[0], v1[1]))
        c1 = add6(self._f6m(a[0], b[1]), self._f6m(a[1], b[0]))
        return tuple(c for f6 in (c0, c1) for f2 in f6 for c in f2)

    def pow(self, x, e):
        out = self.one
        while e:
            if e & 1:
                out = self.mul(out, x)
            x = self.mul(x, x)
            e >>= 1
        return out

    def conj(self, x):
        """x^(q^6): the inverse of an element of GT (unitary)."""
        return tuple(x[:6]) + tuple((-c) % self.q for c in x[6:])
