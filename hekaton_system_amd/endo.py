"""Splitting a scalar along the G2 endomorphism psi (untwist - Frobenius - twist): psi(Q) = [lambda] Q on G2 with
lambda = q mod r = t - 1 (6 x^2 on BN254, x on BLS12-381; csrc/pairing.cuh `g2_psi`).  A scalar c is written as

        c = k0 + k1 lambda + k2 lambda^2 + k3 lambda^3   (mod r),      |kj| ~ r^(1/4) ~ 2^64

so that  c * Q = k0 Q + k1 psi(Q) + k2 psi^2(Q) + k3 psi^3(Q)  runs on one shared doubling chain of ~66 steps instead
of 254 (hk_points_fold_g2: the G2 folds of a TIPA round, where ONE challenge multiplies a whole vector).  The short
lattice basis is found once per curve by a textbook LLL on the 4 x 4 lattice of relations
{(a0..a3) : a0 + a1 lambda + a2 lambda^2 + a3 lambda^3 = 0 mod r}; a scalar is then reduced by Babai rounding.
Exact integer / rational arithmetic (a few hundred big-int operations per scalar, one scalar per fold)."""
from fractions import Fraction

from .cp_groth16 import CURVE_PARAMS

_X = {"bn254": 4965661367192848881, "bls12_381": -0xd201000000010000}


def eigenvalue(curve):
    p = CURVE_PARAMS[curve]
    lam = p["q"] % p["r"]
    x = _X[curve]
    assert lam == (6 * x * x if curve == "bn254" else x) % p["r"]
    return lam


def _lll(basis, delta=Fraction(3, 4)):
    """LLL reduction of integer row vectors (exact rationals; 4 x 4 here)."""
    b = [list(v) for v in basis]
    n = len(b)

    def gso():
        bs, mu = [], [[Fraction(0)] * n for _ in range(n)]
        for i in range(n):
            v = [Fraction(x) for x in b[i]]
            for j in range(i):
                d = sum(x * x for x in bs[j])
                mu[i][j] = sum(Fraction(b[i][t]) * bs[j][t] for t in range(len(v))) / d
                v = [v[t] - mu[i][j] * bs[j][t] for t in range(len(v))]
            bs.append(v)
        return bs, mu
    k = 1
    bs, mu = gso()
    while k < n:
        for j in range(k - 1, -1, -1):
            q = round(mu[k][j])
            if q:
                b[k] = [b[k][t] - q * b[j][t] for t in range(len(b[k]))]
                bs, mu = gso()
        nk = sum(x * x for x in bs[k])
        nk1 = sum(x * x for x in bs[k - 1])
        if nk >= (delta - mu[k][k - 1] ** 2) * nk1:
            k += 1
        else:
            b[k], b[k - 1] = b[k - 1], b[k]
            bs, mu = gso()
            k = max(k - 1, 1)
    return b


def _inverse(m):
    """Inverse of a small integer matrix over the rationals (Gauss-Jordan)."""
    n = len(m)
    a = [[Fraction(x) for x in row] + [Fraction(int(i == j)) for j in range(n)] for i, row in enumerate(m)]
    for c in range(n):
        p = next(r for r in range(c, n) if a[r][c] != 0)
        a[c], a[p] = a[p], a[c]
        inv = 1 / a[c][c]
        a[c] = [x * inv for x in a[c]]
        for r in range(n):
            if r != c and a[r][c] != 0:
                f = a[r][c]
                a[r] = [x - f * y for x, y in zip(a[r], a[c])]
    return [row[n:] for row in a]


class Psi4:
    """decompose(c) -> [k0, k1, k2, k3] (signed ints) with sum kj lambda^j = c mod r."""

    def __init__(self, curve):
        p = CURVE_PARAMS[curve]
        self.r = p["r"]
        self.lam = eigenvalue(curve)
        l1, l2, l3 = self.lam, self.lam ** 2 % self.r, self.lam ** 3 % self.r
        self.basis = _lll([[self.r, 0, 0, 0], [-l1, 1, 0, 0], [-l2, 0, 1, 0], [-l3, 0, 0, 1]])
        for v in self.basis:
            assert (v[0] + v[1] * l1 + v[2] * l2 + v[3] * l3) % self.r == 0
        self.inv = _inverse(self.basis)
        self.bound = max(abs(x) for v in self.basis for x in v)

    def decompose(self, c):
        c %= self.r
        # Babai rounding: (c, 0, 0, 0) = t * B over the rationals, subtract the nearest lattice point
        t = [round(c * self.inv[0][j]) for j in range(4)]
        k = [c, 0, 0, 0]
        for j in range(4):
            for i in range(4):
                k[i] -= t[j] * self.basis[j][i]
        return k


def _g1_mul(P, k, q):
    """k * P on y^2 = x^3 + b over Fq (affine, plain double-and-add; used once per curve to pick phi's eigenvalue)."""
    def add(A, B):
        if A is None:
            return B
        if B is None:
            return A
        if A[0] == B[0]:
            if (A[1] + B[1]) % q == 0:
                return None
            lam = 3 * A[0] * A[0] * pow(2 * A[1], -1, q) % q
        else:
            lam = (B[1] - A[1]) * pow(B[0] - A[0], -1, q) % q
        x = (lam * lam - A[0] - B[0]) % q
        return (x, (lam * (A[0] - x) - A[1]) % q)
    R, Q = None, P
    while k:
        if k & 1:
            R = add(R, Q)
        Q = add(Q, Q)
        k >>= 1
    return R


class Phi2:
    """The GLV endomorphism of G1, phi(x, y) = (beta x, y) with beta the first g^((q-1)/3) != 1 (g = 2, 3, ...: the constant
    csrc/gen_tower_params.py emits as BETA), and its eigenvalue lambda (the root of X^2 + X + 1 mod r with
    phi(G) = [lambda] G, found by trying both).  decompose(c) -> [k0, k1] with k0 + k1 lambda = c mod r, |kj| ~ 2^127."""

    def __init__(self, curve):
        p = CURVE_PARAMS[curve]
        q, r = p["q"], p["r"]
        self.r = r
        self.beta = next(b for b in (pow(g, (q - 1) // 3, q) for g in range(2, 50)) if b != 1)
        G = tuple(p["g1"])
        image = (self.beta * G[0] % q, G[1])
        w = next(x for x in (pow(g, (r - 1) // 3, r) for g in range(2, 50)) if x != 1)
        self.lam = next(l for l in (w, w * w % r) if _g1_mul(G, l, q) == image)
        self.basis = _lll([[r, 0], [-self.lam, 1]])
        for v in self.basis:
            assert (v[0] + v[1] * self.lam) % r == 0
        self.inv = _inverse(self.basis)

    def decompose(self, c):
        c %= self.r
        t = [round(c * self.inv[0][j]) for j in range(2)]
        k = [c, 0]
        for j in range(2):
            for i in range(2):
                k[i] -= t[j] * self.basis[j][i]
        return k


_CACHE = {}


def phi2(curve):
    key = ("phi", curve)
    if key not in _CACHE:
        _CACHE[key] = Phi2(curve)
    return _CACHE[key]


def psi4(curve):
    if curve not in _CACHE:
        _CACHE[curve] = Psi4(curve)
    return _CACHE[curve]
