"""Short-Weierstrass group arithmetic (a = 0) over Fq and Fq2, plain big-int Python.

TEST INFRASTRUCTURE (oracle) — see params.py header.  Restates the group law
that ark-ec 0.4 `short_weierstrass::{Affine,Projective}` implements (used by the
reference at cp-groth16/src/prover.rs:86-147 and committer.rs:87-91).  Points
are affine tuples (x, y); the point at infinity is None.
"""


class Fq:
    """Base field ops on Python ints."""

    def __init__(self, p):
        self.p = p
        self.zero = 0
        self.one = 1

    def add(self, a, b): return (a + b) % self.p
    def sub(self, a, b): return (a - b) % self.p
    def mul(self, a, b): return (a * b) % self.p
    def neg(self, a): return (-a) % self.p
    def inv(self, a): return pow(a, -1, self.p)
    def is_zero(self, a): return a % self.p == 0
    def from_int(self, k): return k % self.p


class Fq2:
    """Fq[u]/(u^2+1); elements are (c0, c1) tuples."""

    def __init__(self, p):
        self.p = p
        self.zero = (0, 0)
        self.one = (1, 0)

    def add(self, a, b): return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)
    def sub(self, a, b): return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)

    def mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def neg(self, a): return ((-a[0]) % self.p, (-a[1]) % self.p)

    def inv(self, a):
        p = self.p
        n = pow(a[0] * a[0] + a[1] * a[1], -1, p)
        return (a[0] * n % p, (-a[1]) * n % p)

    def is_zero(self, a): return a[0] % self.p == 0 and a[1] % self.p == 0
    def from_int(self, k): return (k % self.p, 0)


class _Group:
    def __init__(self, F, b):
        self.F = F
        self.b = b

    def on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.sub(F.mul(y, y), F.add(F.mul(F.mul(x, x), x), self.b)) == F.zero

    def neg(self, P):
        if P is None:
            return None
        return (P[0], self.F.neg(P[1]))

    # --- Jacobian internals -------------------------------------------------
    def _to_jac(self, P):
        if P is None:
            return (self.F.one, self.F.one, self.F.zero)
        return (P[0], P[1], self.F.one)

    def _from_jac(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        zi2 = F.mul(zi, zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def _jdbl(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z) or F.is_zero(Y):
            return (F.one, F.one, F.zero)
        A = F.mul(X, X)
        B = F.mul(Y, Y)
        C = F.mul(B, B)
        t = F.add(X, B)
        D = F.sub(F.sub(F.mul(t, t), A), C)
        D = F.add(D, D)
        E = F.add(F.add(A, A), A)
        Fv = F.mul(E, E)
        X3 = F.sub(Fv, F.add(D, D))
        C8 = F.add(C, C); C8 = F.add(C8, C8); C8 = F.add(C8, C8)
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
        Z3 = F.mul(F.add(Y, Y), Z)
        return (X3, Y3, Z3)

    def _jadd(self, J1, J2):
        F = self.F
        X1, Y1, Z1 = J1
        X2, Y2, Z2 = J2
        if F.is_zero(Z1):
            return J2
        if F.is_zero(Z2):
            return J1
        Z1Z1 = F.mul(Z1, Z1)
        Z2Z2 = F.mul(Z2, Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(F.mul(Y1, Z2), Z2Z2)
        S2 = F.mul(F.mul(Y2, Z1), Z1Z1)
        if U1 == U2:
            if S1 == S2:
                return self._jdbl(J1)
            return (F.one, F.one, F.zero)
        H = F.sub(U2, U1)
        R = F.sub(S2, S1)
        HH = F.mul(H, H)
        HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.mul(R, R), HHH), F.add(V, V))
        Y3 = F.sub(F.mul(R, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    # --- public affine API --------------------------------------------------
    def add(self, P, Q):
        return self._from_jac(self._jadd(self._to_jac(P), self._to_jac(Q)))

    def sub(self, P, Q):
        return self.add(P, self.neg(Q))

    def dbl(self, P):
        return self._from_jac(self._jdbl(self._to_jac(P)))

    def mul(self, P, k):
        """k * P for any integer k >= 0 (not reduced: callers may pass the group order)."""
        if P is None or k == 0:
            return None
        if k < 0:
            return self.mul(self.neg(P), -k)
        J = self._to_jac(P)
        acc = (self.F.one, self.F.one, self.F.zero)
        for bit in bin(k)[2:]:
            acc = self._jdbl(acc)
            if bit == "1":
                acc = self._jadd(acc, J)
        return self._from_jac(acc)

    def sum(self, pts):
        acc = (self.F.one, self.F.one, self.F.zero)
        for P in pts:
            acc = self._jadd(acc, self._to_jac(P))
        return self._from_jac(acc)

    def msm(self, bases, scalars):
        """Naive reference MSM: sum_i scalars[i] * bases[i] over min(len) terms
        (ark-ec `msm_unchecked` zips to the shorter length — SURVEY.md A.3)."""
        acc = (self.F.one, self.F.one, self.F.zero)
        for P, k in zip(bases, scalars):
            if P is None or k == 0:
                continue
            J = self._to_jac(P)
            t = (self.F.one, self.F.one, self.F.zero)
            for bit in bin(k)[2:]:
                t = self._jdbl(t)
                if bit == "1":
                    t = self._jadd(t, J)
            acc = self._jadd(acc, t)
        return self._from_jac(acc)


class G1(_Group):
    def __init__(self, cp):
        super().__init__(Fq(cp.q), cp.g1_b)
        self.cp = cp
        self.gen = cp.g1_gen
        self.coord_ints = 1


class G2(_Group):
    def __init__(self, cp):
        super().__init__(Fq2(cp.q), cp.g2_b)
        self.cp = cp
        self.gen = cp.g2_gen
        self.coord_ints = 2
