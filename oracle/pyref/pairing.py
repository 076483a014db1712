"""Optimal-ate pairing for BN254 and BLS12-381 over the Fq2 / Fq6 / Fq12 tower, plain Python big-ints.

TEST INFRASTRUCTURE (oracle) - see params.py header; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this.  PARITY UNPINNED: the algorithm lives in ark-ec 0.4 (`models/bn/{mod,g2}.rs`,
`models/bls12/{mod,g2}.rs`) and ark-ff 0.4 (`fields/models/{fp2,fp6_3over2,fp12_2over3over2}.rs`), third-party crates
absent from /root/reference; restated from memory of those files and from the papers they cite, reached by the
reference through `E::multi_miller_loop` + `E::final_exponentiation` (distributed-prover/src/pairing_ops.rs:9-29)
and `E::pairing` (aggregation.rs:208-216).  What pins it:
  * bilinearity, non-degeneracy, GT order r (tests/test_oracle_pairing.py);
  * BN254: the value equals the INDEPENDENT affine / flat-basis pairing of pairing_bn254.py raised to ark's hard-part
    multiple 2x(6x^2+3x+1) (Fuentes-Castaneda et al.), and BLS12-381's equals f^((p^12-1)/r) raised to 3
    (Hayashida-Hayasaka-Teruya, eprint 2020/875) - i.e. the remembered addition chains reproduce the closed forms
    the ark comments state;
  * the Groth16 verifier equation (cp-groth16/src/verifier.rs:23-43) accepts every fixture proof on both curves.

Tower (both curves): Fq2 = Fq[u]/(u^2+1); Fq6 = Fq2[v]/(v^3 - xi); Fq12 = Fq6[w]/(w^2 - v);
xi = 9 + u (BN254, D-type twist), 1 + u (BLS12-381, M-type twist).
GT layout (ark `Fp12` = c0: Fp6, c1: Fp6; Fp6 = c0, c1, c2: Fp2; Fp2 = c0, c1): 12 Fq in that nesting order.
"""
from .params import BN254, BLS12_381


class Tower:
    def __init__(self, cp):
        self.cp = cp
        self.p = p = cp.q
        self.r = cp.r
        if cp.name == "bn254":
            self.xi = (9, 1)
            self.x = 4965661367192848881
            self.x_is_negative = False
            self.twist = "D"
            self.loop = 6 * self.x + 2            # ATE_LOOP_COUNT (ark stores its NAF; the value is 6x+2)
        else:
            self.xi = (1, 1)
            self.x = 0xd201000000010000
            self.x_is_negative = True
            self.twist = "M"
            self.loop = self.x
        self.b_twist = cp.g2_b
        self.two_inv = pow(2, -1, p)
        # Frobenius coefficients (ark FROBENIUS_COEFF_FP6_C1/C2, FP12_C1): xi^((p^k-1)/3), xi^(2(p^k-1)/3), xi^((p^k-1)/6)
        self.f6c1 = [self.f2_pow(self.xi, (p ** k - 1) // 3) for k in range(6)]
        self.f6c2 = [self.f2_pow(self.xi, 2 * (p ** k - 1) // 3) for k in range(6)]
        self.f12c1 = [self.f2_pow(self.xi, (p ** k - 1) // 6) for k in range(12)]
        # ark TWIST_MUL_BY_Q_X / _Y (BN only): xi^((p-1)/3), xi^((p-1)/2)
        self.twist_mul_by_q_x = self.f2_pow(self.xi, (p - 1) // 3)
        self.twist_mul_by_q_y = self.f2_pow(self.xi, (p - 1) // 2)

    # ---- Fq2 ------------------------------------------------------------------------------------------
    def f2_add(self, a, b): return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)
    def f2_sub(self, a, b): return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)
    def f2_neg(self, a): return ((-a[0]) % self.p, (-a[1]) % self.p)
    def f2_dbl(self, a): return self.f2_add(a, a)

    def f2_mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def f2_sqr(self, a): return self.f2_mul(a, a)
    def f2_scale(self, a, k): return (a[0] * k % self.p, a[1] * k % self.p)
    def f2_conj(self, a): return (a[0], (-a[1]) % self.p)

    def f2_inv(self, a):
        p = self.p
        n = pow(a[0] * a[0] + a[1] * a[1], -1, p)
        return (a[0] * n % p, (-a[1]) * n % p)

    def f2_pow(self, a, e):
        out = (1, 0)
        while e:
            if e & 1:
                out = self.f2_mul(out, a)
            a = self.f2_mul(a, a)
            e >>= 1
        return out

    def f2_mul_xi(self, a): return self.f2_mul(a, self.xi)
    def f2_frob(self, a, k): return self.f2_conj(a) if k & 1 else a

    # ---- Fq6 = Fq2[v]/(v^3 - xi) ------------------------------------------------------------------------
    F6_ZERO = ((0, 0), (0, 0), (0, 0))
    F6_ONE = ((1, 0), (0, 0), (0, 0))

    def f6_add(self, a, b): return tuple(self.f2_add(x, y) for x, y in zip(a, b))
    def f6_sub(self, a, b): return tuple(self.f2_sub(x, y) for x, y in zip(a, b))
    def f6_neg(self, a): return tuple(self.f2_neg(x) for x in a)

    def f6_mul(self, a, b):
        m, ad, sb, xi = self.f2_mul, self.f2_add, self.f2_sub, self.f2_mul_xi
        a0, a1, a2 = a
        b0, b1, b2 = b
        v0, v1, v2 = m(a0, b0), m(a1, b1), m(a2, b2)
        c0 = ad(v0, xi(sb(sb(m(ad(a1, a2), ad(b1, b2)), v1), v2)))
        c1 = ad(sb(sb(m(ad(a0, a1), ad(b0, b1)), v0), v1), xi(v2))
        c2 = ad(sb(sb(m(ad(a0, a2), ad(b0, b2)), v0), v2), v1)
        return (c0, c1, c2)

    def f6_sqr(self, a): return self.f6_mul(a, a)
    def f6_mul_by_v(self, a): return (self.f2_mul_xi(a[2]), a[0], a[1])        # ark Fp12Config::mul_fp6_by_nonresidue

    def f6_inv(self, a):
        m, sb, ad, xi = self.f2_mul, self.f2_sub, self.f2_add, self.f2_mul_xi
        a0, a1, a2 = a
        t0 = sb(self.f2_sqr(a0), xi(m(a1, a2)))
        t1 = sb(xi(self.f2_sqr(a2)), m(a0, a1))
        t2 = sb(self.f2_sqr(a1), m(a0, a2))
        d = ad(m(a0, t0), xi(ad(m(a2, t1), m(a1, t2))))
        di = self.f2_inv(d)
        return (m(t0, di), m(t1, di), m(t2, di))

    def f6_frob(self, a, k):
        return (self.f2_frob(a[0], k), self.f2_mul(self.f2_frob(a[1], k), self.f6c1[k % 6]),
                self.f2_mul(self.f2_frob(a[2], k), self.f6c2[k % 6]))

    # ---- Fq12 = Fq6[w]/(w^2 - v) ------------------------------------------------------------------------
    def f12_one(self): return (self.F6_ONE, self.F6_ZERO)

    def f12_mul(self, a, b):
        a0, a1 = a
        b0, b1 = b
        v0, v1 = self.f6_mul(a0, b0), self.f6_mul(a1, b1)
        c1 = self.f6_sub(self.f6_sub(self.f6_mul(self.f6_add(a0, a1), self.f6_add(b0, b1)), v0), v1)
        c0 = self.f6_add(v0, self.f6_mul_by_v(v1))
        return (c0, c1)

    def f12_sqr(self, a): return self.f12_mul(a, a)
    def f12_conj(self, a): return (a[0], self.f6_neg(a[1]))                   # = a^(p^6): cyclotomic inverse

    def f12_inv(self, a):
        a0, a1 = a
        d = self.f6_sub(self.f6_sqr(a0), self.f6_mul_by_v(self.f6_sqr(a1)))
        di = self.f6_inv(d)
        return (self.f6_mul(a0, di), self.f6_neg(self.f6_mul(a1, di)))

    def f12_frob(self, a, k):
        c0 = self.f6_frob(a[0], k)
        c1 = self.f6_frob(a[1], k)
        co = self.f12c1[k % 12]
        return (c0, tuple(self.f2_mul(x, co) for x in c1))

    def f12_pow(self, a, e):
        out = self.f12_one()
        while e:
            if e & 1:
                out = self.f12_mul(out, a)
            a = self.f12_mul(a, a)
            e >>= 1
        return out

    def f12_flat(self, a):
        """The 12 Fq coordinates in ark's nesting order (c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1)."""
        return [c for f6 in a for f2 in f6 for c in f2]

    def f12_from_flat(self, v):
        f2 = [(v[2 * i], v[2 * i + 1]) for i in range(6)]
        return ((f2[0], f2[1], f2[2]), (f2[3], f2[4], f2[5]))

    def sparse(self, c0, c1, c2):
        """The line value as a full Fq12 element.  D-type (ark mul_by_034): c0 + (c1 + c2 v) w;
        M-type (ark mul_by_014): (c0 + c1 v) + (c2 v) w."""
        z = (0, 0)
        if self.twist == "D":
            return ((c0, z, z), (c1, c2, z))
        return ((c0, c1, z), (z, c2, z))

    # ---- G2 line steps in homogeneous projective coordinates (ark bn/g2.rs, bls12/g2.rs) ----------------
    def doubling_step(self, R):
        m, sq, ad, sb = self.f2_mul, self.f2_sqr, self.f2_add, self.f2_sub
        X, Y, Z = R
        a = self.f2_scale(m(X, Y), self.two_inv)
        b, c = sq(Y), sq(Z)
        e = m(self.b_twist, ad(ad(c, c), c))
        f = ad(ad(e, e), e)
        g = self.f2_scale(ad(b, f), self.two_inv)
        h = sb(sq(ad(Y, Z)), ad(b, c))
        i = sb(e, b)
        j = sq(X)
        e_sq = sq(e)
        R2 = (m(a, sb(b, f)), sb(sq(g), ad(ad(e_sq, e_sq), e_sq)), m(b, h))
        j3 = ad(ad(j, j), j)
        coeffs = (self.f2_neg(h), j3, i) if self.twist == "D" else (i, j3, self.f2_neg(h))
        return R2, coeffs

    def addition_step(self, R, Q):
        m, sq, ad, sb = self.f2_mul, self.f2_sqr, self.f2_add, self.f2_sub
        X, Y, Z = R
        x2, y2 = Q
        theta = sb(Y, m(y2, Z))
        lam = sb(X, m(x2, Z))
        c, d = sq(theta), sq(lam)
        e = m(lam, d)
        f = m(Z, c)
        g = m(X, d)
        h = sb(ad(e, f), ad(g, g))
        R2 = (m(lam, h), sb(m(theta, sb(g, h)), m(e, Y)), m(Z, e))
        j = sb(m(theta, x2), m(lam, y2))
        coeffs = (lam, self.f2_neg(theta), j) if self.twist == "D" else (j, self.f2_neg(theta), lam)
        return R2, coeffs

    def ell(self, f, coeffs, P):
        c0, c1, c2 = coeffs
        px, py = P
        if self.twist == "D":
            c0, c1 = self.f2_scale(c0, py), self.f2_scale(c1, px)
        else:
            c2, c1 = self.f2_scale(c2, py), self.f2_scale(c1, px)
        return self.f12_mul(f, self.sparse(c0, c1, c2))

    def mul_by_char(self, Q):
        return (self.f2_mul(self.f2_frob(Q[0], 1), self.twist_mul_by_q_x),
                self.f2_mul(self.f2_frob(Q[1], 1), self.twist_mul_by_q_y))

    def prepare_g2(self, Q):
        """ark G2Prepared::from: the list of line coefficients, in the order the Miller loop consumes them."""
        if Q is None:
            return None
        R = (Q[0], Q[1], (1, 0))
        negQ = (Q[0], self.f2_neg(Q[1]))
        out = []
        if self.cp.name == "bn254":
            digits = naf(self.loop)                        # ark ATE_LOOP_COUNT: i8 NAF, least significant first
            for i in range(len(digits) - 1, 0, -1):
                R, c = self.doubling_step(R); out.append(c)
                d = digits[i - 1]
                if d == 1:
                    R, c = self.addition_step(R, Q); out.append(c)
                elif d == -1:
                    R, c = self.addition_step(R, negQ); out.append(c)
            Q1 = self.mul_by_char(Q)
            Q2 = self.mul_by_char(Q1)
            if self.x_is_negative:
                R = (R[0], self.f2_neg(R[1]), R[2])
            Q2 = (Q2[0], self.f2_neg(Q2[1]))
            R, c = self.addition_step(R, Q1); out.append(c)
            R, c = self.addition_step(R, Q2); out.append(c)
        else:
            bits = bin(self.loop)[3:]                      # BitIteratorBE::without_leading_zeros(X).skip(1)
            for bit in bits:
                R, c = self.doubling_step(R); out.append(c)
                if bit == "1":
                    R, c = self.addition_step(R, Q); out.append(c)
        return out

    def multi_miller_loop(self, pairs):
        """pairs: [(P in G1 affine | None, Q in G2 affine | None)].  ark `multi_miller_loop`: pairs with an infinity
        member are skipped; one shared squaring of f per step."""
        prepared = [(P, iter(self.prepare_g2(Q))) for P, Q in pairs if P is not None and Q is not None]
        f = self.f12_one()
        if self.cp.name == "bn254":
            digits = naf(self.loop)
            for i in range(len(digits) - 1, 0, -1):
                if i != len(digits) - 1:
                    f = self.f12_sqr(f)
                for P, it in prepared:
                    f = self.ell(f, next(it), P)
                if digits[i - 1] != 0:
                    for P, it in prepared:
                        f = self.ell(f, next(it), P)
            if self.x_is_negative:
                f = self.f12_conj(f)
            for _ in range(2):
                for P, it in prepared:
                    f = self.ell(f, next(it), P)
        else:
            bits = bin(self.loop)[3:]
            for bit in bits:
                f = self.f12_sqr(f)
                for P, it in prepared:
                    f = self.ell(f, next(it), P)
                if bit == "1":
                    for P, it in prepared:
                        f = self.ell(f, next(it), P)
            if self.x_is_negative:
                f = self.f12_conj(f)
        return f

    # ---- final exponentiation, ark's addition chains ---------------------------------------------------
    def _exp_by_x(self, f):
        """ark `exp_by_x` (bls12) : f^X, conjugated when X is negative."""
        g = self.f12_pow(f, self.x)
        return self.f12_conj(g) if self.x_is_negative else g

    def _exp_by_neg_x(self, f):
        """ark `exp_by_neg_x` (bn): f^X, conjugated when X is NOT negative."""
        g = self.f12_pow(f, self.x)
        return g if self.x_is_negative else self.f12_conj(g)

    def final_exponentiation(self, f):
        mul, conj, frob, sq = self.f12_mul, self.f12_conj, self.f12_frob, self.f12_sqr
        # easy part: r = f^((p^6 - 1)(p^2 + 1))
        f2 = self.f12_inv(f)
        r = mul(conj(f), f2)
        r = mul(frob(r, 2), r)
        if self.cp.name == "bn254":
            # hard part, Fuentes-Castaneda et al.: r^(2x(6x^2+3x+1) (p^4-p^2+1)/r)
            y0 = self._exp_by_neg_x(r)
            y1 = sq(y0)
            y2 = sq(y1)
            y3 = mul(y2, y1)
            y4 = self._exp_by_neg_x(y3)
            y5 = sq(y4)
            y6 = self._exp_by_neg_x(y5)
            y3 = conj(y3)
            y6 = conj(y6)
            y7 = mul(y6, y4)
            y8 = mul(y7, y3)
            y9 = mul(y8, y1)
            y10 = mul(y8, y4)
            y11 = mul(y10, r)
            y12 = frob(y9, 1)
            y13 = mul(y12, y11)
            y8 = frob(y8, 2)
            y14 = mul(y8, y13)
            r = conj(r)
            y15 = frob(mul(r, y9), 3)
            return mul(y15, y14)
        # BLS12, Hayashida-Hayasaka-Teruya (eprint 2020/875): r^(3 (p^4-p^2+1)/r)
        y0 = sq(r)
        y1 = self._exp_by_x(r)
        y2 = conj(r)
        y1 = mul(y1, y2)
        y2 = self._exp_by_x(y1)
        y1 = conj(y1)
        y1 = mul(y1, y2)
        y2 = self._exp_by_x(y1)
        y1 = frob(y1, 1)
        y1 = mul(y1, y2)
        r = mul(r, y0)
        y0 = self._exp_by_x(y1)
        y2 = self._exp_by_x(y0)
        y0 = frob(y1, 2)
        y1 = conj(y1)
        y1 = mul(y1, y2)
        y1 = mul(y1, y0)
        return mul(r, y1)

    @property
    def hard_part_multiple(self):
        """ark's final exponentiation = f^(lambda (p^12-1)/r) with this lambda."""
        x = -self.x if self.x_is_negative else self.x
        return 2 * x * (6 * x * x + 3 * x + 1) if self.cp.name == "bn254" else 3

    def multi_pairing(self, pairs):
        """`pairing()` of distributed-prover/src/pairing_ops.rs:25-29: prod_i e(P_i, Q_i) as ark computes it."""
        return self.final_exponentiation(self.multi_miller_loop(pairs))

    def pairing(self, P, Q):
        return self.multi_pairing([(P, Q)])


def naf(n):
    """Non-adjacent form, least significant digit first (ark `find_naf` / the ATE_LOOP_COUNT table of ark-bn254)."""
    out = []
    while n:
        if n & 1:
            d = 2 - (n % 4)
            n -= d
        else:
            d = 0
        out.append(d)
        n >>= 1
    return out


_TOWERS = {}


def tower(name):
    if name not in _TOWERS:
        _TOWERS[name] = Tower({"bn254": BN254, "bls12_381": BLS12_381}[name])
    return _TOWERS[name]


def verify_proof(curve_name, vk, proof, public_inputs):
    """cp-groth16/src/verifier.rs:64-71 + :23-43 on either curve:
    e(A,B) * e(IC,-gamma) * prod e(D_i,-delta_i) * e(C,-delta_last) == e(alpha,beta)."""
    from . import curve
    from .groth16 import prepare_inputs
    T = tower(curve_name)
    G2 = curve.G2(T.cp)
    ic = prepare_inputs(T.cp, vk, public_inputs)
    lhs = [proof.a, ic] + list(proof.ds) + [proof.c]
    rhs = [proof.b, G2.neg(vk.gamma_h)] + [G2.neg(d) for d in vk.deltas_h]
    if len(lhs) != len(rhs):
        return False
    return T.multi_pairing(list(zip(lhs, rhs))) == T.pairing(vk.alpha_g, vk.beta_h)
