"""BN254 optimal-ate pairing in plain Python big-ints, and the CP-Groth16 verifier
equation on top of it.

TEST INFRASTRUCTURE (oracle) — see params.py header.  This is the strongest pin
the reference offers for the hot path: its own tests accept a proof iff
`verify_proof(&pvk, &proof, &inputs) == true` (cp-groth16/src/lib.rs:179,312),
i.e. iff  e(A,B) * e(IC,-gamma) * prod e(D_i,-delta_i) * e(C,-delta_last) == e(alpha,beta)
(cp-groth16/src/verifier.rs:23-43).  The pairing itself lives in ark-bn254 /
ark-ec (third-party, absent); restated here from the published optimal-ate
construction (Fq12 = Fq[w]/(w^12 - 18 w^6 + 82), D-type twist, loop count 6x+2)
and validated by bilinearity + non-degeneracy self-tests (tests/test_oracle_pairing.py).
"""

from .params import BN254
from . import curve

P = BN254.q
R_ORDER = BN254.r
ATE_LOOP_COUNT = 29793968203157093288       # 6x+2, x = 4965661367192848881
LOG_ATE = 63

_F2 = curve.Fq2(P)
_G2 = curve.G2(BN254)


# ---- Fq12 = Fq[w] / (w^12 - 18 w^6 + 82) ------------------------------------------------

def f12_one():
    return [1] + [0] * 11


def f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                if bj:
                    t[i + j] += ai * bj
    for k in range(22, 11, -1):         # w^k = 18 w^(k-6) - 82 w^(k-12)
        c = t[k]
        if c:
            t[k - 6] += 18 * c
            t[k - 12] -= 82 * c
    return [x % P for x in t[:12]]


def f12_pow(a, e):
    out = f12_one()
    base = a
    while e:
        if e & 1:
            out = f12_mul(out, base)
        base = f12_mul(base, base)
        e >>= 1
    return out


def _embed(c):
    """Fq2 element c0 + c1 u  ->  (c0 - 9 c1) + c1 w^6  (u = w^6 - 9)."""
    return ((c[0] - 9 * c[1]) % P, c[1] % P)


def _line(m, x1, y1, px, py):
    """Line with Fq2 slope m through twisted (x1,y1), evaluated at P=(px,py) in G1:
    -py + (m px) w + (y1 - m x1) w^3."""
    out = [0] * 12
    out[0] = (-py) % P
    a0, a6 = _embed(_F2.mul(m, (px, 0)))
    out[1], out[7] = a0, a6
    b0, b6 = _embed(_F2.sub(y1, _F2.mul(m, x1)))
    out[3], out[9] = b0, b6
    return out


def _vertical(x1, px):
    """px - x1 w^2."""
    out = [0] * 12
    out[0] = px % P
    a0, a6 = _embed(_F2.neg(x1))
    out[2], out[8] = a0, a6
    return out


def _f2_pow(a, e):
    out = (1, 0)
    while e:
        if e & 1:
            out = _F2.mul(out, a)
        a = _F2.mul(a, a)
        e >>= 1
    return out


_XI = (9, 1)
_G_X1 = _f2_pow(_XI, (P - 1) // 3)
_G_Y1 = _f2_pow(_XI, (P - 1) // 2)
_G_X2 = _f2_pow(_XI, (P * P - 1) // 3)
_G_Y2 = _f2_pow(_XI, (P * P - 1) // 2)


def _conj(a):
    return (a[0], (-a[1]) % P)


def _step(R, Q, px, py):
    """Returns (line through R and Q evaluated at P, R+Q) with R, Q affine on the twist."""
    x1, y1 = R
    x2, y2 = Q
    if x1 != x2:
        m = _F2.mul(_F2.sub(y2, y1), _F2.inv(_F2.sub(x2, x1)))
    elif y1 == y2:
        x1s = _F2.mul(x1, x1)
        m = _F2.mul(_F2.add(_F2.add(x1s, x1s), x1s), _F2.inv(_F2.add(y1, y1)))
    else:
        return _vertical(x1, px), None
    x3 = _F2.sub(_F2.sub(_F2.mul(m, m), x1), x2)
    y3 = _F2.sub(_F2.mul(m, _F2.sub(x1, x3)), y1)
    return _line(m, x1, y1, px, py), (x3, y3)


def miller_loop(Q, Pt):
    """Q in G2 (affine Fq2 coords or None), Pt in G1 (affine or None)."""
    if Q is None or Pt is None:
        return f12_one()
    px, py = Pt
    Rp = Q
    f = f12_one()
    for i in range(LOG_ATE, -1, -1):
        l, R2 = _step(Rp, Rp, px, py)
        f = f12_mul(f12_mul(f, f), l)
        Rp = R2
        if ATE_LOOP_COUNT & (1 << i):
            l, R2 = _step(Rp, Q, px, py)
            f = f12_mul(f, l)
            Rp = R2
    Q1 = (_F2.mul(_conj(Q[0]), _G_X1), _F2.mul(_conj(Q[1]), _G_Y1))
    nQ2 = (_F2.mul(Q[0], _G_X2), _F2.neg(_F2.mul(Q[1], _G_Y2)))
    l, R2 = _step(Rp, Q1, px, py)
    f = f12_mul(f, l)
    Rp = R2
    l, _ = _step(Rp, nQ2, px, py)
    f = f12_mul(f, l)
    return f


_FINAL_EXP = (P ** 12 - 1) // R_ORDER


def final_exponentiation(f):
    return f12_pow(f, _FINAL_EXP)


def pairing(Q, Pt):
    return final_exponentiation(miller_loop(Q, Pt))


def multi_pairing(pairs):
    """prod e(P_i, Q_i) for pairs [(P in G1, Q in G2)] — one final exponentiation."""
    f = f12_one()
    for Pt, Q in pairs:
        f = f12_mul(f, miller_loop(Q, Pt))
    return final_exponentiation(f)


def verify_proof(vk, proof, public_inputs):
    """cp-groth16/src/verifier.rs:64-71 (prepare_inputs + the multi-Miller-loop check
    of verify_proof_with_prepared_inputs :23-43) for BN254."""
    from .groth16 import prepare_inputs
    G1 = curve.G1(BN254)
    ic = prepare_inputs(BN254, vk, public_inputs)
    lhs = [proof.a, ic] + list(proof.ds) + [proof.c]
    rhs = [proof.b, _G2.neg(vk.gamma_h)] + [_G2.neg(d) for d in vk.deltas_h]
    if len(lhs) != len(rhs):
        return False
    test = multi_pairing(list(zip(lhs, rhs)))
    return test == pairing(vk.beta_h, vk.alpha_g)
