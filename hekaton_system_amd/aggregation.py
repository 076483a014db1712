"""Host-side mirror of the aggregator's front half over the C ABI.

    IPCommKey / TIPPCommitment.{commit_only_left, commit_only_right, commit_with_ip}, IppCom (+, * scalar)
                                    ark-ip-proofs `ip_commitment::snarkpack::TIPPCommitment` as the reference uses it:
                                    distributed-prover/src/aggregation.rs:18,97-103,167-168; coordinator.rs:339
    tipa_commitment_key             the commitment-key part of `TIPA::setup` (mpi-snark/src/coordinator.rs:80-97)
    AggProvingKey.new               distributed-prover/src/aggregation.rs:60-135
    AggProvingKey.agg_front         distributed-prover/src/aggregation.rs:138-330: everything `agg_subcircuit_proofs` does
                                    BEFORE `TIPA::prove` - commitments, prepared inputs, twisted vectors, the 4 x 4 cross
                                    terms, the s/t combination, com_lr, z_lr - i.e. the TIPA instance and witness

All group arithmetic runs on the GPU: multi-pairings (hk_pairing_products), element-wise scalar multiplications
(hk_scalar_pairing), element-wise linear combinations (hk_points_lincomb).  GT products / powers of commitments
(three per job) are host arithmetic (gt.py).  The Fiat-Shamir challenges r (twist), s, t come from a merlin transcript
(merlin.py, pinned by merlin's known-answer test) when `pt` is given, else they are arguments.  `TIPA::prove / verify`
(the GIPA recursion of the third-party `ripp` crate, absent from /root/reference) are tipa.py; `agg_subcircuit_proofs`
joins the two as aggregation.rs:138-345 does.  PARITY UNPINNED for the snarkpack commitment layout (T = e(A, v1) e(w1, B), U = e(A, v2) e(w2, B),
restated from the SnarkPack paper); what the tests pin is the reference's own debug assertions
(aggregation.rs:208-216,246-253,265-269) holding on proofs made by this prover.
"""
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass

import numpy as np

from .cp_groth16 import CURVE_PARAMS, FrCodec
from .gt import GtField


@dataclass
class IPCommKey:
    """SnarkPack pair commitment key: v1, v2 in G2^n (commit the left / G1 vector), w1, w2 in G1^n (right / G2 vector)."""
    v1: np.ndarray
    v2: np.ndarray
    w1: np.ndarray
    w2: np.ndarray
    n: int


def tipa_commitment_key(ctx, curve, n, a, b):
    """v1 = h^(a^i), v2 = h^(b^i), w1 = g^(a^(n+i)), w2 = g^(b^(n+i)), i < n (the structured key `TIPA::setup` derives from
    its two trapdoors), built with the GPU fixed-base path."""
    p = CURVE_PARAMS[curve]
    fc = FrCodec(curve)
    r = p["r"]
    pa, pb = [1] * (2 * n), [1] * (2 * n)
    for i in range(1, 2 * n):
        pa[i] = pa[i - 1] * a % r
        pb[i] = pb[i - 1] * b % r
    G1, G2 = fc.g1(p["g1"]), fc.g2(p["g2"])
    return IPCommKey(v1=np.asarray(ctx.fixed_base(2, G2, fc.enc(pa[:n]))), v2=np.asarray(ctx.fixed_base(2, G2, fc.enc(pb[:n]))),
                     w1=np.asarray(ctx.fixed_base(1, G1, fc.enc(pa[n:]))), w2=np.asarray(ctx.fixed_base(1, G1, fc.enc(pb[n:]))), n=n)


class IppCom:
    """`Commitment<TIPPCommitment<E>>`: (T, U) in GT x GT and, for commit_with_ip, the inner product.  `+` multiplies
    in GT, `* k` raises to k (the reference's additive notation)."""

    def __init__(self, F, t, u, ip=None, ctx=None):
        self.F, self.t, self.u, self.ip, self.ctx = F, t, u, ip, ctx

    def __add__(self, o):
        ip = None if (self.ip is None and o.ip is None) else self.F.mul(self.ip or self.F.one, o.ip or self.F.one)
        return IppCom(self.F, self.F.mul(self.t, o.t), self.F.mul(self.u, o.u), ip, self.ctx or o.ctx)

    def __mul__(self, k):
        parts = [self.t, self.u] + ([self.ip] if self.ip is not None else [])
        if self.ctx is not None:                       # GT powers on the GPU (hk_gt_pow), one wave per component
            from .cp_groth16 import FrCodec
            fc = FrCodec(self.ctx.curve)
            out = self.ctx.gt_pow(np.frombuffer(b"".join(self.F.encode(x) for x in parts), np.uint8), fc.enc([k] * len(parts)))
            res = [self.F.decode(out[i]) for i in range(len(parts))]
        else:
            res = [self.F.pow(x, k) for x in parts]
        return IppCom(self.F, res[0], res[1], res[2] if len(res) > 2 else None, self.ctx)

    @staticmethod
    def lincomb(terms):
        """sum_k com_k * scalar_k (additive notation) with every GT power of the sum in ONE hk_gt_pow call.
        terms: [(IppCom, int scalar or None for 1)]."""
        F = terms[0][0].F
        ctx = next((c.ctx for c, _ in terms if c.ctx is not None), None)
        has_ip = any(com.ip is not None for com, _ in terms)
        members = [[com.t, com.u] + ([com.ip] if has_ip else []) for com, _ in terms]       # a missing inner product: 1
        scalars = [1 if k is None else k for _, k in terms]
        if ctx is not None:
            # one grouped multi-exponentiation: a group per member (T, U, inner product), a power per term
            from .cp_groth16 import FrCodec
            bases, exps = [], []
            for j in range(3 if has_ip else 2):
                for parts, k in zip(members, scalars):
                    bases.append(parts[j] if parts[j] is not None else F.one)
                    exps.append(k if parts[j] is not None else 0)
            out = ctx.gt_pow_prod(np.frombuffer(b"".join(F.encode(x) for x in bases), np.uint8), FrCodec(ctx.curve).enc(exps),
                                  len(terms))
            res = [F.decode(out[j]) for j in range(len(out))]
        else:
            res = []
            for j in range(3 if has_ip else 2):
                acc = F.one
                for parts, k in zip(members, scalars):
                    if parts[j] is not None:
                        acc = F.mul(acc, parts[j] if k == 1 else F.pow(parts[j], k))
                res.append(acc)
        return IppCom(F, res[0], res[1], res[2] if has_ip else None, ctx)

    def __eq__(self, o):
        return self.t == o.t and self.u == o.u and (self.ip or self.F.one) == (o.ip or self.F.one)

    def to_bytes(self):
        return self.F.encode(self.t) + self.F.encode(self.u) + (self.F.encode(self.ip) if self.ip is not None else b"")

    def serialize_uncompressed(self):
        """Transcript bytes: each GT element as ark-serialize writes a `PairingOutput` (12 canonical little-endian Fq in
        tower order), T then U then the inner product when present.  The member order of the third-party
        `Commitment<TIPPCommitment<E>>` is an assumption (ripp is absent)."""
        parts = [self.t, self.u] + ([self.ip] if self.ip is not None else [])
        return b"".join(self.F.serialize(x) for x in parts)


def rom_challenges(super_com, r_mod):
    """(entry_chal, tr_chal) of a ROM job (`RomRunningEvaluation::new`, rom_transcript.rs:42-75; the whole running
    evaluation object is transcript.RunningEvaluation)."""
    from .transcript import ROM, RunningEvaluation
    return RunningEvaluation.new(ROM, super_com, r_mod).challenges


class TIPPCommitment:
    def __init__(self, ctx, curve):
        self.ctx, self.F = ctx, GtField(curve)

    def commit_only_left(self, ck, left):
        """(e(A, v1), e(A, v2)) - aggregation.rs:97-100,168; coordinator.rs:339 (the super-commitment to all stage-0
        commitments, on the coordinator's critical path between the two rounds)."""
        out = self.ctx.pairing_products([left], [ck.v1, ck.v2], n=ck.n)
        return IppCom(self.F, self.F.decode(out[0, 0]), self.F.decode(out[0, 1]), ctx=self.ctx)

    def commit_only_right(self, ck, right):
        """(e(w1, B), e(w2, B)) - aggregation.rs:101-103."""
        out = self.ctx.pairing_products([ck.w1, ck.w2], [right], n=ck.n)
        return IppCom(self.F, self.F.decode(out[0, 0]), self.F.decode(out[1, 0]), ctx=self.ctx)

    def commit_with_ip(self, ck, left, right):
        """T = e(A, v1) e(w1, B), U = e(A, v2) e(w2, B), Z = e(A, B) - aggregation.rs:167."""
        # five inner products out of 3 x 3 vectors in ONE batched call (each G2 vector's Miller lines once)
        o = self.ctx.pairing_pairs([left, ck.w1, ck.w2], [ck.v1, ck.v2, right], [(0, 0), (1, 2), (0, 1), (2, 2), (0, 2)], n=ck.n)
        F = self.F
        D = F.decode
        return IppCom(F, F.mul(D(o[0]), D(o[1])), F.mul(D(o[2]), D(o[3])), D(o[4]), ctx=self.ctx)


class AggProvingKey:
    """distributed-prover/src/aggregation.rs:23-135.  `vks[i]`: the Groth16 verifying-key parts of subcircuit i's class
    (cp_groth16.VerifyingKey): gamma_abc_g (4 G1), gamma_h, deltas_h (2 G2), alpha_g, beta_h."""

    def __init__(self, ctx, curve, ck, vks):
        self.ctx, self.curve, self.ck = ctx, curve, ck
        self.com = TIPPCommitment(ctx, curve)
        self.F = self.com.F
        self.pool = ThreadPoolExecutor(max_workers=12)
        g1b, g2b = ctx.g1_bytes, ctx.g2_bytes
        cat = lambda xs: np.concatenate([np.asarray(x, np.uint8) for x in xs])
        self.n = len(vks)
        assert self.n == ck.n
        self.s = [cat([vk.gamma_abc_g[j * g1b:(j + 1) * g1b] for vk in vks]) for j in range(4)]      # :84-87
        self.h = cat([vk.gamma_h for vk in vks])                                                    # :88
        self.delta0 = cat([vk.deltas_h[:g2b] for vk in vks])                                        # :89
        self.delta1 = cat([vk.deltas_h[g2b:2 * g2b] for vk in vks])                                 # :90
        self.alpha = cat([vk.alpha_g for vk in vks])                                                # :91
        self.beta = cat([vk.beta_h for vk in vks])                                                  # :92
        # the seven commitments (:97-103) are fourteen inner products between six G1 and five G2 vectors: one batched call
        o = ctx.pairing_pairs(self.s + [ck.w1, ck.w2], [ck.v1, ck.v2, self.h, self.delta0, self.delta1],
                              [(j, k) for j in range(4) for k in range(2)] + [(4 + j, 2 + k) for k in range(3) for j in range(2)], n=ck.n)
        D = self.F.decode
        self.com_s = [IppCom(self.F, D(o[2 * j]), D(o[2 * j + 1]), ctx=ctx) for j in range(4)]
        self.com_h, self.com_delta0, self.com_delta1 = (IppCom(self.F, D(o[8 + 2 * k]), D(o[9 + 2 * k]), ctx=ctx) for k in range(3))

    def agg_subcircuit_proofs(self, pt, super_com, proofs, pub_inputs, srs, tipp=None, check=True):
        """aggregation.rs:138-345 whole: the challenges come from the merlin transcript `pt` (merlin.Transcript) exactly
        where the reference draws them (:219-222, :276-278), then `TIPA::prove` and, as the reference does (:340), the
        aggregator verifies its own proof.  Returns (tipp_proof, instance dict)."""
        from . import tipa
        tipp = tipp or tipa.Tipp(self.ctx, self.curve)
        inst = self.agg_front(super_com, proofs, pub_inputs, pt=pt)
        proof = tipp.prove(srs, inst["left"], inst["right"], inst["twist"], inst["commitment"], inst["output"])
        if check:
            vk = tipa.verifier_key(self.ctx, self.curve, srs)
            assert tipp.verify(vk, inst["commitment"], inst["output"], inst["twist"], proof), "TIPA proof rejected (aggregation.rs:340)"
        return proof, inst

    def agg_front(self, super_com, proofs, pub_inputs, twist=None, s=None, t=None, pt=None):
        """aggregation.rs:138-330 up to the `TIPA::prove` call.  proofs: [cp_groth16.Proof] with one stage-0
        commitment each; pub_inputs: 3 ints; the Fiat-Shamir challenges either as ints (twist, s, t) or drawn from the
        merlin transcript `pt` at the reference's points.  Returns a dict with the
        TIPA instance (`output` = z_lr, `commitment` = com_lr, `twist`), the witness (`left`, `right`) and the 4 x 4
        `cross_terms`; raises AssertionError if the pairing-product equation of :265-269 fails."""
        ctx, F, ck, fc = self.ctx, self.F, self.ck, FrCodec(self.curve)
        r_mod = CURVE_PARAMS[self.curve]["r"]
        n = len(proofs)
        assert n == self.n
        cat = lambda xs: np.concatenate([np.asarray(x, np.uint8) for x in xs])
        a_vals, b_vals = cat([p.a for p in proofs]), cat([p.b for p in proofs])
        c_vals, d_vals = cat([p.c for p in proofs]), cat([p.ds[0] for p in proofs])
        # independent GPU calls of a phase go out together (one lane each), as in tipa.Tipp
        go = self.pool.submit
        x = [v % r_mod for v in pub_inputs]
        f_ab = go(self.com.commit_with_ip, ck, a_vals, b_vals)                                      # :167
        f_c = go(self.com.commit_only_left, ck, c_vals)                                             # :168
        f_in = go(ctx.points_lincomb, 1, self.s, fc.enc([1] + x), n)                                # :192-205
        f_cin = go(IppCom.lincomb, [(self.com_s[0], None), (self.com_s[1], x[0]), (self.com_s[2], x[1]), (self.com_s[3], x[2])])  # :171-174
        com_ab, com_c, com_d = f_ab.result(), f_c.result(), super_com
        if pt is not None:                                                                          # :219-222
            pt.append_serializable(b"AB-commitment", com_ab.serialize_uncompressed())
            pt.append_serializable(b"C-commitment", com_c.serialize_uncompressed())
            pt.append_serializable(b"D-commitment", com_d.serialize_uncompressed())
            twist = pt.challenge_scalar(b"r-random-fiatshamir", r_mod)
        prepared_input, com_prepared_input = f_in.result(), f_cin.result()
        tw = [fc.R % r_mod] * n                               # :224 structured_scalar_power, as Montgomery values R * twist^i
        for i in range(1, n):
            tw[i] = tw[i - 1] * twist % r_mod
        twb = fc.enc_canon(tw)
        # the five twisted G1 vectors (:236-242: five `scalar_pairing` sweeps with the same powers) as ONE sweep of 5 n
        # elements: one launch and one normalisation instead of five side by side
        g1b = ctx.g1_bytes
        swept = ctx.scalar_pairing(1, np.concatenate([a_vals, c_vals, d_vals, self.alpha, np.asarray(prepared_input, np.uint8)]),
                                   np.tile(twb, 5), 5 * n)
        a_r, c_r, d_r, alpha_r, input_r = (swept[j * n * g1b:(j + 1) * n * g1b] for j in range(5))
        f_cross = go(ctx.pairing_products, [a_r, input_r, d_r, c_r], [b_vals, self.h, self.delta0, self.delta1], n)   # :255-263
        f_ab_z = go(ctx.multi_pairing, alpha_r, self.beta, n)
        cross = f_cross.result()
        z = [[F.decode(cross[i, j]) for j in range(4)] for i in range(4)]
        z_alpha_beta = F.decode(f_ab_z.result())
        rhs = F.mul(F.mul(z_alpha_beta, z[1][1]), F.mul(z[2][2], z[3][3]))
        assert z[0][0] == rhs, "pairing-product equation of the twisted proofs does not hold (aggregation.rs:265-269)"
        if pt is not None:                                                                          # :276-278
            # Vec<Vec<PairingOutput>>: u64 length prefixes, elements canonical
            ser = (4).to_bytes(8, "little") + b"".join((4).to_bytes(8, "little") + b"".join(F.serialize(e) for e in row) for row in z)
            pt.append_serializable(b"cross-terms", ser)
            s = pt.challenge_scalar(b"s-random-fiatshamir", r_mod)
            t = pt.challenge_scalar(b"t-random-fiatshamir", r_mod)
        s2, s3, t2, t3 = s * s % r_mod, s * s * s % r_mod, t * t % r_mod, t * t * t % r_mod
        # left = A + S^s + D^(s^2) + C^(s^3), right = B + H^t + ... (:293-326: three `scalar_pairing` sweeps with a constant
        # scalar and element-wise additions each).  The three products of a side go out as ONE sweep of 3 n elements
        # (hk_scalar_pairing: endomorphism-split chains of 128 / 66 doubling steps, K lanes per element), then one
        # element-wise sum - instead of one joint 254-step chain per side (4.5 ms G1 / 13 ms G2).
        g2b = ctx.g2_bytes
        rep = lambda ks: np.concatenate([np.tile(fc.enc1(k), n) for k in ks])
        f_l = go(ctx.scalar_pairing, 1, np.concatenate([np.asarray(prepared_input, np.uint8), d_vals, c_vals]), rep((s, s2, s3)), 3 * n)
        f_r = go(ctx.scalar_pairing, 2, np.concatenate([self.h, self.delta0, self.delta1]), rep((t, t2, t3)), 3 * n)
        f_lr = go(IppCom.lincomb, [(com_ab, None), (com_prepared_input, s), (com_d, s2), (com_c, s3),
                                   (self.com_h, t), (self.com_delta0, t2), (self.com_delta1, t3)])  # :328-332
        # z_lr = twisted_inner_product(left, right) (:334) = prod_ij cross[i][j]^(s^i t^j) by bilinearity - the cross terms
        # ARE the pairings of the twisted components (:255-263) - so it costs 16 GT powers in one batched call instead of
        # another element-wise sweep and another N-pair multi-pairing; the same GT element, bit for bit
        exps = [pow(s, i, r_mod) * pow(t, j, r_mod) % r_mod for i in range(4) for j in range(4)]
        f_z = go(ctx.gt_pow_prod, np.frombuffer(b"".join(F.encode(z[i][j]) for i in range(4) for j in range(4)), np.uint8),
                 fc.enc(exps), 16)
        ones = fc.enc([1, 1, 1, 1])
        split = lambda buf, sz: [buf[j * n * sz:(j + 1) * n * sz] for j in range(3)]
        f_left = go(lambda: ctx.points_lincomb(1, [a_vals] + split(f_l.result(), g1b), ones, n))
        f_right = go(lambda: ctx.points_lincomb(2, [b_vals] + split(f_r.result(), g2b), ones, n))
        z_lr = F.decode(f_z.result()[0])
        left, right = f_left.result(), f_right.result()
        com_lr = f_lr.result()
        return dict(size=n, output=z_lr, commitment=com_lr, twist=twist, left=left, right=right, cross_terms=z,
                    com_ab=com_ab, com_c=com_c, prepared_input=prepared_input)
