"""Shared helpers for the parity tests (oracle side)."""
import random

import numpy as np

from oracle.pyref import curve
from oracle.pyref.codec import Codec
from oracle.pyref.params import CURVES


def running_bases(G, n, s0=3):
    """P_i = (s0 + i) * gen, distinct non-infinity points (SURVEY.md §8d synthetic bases)."""
    out = []
    P = G.mul(G.gen, s0)
    for _ in range(n):
        out.append(P)
        P = G.add(P, G.gen)
    return out


def mixed_scalars(rnd, r, n, dense=False):
    """85 % in {0,1}, 15 % uniform (SHA-like witness mix) or all uniform."""
    out = []
    for _ in range(n):
        if dense or rnd.random() < 0.15:
            out.append(rnd.randrange(r))
        else:
            out.append(rnd.randrange(2))
    return out


def csr_from_rows(cd, rows):
    """[(coeff, col)] rows -> (row_ptr u64, col u32, val Montgomery bytes)."""
    row_ptr = [0]
    cols, vals = [], []
    for row in rows:
        for c, j in row:
            cols.append(j)
            vals.append(c)
        row_ptr.append(len(cols))
    return (np.array(row_ptr, dtype=np.uint64), np.array(cols, dtype=np.uint32), cd.fr_vec_mont(vals))


def synthetic_r1cs(cp, rnd, n_inst, n_free, n_c, two_stage_split=None, nnz=(3, 2)):
    """Satisfiable synthetic R1CS in the shape SURVEY.md §8(d) prescribes: a few non-zeros per row in
    A and B with small +-1 / +-2^k coefficients, C row i = one fresh witness holding <A_i,z><B_i,z>.
    Stage 0 holds the first `two_stage_split` free witnesses when given (two-stage circuit)."""
    from oracle.pyref.groth16 import R1CS
    r = cp.r
    cs = R1CS(r)
    coeffs = [1, 1, 1, r - 1, 2, r - 2, 1 << 7, 1 << 31]
    variables = ["one"]
    values = [1]

    def lc(k):
        out = []
        for _ in range(k):
            idx = rnd.randrange(len(variables))
            out.append((rnd.choice(coeffs), variables[idx]))
        return out

    def ev(l):
        return sum(c * values[variables.index(v)] for c, v in l) % r

    stage0 = two_stage_split or 0
    cs.begin_stage()
    for k in range(n_free):
        if stage0 and k == stage0:
            cs.end_stage()
            cs.begin_stage()
            for _ in range(n_inst - 1):
                val = rnd.randrange(r)
                variables.append(cs.alloc_instance(val)); values.append(val)
        val = rnd.randrange(r) if rnd.random() < 0.3 else rnd.randrange(2)
        variables.append(cs.alloc_witness(val)); values.append(val)
    if not stage0:
        for _ in range(n_inst - 1):
            val = rnd.randrange(r)
            variables.append(cs.alloc_instance(val)); values.append(val)
    # index lookup would be O(n) per term: keep a dict instead
    pos = {v if isinstance(v, str) else tuple(v): i for i, v in enumerate(variables)}

    def ev_fast(l):
        return sum(c * values[pos[v if isinstance(v, str) else tuple(v)]] for c, v in l) % r

    for _ in range(n_c):
        a = lc(nnz[0]); b = lc(nnz[1])
        val = ev_fast(a) * ev_fast(b) % r
        w = cs.alloc_witness(val)
        variables.append(w); values.append(val); pos[tuple(w)] = len(variables) - 1
        cs.enforce(a, b, [(1, w)])
    cs.end_stage()
    return cs


def pk_upload_from_oracle(ctx, cd, pk, cs):
    """Oracle ProvingKey + finalized R1CS -> hk_pk resident on the device."""
    A, B, C = cs.matrices()
    return ctx.pk_upload(
        a_g=cd.g1_vec(pk.a_g), b_g=cd.g1_vec(pk.b_g), b_h=cd.g2_vec(pk.b_h), h_g=cd.g1_vec(pk.h_g),
        ck_stages=[cd.g1_vec(v) for v in pk.ck.deltas_abc_g],
        deltas_g=cd.g1_vec(pk.deltas_g), last_delta_h=cd.g2_vec([pk.last_delta_h()]),
        alpha_g=cd.g1_vec([pk.vk.alpha_g]), beta_g=cd.g1_vec([pk.beta_g]), beta_h=cd.g2_vec([pk.vk.beta_h]),
        matrices=(csr_from_rows(cd, A), csr_from_rows(cd, B), csr_from_rows(cd, C)),
        n_inst=cs.num_instance, n_constraints=cs.num_constraints)
