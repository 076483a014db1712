#!/usr/bin/env python3
"""Generates tests/golden/*.json from the big-int oracle (oracle/pyref).

The reference holds no golden vectors for this path and cannot be run here, so these
vectors are produced by OUR restatement (pinned by the pairing equation + trapdoor checks, see
tests/test_oracle_py.py) — "parity unpinned" by reference constants.  All byte strings use the C-ABI
layouts of include/hekaton.h (Montgomery LE limbs, packed affine, infinity = zeros), hex-encoded.

    python tests/golden/gen_golden.py        # rewrites the json files next to this script
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle.pyref import curve, groth16                      # noqa: E402
from oracle.pyref.codec import Codec                         # noqa: E402
from oracle.pyref.params import CURVES                       # noqa: E402
from oracle.pyref.poly import Domain                         # noqa: E402
from tests.util import running_bases, mixed_scalars, csr_from_rows, synthetic_r1cs   # noqa: E402

SEED = 0x48454B41544F4E31          # "HEKATON1" 


def hx(b):
    return bytes(b).hex()


def field_and_curve_kats():
    out = {}
    for name, cp in CURVES.items():
        cd = Codec(cp)
        rnd = random.Random(SEED ^ cp.cid)
        F2 = curve.Fq2(cp.q)
        e2 = lambda x: cd.fq_mont(x[0]) + cd.fq_mont(x[1])
        ent = {"fr": [], "fq": [], "fq2": [], "g1": [], "g2": []}
        for p, enc, key in ((cp.r, cd.fr_mont, "fr"), (cp.q, cd.fq_mont, "fq")):
            for a, b in [(0, 0), (1, p - 1), (p - 1, p - 1)] + [(rnd.randrange(p), rnd.randrange(p)) for _ in range(5)]:
                ent[key].append({"a": hx(enc(a)), "b": hx(enc(b)), "add": hx(enc(a + b)), "sub": hx(enc(a - b)),
                                 "mul": hx(enc(a * b)), "inv_a": hx(enc(pow(a, -1, p))) if a else None})
        for _ in range(4):
            a = (rnd.randrange(cp.q), rnd.randrange(cp.q)); b = (rnd.randrange(cp.q), rnd.randrange(cp.q))
            ent["fq2"].append({"a": hx(e2(a)), "b": hx(e2(b)), "mul": hx(e2(F2.mul(a, b))),
                               "sqr_a": hx(e2(F2.mul(a, a))), "inv_a": hx(e2(F2.inv(a)))})
        for key, G, enc in (("g1", curve.G1(cp), cd.g1), ("g2", curve.G2(cp), cd.g2)):
            for _ in range(3):
                k1, k2 = rnd.randrange(1, cp.r), rnd.randrange(1, cp.r)
                P, Q = G.mul(G.gen, k1), G.mul(G.gen, k2)
                ent[key].append({"p": hx(enc(P)), "q": hx(enc(Q)), "add": hx(enc(G.add(P, Q))),
                                 "dbl_p": hx(enc(G.dbl(P))), "k": hx(cd.fr_canon(k2)),
                                 "k_times_p": hx(enc(G.mul(P, k2)))})
        out[name] = ent
    return out


def msm_cases():
    out = {}
    for name, cp in CURVES.items():
        cd = Codec(cp)
        cases = []
        for group, G, enc, sizes in (("g1", curve.G1(cp), cd.g1_vec, [1, 2, 31, 32, 33, 200]),
                                     ("g2", curve.G2(cp), cd.g2_vec, [1, 33, 64])):
            for n in sizes:
                rnd = random.Random(SEED ^ (n * 7 + len(group) + cp.cid * 1000))
                bases = running_bases(G, n, s0=5)
                scalars = mixed_scalars(rnd, cp.r, n, dense=(n % 2 == 0))
                if n >= 31:
                    scalars[0], scalars[1], scalars[2] = 0, 1, cp.r - 1
                    bases[3] = None
                    bases[5] = bases[4]
                    bases[7] = G.neg(bases[6]); scalars[7] = scalars[6]
                want = G.msm(bases, scalars)
                e1 = cd.g1 if group == "g1" else cd.g2
                cases.append({"group": group, "n": n, "bases": hx(enc(bases)),
                              "scalars_mont": hx(cd.fr_vec_mont(scalars)),
                              "scalars_canon": hx(cd.fr_vec_canon(scalars)), "expect": hx(e1(want))})
        out[name] = cases
    return out


def ntt_cases():
    out = {}
    for name, cp in CURVES.items():
        cd = Codec(cp)
        cases = []
        for log_m in (1, 2, 3, 10):
            rnd = random.Random(SEED ^ (log_m + 31 * cp.cid))
            m = 1 << log_m
            x = [rnd.randrange(cp.r) for _ in range(m)]
            dom = Domain(cp, m)
            g = cp.fr_generator
            cases.append({"log_m": log_m, "input": hx(cd.fr_vec_mont(x)),
                          "fft": hx(cd.fr_vec_mont(dom.fft(x))), "ifft": hx(cd.fr_vec_mont(dom.ifft(x))),
                          "coset_fft": hx(cd.fr_vec_mont(dom.coset_fft(x, g))),
                          "coset_ifft": hx(cd.fr_vec_mont(dom.coset_ifft(x, g)))})
        out[name] = cases
    return out


def csr_hex(cd, rows):
    rp, col, val = csr_from_rows(cd, rows)
    return {"row_ptr": [int(x) for x in rp], "col": [int(x) for x in col], "val_mont": hx(val)}


def groth16_cases():
    """Full commit + prove on the reference's unit-test circuit shape (cp-groth16/src/lib.rs:30-100)
    and on a small synthetic ROM-shaped subcircuit; includes the trapdoor so tests can re-verify."""
    out = {}
    for name, cp in CURVES.items():
        cd = Codec(cp)
        cases = []
        for label in ("poly_eval_two_stage", "poly_eval_single_stage", "synthetic_two_stage"):
            rnd = random.Random(SEED ^ (hash(label) & 0xffff) ^ cp.cid)
            rnd = random.Random("%s-%s" % (label, name))
            r = cp.r
            if label.startswith("poly_eval"):
                poly = [rnd.randrange(r) for _ in range(10)] + [1]
                cs, inputs = groth16.poly_eval_circuit(cp, poly, rnd.randrange(r), two_stage=label.endswith("two_stage"))
            else:
                cs = synthetic_r1cs(cp, rnd, n_inst=4, n_free=24, n_c=60, two_stage_split=16)
                inputs = cs.instance[1:]
            td_args = dict(alpha=rnd.randrange(1, r), beta=rnd.randrange(1, r), gamma=rnd.randrange(1, r),
                           deltas=[rnd.randrange(1, r) for _ in cs.stage_ranges], t=rnd.randrange(2, r),
                           g1_scalar=5, g2_scalar=11)
            pk, td = groth16.generate_parameters(cp, cs, **td_args)
            kappas = [rnd.randrange(r) for _ in range(len(cs.stage_ranges) - 1)]
            r_, s_ = rnd.randrange(r), rnd.randrange(r)
            comms = [groth16.commit(cp, cs, pk, k, kappas[k]) for k in range(len(kappas))]
            proof = groth16.prove(cp, cs, pk, comms, kappas, r_, s_)
            assert groth16.verify_proof_trapdoor(cp, cs, pk, td, proof, kappas, r_, s_)
            A, B, C = cs.matrices()
            h = groth16.witness_map_from_matrices(cp, A, B, C, cs.num_instance, cs.num_constraints, cs.full_assignment())
            cases.append({
                "label": label, "n_inst": cs.num_instance, "n_constraints": cs.num_constraints,
                "stage_ranges": cs.stage_ranges, "trapdoor": {k: (v if not isinstance(v, list) else v) for k, v in td_args.items()},
                "public_inputs": [int(x) for x in inputs],
                "A": csr_hex(cd, A), "B": csr_hex(cd, B), "C": csr_hex(cd, C),
                "z_mont": hx(cd.fr_vec_mont(cs.full_assignment())),
                "h_mont": hx(cd.fr_vec_mont(h)),
                "pk": {"a_g": hx(cd.g1_vec(pk.a_g)), "b_g": hx(cd.g1_vec(pk.b_g)), "b_h": hx(cd.g2_vec(pk.b_h)),
                       "h_g": hx(cd.g1_vec(pk.h_g)), "ck": [hx(cd.g1_vec(v)) for v in pk.ck.deltas_abc_g],
                       "deltas_g": hx(cd.g1_vec(pk.deltas_g)), "last_delta_h": hx(cd.g2(pk.last_delta_h())),
                       "alpha_g": hx(cd.g1(pk.vk.alpha_g)), "beta_g": hx(cd.g1(pk.beta_g)),
                       "beta_h": hx(cd.g2(pk.vk.beta_h)), "gamma_h": hx(cd.g2(pk.vk.gamma_h)),
                       "gamma_abc_g": hx(cd.g1_vec(pk.vk.gamma_abc_g)), "deltas_h": hx(cd.g2_vec(pk.vk.deltas_h))},
                "r_mont": hx(cd.fr_mont(r_)), "s_mont": hx(cd.fr_mont(s_)),
                "kappas_mont": hx(cd.fr_vec_mont(kappas)),
                "comms": [hx(cd.g1(c)) for c in comms],
                "proof": {"a": hx(cd.g1(proof.a)), "b": hx(cd.g2(proof.b)), "c": hx(cd.g1(proof.c))},
            })
        out[name] = cases
    return out


def main():
    for fname, fn in (("field_curve_kat.json", field_and_curve_kats), ("msm.json", msm_cases),
                      ("ntt.json", ntt_cases), ("groth16.json", groth16_cases)):
        data = fn()
        with open(os.path.join(HERE, fname), "w") as f:
            json.dump(data, f, separators=(",", ":"))
        print("wrote", fname, os.path.getsize(os.path.join(HERE, fname)), "bytes")


if __name__ == "__main__":
    main()
