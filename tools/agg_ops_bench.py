#!/usr/bin/env python3
"""Latency of the aggregation-sized primitives: the KZG-opening MSMs of TIPA
(`G::Group::msm(&srs_powers, &coeffs)`, distributed-prover/src/kzg.rs:151-152 — N and 2N terms for N subcircuits)
through hk_msm_g1 / hk_msm_g2 over caller-supplied bases, and `scalar_pairing` (pairing_ops.rs:32-39) through
hk_scalar_pairing_g1 / g2; `resident` = the same MSM through hk_bases_upload / hk_msm_bases (bases uploaded once with
their shift tables, as a static SRS would be).  Inputs resident in HBM; the CPU column is the oracle's ark-style Pippenger on this
box's host threads (test infrastructure, timed here only as the baseline beside it)."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from hekaton_system_amd import capi  # noqa: E402
from hekaton_system_amd.cp_groth16 import CURVE_PARAMS, FrCodec  # noqa: E402


def main():
    curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
    ctx = capi.Context(curve, 0)
    fc = FrCodec(curve)
    p = CURVE_PARAMS[curve]
    try:
        from oracle.c_oracle import COracle
        co = COracle(curve)
    except Exception:       # noqa: BLE001
        co = None
    rnd = random.Random(1)
    cores = os.cpu_count() or 1
    cpu_threads = min(32, cores)               # the reference's convention: 32 threads per task (run_single_bench:5-7)
    if co is not None:
        co.set_threads(cpu_threads)
    print("# CPU columns: oracle/c restatement on %d of this box's %d host threads" % (cpu_threads, cores))
    # ---- pairings: hk_multi_pairing (one product) and hk_pairing_products (the 4 x 4 cross terms, aggregation.rs:255-263)
    print("%-8s %8s %14s %16s %14s %18s" % ("pairing", "n", "1 product ms", "16 products ms", "cpu 1 prod ms", "cpu 16 prods ms"))
    gen1, gen2 = fc.g1(p["g1"]), fc.g2(p["g2"])
    only_front = bool(os.environ.get("HK_AGG_FRONT_TRACE"))
    for n in () if only_front else (64, 256, 1024, 4096):
        v1 = [ctx.fixed_base(1, gen1, fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)])) for _ in range(4)]
        v2 = [ctx.fixed_base(2, gen2, fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)])) for _ in range(4)]
        d1 = [capi.DeviceBuffer.from_host(ctx, v) for v in v1]
        d2 = [capi.DeviceBuffer.from_host(ctx, v) for v in v2]
        ctx.multi_pairing(d1[0], d2[0], n=n)
        reps = 5
        t0 = time.time()
        for _ in range(reps):
            one = ctx.multi_pairing(d1[0], d2[0], n=n)
        t_one = (time.time() - t0) / reps
        ctx.pairing_products(d1, d2, n=n)
        t0 = time.time()
        for _ in range(reps):
            allp = ctx.pairing_products(d1, d2, n=n)
        t_all = (time.time() - t0) / reps
        assert np.array_equal(allp[0, 0], one)
        t_c1 = t_c16 = float("nan")
        if co is not None:
            t0 = time.time()
            ref = co.multi_pairing(v1[0], v2[0], n=n)
            t_c1 = time.time() - t0
            assert np.array_equal(ref, one), "GPU pairing differs from the CPU restatement"
            if n <= 1024:
                t0 = time.time()
                for a in v1:
                    for b in v2:
                        co.multi_pairing(a, b, n=n)
                t_c16 = time.time() - t0
        print("%-8s %8d %14.3f %16.3f %14.3f %18.3f" % (curve, n, t_one * 1e3, t_all * 1e3, t_c1 * 1e3, t_c16 * 1e3))
        for b in d1 + d2:
            b.free()
    # ---- TIPP prove / verify (tipa.py): the whole GIPA recursion + KZG openings on random vectors
    from hekaton_system_amd import aggregation as agg, tipa
    print("%-8s %8s %14s %14s %14s" % ("tipp", "n", "setup ms", "prove ms", "verify ms"))
    for n in () if only_front else (64, 256, 1024):
        t0 = time.time()
        srs = tipa.setup(ctx, curve, n, rnd.randrange(2, p["r"]), rnd.randrange(2, p["r"]))
        t_setup = time.time() - t0
        A = ctx.fixed_base(1, gen1, fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)]))
        B = ctx.fixed_base(2, gen2, fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)]))
        T = tipa.Tipp(ctx, curve)
        com = T.com.commit_with_ip(srs.ck, A, B)
        twist = rnd.randrange(2, p["r"])
        tw = [pow(twist, i, p["r"]) for i in range(n)]
        z = T.F.decode(ctx.multi_pairing(ctx.scalar_pairing(1, A, fc.enc(tw), n=n), B, n=n))
        T.prove(srs, A, B, twist, com, z)
        t0 = time.time()
        proof = T.prove(srs, A, B, twist, com, z)
        t_prove = time.time() - t0
        vk = tipa.verifier_key(ctx, curve, srs)
        T.verify(vk, com, z, twist, proof)                  # warm (the first call creates the lanes of its ten concurrent calls)
        t0 = time.time()
        ok = T.verify(vk, com, z, twist, proof)
        t_verify = time.time() - t0
        assert ok
        print("%-8s %8d %14.1f %14.1f %14.1f" % (curve, n, t_setup * 1e3, t_prove * 1e3, t_verify * 1e3))
        print("         prove: setup %.1f ms, rounds %.1f ms, openings %.1f ms" % tuple(x * 1e3 for x in T.phase_times))
        # (rounds go two per pairing pass: the second of a pair shows 0.0 ms of pairings)
        print("         rounds (m: pairings / host / folds ms): " + "  ".join(
            "%d: %.1f/%.1f/%.1f" % (m, a * 1e3, b * 1e3, c * 1e3) for m, a, b, c in T.round_times))
        for rb in srs.resident.values():
            rb.free()
    # ---- the aggregator's whole serial tail on REAL proofs (tiny subcircuits, 5 key classes by the big-merkle index map):
    # AggProvingKey::new (7 commitments), the super commitment, agg_subcircuit_proofs = front half + TIPA prove + verify
    from hekaton_system_amd.chacha import ChaCha12Rng
    from hekaton_system_amd.cp_groth16 import Proof, SeededRng, generate_parameters
    from hekaton_system_amd.merlin import Transcript as Merlin
    from hekaton_system_amd.workload import make_config, representative_subcircuit, unique_subcircuits
    print("%-8s %8s %14s %14s %12s %12s %12s %12s" % ("job", "n", "agg key ms", "super com ms", "front ms", "prove ms", "verify ms", "total ms"))
    for n in ((int(os.environ["HK_AGG_FRONT_TRACE"]),) if os.environ.get("HK_AGG_FRONT_TRACE") else (64, 1024)):
        keys = {}
        for rep in unique_subcircuits("big-merkle", n):
            circ = make_config(curve, "tiny", rep)
            pk, _td = generate_parameters(circ, curve, SeededRng(bytes([rep % 251 + 1]) * 32), ctx)
            circ.set_witness_seed(5)
            keys[rep] = (circ, pk, pk.upload(ctx), circ.full_assignment_bytes(), circ.stage0_witness_bytes())
        proofs, coms, vks, pub = [], [], [], None
        for idx in range(n):
            circ, pk, dpk, zb, w0 = keys[representative_subcircuit("big-merkle", n, idx)]
            pub = pub or circ.assignment_ints()[1:4]
            kappa = ChaCha12Rng(idx.to_bytes(4, "little") * 8).fr(p["r"])
            com = dpk.commit(0, w0, fc.enc1(kappa))
            a, b, c = dpk.prove(zb, fc.enc1(rnd.randrange(p["r"])), fc.enc1(rnd.randrange(p["r"])), fc.enc([kappa]), n_v=circ.n_v)
            proofs.append(Proof(a, b, c, [com])); coms.append(com); vks.append(pk.vk)
        srs = tipa.setup(ctx, curve, n, rnd.randrange(2, p["r"]), rnd.randrange(2, p["r"]))
        t0 = time.time()
        apk = agg.AggProvingKey(ctx, curve, srs.ck, vks)
        t_key = time.time() - t0
        t0 = time.time()
        super_com = apk.com.commit_only_left(srs.ck, np.concatenate(coms))
        t_super = time.time() - t0
        T = tipa.Tipp(ctx, curve)
        vk = tipa.verifier_key(ctx, curve, srs)
        apk.agg_subcircuit_proofs(Merlin(b"bench"), super_com, proofs, pub, srs, tipp=T)          # warm
        if os.environ.get("HK_AGG_FRONT_TRACE") == str(n):    # under `rocprofv3 --kernel-trace`: the front half alone after a
            time.sleep(0.3)                                   # pause (tools/tipp_timeline.py show <csv> prints what follows it)
            apk.agg_front(super_com, proofs, pub, pt=Merlin(b"bench"))
            return
        if os.environ.get("HK_AGG_CPROFILE"):                 # where the HOST time of the front half and the verifier goes
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            inst_ = apk.agg_front(super_com, proofs, pub, pt=Merlin(b"bench"))
            pr.disable()
            pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(18)
        t0 = time.time()
        inst = apk.agg_front(super_com, proofs, pub, pt=Merlin(b"bench"))
        t1 = time.time()
        proof = T.prove(srs, inst["left"], inst["right"], inst["twist"], inst["commitment"], inst["output"])
        t2 = time.time()
        assert T.verify(vk, inst["commitment"], inst["output"], inst["twist"], proof)
        t3 = time.time()
        print("%-8s %8d %14.1f %14.1f %12.1f %12.1f %12.1f %12.1f" % (curve, n, t_key * 1e3, t_super * 1e3, (t1 - t0) * 1e3,
                                                                   (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3))
        for _c, _pk, dpk, _z, _w in keys.values():
            dpk.free()
        for rb in srs.resident.values():
            rb.free()
    print("%-8s %8s %12s %14s %12s %14s %18s" % ("group", "n", "msm ms", "resident ms", "cpu msm ms", "scalar_pair ms", "cpu scalar_pair ms"))
    for group in (1, 2):
        gen = fc.g1(p["g1"]) if group == 1 else fc.g2(p["g2"])
        pb = ctx.g1_bytes if group == 1 else ctx.g2_bytes
        for n in (64, 256, 1024, 2048, 8192, 65536):
            ks = fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)])
            bases = capi.DeviceBuffer(ctx, n * pb)
            ctx.fixed_base(group, gen, ks, out=bases)
            scal_h = fc.enc([rnd.randrange(p["r"]) for _ in range(n)])
            scal = capi.DeviceBuffer.from_host(ctx, scal_h)
            msm = ctx.msm_g1 if group == 1 else ctx.msm_g2
            msm(bases, scal, n_bases=n, n_scalars=n)
            reps = 20 if n <= 8192 else 5
            t0 = time.time()
            for _ in range(reps):
                msm(bases, scal, n_bases=n, n_scalars=n)
            t_msm = (time.time() - t0) / reps
            rb = ctx.bases_upload(group, bases, n=n)
            rb.msm(scal, n_scalars=n)
            t0 = time.time()
            for _ in range(reps):
                rb.msm(scal, n_scalars=n)
            t_res = (time.time() - t0) / reps
            rb.free()
            ctx.scalar_pairing(group, bases, scal, n=n)
            t0 = time.time()
            for _ in range(reps):
                ctx.scalar_pairing(group, bases, scal, n=n)
            t_sp = (time.time() - t0) / reps
            t_cpu = t_csp = float("nan")
            if co is not None and n <= 8192:
                bh = bases.to_host()
                t0 = time.time()
                co.msm(group, bh, scal_h)
                t_cpu = time.time() - t0
                t0 = time.time()
                co.scalar_mul_each(group, bh, scal_h)
                t_csp = time.time() - t0
            print("%-8s %8d %12.3f %14.3f %12.3f %14.3f %18.3f" % ("G%d" % group, n, t_msm * 1e3, t_res * 1e3, t_cpu * 1e3, t_sp * 1e3, t_csp * 1e3))
            bases.free()
            scal.free()


if __name__ == "__main__":
    main()
