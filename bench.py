#!/usr/bin/env python3
"""bench.py — subcircuit Groth16 proofs/sec on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic subcircuits per GPU: for every
subcircuit one stage-0 commit (hk_commit) and one stage-1 prove (hk_prove), exactly the unit of work
`process_stage0_request` / `process_stage1_request` do in the reference
(distributed-prover/src/worker.rs:91-146,150-195).  Subcircuits are sharded contiguously over ranks
(mpi-snark/src/bin/node.rs:471-472,490-493); there is no data-path collective (weak scaling: every
GPU proves `--subcircuits` of them per step).  Inputs (proving key with shift tables, matrices,
assignments) are resident in HBM before the timed region; outputs are 3 affine points per proof.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="big-merkle-64x32",
                    help="workload (hekaton_system_amd/workload.py CONFIGS); default = BASELINE configs[1]")
    ap.add_argument("--curve", default="bn254", help="bn254 is what the reference instantiates (SURVEY F1)")
    ap.add_argument("--subcircuits", type=int, default=64,
                    help="subcircuits per GPU per step (default: the 64 subcircuits of BASELINE configs[1], "
                         "big-merkle N=64; a step is that whole batch, proved 8 at a time)")
    ap.add_argument("--threads", type=int, default=8, help="host threads (= GPU lanes) proving concurrently")
    ap.add_argument("--witnesses", type=int, default=4, help="distinct assignments cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-inputs", action="store_true",
                    help="hand every assignment over from host memory (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--device", type=int, default=None,
                    help="force a device index (rehearsing N > 1 ranks on a 1-GPU box with --backend gloo)")
    return ap.parse_args()


def cpu_baseline(args, circ, pk_host, fc):
    """The CPU restatement of the ark-ec/ark-poly path (oracle/c, kind "port") timed on this box's
    host cores over a bounded sample of the same workload: three subcircuits, commit + prove each."""
    import shutil
    import tempfile
    from oracle import c_oracle
    lib = None
    if shutil.which("g++"):
        try:
            lib = c_oracle.build(native=True, out=os.path.join(tempfile.mkdtemp(prefix="hk_oracle_"), "libhk_oracle.so"))
        except Exception as e:       # noqa: BLE001
            log("native oracle build failed, using prebuilt:", e)
    co = c_oracle.COracle(args.curve, lib_path=lib)
    cores = co.threads()
    view = co.pk_view(**pk_host["points"])
    A, B, C = pk_host["matrices"]
    kap = fc.enc([7])
    n_sample = 3                                   # ~15 s of CPU work on the GPU box's host cores
    inputs = []
    for seed in range(1, n_sample + 1):
        circ.set_witness_seed(seed)
        inputs.append((circ.full_assignment_bytes(), circ.stage0_witness_bytes()))
    t0 = time.time()
    for z, w0 in inputs:
        co.commit(view, 0, w0, kap)
        co.prove(view, A, B, C, circ.N_INST, circ.n_c, z, fc.enc1(11), fc.enc1(13), kap)
    dt = time.time() - t0
    return {"value": n_sample / dt, "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": "%d subcircuits (1 commit + 1 prove each) of %s, %.1f s" % (n_sample, args.config, dt)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    from hekaton_system_amd import capi          # first: exports GPU_MAX_HW_QUEUES before anything initialises HIP
    import torch
    import torch.distributed as dist
    dev = local_rank if args.device is None else args.device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    from hekaton_system_amd.cp_groth16 import FrCodec, SeededRng, generate_parameters
    from hekaton_system_amd.workload import make_config

    ctx = capi.Context(args.curve, dev)
    fc = FrCodec(args.curve)
    circ = make_config(args.curve, args.config)
    log("rank %d: setup %s n_c=%d n_v=%d" % (rank, args.config, circ.n_c, circ.n_v))
    t0 = time.time()
    keep_host = (rank == 0 and world == 1 and not args.no_cpu_baseline)
    pk, td = generate_parameters(circ, args.curve, SeededRng(b"HEKATON1" * 4), ctx, keep_on_device=not keep_host)
    dpk = pk.upload(ctx)
    pk_host = None
    if keep_host:
        pk_host = {"points": dict(a_g=pk.a_g, b_g=pk.b_g, b_h=pk.b_h, h_g=pk.h_g, ck_stages=pk.ck.deltas_abc_g,
                                  deltas_g=pk.deltas_g, last_delta_h=pk.vk.last_delta_h, alpha_g=pk.vk.alpha_g,
                                  beta_g=pk.beta_g, beta_h=pk.vk.beta_h),
                   "matrices": pk.matrices}
    else:
        for b in (pk.a_g, pk.b_g, pk.b_h, pk.h_g):
            if isinstance(b, capi.DeviceBuffer):
                b.free()
    log("rank %d: key generated + resident in %.1f s" % (rank, time.time() - t0))
    # assignments resident in HBM
    zs, w0s = [], []
    for k in range(args.witnesses):
        circ.set_witness_seed(1000 * rank + k + 1)
        zb, wb = circ.full_assignment_bytes(), circ.stage0_witness_bytes()
        zs.append(zb if args.host_inputs else capi.DeviceBuffer.from_host(ctx, zb))
        w0s.append(wb if args.host_inputs else capi.DeviceBuffer.from_host(ctx, wb))
    r_b, s_b, kap = fc.enc1(0x1234567), fc.enc1(0x7654321), fc.enc([0x5555])
    ctx.set_profiling(True)
    accum_ms, accum_n, accum_h, phase = [], [], [], {}

    def one(i):
        k = i % args.witnesses
        dpk.commit(0, w0s[k], kap, n=circ.n0)
        tc = ctx.last_timings()
        out = dpk.prove(zs[k], r_b, s_b, kap, n_v=circ.n_v)
        t = ctx.last_timings()
        t["accum_kernel_ms"] += tc["accum_kernel_ms"]            # the commit's launch of the same kernel
        t["accum_kernel_launches"] += tc["accum_kernel_launches"]
        return out, t

    pool = ThreadPoolExecutor(max_workers=args.threads)

    def step(record):
        res = list(pool.map(one, range(args.subcircuits)))
        if record:
            for _, t in res:
                accum_ms.append(t["accum_kernel_ms"])
                accum_n.append(t["accum_kernel_launches"])
                accum_h.append(t["accum_h_ms"])
                for key, v in t.items():
                    phase[key] = phase.get(key, 0.0) + v
        return res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        last = step(True)
    ctx.sync()
    barrier()
    dt = time.time() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    proofs = world * args.subcircuits * args.steps
    if rank == 0:
        m = 1
        while m < circ.n_c + circ.N_INST:
            m *= 2
        g1 = ctx.g1_bytes
        # dominant kernel = k_msm_accum0<Fq> (bucket accumulation).  It is launched 5 times per subcircuit:
        # H query (m-1 dense terms), A / B1 / L queries (n_v-1, n_v-1, n1 terms) and the stage-0 commitment.
        # Algorithmic bytes per launch = terms * (32 + S1)  (SURVEY.md §8d "MSM-G1 = n*(32+S1)"), averaged
        # over the same launches whose durations are averaged — the population rocprofv3 --stats averages.
        n1 = circ.n_v - circ.N_INST - circ.n0
        fq_limbs = ctx.fq_bytes // 4
        nwin = (ctx.fr_bytes * 8 + 1 + 15) // 16 if args.curve == "bls12_381" else 16      # ceil((bits+2)/16): 16 / 17
        terms = [m - 1, circ.n_v - 1, circ.n_v - 1, n1, circ.n0]
        alg_bytes = sum(terms) * (32 + g1) / len(terms)
        avg_ms = float(np.sum(accum_ms) / max(1, np.sum(accum_n))) if accum_ms else float("nan")
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        h_ms = float(np.mean(accum_h)) if accum_h else float("nan")
        nprov = max(1, len(accum_ms))
        # HBM traffic of the same kernel from rocprofv3 PMC passes (profiles/, collected offline with
        # `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` on this command; cannot be read live)
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "r01_pmc_accum0.json")) as f:
                pmc = json.load(f)
            if args.config == "big-merkle-64x32" and args.curve == "bn254":
                traffic = (pmc["FETCH_SIZE_KiB_avg"] + pmc["WRITE_SIZE_KiB_avg"]) * 1024.0
        except Exception:       # noqa: BLE001
            pass
        out = {
            "metric": "subcircuit Groth16 proofs/sec (whole node), big-merkle",
            "value": proofs / dt, "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (Montgomery mod-p integers)",
            "data": "synthetic (SHA-like 85%% small / 15%% full-width witnesses, genuine Groth16 SRS from a "
                    "seeded trapdoor; %d distinct assignments cycled%s)" % (
                        args.witnesses, "; assignments copied from pageable host memory per proof" if args.host_inputs else ""),
            "config": {"workload": args.config, "curve": args.curve, "subcircuits_per_gpu_per_step": args.subcircuits,
                       "n_constraints": circ.n_c, "n_variables": circ.n_v, "domain": m,
                       "host_threads_per_gpu": args.threads, "sharding": "subcircuits/%d per rank" % world},
            "roofline": {"bound": "hbm", "kernel": "k_msm_accum0<Fp<%s>> (bucket accumulation; avg over its 5 launches per subcircuit)" % ("Bn254FqP" if args.curve == "bn254" else "Bls381FqP"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms,
                         "h_query_launch": {"alg_bytes": (m - 1) * (32 + g1), "avg_ms": h_ms,
                                            "achieved": (m - 1) * (32 + g1) / (h_ms * 1e-3) / 1e9},
                         # the bound that actually binds: integer multiply issue.  One mixed add = 10 Montgomery
                         # products of 2*N^2 v_mad_u64_u32 each (N = 8 / 12 limbs); peak = the measured chip-wide
                         # v_mad_u64_u32 rate (tools/ubench.hip: 24.7 Top/s) / mads per product
                         "valu": {"unit": "G field mults/s",
                                  "achieved": (m - 1) * nwin * 10 / (h_ms * 1e-3) / 1e9,
                                  "peak": 24.7e3 / (2 * (fq_limbs ** 2)),
                                  "frac": (m - 1) * nwin * 10 / (h_ms * 1e-3) / 1e9 / (24.7e3 / (2 * (fq_limbs ** 2))),
                                  "note": "H-query launch: (m-1) scalars x %d signed 16-bit digits x 10 products per mixed add" % nwin}},
            "phase_ms_per_proof": {k: v / nprov for k, v in phase.items() if k.endswith("_ms")},
        }
        if keep_host:
            try:
                out["cpu_baseline"] = cpu_baseline(args, circ, pk_host, fc)
            except Exception as e:       # noqa: BLE001
                log("cpu baseline failed:", e)
                out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
