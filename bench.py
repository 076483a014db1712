#!/usr/bin/env python3
"""bench.py — subcircuit Groth16 proofs/sec on N MI355X (BASELINE.json metric).

A "step" is one job of the reference's `node work` flow over one batch of synthetic subcircuits per GPU
(mpi-snark/src/bin/node.rs:461-621), without the coordinator's own work:

    round 1  every subcircuit of the rank's shard: stage-0 commitment   (worker.rs:91-146  -> hk_commit)
             gather the 104-byte Stage0Response records of all ranks    (node.rs:500-506)
    round 2  every subcircuit of the rank's shard: stage-1 proof        (worker.rs:150-195 -> hk_prove)
             gather the 328-byte Stage1Response records of all ranks    (node.rs:526-533)

Subcircuits are sharded contiguously over ranks (node.rs:471-472,490-493); the two gathers are the only
exchange steps (torch.distributed all_gather: RCCL on the GPUs, gloo for the CPU rehearsal).  Every rank holds
the proving-key classes its shard needs, chosen per subcircuit index by the circuit family's
`representative_subcircuit` (big-merkle: tree_hash_circuit.rs:192-216 - five classes).  Weak scaling: every GPU
proves `--subcircuits` subcircuits per step (BASELINE configs[1] = 64 on one GPU, configs[2] = 512 on eight).
Inputs (keys with shift tables, matrices, assignments) are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 2 --warmup 1
    python bench.py --gpus 4                      # starts 4 rank processes itself (rank r -> device r)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 2 --device 0           # rehearsal: 2 ranks on ONE GPU (gloo gathers)
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# measured ceilings of the field multiply loop (tools/ubench.hip, profiles/r01_ubench_asm_addsub.txt): G products/s
VALU_PRODUCT_CEILING = {"bn254": 134.0, "bls12_381": 63.0}
VMAD_RATE_TOPS = 24.7      # chip-wide v_mad_u64_u32 issue rate (profiles/r01_ubench_instruction_rates.txt)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="big-merkle-64x32",
                    help="workload (hekaton_system_amd/workload.py CONFIGS); default = BASELINE configs[1]")
    ap.add_argument("--curve", default="bn254", help="bn254 is what the reference instantiates (SURVEY F1)")
    ap.add_argument("--subcircuits", type=int, default=64,
                    help="subcircuits per GPU per step (default: the 64 subcircuits of BASELINE configs[1])")
    ap.add_argument("--threads", type=int, default=8, help="host threads (= GPU lanes) proving concurrently")
    ap.add_argument("--witnesses", type=int, default=4, help="distinct assignments per proving-key class")
    ap.add_argument("--single-class", action="store_true",
                    help="prove every subcircuit against ONE proving-key class (debug; the default holds every class "
                         "of the rank's shard resident and picks by the reference's index -> class map)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the BLS12-381 secondary measurement (N = 1 only)")
    ap.add_argument("--no-synthesis-leg", action="store_true",
                    help="skip the `with_synthesis` object (N = 1 only): real SHA-256 big-merkle subcircuits whose "
                         "assignments are generated INSIDE the timed step - the reference's timer includes synthesis "
                         "(node.rs:589-596, prover.rs:70-75)")
    ap.add_argument("--no-batch-commit", action="store_true",
                    help="round 1 as one hk_commit per subcircuit instead of one hk_commit_batch per key class (A/B)")
    ap.add_argument("--no-verify", action="store_true", help="skip the post-timing checks of the timed proofs")
    ap.add_argument("--no-e2e", action="store_true", help="skip the post-timing aggregation of the last step's proofs")
    ap.add_argument("--witness-gen", action="store_true",
                    help="real-SHA configs only: generate every subcircuit's witness inside the step, on the GPU, from the "
                         "subcircuit's inputs (hk_wprog_run: the class's word program + column map, csrc/witness.cuh); the "
                         "reference's timed region includes its synthesis (prover.rs:70-75)")
    ap.add_argument("--host-inputs", action="store_true",
                    help="hand every assignment over from host memory (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--backend", default="auto",
                    help="torch.distributed backend for N > 1: nccl (= RCCL), gloo, or auto (nccl unless --device "
                         "puts several ranks on one GPU)")
    ap.add_argument("--device", type=int, default=None,
                    help="force a device index for every rank (rehearsing N > 1 ranks on a 1-GPU box)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------- self-launch
def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (rank r -> device r) BEFORE
    anything in this process touches HIP, forward their output, return the worst exit code.  The parent never
    initialises the GPU (a process that has must not exec or fork workers on this pool)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in procs:              # one rank died: the others would wait in a collective forever
                        q.terminate()
            time.sleep(0.2)
    except KeyboardInterrupt:
        for p in procs:
            p.terminate()
        rc = 130
    return rc


# ------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(args, circ, pk_host, fc):
    """The CPU restatement of the ark-ec/ark-poly path (oracle/c, kind "port") on this box's host cores, run the
    way the reference fills a node: floor(cores / 32) proofs in flight, 32 threads each (node.rs:745-795 with
    slurm_scripts/run_single_bench:5-7).  Also the latency of one proof alone with 32 threads."""
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
    cores = os.cpu_count() or 1
    per_task = min(32, cores)
    co.set_threads(per_task)
    conc = max(1, cores // 32)
    view = co.pk_view(**pk_host["points"])
    A, B, C = pk_host["matrices"]
    kap = fc.enc([7])
    inputs = []
    for seed in range(1, min(conc, 4) + 1):
        circ.set_witness_seed(seed)
        inputs.append((circ.full_assignment_bytes(), circ.stage0_witness_bytes()))

    def one(i):
        z, w0 = inputs[i % len(inputs)]
        co.commit(view, 0, w0, kap)
        co.prove(view, A, B, C, circ.N_INST, circ.n_c, z, fc.enc1(11), fc.enc1(13), kap)

    t0 = time.time()
    one(0)
    lat = time.time() - t0
    # bounded sample: enough rounds of `conc` concurrent proofs for ~15-25 s of wall clock
    rounds = 1 if conc >= 4 else max(1, min(3, int(20.0 / max(lat, 1e-3))))
    n = conc * rounds
    t0 = time.time()
    with ThreadPoolExecutor(max_workers=conc) as pool:
        list(pool.map(one, range(n)))
    dt = time.time() - t0
    return {"value": n / dt, "unit": "proofs/s", "cores": cores, "kind": "port",
            "concurrent_proofs": conc, "threads_per_proof": per_task, "single_proof_latency_s": lat,
            "sample": "%d subcircuits (1 commit + 1 prove each) of %s, %d in flight x %d threads, %.1f s; one proof "
                      "alone: %.1f s" % (n, args.config, conc, per_task, dt, lat)}


# ------------------------------------------------------------------------------------------- one measured job
def plan_classes(args, rank, world, single_class):
    """Which subcircuit indices this rank proves, the proving-key class of each, and the host-side setup jobs."""
    from hekaton_system_amd.workload import config_classes, representative_subcircuit
    from hekaton_system_amd.worker import shard_range
    family, _n_default, _reps = config_classes(args.config)
    n_total = args.subcircuits * world
    shard = list(shard_range(n_total, world, rank))
    if single_class:
        class_of = {i: 1 for i in shard}
    else:
        class_of = {i: representative_subcircuit(family, n_total, i) for i in shard}
    return n_total, shard, class_of


def job_instance():
    """The job's three public inputs (entry_chal, tr_chal, root): one triple for every subcircuit, as in the reference
    (aggregation.rs:192-205 combines the proofs under ONE set of public inputs)."""
    return [int.from_bytes(hashlib.sha256(b"hekaton bench public input %d" % k).digest()[:31], "little") for k in range(3)]


def prepare_job_host(args, curve, rank, world, single_class=False, witnesses=None):
    """Host half of a Job (no device, no HIP): per proving-key class of the rank's shard, the trusted setup's scalar
    work and the assignments, in parallel worker processes.  Called BEFORE anything initialises the GPU."""
    from hekaton_system_amd.workload import prepare_classes_host
    n_total, shard, class_of = plan_classes(args, rank, world, single_class)
    need = sorted(set(class_of.values()))
    nw = witnesses or args.witnesses
    jobs = []
    for rep in need:
        members = [i for i in shard if class_of[i] == rep]
        k = min(nw, len(members))
        seed = hashlib.sha256(b"HEKATON1 class %d" % rep).digest()
        jobs.append((curve, args.config, rep, seed, [1000 * rep + j + 1 for j in range(k)], n_total, job_instance()))
    t0 = time.time()
    out = {rep: (hs, assigns) for rep, hs, assigns in prepare_classes_host(jobs)}
    log("rank %d: %s host setup of %d proving-key class(es) %s in %.1f s" % (rank, curve, len(need), need, time.time() - t0))
    return dict(n_total=n_total, shard=shard, class_of=class_of, classes=out, curve=curve)


class Job:
    """Keys, assignments and the step function of one (curve, config) on one rank."""

    def __init__(self, args, prepared, rank, world, dev, backend, keep_host=False):
        from hekaton_system_amd import capi
        from hekaton_system_amd.cp_groth16 import FrCodec, setup_device
        from hekaton_system_amd.workload import make_config
        curve = prepared["curve"]
        self.args, self.curve, self.rank, self.world, self.backend = args, curve, rank, world, backend
        self.capi = capi
        self.ctx = capi.Context(curve, dev)
        self.fc = FrCodec(curve)
        self.n_total, self.shard, self.class_of = prepared["n_total"], prepared["shard"], prepared["class_of"]
        need = sorted(prepared["classes"])
        count = {rep: sum(1 for i in self.shard if self.class_of[i] == rep) for rep in need}
        self.main_class = max(need, key=lambda rep: (count[rep], -rep))      # the shard's most common class
        keep_host_class = self.main_class if keep_host else None
        self.classes = {}
        t0 = time.time()
        for rep in need:
            hs, assigns = prepared["classes"][rep]
            circ = make_config(curve, args.config, rep, self.n_total)
            keep = keep_host_class is not None and rep == keep_host_class
            pk, td = setup_device(hs, self.ctx, keep_on_device=not keep)
            dpk = pk.upload(self.ctx)
            host = None
            if keep:
                host = {"points": dict(a_g=pk.a_g, b_g=pk.b_g, b_h=pk.b_h, h_g=pk.h_g, ck_stages=pk.ck.deltas_abc_g,
                                       deltas_g=pk.deltas_g, last_delta_h=pk.vk.last_delta_h, alpha_g=pk.vk.alpha_g,
                                       beta_g=pk.beta_g, beta_h=pk.vk.beta_h),
                        "matrices": pk.matrices}
            else:
                for b in (pk.a_g, pk.b_g, pk.b_h, pk.h_g):
                    if isinstance(b, capi.DeviceBuffer):
                        b.free()
            members = [i for i in self.shard if self.class_of[i] == rep]
            zs, w0s, seeds = [], [], []
            for ws, zb, wb in assigns:
                zs.append(zb if args.host_inputs else capi.DeviceBuffer.from_host(self.ctx, zb))
                w0s.append(wb if args.host_inputs else capi.DeviceBuffer.from_host(self.ctx, wb))
                seeds.append(ws)
            extra = {}
            if args.witness_gen:
                # the requests of this rank's subcircuits (leaf bytes / child hashes / portal entries: INPUT data, made
                # before the timed region) and the class's word program on the device
                from hekaton_system_amd.poseidon import device_params
                from hekaton_system_amd.sha_circuit import example_witness, program_inputs
                ws = [example_witness(circ, seed=seeds[pos % len(seeds)], entry_chal=0x1234567, tr_chal=0x7654321)
                      for pos in range(len(members))]
                ops, refs, vmap = circ.tape.word_program(circ.n_v)
                if not hasattr(self, "poseidon_params"):
                    pp = device_params(curve, self.fc)
                    self.poseidon_params = (capi.DeviceBuffer.from_host(self.ctx, pp[0]),) + pp[1:]
                extra = dict(wg_ws=ws, wg_inputs=program_inputs(circ, ws),
                             wprog=self.ctx.wprog_upload(ops, refs, vmap, circ.tape.n_values, circ.tape.n_inputs),
                             zbig=capi.DeviceBuffer(self.ctx, len(members) * circ.n_v * self.ctx.fr_bytes))
            self.classes[rep] = dict(circ=circ, pk=pk, td=td, dpk=dpk, host=host, zs=zs, w0s=w0s, seeds=seeds,
                                     members=members, matrices=pk.matrices, w0_host=[wb for _ws, _zb, wb in assigns], **extra)
            log("rank %d: %s class %d (%d subcircuits of this shard): key + %d assignments resident, %.1f s" % (
                rank, curve, rep, len(members), len(zs), time.time() - t0))
        prepared["classes"] = None                   # the host copies are no longer needed
        self.circ = self.classes[self.main_class]["circ"]
        from hekaton_system_amd.workload import SyntheticSubcircuit
        self.synthetic = isinstance(self.circ, SyntheticSubcircuit)
        # which assignment a subcircuit uses: its position within its class, cycled
        self.assign_of = {}
        for rep, c in self.classes.items():
            for pos, i in enumerate(c["members"]):
                self.assign_of[i] = pos % len(c["zs"])
        # fixed per-subcircuit randomness: com_seed (worker.rs:129) and the prover's r, s (prover.rs:28-29) derive from
        # the index only, so a subcircuit's responses are byte-identical in every step (checked after the timed region)
        p_mod = self.fc.r
        from hekaton_system_amd.chacha import ChaCha12Rng
        self.rand = {}
        for i in self.shard:
            com_seed = hashlib.sha256(b"com_seed %d" % i).digest()
            kappa = ChaCha12Rng(com_seed).fr(p_mod)                    # mpi-snark/src/worker.rs:63-66
            rr = int.from_bytes(hashlib.sha256(b"r %d" % i).digest(), "little") % p_mod
            ss = int.from_bytes(hashlib.sha256(b"s %d" % i).digest(), "little") % p_mod
            self.rand[i] = dict(com_seed=com_seed, kappa=kappa, r=rr, s=ss, kappa_b=self.fc.enc1(kappa),
                                r_b=self.fc.enc1(rr), s_b=self.fc.enc1(ss))
        self.ctx.set_profiling(True)
        self.pool = ThreadPoolExecutor(max_workers=args.threads)
        self.wg_pool = ThreadPoolExecutor(max_workers=5)        # the classes' witness programs, beside the stage-0 round
        self.accum_ms, self.accum_n, self.accum_h, self.phase = [], [], [], {}
        self.gather_s = 0.0
        self.wait_s = 0.0
        self.wg_s = 0.0
        self.last_records = None
        self.prev_records = None

    # -- round 1 for every subcircuit of a key class in one call (hk_commit_batch) ------------------------
    def _stage0_class(self, c):
        from hekaton_system_amd.worker import Stage0Response
        import numpy as np
        members = c["members"]
        if "w0_rows" not in c:              # the requests' stage-0 witnesses row after row: input data, laid out once
            rows = np.concatenate([np.asarray(c["w0_host"][self.assign_of[i]], np.uint8).reshape(-1) for i in members])
            c["w0_rows"] = rows if self.args.host_inputs else self.capi.DeviceBuffer.from_host(self.ctx, rows)
            c["kappa_rows"] = np.concatenate([np.asarray(self.rand[i]["kappa_b"], np.uint8).reshape(-1) for i in members])
        coms = c["dpk"].commit_batch(0, c["w0_rows"], c["kappa_rows"], c["circ"].n0, len(members))
        t = self.ctx.last_timings()
        return [(i, Stage0Response(i, coms[k], self.rand[i]["com_seed"]).to_record(), t) for k, i in enumerate(members)]

    def _stage0_one(self, i):
        rec, t = self._stage0(i)
        return [(i, rec, t)]

    # -- the two rounds of one subcircuit ----------------------------------------------------------------
    def _stage0(self, i):
        from hekaton_system_amd.worker import Stage0Response
        c = self.classes[self.class_of[i]]
        k = self.assign_of[i]
        rnd = self.rand[i]
        com = c["dpk"].commit(0, c["w0s"][k], rnd["kappa_b"], n=c["circ"].n0)
        t = self.ctx.last_timings()
        return Stage0Response(i, com, rnd["com_seed"]).to_record(), t

    def _stage1(self, i_com):
        from hekaton_system_amd.cp_groth16 import Proof
        from hekaton_system_amd.worker import Stage1Response
        i, com = i_com
        c = self.classes[self.class_of[i]]
        k = self.assign_of[i]
        rnd = self.rand[i]
        z = c["zs"][k]
        if self.args.witness_gen:
            z = c["zbig"].ptr + c["members"].index(i) * c["circ"].n_v * self.ctx.fr_bytes     # generated this step
        a, b, cc = c["dpk"].prove(z, rnd["r_b"], rnd["s_b"], rnd["kappa_b"], n_v=c["circ"].n_v)
        t = self.ctx.last_timings()
        return Stage1Response(i, Proof(a, b, cc, [com])).to_record(), t

    def _start_witness_programs(self):
        """The challenge-independent part of stage-1 witness generation - each class's word program and the expansion of
        its bits into the assignments (16 ms, one dependent chain per subcircuit) - issued when the step begins, beside the
        stage-0 round: a subcircuit's SHA-256 trace does not depend on the round's challenges, its running evaluations
        and its execution-tree leaf do (_finish_witnesses)."""
        self._wg_t0 = time.time()
        self._wg_futs = [self.wg_pool.submit(lambda c=c: c["wprog"].run(c["wg_inputs"], [], [], out=c["zbig"]))
                         for c in self.classes.values()]

    def _finish_witnesses(self):
        """After the first round: per class the ~50 full-width values the host computes (portal entries, running
        evaluations, address-step flags: hk_assignment_scatter) and the membership block (execution-tree leaf hash + path,
        subcircuit_circuit.rs:233-252: hk_poseidon_path), classes side by side and beside whatever is left of the programs.
        Returns the seconds this added to the step."""
        from hekaton_system_amd.sha_circuit import full_values, poseidon_inputs
        t0 = time.time()
        trace = bool(os.environ.get("HK_WG_TRACE"))
        # the membership kernel is the longest piece (a chain of eight Poseidon permutations on one lane per subcircuit, 5 ms):
        # its inputs first and straight to the GPU, beside the programs - the expansion of a program's bits writes the
        # bit-valued columns only (k_witness_expand), these calls the full-width ones only -, then the host computes the
        # full-width values while it runs, then the scatter
        def membership(c):
            circ = c["circ"]
            leaves, sibs, idx = poseidon_inputs(circ, c["wg_ws"])
            self.ctx.poseidon_path(self.poseidon_params, leaves, sibs, idx, circ.n_v, circ.pos_col0, c["zbig"])

        def full(c):
            cols, vals = c["_wg_full"]
            c["wprog"].scatter(cols, vals, c["zbig"])

        mem = [self.pool.submit(membership, c) for c in self.classes.values()]
        for c in self.classes.values():
            c["_wg_full"] = full_values(c["circ"], c["wg_ws"])
        t1 = time.time()
        list(self.pool.map(full, self.classes.values()))
        for f in mem:
            f.result()
        for f in self._wg_futs:
            f.result()
        if trace:
            log("witness gen: programs started %.1f ms before the gather ended, host values %.1f ms, scatter + membership + programs' rest %.1f ms"
                % ((t0 - self._wg_t0) * 1e3, (t1 - t0) * 1e3, (time.time() - t1) * 1e3))
        return time.time() - t0

    def _gather(self, records):
        """The exchange step, timed in two parts: the wait for the slowest rank (a barrier of its own before the
        collective - rank skew, not communication) and the all_gather itself."""
        from hekaton_system_amd.worker import gather_records
        if self.world > 1 or getattr(self, "force_dist", False):
            import torch.distributed as dist
            tw = time.time()
            dist.barrier()
            self.wait_s += time.time() - tw
        t0 = time.time()
        if getattr(self, "force_dist", False):
            import torch
            import torch.distributed as dist
            import numpy as np
            local = torch.from_numpy(np.stack(records)).to("cuda" if self.backend == "nccl" else "cpu")
            bufs = [torch.empty_like(local)]
            dist.all_gather(bufs, local)
            out = bufs[0].cpu().numpy()
        else:
            out = gather_records(records, self.world, "cuda" if (self.world > 1 and self.backend == "nccl") else "cpu")
        self.gather_s += time.time() - t0
        return out

    def step(self, record):
        g1b = self.ctx.g1_bytes
        if self.args.witness_gen:
            self._start_witness_programs()
        # round 1: key classes with a SHORT stage 0 (big-merkle: 16 witnesses) take one hk_commit_batch call each; long
        # stages (vm: 217 280 terms, vkd: 8 192) stay one hk_commit per subcircuit, spread over the lanes as before
        short = lambda c: (c["circ"].n0 + 1) * len(c["members"]) * 2 <= 65536 and not self.args.no_batch_commit
        jobs = [(self._stage0_class, c) for c in self.classes.values() if short(c)]
        jobs += [(self._stage0_one, i) for c in self.classes.values() if not short(c) for i in c["members"]]
        by_i = {i: (rec, t) for part in self.pool.map(lambda j: j[0](j[1]), jobs) for i, rec, t in part}
        r0 = [by_i[i] for i in self.shard]
        all0 = self._gather([r for r, _ in r0])                           # node.rs:500-506
        assert len(all0) == self.n_total
        coms = [r[8:8 + g1b] for r, _ in r0]
        if self.args.witness_gen:
            self.wg_s += self._finish_witnesses()
        r1 = list(self.pool.map(self._stage1, zip(self.shard, coms)))
        all1 = self._gather([r for r, _ in r1])                           # node.rs:526-533
        assert len(all1) == self.n_total
        if record:
            for (_, tc), (_, t) in zip(r0, r1):
                self.accum_ms.append(t["accum_kernel_ms"] + tc["accum_kernel_ms"])
                self.accum_n.append(t["accum_kernel_launches"] + tc["accum_kernel_launches"])
                self.accum_h.append(t["accum_h_ms"])
                for key, v in t.items():
                    self.phase[key] = self.phase.get(key, 0.0) + v
        self.prev_records, self.last_records = self.last_records, (all0, all1)
        return all0, all1

    # -- post-timing checks of the timed proofs (never inside the timed region) -------------------------
    def verify_last_step(self):
        """(i) a subcircuit's records are byte-identical in the last two timed steps; (ii) subcircuits with distinct
        assignments / randomness have distinct proofs; (iii) one timed proof of this rank passes the trapdoor form of
        the Groth16 verifier equation (cp_groth16.trapdoor_verify), with h recomputed by hk_witness_map."""
        from hekaton_system_amd.cp_groth16 import trapdoor_verify
        import numpy as np
        out = {}
        all0, all1 = self.last_records
        if self.prev_records is not None:
            p0, p1 = self.prev_records
            assert np.array_equal(all0, p0) and np.array_equal(all1, p1), "responses changed between timed steps"
            out["repeat_identical"] = True
        proofs = {bytes(r[8:]) for r in all1}
        assert len(proofs) == self.n_total, "distinct subcircuits gave identical proofs"
        out["distinct_proofs"] = len(proofs)
        idx = [int.from_bytes(bytes(r[:8]), "little") for r in all1]
        assert idx == list(range(self.n_total)), "gathered records are not in subcircuit order"
        # trapdoor check of the LAST subcircuit of this rank's shard, from the gathered record
        i = self.shard[-1]
        c = self.classes[self.class_of[i]]
        circ, td = c["circ"], c["td"]
        if self.synthetic:
            circ.set_witness_seed(c["seeds"][self.assign_of[i]], job_instance())
        else:
            circ.set_witness_seed(c["seeds"][self.assign_of[i]])
        z_ints = circ.assignment_ints_current() if hasattr(circ, "assignment_ints_current") else circ.assignment_ints()
        A, B, C = c["matrices"]
        h_b, m = self.ctx.witness_map(A, B, C, circ.N_INST, circ.n_c, c["zs"][self.assign_of[i]], n_v=circ.n_v)
        h = self.fc.dec(h_b)
        rec = all1[i]
        g1b, g2b = self.ctx.g1_bytes, self.ctx.g2_bytes
        a, b, cc = rec[8:8 + g1b], rec[8 + g1b:8 + g1b + g2b], rec[8 + g1b + g2b:8 + 2 * g1b + g2b]
        com = all0[i][8:8 + g1b]
        rnd = self.rand[i]
        trapdoor_verify(self.ctx, self.curve, td, circ.N_INST, td.stage_ranges, z_ints, h, [com], [rnd["kappa"]],
                        rnd["r"], rnd["s"], (a, b, cc))
        out["trapdoor_verified_subcircuit"] = i
        return out

    def close(self):
        self.pool.shutdown()
        for c in self.classes.values():
            for b in c["zs"] + c["w0s"] + ([c["zbig"]] if c.get("zbig") else []) + ([c["w0_rows"]] if c.get("w0_rows") is not None else []):
                if isinstance(b, self.capi.DeviceBuffer):
                    b.free()
            c["dpk"].free()
            if c.get("wprog"):
                c["wprog"].free()
        self.ctx.close()


def end_to_end(job, use_dist, two_rounds_s):
    """After the timed region, outside `value`: the rest of the job the reference runs once the proofs are in
    (mpi-snark/src/coordinator.rs, distributed-prover/src/aggregation.rs:138-345) on the proofs of the LAST TIMED STEP -
    super commitment, `agg_subcircuit_proofs` (IPP commitments, twisted vectors, 4 x 4 cross terms, TIPA prove) and the
    verifier's TIPA check.  Its pairing-product assertion (aggregation.rs:265-269) holds only if EVERY timed proof
    satisfies the Groth16 verifier equation under its class's verifying key, so this is also the check of all timed
    proofs.  Key generation (TIPA SRS, AggProvingKey::new) is inside total_s, as in the reference's `work`."""
    import numpy as np
    from hekaton_system_amd import aggregation as agg, tipa
    from hekaton_system_amd.cp_groth16 import Proof
    from hekaton_system_amd.merlin import Transcript
    from hekaton_system_amd.workload import config_classes, representative_subcircuit
    n = job.n_total
    local = {rep: c["pk"].vk for rep, c in job.classes.items()}
    if use_dist:
        import torch.distributed as dist
        parts = [None] * job.world
        dist.all_gather_object(parts, local)
        vk_of = {}
        for d in parts:
            vk_of.update(d)
    else:
        vk_of = local
    if job.rank != 0:
        return None
    family, _n, _reps = config_classes(job.args.config)
    cls = (lambda i: 1) if job.args.single_class else (lambda i: representative_subcircuit(family, n, i))
    ctx, fc, r = job.ctx, job.fc, job.fc.r
    g1b, g2b = ctx.g1_bytes, ctx.g2_bytes
    all0, all1 = job.last_records
    coms = [np.asarray(rec[8:8 + g1b]) for rec in all0]
    proofs = []
    for i, rec in enumerate(all1):
        a, b, c = rec[8:8 + g1b], rec[8 + g1b:8 + g1b + g2b], rec[8 + g1b + g2b:8 + 2 * g1b + g2b]
        proofs.append(Proof(np.asarray(a), np.asarray(b), np.asarray(c), [coms[i]]))
    rnd = lambda tag: int.from_bytes(hashlib.sha256(tag).digest(), "little") % r
    t0 = time.time()
    srs = tipa.setup(ctx, job.curve, n, rnd(b"tipa alpha"), rnd(b"tipa beta"))
    t1 = time.time()
    apk = agg.AggProvingKey(ctx, job.curve, srs.ck, [vk_of[cls(i)] for i in range(n)])
    t2 = time.time()
    tipp = tipa.Tipp(ctx, job.curve)
    vk = tipa.verifier_key(ctx, job.curve, srs)
    pub = job_instance()
    res = {}
    for attempt in ("warm", "timed"):
        ta = time.time()
        super_com = apk.com.commit_only_left(srs.ck, np.concatenate(coms))              # coordinator.rs:339
        tb = time.time()
        proof, inst = apk.agg_subcircuit_proofs(Transcript(b"hekaton-bench"), super_com, proofs, pub, srs, tipp=tipp, check=False)
        tc = time.time()
        ok = tipp.verify(vk, inst["commitment"], inst["output"], inst["twist"], proof)
        td = time.time()
        assert ok, "TIPA proof of the timed proofs rejected"
        res = {"super_commitment_s": tb - ta, "aggregate_s": tc - tb, "verify_s": td - tc}
    for rb in srs.resident.values():
        rb.free()
    setup_s = (t1 - t0) + (t2 - t1)
    proving_s = two_rounds_s + res["super_commitment_s"] + res["aggregate_s"]
    res.update({"subcircuits": n, "two_rounds_s": two_rounds_s,
                # the reference's `work` builds the aggregation key inside the job and times it (mpi-snark/src/coordinator.rs:
                # 27-40,80-97: generate_agg_key -> TIPA::setup(N) + AggProvingKey::new): it is part of total_s
                "tipa_setup_s": t1 - t0, "agg_key_s": t2 - t1,
                "total_s": setup_s + proving_s, "total_without_agg_key_setup_s": proving_s,
                "all_timed_proofs_pass_the_pairing_product_equation": True, "tipa_proof_verified": True,
                "note": "two_rounds_s = one timed step (commit round, gather, prove round, gather); aggregator key setup and "
                        "aggregation of that step's proofs measured once after the timed region (aggregation: second of two runs); "
                        "total_s = job wall-clock up to the aggregate proof as the reference's `work` counts it (setup included); "
                        "total_without_agg_key_setup_s = the figure earlier rounds reported as total_s"})
    return res


def timed_run(job, steps, warmup, barrier):
    for _ in range(warmup):
        job.step(False)
    job.gather_s = 0.0
    job.wait_s = 0.0
    job.wg_s = 0.0
    barrier()
    t0 = time.time()
    for _ in range(steps):
        job.step(True)
    job.ctx.sync()
    barrier()
    return time.time() - t0


def roofline_of(job, curve, ms_per_proof=None):
    """Dominant kernel = k_msm_accum0<Fq> (bucket accumulation), launched 5 times per subcircuit: H query (m-1 dense
    terms), A / B1 / L queries (n_v-1, n_v-1, n1 terms) and the stage-0 commitment.  Algorithmic bytes per launch =
    terms * (32 + S1) (SURVEY.md §8d "MSM-G1 = n*(32+S1)"), averaged over the same launches whose durations are
    averaged - the population rocprofv3 --stats averages."""
    import numpy as np
    circ, ctx = job.circ, job.ctx
    m = 1
    while m < circ.n_c + circ.N_INST:
        m *= 2
    g1 = ctx.g1_bytes
    n1 = circ.n_v - circ.N_INST - circ.n0
    fq_limbs = ctx.fq_bytes // 4
    # signed 16-bit windows of the H query (csrc/msm.cuh msm_num_windows): 16 on both curves - BLS12-381's r = 0.906 * 2^255
    # leaves room for the "+2^15 per window" constant inside 256 bits, so there is no 17th digit
    nwin = 16
    terms = [m - 1, circ.n_v - 1, circ.n_v - 1, n1, circ.n0]
    # launches of the kernel per subcircuit in the timed steps: H, A, B1, L of stage 1, + the stage-0 commitment's unless
    # round 1 ran as hk_commit_batch (short stages take no bucket pass there)
    per_sub = int(round(np.sum(job.accum_n) / max(1, len(job.accum_ms)))) if job.accum_ms else 5
    per_sub = min(max(per_sub, 1), 5)
    alg_bytes = sum(terms[:per_sub]) * (32 + g1) / per_sub
    avg_ms = float(np.sum(job.accum_ms) / max(1, np.sum(job.accum_n))) if job.accum_ms else float("nan")
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
    h_ms = float(np.mean(job.accum_h)) if job.accum_h else float("nan")
    traffic, traffic_source = None, None
    try:      # HBM traffic of the same kernel from separate rocprofv3 --pmc passes (cannot be read live): a COMMITTED
        # measurement of an earlier build of this kernel, labelled as such
        with open(os.path.join(ROOT, "profiles", "pmc_accum0.json")) as f:
            pmc = json.load(f)
        key = "%s/%s" % (job.args.config, curve)
        if key in pmc:
            traffic = (pmc[key]["FETCH_SIZE_KiB_avg"] + pmc[key]["WRITE_SIZE_KiB_avg"]) * 1024.0
            traffic_source = "profiles/pmc_accum0.json (%s): FETCH_SIZE + WRITE_SIZE per launch from separate rocprofv3 --pmc " \
                             "passes, not read during this run" % pmc.get("_collected", "collection not recorded")
    except Exception:       # noqa: BLE001
        pass
    alone = None
    al = getattr(job, "alone", None)
    if al:
        ah = float(np.mean([t["accum_h_ms"] for t in al]))
        aavg = float(np.sum([t["accum_kernel_ms"] for t in al]) / max(1, np.sum([t["accum_kernel_launches"] for t in al])))
        ap = (m - 1) * nwin * 10 / (ah * 1e-3) / 1e9
        # the uncontended proofs run stage 1 only: the same 4 launches (H, A, B1, L) on both sides of the division
        alg4 = sum(terms[:4]) * (32 + g1) / 4
        alone = {"note": "the same kernel with ONE proof on the GPU (sequential stage-1 proofs on one lane, after the timed "
                         "region); bytes and time both over its 4 launches per proof (H, A, B1, L)",
                 "avg_launch_ms": aavg, "alg_bytes_per_launch": alg4, "achieved": alg4 / (aavg * 1e-3) / 1e9,
                 "frac": alg4 / (aavg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                 "h_query_ms": ah, "valu_achieved": ap, "valu_frac": ap / VALU_PRODUCT_CEILING[curve],
                 "proof_latency_ms": float(np.mean([t["total_ms"] for t in al]))}
    prods = (m - 1) * nwin * 10 / (h_ms * 1e-3) / 1e9
    whole = None
    if ms_per_proof and hasattr(circ, "query_density") and getattr(job, "synthetic", False):
        # every kernel of a proof shares the chip with seven other proofs' kernels, so a launch's own duration grows
        # with what runs beside it; the whole step's products over the whole step's time does not have that bias.
        # Counted (in Fq products; an Fq2 product = 3; an Fr product = (Fr limbs / Fq limbs)^2: 1 on BN254, 4/9 on BLS12-381): one
        # mixed add (10 products) per non-zero signed digit of a scalar whose base is not the point at infinity -
        # full-width scalars have 16 digits, a scalar 1 has one, a scalar 0 none -, the butterflies of the quotient
        # map's six transforms + its element-wise passes, the matrix-vector products.  NOT counted: bucket reductions,
        # window sums, k_finish, the digit sort (no products) - the fraction is a lower bound
        da, db = circ.query_density()
        full = 1.0 - circ.bit_fraction if hasattr(circ, "bit_fraction") else 0.15
        per_scalar = (1.0 - full) * 0.5 + full * nwin
        logm = m.bit_length() - 1
        frw = (ctx.fr_bytes / ctx.fq_bytes) ** 2
        parts = {"h_query": (m - 1) * nwin * 10.0,
                 "a_query": (circ.n_v - 1) * da * per_scalar * 10.0,
                 "b_g1_query": (circ.n_v - 1) * db * per_scalar * 10.0,
                 "b_g2_query": (circ.n_v - 1) * db * per_scalar * 10.0 * 3.0,
                 "l_query": n1 * per_scalar * 10.0,
                 "quotient_map": frw * (6.0 * (m // 2) * logm + 4.0 * m),
                 "matrix_vector": frw * 6.0 * circ.n_c}
        tot = sum(parts.values())
        rate = tot / (ms_per_proof * 1e-3) / 1e9
        whole = {"products_per_proof": tot, "parts": parts, "ms_per_proof": ms_per_proof, "achieved": rate,
                 "frac": rate / VALU_PRODUCT_CEILING[curve],
                 "note": "all counted products of a proof / (step time / proofs per step): a lower bound, reductions and "
                         "k_finish not counted; expected digit counts of the workload's scalar mixture"}
    mad_only = VMAD_RATE_TOPS * 1e3 / (2 * (fq_limbs ** 2))
    return m, {
        "bound": "hbm",
        "kernel": "k_msm_accum0<Fp<%s>> (bucket accumulation; avg over its %d launches per subcircuit)" % (
            "Bn254FqP" if curve == "bn254" else "Bls381FqP", per_sub),
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_source": traffic_source, "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms,
        "proofs_in_flight": job.args.threads, "alone": alone,
        "h_query_launch": {"alg_bytes": (m - 1) * (32 + g1), "avg_ms": h_ms,
                           "achieved": (m - 1) * (32 + g1) / (h_ms * 1e-3) / 1e9},
        # the bound that actually binds: integer VALU issue.  One mixed add = 10 Montgomery products; peak = the
        # MEASURED ceiling of the whole product loop (mads + carries + moves, tools/ubench.hip), with the
        # v_mad_u64_u32-only issue limit (24.7 Top/s / 2 N^2 mads per product) kept beside it
        "valu": {"unit": "G field mults/s", "achieved": prods, "peak": VALU_PRODUCT_CEILING[curve],
                 "frac": prods / VALU_PRODUCT_CEILING[curve], "peak_mad_only": mad_only,
                 "frac_mad_only": prods / mad_only,
                 "note": "H-query launch: (m-1) scalars x %d signed 16-bit digits x 10 products per mixed add" % nwin,
                 "whole_step": whole}}


def main():
    args = parse()
    # before anything imports the binding or touches HIP: with one stream per lane (HK_SERIAL_STREAMS=1, the experiment of
    # DESIGN.md section 5) every lane gets its own hardware queue; the default forked form keeps the binding's 20
    if os.environ.get("HK_SERIAL_STREAMS"):
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(args.threads))
    os.environ.setdefault("HK_MAX_LANES", str(max(args.threads, 8)))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))                     # before any HIP call
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or run without a launcher and let "
                 "bench.py start the ranks itself)" % (args.gpus, world))
    # stdout carries exactly one line (rank 0's JSON): everything libraries print there (gloo's "[Gloo] Rank 0 is
    # connected ..." chatter, RCCL banners) is sent to stderr for the whole run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("HK_BENCH_ECHO_RANK"):
        log("rank %d of %d starting (pid %d)" % (rank, world, os.getpid()))
    # host half of every job first: spawned worker processes, before this process touches HIP
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    want_secondary = rank == 0 and world == 1 and not args.no_secondary and args.curve == "bn254"
    prep = prepare_job_host(args, args.curve, rank, world, single_class=args.single_class)
    prep2 = prepare_job_host(args, "bls12_381", 0, 1, single_class=True, witnesses=2) if want_secondary else None
    # third leg: the same job shape on REAL SHA-256 subcircuits with every stage-1 assignment generated inside the step
    want_synth = (rank == 0 and world == 1 and not args.no_synthesis_leg and args.curve == "bn254"
                  and args.config == "big-merkle-64x32" and not args.witness_gen and args.subcircuits == 64)
    args3 = prep3 = None
    if want_synth:
        args3 = argparse.Namespace(**vars(args))
        args3.config, args3.witness_gen, args3.host_inputs = "big-merkle-sha-64x32", True, False
        prep3 = prepare_job_host(args3, "bn254", 0, 1)
    from hekaton_system_amd import capi          # noqa: F401  first: exports GPU_MAX_HW_QUEUES before HIP initialises
    import torch
    import torch.distributed as dist
    dev = local_rank if args.device is None else args.device
    backend = args.backend
    if backend == "auto":
        backend = "gloo" if (args.device is not None and world > 1) else "nccl"
    # HK_BENCH_FORCE_DIST=1: initialise the process group and route the gathers through it even with one rank (checks
    # that RCCL and this library's streams coexist in one process on a 1-GPU box)
    use_dist = world > 1 or bool(os.environ.get("HK_BENCH_FORCE_DIST"))
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        log("rank %d/%d on device %d, %s backend sees %d ranks" % (rank, world, dev, backend, dist.get_world_size()))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    job = Job(args, prep, rank, world, dev, backend, keep_host=want_cpu)
    job.force_dist = use_dist and world == 1
    circ = job.circ
    dt_local = timed_run(job, args.steps, args.warmup, barrier)
    dt = dt_local
    per_rank = [args.subcircuits * args.steps / dt_local]
    gather_ms = [job.gather_s / args.steps * 1e3]
    wait_ms = [job.wait_s / args.steps * 1e3]
    rank_ms = [dt_local / args.steps * 1e3]
    rank_classes = [sorted(job.classes)]
    if use_dist:
        on = "cuda" if backend == "nccl" else "cpu"
        tt = torch.tensor([dt_local], dtype=torch.float64, device=on)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        mine = torch.tensor([per_rank[0], gather_ms[0], wait_ms[0], rank_ms[0]], dtype=torch.float64, device=on)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(x[0]) for x in allr]
        gather_ms = [float(x[1]) for x in allr]
        wait_ms = [float(x[2]) for x in allr]
        rank_ms = [float(x[3]) for x in allr]
        parts = [None] * world
        dist.all_gather_object(parts, sorted(job.classes))
        rank_classes = parts
    # the dominant kernel WITHOUT other proofs sharing the chip: three proofs one after another on one lane, after the
    # timed region (reported beside the timed-region figure, never instead of it)
    alone = []
    try:
        if args.no_verify:                       # the counter-collection passes keep exactly 5 launches per subcircuit
            raise RuntimeError("skipped with --no-verify")
        i0 = job.shard[0]
        com0 = job.last_records[0][i0][8:8 + job.ctx.g1_bytes]
        for _ in range(3):
            _rec, t = job._stage1((i0, com0))
            alone.append(t)
        job.alone = alone[1:]
    except Exception as e:       # noqa: BLE001
        if not args.no_verify:
            log("rank %d: uncontended measurement failed: %r" % (rank, e))
        job.alone = []
    checks = None
    if not args.no_verify:
        t0 = time.time()
        checks = job.verify_last_step()              # every rank checks its own shard; raises on failure
        checks["seconds"] = time.time() - t0
        log("rank %d: timed proofs verified: %s" % (rank, checks))
    proofs = world * args.subcircuits * args.steps
    e2e = None
    n_tot = world * args.subcircuits
    if not args.no_e2e and job.synthetic and n_tot & (n_tot - 1) == 0:
        try:
            e2e = end_to_end(job, use_dist, dt / args.steps)          # every rank: the verifying keys are gathered
            if rank == 0:
                log("end to end: %s" % {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e2e.items() if k != "note"})
        except Exception as e:       # noqa: BLE001
            log("rank %d: end-to-end leg failed: %r" % (rank, e))
            e2e = {"error": repr(e)}
    if rank == 0:
        m, roof = roofline_of(job, args.curve, dt / (args.subcircuits * args.steps) * 1e3)
        nprov = max(1, len(job.accum_ms))
        out = {
            "metric": "subcircuit Groth16 proofs/sec (whole node), big-merkle",
            "value": proofs / dt, "unit": "proofs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32 limbs (Montgomery mod-p integers)",
            "data": "synthetic (SHA-like 85%% small / 15%% full-width witnesses, genuine Groth16 SRS from a seeded "
                    "trapdoor per proving-key class; up to %d distinct assignments per class%s)" % (
                        args.witnesses, "; assignments copied from pageable host memory per proof" if args.host_inputs else ""),
            "config": {"workload": args.config, "curve": args.curve, "subcircuits_per_gpu_per_step": args.subcircuits,
                       "subcircuits_per_step": job.n_total, "n_constraints": circ.n_c, "n_variables": circ.n_v,
                       "domain": m, "host_threads_per_gpu": args.threads,
                       "sharding": "contiguous, %d subcircuits per rank (node.rs:471-493)" % args.subcircuits,
                       "pk_classes_rank0": {str(k): len(v["members"]) for k, v in job.classes.items()},
                       "exchange": "2 all_gathers per step (104 B + 328 B records), %s" % (backend if world > 1 else "local")},
            "roofline": roof,
            "per_rank_proofs_per_s": per_rank,
            # the two exchange steps of a job, per rank: `collective` is the time inside the all_gathers with every rank
            # already there; `waiting_for_slowest_rank` is the barrier in front of them (rank skew: a rank with lighter
            # proving-key classes finishes its round early and waits here - not communication)
            "exchange_ms_per_step": {"collective": gather_ms, "waiting_for_slowest_rank": wait_ms},
            "per_rank_ms_per_step": rank_ms, "per_rank_pk_classes": rank_classes,
            "witness_gen_ms_per_step": (job.wg_s / args.steps * 1e3) if args.witness_gen else None,
            "timed_proofs_check": checks,
            "end_to_end": e2e,
            "phase_ms_per_proof": {k: v / nprov for k, v in job.phase.items() if k.endswith("_ms")},
        }
        if want_cpu:
            try:
                mc = job.classes[job.main_class]
                out["cpu_baseline"] = cpu_baseline(args, mc["circ"], mc["host"], job.fc)
            except Exception as e:       # noqa: BLE001
                log("cpu baseline failed:", e)
                out["cpu_baseline"] = None
        if want_secondary:
            # the curve BASELINE.json's north_star names, same workload, one proving-key class, shorter run
            try:
                job.close()
                job = None
                j2 = Job(args, prep2, 0, 1, dev, backend)
                s2 = max(2, min(args.steps, 4))
                dt2 = timed_run(j2, s2, 1, barrier)
                chk2 = None if args.no_verify else j2.verify_last_step()
                _m2, roof2 = roofline_of(j2, "bls12_381", dt2 / (args.subcircuits * s2) * 1e3)
                out["secondary"] = {"curve": "bls12_381", "value": args.subcircuits * s2 / dt2, "unit": "proofs/s",
                                    "steps": s2, "warmup": 1, "ms_per_step": dt2 / s2 * 1e3,
                                    "pk_classes": 1, "timed_proofs_check": chk2,
                                    "roofline": {k: roof2[k] for k in ("achieved", "peak", "unit", "frac", "avg_launch_ms", "valu")}}
                j2.close()
            except Exception as e:       # noqa: BLE001
                log("secondary (BLS12-381) run failed:", repr(e))
                out["secondary"] = None
        if want_synth:
            try:
                if job is not None:
                    job.close()
                    job = None
                j3 = Job(args3, prep3, 0, 1, dev, backend)
                s3 = max(2, min(args.steps, 4))
                dt3 = timed_run(j3, s3, 1, barrier)
                chk3 = None if args.no_verify else j3.verify_last_step()
                c3 = j3.circ
                out["with_synthesis"] = {
                    "workload": "big-merkle-sha-64x32: 64 real SHA-256 big-merkle subcircuits (32 iterations, 5 proving-key "
                                "classes, own gadget set - DESIGN.md section 4c), every stage-1 assignment generated on the "
                                "GPU inside the step from the subcircuit's inputs (the reference times synthesis inside "
                                "stage 1: node.rs:589-596, prover.rs:70-75)",
                    "curve": "bn254", "value": args.subcircuits * s3 / dt3, "unit": "proofs/s", "steps": s3, "warmup": 1,
                    "ms_per_step": dt3 / s3 * 1e3, "witness_gen_ms_per_step": j3.wg_s / s3 * 1e3,
                    "n_constraints": c3.n_c, "n_variables": c3.n_v,
                    "pk_classes": {str(k): len(v["members"]) for k, v in j3.classes.items()},
                    "timed_proofs_check": chk3}
                j3.close()
            except Exception as e:       # noqa: BLE001
                log("with_synthesis (real SHA-256, witness generation in the step) run failed:", repr(e))
                out["with_synthesis"] = None
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if job is not None:
        job.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
