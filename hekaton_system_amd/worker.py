"""Host-side mirror of the reference's worker protocol for the hot path (one process per GPU).

    Stage0Request/Response, Stage1Request/Response     distributed-prover/src/worker.rs:20-71,
                                                       coordinator.rs:195-199,520-532
    process_stage0_request_get_cb                      distributed-prover/src/worker.rs:91-146
    process_stage1_request_with_cb                     distributed-prover/src/worker.rs:150-195
    WorkerState.{stage_0,stage_1}                      mpi-snark/src/worker.rs:25-87
    shard / gather                                     mpi-snark/src/bin/node.rs:471-472,490-506,523-533

The reference scatters requests and gathers responses with MPI (rank 0 = coordinator).  Here every
rank is a worker on its own GPU; subcircuits are sharded contiguously and the fixed-size response
records are all-gathered with torch.distributed (RCCL on GPUs, gloo in the CPU tests).  There is no
collective on the data path: a response record is 104 B (stage 0) / 328 B (stage 1: idx + A + B + C + one
stage-0 commitment) for BN254; the ark-serialize framings of the same responses are 104 B / 336 B
(ark_serialize.py: the `Vec` of commitments carries an 8-byte length).
"""
from dataclasses import dataclass, field

import numpy as np

from .chacha import ChaCha12Rng
from .cp_groth16 import CommitmentBuilder, Proof


@dataclass
class Stage0Request:                 # coordinator.rs:195-199; ROM entries as (addr u64, val int) pairs (rom_transcript.rs:222-226)
    subcircuit_idx: int
    time_ordered_subtrace: list = field(default_factory=list)
    addr_ordered_subtrace: list = field(default_factory=list)


@dataclass
class Stage0Response:                # worker.rs:20-25
    subcircuit_idx: int
    com: np.ndarray                  # packed G1
    com_seed: bytes                  # 32 bytes

    def to_record(self):
        return np.concatenate([np.frombuffer(int(self.subcircuit_idx).to_bytes(8, "little"), np.uint8),
                               np.asarray(self.com, np.uint8), np.frombuffer(self.com_seed, np.uint8)])

    @classmethod
    def from_record(cls, rec, g1_bytes):
        rec = np.asarray(rec, np.uint8)
        return cls(int.from_bytes(rec[:8].tobytes(), "little"), rec[8:8 + g1_bytes].copy(),
                   rec[8 + g1_bytes:8 + g1_bytes + 32].tobytes())


@dataclass
class Stage1Request:                 # coordinator.rs:520-532
    subcircuit_idx: int
    witness_seed: int = 0            # selects the synthetic subcircuit's stage-1 assignment (synthetic workloads)
    # the reference's fields (ROM circuits, Poseidon exec tree over Fr), carried for the wire format:
    time_ordered_eval: int = 1       # cur_leaf.evals (eval_tree.rs:53-58, rom_transcript.rs:18-27)
    addr_ordered_eval: int = 1
    challenges: tuple = None         # Option<(entry_chal, tr_chal)>
    last_subtrace_entry: tuple = (0, 0)
    leaf_sibling_hash: int = 0       # next_leaf_membership: ark-crypto-primitives merkle_tree::Path
    auth_path: list = field(default_factory=list)
    leaf_index: int = 0
    root: int = 0
    serialized_witnesses: bytes = b""
    circ_params: tuple = (0, 0, 0)   # MerkleTreeCircuitParams { num_leaves, num_sha_iters_per_subcircuit, num_portals_per_subcircuit }


@dataclass
class Stage1Response:                # worker.rs:49-52
    subcircuit_idx: int
    proof: Proof

    def to_record(self):
        p = self.proof
        parts = [np.frombuffer(int(self.subcircuit_idx).to_bytes(8, "little"), np.uint8),
                 np.asarray(p.a, np.uint8), np.asarray(p.b, np.uint8), np.asarray(p.c, np.uint8)]
        parts += [np.asarray(d, np.uint8) for d in p.ds]
        return np.concatenate(parts)

    @classmethod
    def from_record(cls, rec, g1_bytes, g2_bytes):
        rec = np.asarray(rec, np.uint8)
        o = 8
        a = rec[o:o + g1_bytes].copy(); o += g1_bytes
        b = rec[o:o + g2_bytes].copy(); o += g2_bytes
        c = rec[o:o + g1_bytes].copy(); o += g1_bytes
        ds = [rec[k:k + g1_bytes].copy() for k in range(o, len(rec), g1_bytes)]
        return cls(int.from_bytes(rec[:8].tobytes(), "little"), Proof(a, b, c, ds))


def process_stage0_request_get_cb(rng, pk, req, circuit):
    """worker.rs:91-146: draw com_seed from `rng`, commit stage 0 with ChaCha(com_seed)'s first draw as
    the randomness, return (response, commitment builder)."""
    circuit.subcircuit_idx = req.subcircuit_idx
    com_seed = rng.gen_seed()                                   # worker.rs:129
    subcircuit_rng = ChaCha12Rng(com_seed)                        # worker.rs:130
    cb = CommitmentBuilder.new(circuit, pk)
    com, _ = cb.commit(subcircuit_rng)                          # worker.rs:134-137
    return Stage0Response(req.subcircuit_idx, com, com_seed), cb


def process_stage0_requests_batch(rngs, pks, reqs, circuits):
    """The stage-0 pass of a worker over ALL the subcircuits it holds (the loop of mpi-snark/src/bin/node.rs:500-506 around
    worker.rs:91-146) with the commitments of every proving-key class in one hk_commit_batch call: the same draws from every
    subcircuit's `rng`, the same responses and commitment builders as process_stage0_request_get_cb request by request.
    rngs, pks, reqs, circuits: one entry per request.  Returns [(Stage0Response, CommitmentBuilder)] in request order."""
    from .cp_groth16 import FrCodec
    import numpy as np
    prepared, groups = [], {}
    for k, (rng, pk, req, circuit) in enumerate(zip(rngs, pks, reqs, circuits)):
        circuit.subcircuit_idx = req.subcircuit_idx
        com_seed = rng.gen_seed()                               # worker.rs:129
        cb = CommitmentBuilder.new(circuit, pk)
        cb.circuit.generate_constraints(cb.cur_stage, cb.cs)    # committer.rs:61: stage-0 synthesis stays on the host
        w = cb.cs.current_stage_witness_assignment()
        kappa = ChaCha12Rng(com_seed).fr(cb.cs.r)               # committer.rs:85: the FIRST draw of ChaCha(com_seed)
        prepared.append((req, cb, com_seed, w, kappa))
        groups.setdefault(id(pk), (pk, []))[1].append(k)
    coms = [None] * len(prepared)
    for pk, members in groups.values():
        fc = FrCodec(pk.device.ctx.curve)
        n = len(prepared[members[0]][3])
        rows = np.concatenate([np.frombuffer(bytes(fc.enc(prepared[k][3])), np.uint8) for k in members]) if n else np.zeros(0, np.uint8)
        kap = np.frombuffer(bytes(fc.enc([prepared[k][4] for k in members])), np.uint8)
        out = pk.device.commit_batch(0, rows, kap, n, len(members))
        for j, k in enumerate(members):
            coms[k] = out[j].copy()
    res = []
    for (req, cb, com_seed, _w, _kappa), com in zip(prepared, coms):
        cb.cur_stage += 1
        res.append((Stage0Response(req.subcircuit_idx, com, com_seed), cb))
    return res


def process_stage1_request_with_cb(rng, cb, com, rand, stage1_req):
    """worker.rs:150-195."""
    assert getattr(cb.circuit, "subcircuit_idx", stage1_req.subcircuit_idx) == stage1_req.subcircuit_idx
    proof = cb.prove([com], [rand], rng)                        # worker.rs:189
    return Stage1Response(stage1_req.subcircuit_idx, proof)


class WorkerState:
    """mpi-snark/src/worker.rs:25-87: keeps the commitment builder between the two rounds and
    re-derives the commitment randomness from `com_seed` (worker.rs:63-66)."""

    def __init__(self, num_subcircuits, pk_for_idx, circuit_for_idx, r_mod):
        self.num_subcircuits = num_subcircuits
        self.pk_for_idx = pk_for_idx
        self.circuit_for_idx = circuit_for_idx
        self.r_mod = r_mod
        self.cb = None
        self.com = None
        self.com_rand = None

    def stage_0(self, rng, stage0_req):
        pk = self.pk_for_idx(stage0_req.subcircuit_idx)
        resp, cb = process_stage0_request_get_cb(rng, pk, stage0_req, self.circuit_for_idx(stage0_req.subcircuit_idx))
        self.cb, self.com = cb, resp.com
        self.com_rand = ChaCha12Rng(resp.com_seed).fr(self.r_mod)
        return resp

    def stage_1(self, rng, stage1_req):
        return process_stage1_request_with_cb(rng, self.cb, self.com, self.com_rand, stage1_req)


# ---- sharding and gathering (the only multi-GPU logic this path has) ---------------------------------

def shard_range(num_subcircuits, num_workers, worker):
    """node.rs:471-472,490-493: contiguous chunks of num_subcircuits / num_workers (must divide)."""
    if num_subcircuits % num_workers != 0:
        raise AssertionError("num_subcircuits % num_workers == 0")     # node.rs:472
    per = num_subcircuits // num_workers
    return range(worker * per, (worker + 1) * per)


def gather_records(local_records, world_size, device="cpu"):
    """All ranks contribute the same number of equal-sized uint8 records; every rank gets all of them in
    subcircuit order (node.rs:500-506,526-533 gather on the root; fixed-size records instead of the
    reference's length-prefixed `Vec<Response>` framing)."""
    import torch
    import torch.distributed as dist
    local = torch.from_numpy(np.stack(local_records)).to(device)
    if world_size == 1:
        return local.cpu().numpy()
    out = [torch.empty_like(local) for _ in range(world_size)]
    dist.all_gather(out, local)
    return torch.cat(out, dim=0).cpu().numpy()
