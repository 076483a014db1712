"""CPU, world_size 2 (gloo): the only multi-rank logic of this path — contiguous sharding of
subcircuits over workers (mpi-snark/src/bin/node.rs:471-472,490-493) and the gather of fixed-size
response records (node.rs:500-506,526-533)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, n_sub, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from hekaton_system_amd.cp_groth16 import Proof
    from hekaton_system_amd.worker import (Stage0Response, Stage1Response, shard_range, gather_records)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_range(n_sub, world, rank)
    g1, g2 = 64, 128
    s0 = [Stage0Response(i, np.full(g1, i % 251, np.uint8), bytes([i % 256]) * 32).to_record() for i in mine]
    s1 = [Stage1Response(i, Proof(np.full(g1, 1 + i % 200, np.uint8), np.full(g2, 2 + i % 200, np.uint8),
                                  np.full(g1, 3 + i % 200, np.uint8), [np.full(g1, i % 251, np.uint8)])).to_record()
          for i in mine]
    all0 = gather_records(s0, world)
    all1 = gather_records(s1, world)
    dist.barrier()
    if rank == 0:
        r0 = [Stage0Response.from_record(r, g1) for r in all0]
        r1 = [Stage1Response.from_record(r, g1, g2) for r in all1]
        ok = [x.subcircuit_idx for x in r0] == list(range(n_sub)) and [x.subcircuit_idx for x in r1] == list(range(n_sub))
        ok &= all(x.com[0] == x.subcircuit_idx % 251 and x.com_seed == bytes([x.subcircuit_idx % 256]) * 32 for x in r0)
        ok &= all(x.proof.b[0] == 2 + x.subcircuit_idx % 200 and len(x.proof.ds) == 1 for x in r1)
        ok &= all0.shape == (n_sub, 8 + g1 + 32) and all1.shape == (n_sub, 8 + 3 * g1 + g2)
        q.put(bool(ok))
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, 8, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_range_semantics():
    sys.path.insert(0, ROOT)
    from hekaton_system_amd.worker import shard_range
    assert list(shard_range(64, 8, 3)) == list(range(24, 32))
    assert [len(shard_range(512, 8, r)) for r in range(8)] == [64] * 8
    with pytest.raises(AssertionError):            # node.rs:472 assert
        shard_range(10, 4, 0)
