"""GPU: hk_poseidon_path (csrc/witness.cuh k_poseidon_path) against hekaton_system_amd/poseidon.py on both curves: the
membership block of a batch of subcircuits - leaf hash, per level (bit, sibling, left), two-to-one hashes - equals
`sha_circuit.poseidon_path_trace` value for value, lands at the requested columns and nowhere else, and its last state
element is the root the host tree computes.  Malformed descriptors / blocks that do not fit are refused."""
import random

import numpy as np
import pytest

from hekaton_system_amd import capi
from hekaton_system_amd.cp_groth16 import CURVE_PARAMS, FrCodec
from hekaton_system_amd.poseidon import ExecTree, device_params, merkle_params
from hekaton_system_amd.sha_circuit import poseidon_path_trace

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_membership_block_equals_the_host_trace(cname, ctx_bn254, ctx_bls):
    ctx = ctx_bn254 if cname == "bn254" else ctx_bls
    fc = FrCodec(cname)
    r = CURVE_PARAMS[cname]["r"]
    rnd = random.Random(11)
    n = 64                                                     # a 64-leaf execution tree: depth 6, as BASELINE configs[1]
    leaves = [[rnd.randrange(r) for _ in range(4)] for _ in range(n)]
    leaves[5] = [0, 0, 0, 0]
    leaves[6] = [r - 1, 1, r - 1, 0]
    tree = ExecTree(cname, leaves)
    leaf_cfg, node_cfg = merkle_params(cname)
    batch = 70                                                 # more than one wavefront; some leaves twice
    which = [i % n for i in range(batch)]
    traces = [poseidon_path_trace(leaf_cfg, node_cfg, leaves[i], *tree.path(i)) for i in which]
    block = len(traces[0])
    col0, tail = 7, 5
    n_v = col0 + block + tail
    z = capi.DeviceBuffer.from_host(ctx, np.full(batch * n_v * ctx.fr_bytes, 0xA5, np.uint8))
    params = device_params(cname, fc)
    leaf_b = np.stack([fc.enc(leaves[i]) for i in which])
    sib_b = np.stack([fc.enc(tree.path(i)[0]) for i in which])
    idx = np.array(which, np.uint32)
    ctx.poseidon_path(params, leaf_b, sib_b, idx, n_v, col0, z)
    got = z.to_host().reshape(batch, n_v, ctx.fr_bytes)
    for b in range(batch):
        assert fc.dec(got[b, col0:col0 + block].reshape(-1)) == traces[b], (cname, b)
        assert traces[b][-2] == tree.root                      # state[1] of the last permutation
    assert (got[:, :col0] == 0xA5).all() and (got[:, col0 + block:] == 0xA5).all()      # nothing outside the block
    # constants resident on the device give the same result
    cbuf = capi.DeviceBuffer.from_host(ctx, params[0])
    z2 = capi.DeviceBuffer.from_host(ctx, np.zeros(batch * n_v * ctx.fr_bytes, np.uint8))
    ctx.poseidon_path((cbuf,) + params[1:], leaf_b, sib_b, idx, n_v, col0, z2)
    assert np.array_equal(z2.to_host().reshape(batch, n_v, -1)[:, col0:col0 + block], got[:, col0:col0 + block])
    # refused: a block that does not fit, a width the kernel has no state for, constants that end too early
    with pytest.raises(capi.HekatonError) as e:
        ctx.poseidon_path(params, leaf_b, sib_b, idx, n_v - tail - 1, col0, z)
    assert e.value.status == capi.HK_ERR_ARG
    bad = (params[0], params[1], (5,) + params[2][1:], params[3])
    with pytest.raises(capi.HekatonError):
        ctx.poseidon_path(bad, leaf_b, sib_b, idx, n_v, col0, z)
    with pytest.raises(capi.HekatonError):
        ctx.poseidon_path((params[0], params[1] - 4, params[2], params[3]), leaf_b, sib_b, idx, n_v, col0, z)
    for x in (z, z2, cbuf):
        x.free()
