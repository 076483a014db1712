"""CPU: the re-implemented big-merkle gadget set (hekaton_system_amd/sha_circuit.py): the trace -> assignment generator
computes real SHA-256 chains (checked against hashlib), and the assignment it emits satisfies every row of the R1CS the
same program builds, for every subcircuit kind; tampering breaks it."""
import hashlib

import numpy as np
import pytest

from hekaton_system_amd.cp_groth16 import FrCodec
from hekaton_system_amd.sha_circuit import (INNER_HASH_SIZE, ShaMerkleSubcircuit, example_witness, iterated_sha256,
                                            node_hash_field)


def _check_r1cs(circ, z):
    fc = circ.fc
    r = circ.r
    vals = []
    for (rp, col, _val, vidx, table) in circ._csr or (circ.csr(fc) and circ._csr):
        rp, col, vidx = rp.tolist(), col.tolist(), vidx.tolist()
        out = []
        for i in range(circ.n_c):
            acc = 0
            for k in range(rp[i], rp[i + 1]):
                acc += table[vidx[k]] * z[col[k]]
            out.append(acc % r)
        vals.append(out)
    bad = [i for i in range(circ.n_c) if vals[0][i] * vals[1][i] % r != vals[2][i]]
    return bad


@pytest.mark.parametrize("kind,ns,first,last", [("leaf", 1, True, False), ("leaf", 2, False, False), ("parent", 1, False, False),
                                                ("root", 2, False, False), ("padding", 1, False, True)])
def test_generated_assignment_satisfies_the_r1cs(kind, ns, first, last):
    circ = ShaMerkleSubcircuit("bn254", kind, ns, n_portals=4, first=first, last=last)
    circ.csr(circ.fc)
    ws = [example_witness(circ, seed=s, entry_chal=12345, tr_chal=67890) for s in (1, 2, 3)]
    zs = circ.assignment_ints(ws)
    assert len(zs[0]) == circ.n_v and circ.n_c > 26000 * ns
    for z in zs[:2]:
        assert _check_r1cs(circ, z) == []
    # the chain is real SHA-256
    _bits, _full, digests = circ.witness_batch(ws)
    for w, d in zip(ws, digests):
        if kind in ("leaf", "padding"):
            assert d == iterated_sha256(w["leaf"], ns)
        else:
            kids = b"".join(int(w["time"][k][1]).to_bytes(INNER_HASH_SIZE, "little") for k in range(2))
            assert d == iterated_sha256(kids, ns)
        if kind in ("leaf", "parent"):
            assert w["time"][-1][1] == node_hash_field(d)
    # a flipped bit somewhere in the middle of the trace violates some row
    z = list(zs[0])
    mid = circ.N_INST + circ.n0 + (circ.n_wit - circ.n0) // 2
    z[mid] = 1 - z[mid] if z[mid] in (0, 1) else z[mid] + 1
    assert _check_r1cs(circ, z) != []


def test_montgomery_bytes_match_the_ints():
    circ = ShaMerkleSubcircuit("bn254", "leaf", 1, n_portals=4)
    w = example_witness(circ, seed=9)
    z = circ.assignment_ints(w)[0]
    zb = circ.assignment_bytes(w)[0]
    fc = FrCodec("bn254")
    assert fc.dec(zb[:40 * 32]) == z[:40]
    assert fc.dec(zb[-8 * 32:]) == z[-8:]
    assert hashlib.sha256(b"abc").hexdigest().startswith("ba7816bf")


def test_whole_job_witness_generation():
    """An 8-subcircuit job (4 leaves): the tree hashes are hashlib's, the root subcircuit's digest packs to the public
    root, every subcircuit's generated assignment satisfies ITS class's R1CS with the evals threading through, and the
    final permutation check (time eval == addr eval) holds at the padding subcircuit."""
    import os
    from hekaton_system_amd.sha_circuit import ShaMerkleJob
    leaves = [bytes([17 * i + k & 0xff for k in range(64)]) for i in range(4)]
    job = ShaMerkleJob("bn254", 8, 1, 4, leaves, entry_chal=0xabc, tr_chal=0xdef123)
    # plain Merkle root over truncated digests
    h = [hashlib.sha256(l).digest() for l in leaves]
    l1 = [hashlib.sha256(h[0][:27] + h[1][:27]).digest(), hashlib.sha256(h[2][:27] + h[3][:27]).digest()]
    root = hashlib.sha256(l1[0][:27] + l1[1][:27]).digest()
    assert job.digest[6] == root and job.sha_root == int.from_bytes(root[:27], "little")
    # the public root is the EXECUTION tree's (Poseidon over the leaves (evals after subcircuit i, last addr-ordered entry)):
    # every subcircuit's path leads to it
    assert all(job.tree.verify(job.tree.leaves[i], *job.tree.path(i)) for i in range(8)) and job.root == job.tree.root
    assert [job.class_of(i)[0] for i in range(8)] == ["leaf"] * 4 + ["parent", "parent", "root", "padding"]
    classes = {}
    for idx in range(8):
        key = job.class_of(idx)
        classes.setdefault(key, (job.make_class(idx), []))[1].append(idx)
    assert len(classes) == 5                       # first leaf, leaf, parent, root, padding
    for (kind, first, last), (circ, members) in classes.items():
        circ.csr(circ.fc)
        zs = circ.assignment_ints([job.inputs(i) for i in members])
        _b, _f, digests = circ.witness_batch([job.inputs(i) for i in members])
        for i, z, d in zip(members, zs, digests):
            assert d == job.digest[i]
            assert _check_r1cs(circ, z) == [], (kind, i)
    # a job with one leaf changed has a different data root, and the old one is then unprovable in the root subcircuit
    job2 = ShaMerkleJob("bn254", 8, 1, 4, [leaves[0], leaves[1], leaves[2], bytes(64)], entry_chal=0xabc, tr_chal=0xdef123)
    assert job2.sha_root != job.sha_root and job2.root != job.root
    circ = job2.make_class(6)
    circ.csr(circ.fc)
    w = job2.inputs(6)
    w["sha_root"] = job.sha_root
    assert _check_r1cs(circ, circ.assignment_ints(w)[0]) != []
    # the membership check: another public root, a sibling of the path changed, the index flipped - each unprovable
    circ, members = classes[("leaf", False, False)]
    for tamper in ("root", "sibling", "index"):
        w = dict(job.inputs(members[0]))
        sib, idx = w["path"]
        if tamper == "root":
            w["root"] = (w["root"] + 1) % circ.r
        elif tamper == "sibling":
            w["path"] = ([sib[0], (sib[1] + 1) % circ.r] + sib[2:], idx)
        else:
            w["path"] = (sib, idx ^ 2)
        assert _check_r1cs(circ, circ.assignment_ints(w)[0]) != [], tamper
    # the address-step check (rom_portal_manager.rs:151-165): a slice whose address jumps by two, or whose equal addresses
    # carry different values, is unprovable even with evals and leaf recomputed consistently
    from hekaton_system_amd.sha_circuit import poseidon_path_root
    def reroot(w):
        ech, tr, r = w["entry_chal"], w["tr_chal"], circ.r
        te, ae = w["time_eval0"], w["addr_eval0"]
        for a, v in w["time"]:
            te = te * ((tr - (v + ech * a)) % r) % r
        for a, v in w["addr"]:
            ae = ae * ((tr - (v + ech * a)) % r) % r
        w["root"] = poseidon_path_root(circ.leaf_cfg, circ.node_cfg, [te, ae, w["addr"][-1][0] % r, w["addr"][-1][1] % r], *w["path"])
        return w
    w = dict(job.inputs(members[0]))
    assert _check_r1cs(circ, circ.assignment_ints(reroot(dict(w)))[0]) == []          # the helper itself is sound
    jump = dict(w); jump["addr"] = list(w["addr"]); jump["addr"][-1] = (w["addr"][-1][0] + 2, w["addr"][-1][1])
    assert _check_r1cs(circ, circ.assignment_ints(reroot(jump))[0]) != []
    k = next(k for k in range(1, 4) if w["addr"][k][0] == w["addr"][k - 1][0])
    diff = dict(w); diff["addr"] = list(w["addr"]); diff["addr"][k] = (w["addr"][k][0], w["addr"][k][1] + 1)
    assert _check_r1cs(circ, circ.assignment_ints(reroot(diff))[0]) != []


def test_macro_entries_produce_the_values_of_their_gadget_entries(monkeypatch):
    """One program entry per SHA-256 round / message-schedule step (OP_SHA_ROUND / OP_SHA_SCHED, the default) against one
    entry per gadget (HK_WPROG_NO_MACRO=1): the same value table, value for value, the same column map and matrices; the
    macro form is ~9 x shorter."""
    from hekaton_system_amd.sha_circuit import OP_SHA_ROUND, OP_SHA_SCHED, program_inputs, run_word_program
    built = {}
    for macro in (True, False):
        if not macro:
            monkeypatch.setenv("HK_WPROG_NO_MACRO", "1")
        circ = ShaMerkleSubcircuit("bn254", "parent", 2, n_portals=4)
        ws = [example_witness(circ, seed=s, entry_chal=77, tr_chal=99) for s in (1, 2, 3)]
        ops, refs, vmap = circ.tape.word_program(circ.n_v)
        built[macro] = (ops, vmap, circ.tape.n_values, run_word_program(ops, refs, circ.tape.n_values, program_inputs(circ, ws)),
                        circ.n_c, circ.n_v)
    (ops1, vmap1, nv1, vals1, nc1, nvar1), (ops0, vmap0, nv0, vals0, nc0, nvar0) = built[True], built[False]
    assert (nv1, nc1, nvar1) == (nv0, nc0, nvar0) and np.array_equal(vmap1, vmap0)
    assert np.array_equal(vals1, vals0)
    assert set(ops1[:, 0].tolist()) >= {OP_SHA_ROUND, OP_SHA_SCHED} and not np.any(ops0[:, 0] >= 8)
    n_round, n_sched = int(np.sum(ops1[:, 0] == OP_SHA_ROUND)), int(np.sum(ops1[:, 0] == OP_SHA_SCHED))
    assert n_round % 64 == 0 and n_sched == n_round // 64 * 48
    assert len(ops0) - len(ops1) == n_round * 8 + n_sched * 4            # 9 gadget entries -> 1, 5 -> 1


@pytest.mark.parametrize("kind", ["leaf", "parent", "root", "padding"])
def test_word_program_reproduces_the_trace(kind):
    """The word program the tape records (what the GPU interprets) + the column map + the host-side full-width values
    rebuild exactly the assignment the numpy EVAL interpreter emits."""
    from hekaton_system_amd.sha_circuit import full_values, program_inputs, run_word_program
    circ = ShaMerkleSubcircuit("bn254", kind, 2, n_portals=4, last=(kind == "padding"))
    ws = [example_witness(circ, seed=s, entry_chal=77, tr_chal=99) for s in (4, 5, 6)]
    ops, refs, vmap = circ.tape.word_program(circ.n_v)
    vals = run_word_program(ops, refs, circ.tape.n_values, program_inputs(circ, ws))
    bits_ref, full_ref, _ = circ.witness_batch(ws)
    mask = vmap != 0xffffffff
    got = ((vals[vmap[mask] >> 5] >> (vmap[mask] & 31)[:, None]) & 1).T.astype(np.uint8)
    assert np.array_equal(got, bits_ref[:, mask])
    cols, fv = full_values(circ, ws)
    # full-width columns = the host-computed values + the membership block (filled by k_poseidon_path on the device, by
    # poseidon_path_trace here)
    block = list(range(circ.pos_col0, circ.pos_col0 + circ.pos_cols))
    assert sorted(cols.tolist() + block) == sorted([1, 2, 3] + list(full_ref.keys()))
    assert set(np.nonzero(~mask)[0].tolist()) == set(cols.tolist()) | set(block) | {0}
    zs = circ.assignment_ints(ws)
    fc = circ.fc
    from hekaton_system_amd.sha_circuit import poseidon_inputs, poseidon_path_trace
    leaves, sibs, idx = poseidon_inputs(circ, ws)
    for b in range(3):
        dec = fc.dec(fv[b])
        assert all(dec[k] == zs[b][c] for k, c in enumerate(cols.tolist()))
        tr = poseidon_path_trace(circ.leaf_cfg, circ.node_cfg, fc.dec(leaves[b]), fc.dec(sibs[b]), int(idx[b]))
        assert len(tr) == circ.pos_cols and tr == zs[b][circ.pos_col0:circ.pos_col0 + circ.pos_cols]
        assert tr[-2] == ws[b]["root"]                   # state[1] of the last permutation = the root
