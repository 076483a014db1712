"""GPU: one whole job, the shape of the reference's end-to-end test (distributed-prover/src/subcircuit_circuit.rs:405-438,
coordinator.rs:362-420), on REAL SHA-256 big-merkle subcircuits (hekaton_system_amd/sha_circuit.py, 8 subcircuits, 5 key
classes):

  setup        5 proving-key classes (`generate_parameters`), the TIPA SRS, the aggregation key
  round 1      every subcircuit commits to its portal subtraces (hk_commit, kappa = first draw of ChaCha12Rng(seed))
  coordinator  super_com = commit_only_left(coms) (coordinator.rs:339); (entry_chal, tr_chal) = SHA-256 of it
               (rom_transcript.rs:42-75); running evaluations
  round 2      every subcircuit proves (hk_prove) from its generated assignment; the commitment inside the proof is the
               round-1 commitment (same seed)
  aggregate    `agg_subcircuit_proofs` under a merlin transcript labelled as in coordinator.rs:411; TIPA verifies
  verifier     re-derives every challenge from public values and checks z_lr = prod cross[i][j]^(s^i t^j) with GPU GT powers

Nothing here is pinned by reference bytes (no golden vectors exist for this path); what is pinned is that every one of
the reference's own assertions holds on this prover's outputs: aggregation.rs:208-216 (each proof satisfies its class's
verifier equation - via :265-269), :265-269, :340."""
import numpy as np
import pytest

from hekaton_system_amd import aggregation as agg, tipa
from hekaton_system_amd.chacha import ChaCha12Rng
from hekaton_system_amd.cp_groth16 import FrCodec, Proof, SeededRng, generate_parameters, CURVE_PARAMS
from hekaton_system_amd.merlin import Transcript as Merlin
from hekaton_system_amd.sha_circuit import ShaMerkleJob

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_whole_job_commit_prove_aggregate_verify(cname, ctx_bn254, ctx_bls):
    ctx = ctx_bn254 if cname == "bn254" else ctx_bls
    r = CURVE_PARAMS[cname]["r"]
    fc = FrCodec(cname)
    n, ns, n_portals = 8, 1, 4
    leaves = [bytes([(29 * i + 3 * k) & 0xff for k in range(64)]) for i in range(n // 2)]
    job = ShaMerkleJob(cname, n, ns, n_portals, leaves)
    # ---- setup
    classes = {}
    for idx in range(n):
        key = job.class_of(idx)
        if key not in classes:
            circ = job.make_class(idx)
            pk, _td = generate_parameters(circ, cname, SeededRng(bytes([len(classes) + 1]) * 32), ctx)
            classes[key] = (circ, pk, pk.upload(ctx))
    assert len(classes) == 5
    rng = ChaCha12Rng(b"\x07" * 32)
    srs = tipa.setup(ctx, cname, n, rng.fr(r), rng.fr(r))
    apk = agg.AggProvingKey(ctx, cname, srs.ck, [classes[job.class_of(i)][1].vk for i in range(n)])
    # ---- round 1
    seeds = [rng.gen_seed() for _ in range(n)]
    kappas = [ChaCha12Rng(sd).fr(r) for sd in seeds]
    coms = [classes[job.class_of(i)][2].commit(0, fc.enc(job.stage0_ints(i)), fc.enc1(kappas[i])) for i in range(n)]
    # ---- coordinator
    super_com = apk.com.commit_only_left(srs.ck, np.concatenate(coms))
    entry_chal, tr_chal = agg.rom_challenges(super_com, r)
    job.set_challenges(entry_chal, tr_chal)
    # ---- round 2
    proofs = []
    for i in range(n):
        circ, pk, dpk = classes[job.class_of(i)]
        w = job.inputs(i)
        z = circ.assignment_ints(w)[0]
        assert z[1:4] == [entry_chal, tr_chal, job.root]
        assert z[circ.N_INST:circ.N_INST + circ.n0] == job.stage0_ints(i)
        a, b, c = dpk.prove(circ.assignment_bytes(w)[0], fc.enc1(rng.fr(r)), fc.enc1(rng.fr(r)), fc.enc([kappas[i]]), n_v=circ.n_v)
        proofs.append(Proof(a, b, c, [coms[i]]))
    # ---- aggregate (asserts aggregation.rs:265-269 and :340 on the way)
    pub = [entry_chal, tr_chal, job.root]
    proof, inst = apk.agg_subcircuit_proofs(Merlin(b"test-e2e"), super_com, proofs, pub, srs)
    # ---- verifier: challenges from the public messages, instance from the cross terms
    F = apk.F
    pt = Merlin(b"test-e2e")
    pt.append_serializable(b"AB-commitment", inst["com_ab"].serialize_uncompressed())
    pt.append_serializable(b"C-commitment", inst["com_c"].serialize_uncompressed())
    pt.append_serializable(b"D-commitment", super_com.serialize_uncompressed())
    twist = pt.challenge_scalar(b"r-random-fiatshamir", r)
    z = inst["cross_terms"]
    ser = (4).to_bytes(8, "little") + b"".join((4).to_bytes(8, "little") + b"".join(F.serialize(e) for e in row) for row in z)
    pt.append_serializable(b"cross-terms", ser)
    s = pt.challenge_scalar(b"s-random-fiatshamir", r)
    t = pt.challenge_scalar(b"t-random-fiatshamir", r)
    assert twist == inst["twist"]
    exps = [pow(s, i, r) * pow(t, j, r) % r for i in range(4) for j in range(4)]
    pw = ctx.gt_pow(np.frombuffer(b"".join(F.encode(z[i][j]) for i in range(4) for j in range(4)), np.uint8), fc.enc(exps))
    z_lr = F.one
    for k in range(16):
        z_lr = F.mul(z_lr, F.decode(pw[k]))
    assert z_lr == inst["output"]
    s2, s3, t2, t3 = s * s % r, pow(s, 3, r), t * t % r, pow(t, 3, r)
    com_in = apk.com_s[0] + apk.com_s[1] * pub[0] + apk.com_s[2] * pub[1] + apk.com_s[3] * pub[2]
    com_lr = (inst["com_ab"] + com_in * s + super_com * s2 + inst["com_c"] * s3) + \
             (apk.com_h * t + apk.com_delta0 * t2 + apk.com_delta1 * t3)
    assert com_lr == inst["commitment"]
    T = tipa.Tipp(ctx, cname)
    assert T.verify(tipa.verifier_key(ctx, cname, srs), com_lr, z_lr, twist, proof)
    # ---- a job whose subcircuit 5 proved against other challenges (a stale coordinator message) does not aggregate
    circ, pk, dpk = classes[job.class_of(5)]
    stale = ShaMerkleJob(cname, n, ns, n_portals, leaves, entry_chal + 1, tr_chal)
    a, b, c = dpk.prove(circ.assignment_bytes(stale.inputs(5))[0], fc.enc1(3), fc.enc1(4), fc.enc([kappas[5]]), n_v=circ.n_v)
    bad = list(proofs)
    bad[5] = Proof(a, b, c, [coms[5]])
    with pytest.raises(AssertionError):
        apk.agg_front(super_com, bad, pub, pt=Merlin(b"test-e2e"))
    for _c, _pk, dpk in classes.values():
        dpk.free()
    for rb in srs.resident.values():
        rb.free()
