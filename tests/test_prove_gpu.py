"""GPU parity for the rest of the hot path: NTT, witness map, commit, prove (C ABI vs big-int oracle).

The end-to-end cases read like the reference's own tests (cp-groth16/src/lib.rs:140-180 two-stage
PolyEvalCircuit: commit -> prove -> verify_proof; :274-313 single stage) with the randomness injected
so that outputs can also be compared bit for bit with the oracle's restatement of prover.rs.
"""
import random

import numpy as np
import pytest

from hekaton_system_amd import capi
from oracle.pyref import curve, groth16, pairing_bn254
from oracle.pyref.codec import Codec
from oracle.pyref.params import CURVES, BN254
from oracle.pyref.poly import Domain
from tests.util import csr_from_rows, synthetic_r1cs, pk_upload_from_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log_m", [1, 2, 3, 5, 10, 11, 12, 13])
def test_ntt_matches_ark_poly_conventions(log_m, ctx_bn254):
    cp = BN254
    cd = Codec(cp)
    rnd = random.Random(log_m)
    m = 1 << log_m
    x = [rnd.randrange(cp.r) for _ in range(m)]
    dom = Domain(cp, m)
    g = cp.fr_generator
    for inverse, coset, want in ((0, 0, dom.fft(x)), (1, 0, dom.ifft(x)), (0, 1, dom.coset_fft(x, g)),
                                 (1, 1, dom.coset_ifft(x, g))):
        buf = cd.fr_vec_mont(x)
        ctx_bn254.ntt(buf, log_m, inverse=inverse, coset=coset)
        assert cd.fr_vec_from_mont(buf) == want, (inverse, coset)


def test_ntt_domain_too_large(ctx_bn254):
    buf = np.zeros(64, dtype=np.uint8)
    with pytest.raises(capi.HekatonError) as e:
        ctx_bn254.ntt(buf, 29)           # BN254 TWO_ADICITY = 28 -> PolynomialDegreeTooLarge
    assert e.value.status == capi.HK_ERR_DOMAIN_TOO_LARGE


@pytest.mark.parametrize("n_c,n_inst", [(3, 2), (24, 3), (200, 4), (1021, 4)])
def test_witness_map_vs_oracle(n_c, n_inst, ctx_bn254):
    cp = BN254
    cd = Codec(cp)
    rnd = random.Random(n_c)
    cs = synthetic_r1cs(cp, rnd, n_inst, 10, n_c)
    assert cs.is_satisfied()
    A, B, C = cs.matrices()
    z = cs.full_assignment()
    want = groth16.witness_map_from_matrices(cp, A, B, C, cs.num_instance, n_c, z)
    assert want[-1] == 0
    got, m = ctx_bn254.witness_map(csr_from_rows(cd, A), csr_from_rows(cd, B), csr_from_rows(cd, C),
                                   cs.num_instance, n_c, cd.fr_vec_mont(z))
    assert m == len(want)
    assert cd.fr_vec_from_mont(got) == want


def _setup(cp, cs, rnd):
    r = cp.r
    return groth16.generate_parameters(
        cp, cs, rnd.randrange(1, r), rnd.randrange(1, r), rnd.randrange(1, r),
        [rnd.randrange(1, r) for _ in cs.stage_ranges], rnd.randrange(2, r), 5, 11)


def test_poly_commit_two_stage_like_reference(ctx_bn254):
    """cp-groth16/src/lib.rs:140-180: commit stage 0, prove, verify_proof == true."""
    cp = BN254
    cd = Codec(cp)
    rnd = random.Random(42)
    poly = [rnd.randrange(cp.r) for _ in range(10)] + [1]
    cs, inputs = groth16.poly_eval_circuit(cp, poly, rnd.randrange(cp.r), two_stage=True)
    pk, td = _setup(cp, cs, rnd)
    dpk = pk_upload_from_oracle(ctx_bn254, cd, pk, cs)
    kappa, r_, s_ = rnd.randrange(cp.r), rnd.randrange(cp.r), rnd.randrange(cp.r)
    # commit (committer.rs:55-98)
    com = cd.g1_from(dpk.commit(0, cd.fr_vec_mont(cs.stage_witness(0)), cd.fr_vec_mont([kappa])))
    assert com == groth16.commit(cp, cs, pk, 0, kappa)
    # wrong witness length -> the committer.rs:83 assert
    with pytest.raises(capi.HekatonError) as e:
        dpk.commit(0, cd.fr_vec_mont(cs.stage_witness(0)[:-1]), cd.fr_vec_mont([kappa]))
    assert e.value.status == capi.HK_ERR_LEN
    # prove (prover.rs:53-156 + committer.rs:100-123)
    a, b, c = dpk.prove(cd.fr_vec_mont(cs.full_assignment()), cd.fr_vec_mont([r_]), cd.fr_vec_mont([s_]),
                        cd.fr_vec_mont([kappa]))
    proof = groth16.Proof(cd.g1_from(a), cd.g2_from(b), cd.g1_from(c), [com])
    want = groth16.prove(cp, cs, pk, [com], [kappa], r_, s_)
    assert (proof.a, proof.b, proof.c) == (want.a, want.b, want.c)
    assert groth16.verify_proof_trapdoor(cp, cs, pk, td, proof, [kappa], r_, s_)
    assert pairing_bn254.verify_proof(pk.vk, proof, inputs)          # lib.rs:179
    # comm_rands of the wrong length -> committer.rs:112 assert
    with pytest.raises(capi.HekatonError):
        dpk.prove(cd.fr_vec_mont(cs.full_assignment()), cd.fr_vec_mont([r_]), cd.fr_vec_mont([s_]),
                  np.zeros(0, np.uint8))
    dpk.free()


def test_poly_commit_single_stage_like_reference(ctx_bn254):
    """cp-groth16/src/lib.rs:274-313: no commitments, `cb.prove(&[], &[], rng)`; also r = s = 0
    (prove_last_stage_without_zk, prover.rs:36-50)."""
    cp = BN254
    cd = Codec(cp)
    rnd = random.Random(43)
    poly = [rnd.randrange(cp.r) for _ in range(10)] + [1]
    cs, inputs = groth16.poly_eval_circuit(cp, poly, rnd.randrange(cp.r), two_stage=False)
    pk, td = _setup(cp, cs, rnd)
    dpk = pk_upload_from_oracle(ctx_bn254, cd, pk, cs)
    for r_, s_ in ((rnd.randrange(cp.r), rnd.randrange(cp.r)), (0, 0)):
        a, b, c = dpk.prove(cd.fr_vec_mont(cs.full_assignment()), cd.fr_vec_mont([r_]),
                            cd.fr_vec_mont([s_]), np.zeros(0, np.uint8))
        proof = groth16.Proof(cd.g1_from(a), cd.g2_from(b), cd.g1_from(c), [])
        want = groth16.prove(cp, cs, pk, [], [], r_, s_)
        assert (proof.a, proof.b, proof.c) == (want.a, want.b, want.c)
        assert pairing_bn254.verify_proof(pk.vk, proof, inputs)
    dpk.free()


def test_prove_synthetic_two_stage(ctx_bn254):
    """Larger synthetic subcircuit shape (ROM-like: 4 instance variables, stage-0 subtrace witnesses)."""
    cp = BN254
    cd = Codec(cp)
    rnd = random.Random(44)
    cs = synthetic_r1cs(cp, rnd, n_inst=4, n_free=40, n_c=200, two_stage_split=16)
    pk, td = _setup(cp, cs, rnd)
    dpk = pk_upload_from_oracle(ctx_bn254, cd, pk, cs)
    kappa, r_, s_ = rnd.randrange(cp.r), rnd.randrange(cp.r), rnd.randrange(cp.r)
    com = cd.g1_from(dpk.commit(0, cd.fr_vec_mont(cs.stage_witness(0)), cd.fr_vec_mont([kappa])))
    assert com == groth16.commit(cp, cs, pk, 0, kappa)
    zdev = capi.DeviceBuffer.from_host(ctx_bn254, cd.fr_vec_mont(cs.full_assignment()))
    a, b, c = dpk.prove(zdev, cd.fr_vec_mont([r_]), cd.fr_vec_mont([s_]), cd.fr_vec_mont([kappa]),
                        n_v=len(cs.full_assignment()))
    proof = groth16.Proof(cd.g1_from(a), cd.g2_from(b), cd.g1_from(c), [com])
    assert groth16.verify_proof_trapdoor(cp, cs, pk, td, proof, [kappa], r_, s_)
    assert pairing_bn254.verify_proof(pk.vk, proof, cs.instance[1:])
    zdev.free(); dpk.free()


def test_witness_map_on_reference_circom_r1cs(ctx_bn254):
    """The reference's own `.r1cs` known-answer blob (circom-compat/src/lib.rs:549-606) as the matrix input
    path: hekaton_system_amd.circom -> hk_csr -> hk_witness_map, against the oracle's restatement."""
    import os
    from hekaton_system_amd import circom
    from hekaton_system_amd.cp_groth16 import FrCodec
    here = os.path.dirname(os.path.abspath(__file__))
    file = circom.R1CSFile.new(open(os.path.join(here, "golden", "circom_sample.r1cs"), "rb").read())
    fc = FrCodec("bn254")
    (A, B, C), n_inst, n_wit = file.to_csr(fc)
    cd = Codec(BN254)
    rnd = random.Random(11)
    z = [1] + [rnd.randrange(BN254.r) for _ in range(n_inst + n_wit - 1)]

    def rows(M):
        rp, col, val = M
        vals = fc.dec(val)
        return [[(vals[k], int(col[k])) for k in range(int(rp[i]), int(rp[i + 1]))] for i in range(len(rp) - 1)]

    want = groth16.witness_map_from_matrices(BN254, rows(A), rows(B), rows(C), n_inst, file.header.n_constraints, z)
    got, m = ctx_bn254.witness_map(A, B, C, n_inst, file.header.n_constraints, cd.fr_vec_mont(z))
    assert m == len(want) == 8
    assert cd.fr_vec_from_mont(got) == want
