"""GPU: edge cases of the C ABI that the reference's domain has — empty / ragged inputs, size-1 domains,
single-stage keys, wrong lengths (the reference asserts / unwraps there), repeated uploads."""
import random

import numpy as np
import pytest

from hekaton_system_amd import capi
from oracle.pyref import curve, groth16
from oracle.pyref.codec import Codec
from oracle.pyref.params import BN254
from oracle.pyref.poly import Domain
from tests.util import synthetic_r1cs, pk_upload_from_oracle, csr_from_rows

pytestmark = pytest.mark.gpu


def test_ntt_size_one_and_two(ctx_bn254):
    cd = Codec(BN254)
    for log_m in (0, 1):
        x = [7, 9][: 1 << log_m]
        for inv, coset in ((0, 0), (1, 0), (0, 1), (1, 1)):
            buf = cd.fr_vec_mont(x)
            ctx_bn254.ntt(buf, log_m, inverse=inv, coset=coset)
            dom = Domain(BN254, 1 << log_m)
            g = BN254.fr_generator
            want = {(0, 0): dom.fft, (1, 0): dom.ifft}.get((inv, coset), None)
            want = want(x) if want else (dom.coset_fft(x, g) if not inv else dom.coset_ifft(x, g))
            assert cd.fr_vec_from_mont(buf) == want


def test_msm_all_zero_scalars_and_all_infinity_bases(ctx_bn254):
    cd = Codec(BN254)
    G = curve.G1(BN254)
    bases = [G.mul(G.gen, k + 2) for k in range(40)]
    assert cd.g1_from(ctx_bn254.msm_g1(cd.g1_vec(bases), cd.fr_vec_mont([0] * 40))) is None
    assert cd.g1_from(ctx_bn254.msm_g1(cd.g1_vec([None] * 40), cd.fr_vec_mont(list(range(1, 41))))) is None
    # sum cancels exactly: P*(r-1) + P = infinity
    assert cd.g1_from(ctx_bn254.msm_g1(cd.g1_vec([bases[0], bases[0]]), cd.fr_vec_mont([BN254.r - 1, 1]))) is None


def test_prove_with_empty_stage0_and_bad_lengths(ctx_bn254):
    """A two-stage key whose stage 0 allocates no witnesses (an empty subtrace) must still commit
    (com = kappa * delta) and prove; wrong n_v / h_len are length errors (prover.rs:128, committer.rs:83)."""
    cp = BN254
    cd = Codec(cp)
    rnd = random.Random(99)
    cs = groth16.R1CS(cp.r)
    cs.begin_stage(); cs.end_stage()                       # empty stage 0
    cs.begin_stage()
    x = cs.alloc_instance(5)
    w = cs.alloc_witness(25)
    cs.enforce([(1, x)], [(1, x)], [(1, w)])
    cs.end_stage()
    assert cs.is_satisfied()
    pk, td = groth16.generate_parameters(cp, cs, 3, 5, 7, [11, 13], 17, 2, 3)
    dpk = pk_upload_from_oracle(ctx_bn254, cd, pk, cs)
    kappa, r_, s_ = 21, 22, 23
    com = cd.g1_from(dpk.commit(0, np.zeros(0, np.uint8), cd.fr_vec_mont([kappa]), n=0))
    assert com == groth16.commit(cp, cs, pk, 0, kappa)
    a, b, c = dpk.prove(cd.fr_vec_mont(cs.full_assignment()), cd.fr_vec_mont([r_]), cd.fr_vec_mont([s_]),
                        cd.fr_vec_mont([kappa]))
    proof = groth16.Proof(cd.g1_from(a), cd.g2_from(b), cd.g1_from(c), [com])
    assert groth16.verify_proof_trapdoor(cp, cs, pk, td, proof, [kappa], r_, s_)
    with pytest.raises(capi.HekatonError) as e:              # assignment of the wrong length
        dpk.prove(cd.fr_vec_mont(cs.full_assignment() + [1]), cd.fr_vec_mont([r_]), cd.fr_vec_mont([s_]),
                  cd.fr_vec_mont([kappa]))
    assert e.value.status == capi.HK_ERR_LEN
    dpk.free()
    # h_g one element short -> the prover.rs:128 assert, reported at upload
    A, B, C = cs.matrices()
    with pytest.raises(capi.HekatonError) as e:
        ctx_bn254.pk_upload(a_g=cd.g1_vec(pk.a_g), b_g=cd.g1_vec(pk.b_g), b_h=cd.g2_vec(pk.b_h),
                            h_g=cd.g1_vec(pk.h_g[:-1]), ck_stages=[cd.g1_vec(v) for v in pk.ck.deltas_abc_g],
                            deltas_g=cd.g1_vec(pk.deltas_g), last_delta_h=cd.g2_vec([pk.last_delta_h()]),
                            alpha_g=cd.g1_vec([pk.vk.alpha_g]), beta_g=cd.g1_vec([pk.beta_g]),
                            beta_h=cd.g2_vec([pk.vk.beta_h]),
                            matrices=(csr_from_rows(cd, A), csr_from_rows(cd, B), csr_from_rows(cd, C)),
                            n_inst=cs.num_instance, n_constraints=cs.num_constraints)
    assert e.value.status == capi.HK_ERR_LEN


def test_prove_with_empty_last_stage(ctx_bn254):
    """The mirror case: every witness is committed in stage 0 and the last stage allocates none, so the L query of
    prover.rs:111-118 is an MSM over ZERO bases (its table is empty: the accumulate loop must not touch it)."""
    cp = BN254
    cd = Codec(cp)
    cs = groth16.R1CS(cp.r)
    cs.begin_stage()
    w = cs.alloc_witness(36)
    v = cs.alloc_witness(6)
    cs.end_stage()
    cs.begin_stage()
    x = cs.alloc_instance(6)
    cs.enforce([(1, x)], [(1, v)], [(1, w)])
    cs.enforce([(1, v)], [(1, "one")], [(1, x)])
    cs.end_stage()
    assert cs.is_satisfied()
    pk, td = groth16.generate_parameters(cp, cs, 3, 5, 7, [11, 13], 17, 2, 3)
    assert len(pk.ck.deltas_abc_g[-1]) == 0
    dpk = pk_upload_from_oracle(ctx_bn254, cd, pk, cs)
    kappa, r_, s_ = 31, 32, 33
    com = cd.g1_from(dpk.commit(0, cd.fr_vec_mont([36, 6]), cd.fr_vec_mont([kappa])))
    assert com == groth16.commit(cp, cs, pk, 0, kappa)
    a, b, c = dpk.prove(cd.fr_vec_mont(cs.full_assignment()), cd.fr_vec_mont([r_]), cd.fr_vec_mont([s_]),
                        cd.fr_vec_mont([kappa]))
    proof = groth16.Proof(cd.g1_from(a), cd.g2_from(b), cd.g1_from(c), [com])
    assert groth16.verify_proof_trapdoor(cp, cs, pk, td, proof, [kappa], r_, s_)
    want = groth16.prove(cp, cs, pk, [com], [kappa], r_, s_)
    assert (proof.a, proof.b, proof.c) == (want.a, want.b, want.c)
    dpk.free()


def test_two_keys_resident_at_once(ctx_bn254):
    """Several proving-key classes live on the device together (big-merkle has 5, tree_hash_circuit.rs:192-216)."""
    cp = BN254
    cd = Codec(cp)
    rnd = random.Random(5)
    keys = []
    for k in range(2):
        cs = synthetic_r1cs(cp, rnd, n_inst=4, n_free=20 + 10 * k, n_c=40 + 30 * k, two_stage_split=8)
        pk, td = groth16.generate_parameters(cp, cs, 3 + k, 5, 7, [11, 13], 17 + k, 2, 3)
        keys.append((cs, pk, td, pk_upload_from_oracle(ctx_bn254, cd, pk, cs)))
    for cs, pk, td, dpk in reversed(keys):
        kappa, r_, s_ = 1 + rnd.randrange(cp.r - 1), rnd.randrange(cp.r), rnd.randrange(cp.r)
        com = cd.g1_from(dpk.commit(0, cd.fr_vec_mont(cs.stage_witness(0)), cd.fr_vec_mont([kappa])))
        a, b, c = dpk.prove(cd.fr_vec_mont(cs.full_assignment()), cd.fr_vec_mont([r_]), cd.fr_vec_mont([s_]),
                            cd.fr_vec_mont([kappa]))
        proof = groth16.Proof(cd.g1_from(a), cd.g2_from(b), cd.g1_from(c), [com])
        assert groth16.verify_proof_trapdoor(cp, cs, pk, td, proof, [kappa], r_, s_)
    for *_, dpk in keys:
        dpk.free()


@pytest.mark.parametrize("group", [1, 2])
def test_scalar_pairing_like_reference(group, ctx_bn254):
    """pairing_ops.rs:32-39 `scalar_pairing`: element-wise scalar multiplication + batch normalisation, as the
    aggregator calls it with powers of a challenge (aggregation.rs:236-242)."""
    cp = BN254
    cd = Codec(cp)
    G = curve.G1(cp) if group == 1 else curve.G2(cp)
    rnd = random.Random(group)
    n = 37
    pts = [G.mul(G.gen, rnd.randrange(1, cp.r)) for _ in range(n)]
    pts[3] = None
    twist = rnd.randrange(cp.r)
    scal = [pow(twist, i, cp.r) for i in range(n)]            # structured_scalar_power(num, s)
    scal[5] = 0
    enc = cd.g1_vec if group == 1 else cd.g2_vec
    dec = cd.g1_from if group == 1 else cd.g2_from
    pb = cd.g1_bytes if group == 1 else cd.g2_bytes
    out = ctx_bn254.scalar_pairing(group, enc(pts), cd.fr_vec_mont(scal))
    got = [dec(out[i * pb:(i + 1) * pb]) for i in range(n)]
    assert got == [G.mul(p, s) if p is not None else None for p, s in zip(pts, scal)]


@pytest.mark.parametrize("group", [1, 2])
def test_fixed_base_table_cache(group):
    """hk_fixed_base keeps the window table of a base it has multiplied before (per context, at most 8 bases): the first
    call (table built into a cache slot), the second (cache hit), a call with the cache switched off and the oracle agree;
    twelve distinct bases - more than the cache holds - each against the oracle; the infinity base."""
    import os
    cp = BN254
    cd = Codec(cp)
    G = curve.G1(cp) if group == 1 else curve.G2(cp)
    enc = cd.g1_vec if group == 1 else cd.g2_vec
    dec = cd.g1_from if group == 1 else cd.g2_from
    pb = cd.g1_bytes if group == 1 else cd.g2_bytes
    rnd = random.Random(70 + group)
    ctx = capi.Context("bn254", 0)
    try:
        ks = [0, 1, cp.r - 1] + [rnd.randrange(cp.r) for _ in range(30)]
        sc = cd.fr_vec_mont(ks)
        base = G.mul(G.gen, 12345)
        first = ctx.fixed_base(group, enc([base]), sc).copy()
        second = ctx.fixed_base(group, enc([base]), sc).copy()
        os.environ["HK_FB_NO_CACHE"] = "1"
        try:
            plain = ctx.fixed_base(group, enc([base]), sc).copy()
        finally:
            del os.environ["HK_FB_NO_CACHE"]
        assert np.array_equal(first, second) and np.array_equal(first, plain)
        assert [dec(first[i * pb:(i + 1) * pb]) for i in range(len(ks))] == [G.mul(base, k) for k in ks]
        for j in range(12):
            bj = G.mul(G.gen, 1000 + j)
            for _ in range(2):                                        # miss (or private build), then hit when cached
                out = ctx.fixed_base(group, enc([bj]), cd.fr_vec_mont([j + 2, cp.r - 1 - j]))
                assert [dec(out[:pb]), dec(out[pb:])] == [G.mul(bj, j + 2), G.mul(bj, cp.r - 1 - j)], j
        out = ctx.fixed_base(group, enc([None]), cd.fr_vec_mont([5, 7]))
        assert dec(out[:pb]) is None and dec(out[pb:]) is None
    finally:
        ctx.close()


@pytest.mark.parametrize("group", [1, 2])
def test_resident_bases_msm_matches_plain_msm_and_oracle(group, ctx_bn254):
    """hk_bases_upload / hk_msm_bases (the aggregator's static-SRS MSMs, kzg.rs:151-152): same group element as the
    one-off hk_msm over the same bases and as the oracle's Pippenger; ark length semantics."""
    from hekaton_system_amd.cp_groth16 import CURVE_PARAMS, FrCodec
    from oracle.c_oracle import COracle
    fc = FrCodec("bn254")
    p = CURVE_PARAMS["bn254"]
    rnd = random.Random(40 + group)
    gen = fc.g1(p["g1"]) if group == 1 else fc.g2(p["g2"])
    co = COracle("bn254")
    msm = ctx_bn254.msm_g1 if group == 1 else ctx_bn254.msm_g2
    pb = ctx_bn254.g1_bytes if group == 1 else ctx_bn254.g2_bytes
    for n in (1, 2, 64, 1000, 5000) + ((9000,) if group == 1 else ()):      # G1: shift tables from 8 193 bases on
        bases = ctx_bn254.fixed_base(group, gen, fc.enc([rnd.randrange(1, p["r"]) for _ in range(n)]))
        if n > 2:
            bases[pb:2 * pb] = 0                                    # an infinity base in the set
        rb = ctx_bn254.bases_upload(group, bases)
        for dense in (True, False):
            sc = fc.enc([rnd.randrange(p["r"]) if (dense or rnd.random() < 0.15) else rnd.randrange(2) for _ in range(n)])
            got = rb.msm(sc)
            assert np.array_equal(got, msm(bases, sc))
            assert np.array_equal(got, co.msm(group, bases, sc))
        # canonical (non-Montgomery) scalars, as msm_bigint takes them
        ints = [rnd.randrange(p["r"]) for _ in range(n)]
        assert np.array_equal(rb.msm(fc.enc_canon(ints), montgomery=False), rb.msm(fc.enc(ints)))
        if n >= 64:
            short = fc.enc([rnd.randrange(p["r"]) for _ in range(n - 7)])
            with pytest.raises(capi.HekatonError) as e:             # msm: Err(min_len)
                rb.msm(short)
            assert e.value.status == capi.HK_ERR_LEN
            assert np.array_equal(rb.msm(short, checked=False), msm(bases, short, checked=False))   # msm_unchecked zips
        rb.free()
    empty = ctx_bn254.bases_upload(group, np.zeros(0, np.uint8), n=0)
    assert not empty.msm(np.zeros(0, np.uint8)).any()
    empty.free()


def _tiny_key(ctx, seed=5):
    cp = BN254
    cd = Codec(cp)
    rnd = random.Random(seed)
    cs = synthetic_r1cs(cp, rnd, n_inst=2, n_free=6, n_c=12, two_stage_split=2)
    pk, td = groth16.generate_parameters(cp, cs, 3, 5, 7, [11, 13], 17, 2, 3)
    return cp, cd, cs, pk


def test_malformed_csr_is_an_argument_error_not_a_device_fault(ctx_bn254):
    """hk_witness_map / hk_pk_upload validate the matrices they are about to index with: a column
    >= n_v, a non-monotone row_ptr, a row_ptr that does not end at nnz, n_inst = 0 -> HK_ERR_ARG."""
    cp, cd, cs, pk = _tiny_key(ctx_bn254)
    A, B, C = (csr_from_rows(cd, M) for M in cs.matrices())
    z = cd.fr_vec_mont(cs.full_assignment())
    n_v = len(cs.full_assignment())
    ok, _m = ctx_bn254.witness_map(A, B, C, cs.num_instance, cs.num_constraints, z)
    assert len(ok)

    def expect_arg(A_, B_, C_, n_inst=cs.num_instance):
        with pytest.raises(capi.HekatonError) as e:
            ctx_bn254.witness_map(A_, B_, C_, n_inst, cs.num_constraints, z)
        assert e.value.status == capi.HK_ERR_ARG

    bad_col = (A[0], A[1].copy(), A[2]); bad_col[1][0] = n_v            # column index one past the assignment
    expect_arg(bad_col, B, C)
    bad_rp = (B[0].copy(), B[1], B[2]); bad_rp[0][1], bad_rp[0][2] = bad_rp[0][2] + 1, bad_rp[0][1]   # not monotone
    expect_arg(A, bad_rp, C)
    bad_end = (C[0].copy(), C[1], C[2]); bad_end[0][-1] += 1                                          # != nnz
    expect_arg(A, B, bad_end)
    expect_arg(A, B, C, n_inst=0)
    # the same matrices refused at key upload
    with pytest.raises(capi.HekatonError) as e:
        ctx_bn254.pk_upload(
            a_g=cd.g1_vec(pk.a_g), b_g=cd.g1_vec(pk.b_g), b_h=cd.g2_vec(pk.b_h), h_g=cd.g1_vec(pk.h_g),
            ck_stages=[cd.g1_vec(v) for v in pk.ck.deltas_abc_g], deltas_g=cd.g1_vec(pk.deltas_g),
            last_delta_h=cd.g2_vec([pk.last_delta_h()]), alpha_g=cd.g1_vec([pk.vk.alpha_g]),
            beta_g=cd.g1_vec([pk.beta_g]), beta_h=cd.g2_vec([pk.vk.beta_h]), matrices=(bad_col, B, C),
            n_inst=cs.num_instance, n_constraints=cs.num_constraints)
    assert e.value.status == capi.HK_ERR_ARG
    # and the context still works afterwards
    again, _ = ctx_bn254.witness_map(A, B, C, cs.num_instance, cs.num_constraints, z)
    assert np.array_equal(ok, again)


def test_key_of_another_context_is_refused(ctx_bn254):
    """hk_commit / hk_prove check pk->ctx == ctx, as hk_msm_bases does."""
    cp, cd, cs, pk = _tiny_key(ctx_bn254, seed=6)
    dpk = pk_upload_from_oracle(ctx_bn254, cd, pk, cs)
    other = capi.Context("bn254", 0)
    try:
        stolen = capi.DevicePk(other, dpk.handle)
        w0 = cd.fr_vec_mont(cs.stage_witness(0))
        with pytest.raises(capi.HekatonError) as e:
            stolen.commit(0, w0, cd.fr_vec_mont([3]))
        assert e.value.status == capi.HK_ERR_ARG
        with pytest.raises(capi.HekatonError) as e:
            stolen.prove(cd.fr_vec_mont(cs.full_assignment()), cd.fr_vec_mont([1]), cd.fr_vec_mont([2]), cd.fr_vec_mont([3]))
        assert e.value.status == capi.HK_ERR_ARG
    finally:
        other.close()
        dpk.free()


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_commit_batch_equals_one_commit_per_row(cname, ctx_bn254, ctx_bls, monkeypatch):
    """hk_commit_batch (every subcircuit of a key class in one call: element-wise products over the endomorphism + one sum
    per commitment for short stages) = hk_commit row by row: random witnesses, an all-zero row, kappa = 0, a batch of one,
    and a batch too long for the one-launch form (falls back to per-row calls); HK_ERR_LEN for a wrong stage length."""
    from hekaton_system_amd.cp_groth16 import FrCodec, SeededRng, generate_parameters
    from hekaton_system_amd.workload import make_config
    ctx = ctx_bn254 if cname == "bn254" else ctx_bls
    fc = FrCodec(cname)
    circ = make_config(cname, "tiny", 1)
    pk, _td = generate_parameters(circ, cname, SeededRng(b"\x07" * 32), ctx)
    dpk = pk.upload(ctx)
    n0, r = circ.n0, fc.r if hasattr(fc, "r") else None
    rnd = random.Random(3)
    from hekaton_system_amd.cp_groth16 import CURVE_PARAMS
    rmod = CURVE_PARAMS[cname]["r"]
    try:
        for batch in (1, 5, 64):
            rows = [[rnd.randrange(rmod) for _ in range(n0)] for _ in range(batch)]
            kappas = [rnd.randrange(rmod) for _ in range(batch)]
            if batch > 1:
                rows[1] = [0] * n0
                kappas[2 % batch] = 0
            wb = np.concatenate([np.frombuffer(bytes(fc.enc(rw)), np.uint8) for rw in rows])
            kb = np.frombuffer(bytes(fc.enc(kappas)), np.uint8)
            want = np.stack([dpk.commit(0, np.frombuffer(bytes(fc.enc(rows[b])), np.uint8), np.frombuffer(bytes(fc.enc1(kappas[b])), np.uint8), n=n0)
                             for b in range(batch)])
            got = dpk.commit_batch(0, wb, kb, n0, batch)
            assert np.array_equal(got, want), (cname, batch)
            dev = capi.DeviceBuffer.from_host(ctx, wb)
            assert np.array_equal(dpk.commit_batch(0, dev, kb, n0, batch), want), (cname, batch, "device rows")
            dev.free()
            monkeypatch.setenv("HK_MSM_NO_SMALL", "1")               # the per-row fallback
            assert np.array_equal(dpk.commit_batch(0, wb, kb, n0, batch), want), (cname, batch, "fallback")
            monkeypatch.delenv("HK_MSM_NO_SMALL")
        with pytest.raises(capi.HekatonError) as e:
            dpk.commit_batch(0, wb, kb, n0 + 1, 2)
        assert e.value.status == capi.HK_ERR_LEN
    finally:
        dpk.free()
