"""CPU: ark-serialize framing, ChaCha12Rng and Fr::rand restatements (hekaton_system_amd/ark_serialize.py).

PARITY UNPINNED by reference bytes (the reference holds no serialised fixture and its crates are absent).
Pinned instead by published vectors: the BLS12-381 generator's zcash encodings, the IETF/eSTREAM ChaCha zero-key
keystreams, the record sizes the reference's own types imply (104 B / 336 B), and round trips."""
import random

import numpy as np
import pytest

from hekaton_system_amd.ark_serialize import (ArkCodec, ChaChaRng, ChaCha12Rng, ProvingKeys, Reader, Writer,
                                              SerializationError, commitment_randomness, fr_rand_mont, to_packed,
                                              _chacha_block)
from hekaton_system_amd.cp_groth16 import (CURVE_PARAMS, FrCodec, Proof, ProvingKey, VerifyingKey, CommitterKey)
from hekaton_system_amd.worker import Stage0Response, Stage1Response
from oracle.pyref.params import CURVES
from oracle.pyref import curve as oc


def _rand_points(cname, group, n, seed):
    cp = CURVES[cname]
    rnd = random.Random(seed)
    G = oc.G1(cp) if group == 1 else oc.G2(cp)
    gen = cp.g1_gen if group == 1 else cp.g2_gen
    return [G.mul(gen, rnd.randrange(1, cp.r)) for _ in range(n)]


def _abi(cname, group, pts):
    fc = FrCodec(cname)
    pb = (2 if group == 1 else 4) * fc.qb
    out = []
    for P in pts:
        out.append(np.zeros(pb, np.uint8) if P is None else (fc.g1(P) if group == 1 else fc.g2(P)))
    return np.concatenate(out) if out else np.zeros(0, np.uint8)


def test_chacha_keystream_known_answers():
    """Zero key, zero nonce, block 0: ChaCha20 (RFC 7539 A.1 #1 / the original djb vectors) and ChaCha12/ChaCha8
    (eSTREAM-style vectors as published with the reference implementations)."""
    z = (0,) * 8
    ks = lambda rounds: b"".join(w.to_bytes(4, "little") for w in _chacha_block(z, 0, 0, rounds))
    assert ks(20)[:32].hex() == "76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
    assert ks(12)[:32].hex() == "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f"
    assert ks(8)[:32].hex() == "3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e"


def test_chacha_rng_word_discipline():
    """BlockRng: next_u64 = two consecutive u32 words (low first), also across the 64-word buffer boundary."""
    seed = bytes(range(32))
    a, b = ChaCha12Rng(seed), ChaCha12Rng(seed)
    words = [a.next_u32() for _ in range(200)]
    assert [b.next_u64() for _ in range(4)] == [words[2 * i] | (words[2 * i + 1] << 32) for i in range(4)]
    c = ChaCha12Rng(seed)
    for _ in range(63):
        c.next_u32()
    assert c.next_u64() == words[63] | (words[64] << 32)          # straddles the refill
    assert c.next_u32() == words[65]
    # blocks are consecutive counters of one keystream
    key = tuple(int.from_bytes(seed[4 * i:4 * i + 4], "little") for i in range(8))
    assert words[16:32] == _chacha_block(key, 1, 0, 12) and words[64:80] == _chacha_block(key, 4, 0, 12)
    assert ChaChaRng(seed, 20).next_u32() != words[0]


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
def test_fr_rand_is_masked_rejection_sampling_of_the_montgomery_limbs(cname):
    r = CURVE_PARAMS[cname]["r"]
    seed = bytes([7] * 32)
    k = commitment_randomness(cname, seed)
    ks = ChaCha12Rng(seed).fill_bytes(256)
    shave = 256 - r.bit_length()
    off = 0
    while True:                                        # first 32-byte draw whose masked value is < r
        v = int.from_bytes(ks[off:off + 32], "little") & ((1 << (256 - shave)) - 1)
        if v < r:
            break
        off += 32
    assert int.from_bytes(k.tobytes(), "little") == v
    assert commitment_randomness(cname, seed).tobytes() == k.tobytes()
    # the bytes ARE the ABI (Montgomery) form: as an Fr value it is v / R
    assert FrCodec(cname).dec(k) == [v * pow(1 << 256, -1, r) % r]


def test_bn254_sw_flags_known_answers():
    cd = ArkCodec("bn254")
    q = cd.q
    g = _abi("bn254", 1, [(1, 2)])
    assert cd.points_to_wire(1, g) == (1).to_bytes(32, "little") + (2).to_bytes(32, "little")       # y <= -y: no flag
    assert cd.points_to_wire(1, g, compress=True) == (1).to_bytes(32, "little")
    neg = _abi("bn254", 1, [(1, q - 2)])
    w = cd.points_to_wire(1, neg)
    assert w[:32] == (1).to_bytes(32, "little") and w[63] == ((q - 2) >> 248) | 0x80
    assert cd.points_to_wire(1, neg, compress=True)[31] == 0x80
    inf = np.zeros(64, np.uint8)
    assert cd.points_to_wire(1, inf) == bytes(63) + b"\x40"
    assert cd.points_to_wire(1, inf, compress=True) == bytes(31) + b"\x40"
    assert cd.points_to_wire(2, np.zeros(128, np.uint8)) == bytes(127) + b"\x40"
    # Fp2 ordering looks at c1 first: y = (c0, c1) with c1 small and c0 large is "positive"
    gx, gy = CURVE_PARAMS["bn254"]["g2"]
    w2 = cd.points_to_wire(2, _abi("bn254", 2, [(gx, gy)]))
    assert len(w2) == 128 and (w2[127] & 0x80 != 0) == (gy[1] > (q - 1) // 2)
    with pytest.raises(SerializationError):
        cd.points_from_wire(1, bytes(63) + b"\xc0", 1)                    # both flags
    with pytest.raises(SerializationError):
        cd.points_from_wire(1, q.to_bytes(32, "little") + bytes(32), 1)   # x not reduced


def test_bls12_381_zcash_generator_encodings():
    """The compressed / uncompressed encodings of the BLS12-381 generators are published constants
    (zcash/pairing, IETF pairing-friendly-curves draft)."""
    cd = ArkCodec("bls12_381")
    p = CURVE_PARAMS["bls12_381"]
    g1 = _abi("bls12_381", 1, [p["g1"]])
    c = cd.points_to_wire(1, g1, compress=True)
    assert c.hex() == ("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac58"
                       "6c55e83ff97a1aeffb3af00adb22c6bb")
    u = cd.points_to_wire(1, g1)
    assert u.hex() == ("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
                       "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1")
    g2 = _abi("bls12_381", 2, [p["g2"]])
    c2 = cd.points_to_wire(2, g2, compress=True)
    assert c2.hex() == ("93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
                        "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8")
    assert cd.points_to_wire(1, np.zeros(96, np.uint8), compress=True) == b"\xc0" + bytes(47)
    assert cd.points_to_wire(1, np.zeros(96, np.uint8)) == b"\x40" + bytes(95)
    for grp, abi in ((1, g1), (2, g2)):
        for comp in (False, True):
            w = cd.points_to_wire(grp, abi, comp)
            assert cd.points_from_wire(grp, w, 1, comp).tobytes() == abi.tobytes()


@pytest.mark.parametrize("cname", ["bn254", "bls12_381"])
@pytest.mark.parametrize("group", [1, 2])
def test_points_round_trip_both_modes(cname, group):
    cd = ArkCodec(cname)
    pts = _rand_points(cname, group, 6, 11) + [None]
    G = oc.G1(CURVES[cname]) if group == 1 else oc.G2(CURVES[cname])
    pts.append(G.neg(pts[0]))
    abi = _abi(cname, group, pts)
    for comp in (False, True):
        w = cd.points_to_wire(group, abi, comp)
        assert len(w) == len(pts) * cd.point_size(group, comp)
        assert cd.points_from_wire(group, w, len(pts), comp).tobytes() == abi.tobytes()
    # P and -P differ only in the sign flag when compressed
    sz = cd.point_size(group, True)
    w = cd.points_to_wire(group, abi, True)
    a, b = bytearray(w[:sz]), bytearray(w[-sz:])
    if cname == "bn254":
        assert a[-1] ^ b[-1] == 0x80 and a[:-1] == b[:-1]
    else:
        assert a[0] ^ b[0] == 0x20 and a[1:] == b[1:]


def test_response_records_have_the_reference_sizes_and_round_trip():
    """Stage0Response 104 B, Stage1Response 336 B (BN254, uncompressed)."""
    cd = ArkCodec("bn254")
    g1s = _abi("bn254", 1, _rand_points("bn254", 1, 4, 5))
    g2s = _abi("bn254", 2, _rand_points("bn254", 2, 1, 6))
    r0 = Stage0Response(0x0102030405, g1s[:64].copy(), bytes(range(32)))
    w0 = cd.stage0_response_to_wire(r0)
    assert len(w0) == 104 == cd.stage0_response_size()
    assert w0[:8] == (0x0102030405).to_bytes(8, "little") and w0[72:] == bytes(range(32))
    b0 = cd.stage0_response_from_wire(w0)
    assert (b0.subcircuit_idx, b0.com.tobytes(), b0.com_seed) == (r0.subcircuit_idx, r0.com.tobytes(), r0.com_seed)
    proof = Proof(g1s[64:128].copy(), g2s.copy(), g1s[128:192].copy(), [g1s[192:256].copy()])
    r1 = Stage1Response(77, proof)
    w1 = cd.stage1_response_to_wire(r1)
    assert len(w1) == 336 == cd.stage1_response_size()
    assert w1[8 + 64 + 128 + 64:8 + 64 + 128 + 64 + 8] == (1).to_bytes(8, "little")       # ds: Vec length prefix
    b1 = cd.stage1_response_from_wire(w1)
    assert b1.subcircuit_idx == 77
    for x, y in ((b1.proof.a, proof.a), (b1.proof.b, proof.b), (b1.proof.c, proof.c), (b1.proof.ds[0], proof.ds[0])):
        assert x.tobytes() == y.tobytes()
    # MPI framing: 256-byte Packed chunks, then chunks_exact(item_size) on the gathered buffer
    assert len(to_packed(w0)) == 256 and len(to_packed(w1)) == 512 and to_packed(w1)[:336] == w1
    flat = w0 + w0 + b"\0\0\0"
    assert cd.split_flattened(flat, 104) == [w0, w0]
    # Default::default() of the reference's records: infinity points (flag 0x40 on each y)
    d = Stage1Response(0, Proof(np.zeros(64, np.uint8), np.zeros(128, np.uint8), np.zeros(64, np.uint8), [np.zeros(64, np.uint8)]))
    wd = cd.stage1_response_to_wire(d)
    assert len(wd) == 336 and wd[8 + 63] == 0x40 and wd[8 + 64 + 127] == 0x40 and sum(wd) == 0x40 * 4 + 1


def test_proving_keys_file_round_trip():
    cd = ArkCodec("bn254")
    g1 = _abi("bn254", 1, _rand_points("bn254", 1, 12, 21) + [None])
    g2 = _abi("bn254", 2, _rand_points("bn254", 2, 5, 22))
    P1 = lambda i: g1[64 * i:64 * (i + 1)].copy()
    P2 = lambda i: g2[128 * i:128 * (i + 1)].copy()

    def mk(k):
        vk = VerifyingKey(P1(0), P2(0), P2(1), P2(2), g1[64:64 * 3].copy(), g2[128 * 3:128 * 5].copy())
        ck = CommitterKey(P1(3), [g1[64 * 4:64 * 6].copy(), g1[64 * 6:64 * (9 + k)].copy()])
        return ProvingKey(vk, P1(9), g1[:64 * 5].copy(), g1[64 * 5:64 * 10].copy(), g2[:128 * 5].copy(),
                          g1[64 * 2:64 * 13].copy(), ck, g1[64 * 10:64 * 12].copy())
    pks = ProvingKeys("BigMerkle circuit", b"\x01\x02\x03", {0: mk(0), 5: mk(1)}, {0: 0, 1: 0, 2: 5, 3: 5})
    for with_id in (True, False):
        blob = pks.serialize(cd, with_id)
        back = ProvingKeys.deserialize(cd, blob, with_id)
        assert back.serialize(cd, with_id) == blob
        assert back.num_subcircuits() == 4 and back.get_pk(3).h_g.tobytes() == mk(1).h_g.tobytes()
        assert back.get_pk(2).ck.deltas_abc_g[1].size == 64 * 4
    blob = pks.serialize(cd)
    assert blob[:8] == (17).to_bytes(8, "little") and blob[8:25] == b"BigMerkle circuit"
    assert ProvingKeys.deserialize(cd, blob).get_id_str() == "BigMerkle circuit"
    with pytest.raises(KeyError):
        pks.get_pk(9)
    with pytest.raises(SerializationError):
        ProvingKeys.deserialize(cd, blob[:-3])


def test_request_records_roundtrip_and_layout():
    """Stage0Request / Stage1Request framing (coordinator.rs:195-261,520-622; ROM circuits): hand-laid bytes, sizes,
    round trips, and rejection of what ark's deserializer rejects (unreduced field element, RAM tag)."""
    from hekaton_system_amd.ark_serialize import ArkCodec, SerializationError
    from hekaton_system_amd.worker import Stage0Request, Stage1Request
    from hekaton_system_amd.cp_groth16 import CURVE_PARAMS
    codec = ArkCodec("bn254")
    r = CURVE_PARAMS["bn254"]["r"]
    req = Stage0Request(5, [(0x1122, 7), (9, r - 1)], [(9, r - 1)])
    b = codec.stage0_request_to_wire(req)
    want = (5).to_bytes(8, "little") + (2).to_bytes(8, "little") + b"\x00" + (0x1122).to_bytes(8, "little") + (7).to_bytes(32, "little") + \
        b"\x00" + (9).to_bytes(8, "little") + (r - 1).to_bytes(32, "little") + (1).to_bytes(8, "little") + b"\x00" + (9).to_bytes(8, "little") + \
        (r - 1).to_bytes(32, "little")
    assert b == want and len(b) == 8 + 8 + 2 * 41 + 8 + 41
    back = codec.stage0_request_from_wire(b)
    assert (back.subcircuit_idx, back.time_ordered_subtrace, back.addr_ordered_subtrace) == (5, req.time_ordered_subtrace, req.addr_ordered_subtrace)
    bad = bytearray(b); bad[16] = 1                                   # a RAM-tagged entry
    with pytest.raises(SerializationError):
        codec.stage0_request_from_wire(bytes(bad))
    bad = bytearray(b); bad[25:57] = r.to_bytes(32, "little")          # val = r: not reduced
    with pytest.raises(SerializationError):
        codec.stage0_request_from_wire(bytes(bad))
    s1 = Stage1Request(3, 0, time_ordered_eval=11, addr_ordered_eval=12, challenges=(13, 14), last_subtrace_entry=(4, 15),
                       leaf_sibling_hash=16, auth_path=[17, 18, 19], leaf_index=6, root=20, serialized_witnesses=b"\xaa" * 64,
                       circ_params=(32, 32, 4))
    w = codec.stage1_request_to_wire(s1)
    assert len(w) == 8 + (1 + 32 + 32 + 1 + 64) + 41 + 32 + (8 + 3 * 32) + 8 + 32 + (8 + 64) + 24
    back = codec.stage1_request_from_wire(w)
    assert back == s1
    none = Stage1Request(3, 0, challenges=None)
    assert codec.stage1_request_from_wire(codec.stage1_request_to_wire(none)).challenges is None
