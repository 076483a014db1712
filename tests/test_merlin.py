"""merlin transcript restatement (hekaton_system_amd/merlin.py; the reference's `ProtoTranscript`,
distributed-prover/src/util.rs:22,41-75).  merlin is a third-party crate absent from /root/reference; pinned by
  * the Keccak-f[1600] permutation reproducing hashlib's SHA3-256 / SHAKE-128,
  * merlin 3.0's own known-answer test (`equivalence_simple`: protocol "test protocol", message "some label" / "some data",
    32 challenge bytes under "challenge"),
  * determinism / domain-separation properties."""
import hashlib

from hekaton_system_amd.cp_groth16 import CURVE_PARAMS
from hekaton_system_amd.merlin import Transcript, keccak_f1600


def _sponge(msg, rate, pad, n_out):
    st = bytearray(200)
    m = bytearray(msg) + bytes([pad])
    m += b"\x00" * ((-len(m)) % rate)
    m[-1] |= 0x80
    for i in range(0, len(m), rate):
        for j in range(rate):
            st[j] ^= m[i + j]
        keccak_f1600(st)
    out = b""
    while len(out) < n_out:
        out += bytes(st[:rate])
        keccak_f1600(st)
    return out[:n_out]


def test_keccak_permutation_reproduces_sha3_and_shake():
    for msg in (b"", b"abc", bytes(range(200)), b"x" * 135, b"y" * 136, b"z" * 137):
        assert _sponge(msg, 136, 0x06, 32) == hashlib.sha3_256(msg).digest()
        assert _sponge(msg, 168, 0x1F, 400) == hashlib.shake_128(msg).digest(400)


def test_merlin_known_answer():
    t = Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_transcript_is_deterministic_and_domain_separated():
    def run(proto, label, data, clabel, n=32):
        t = Transcript(proto)
        t.append_message(label, data)
        return t.challenge_bytes(clabel, n)
    base = run(b"p", b"l", b"d" * 500, b"c")
    assert base == run(b"p", b"l", b"d" * 500, b"c")
    assert len({base, run(b"q", b"l", b"d" * 500, b"c"), run(b"p", b"m", b"d" * 500, b"c"),
                run(b"p", b"l", b"d" * 499 + b"e", b"c"), run(b"p", b"l", b"d" * 500, b"k")}) == 5
    # message framing: (label, data) boundaries matter
    assert run(b"p", b"ab", b"c", b"x") != run(b"p", b"a", b"bc", b"x")
    # long squeezes cross the rate boundary consistently
    t1, t2 = Transcript(b"p"), Transcript(b"p")
    a = t1.challenge_bytes(b"c", 400)
    assert len(a) == 400 and a != t2.challenge_bytes(b"c", 399) + b"\x00"
    # successive challenges differ and depend on everything absorbed before
    t = Transcript(b"p")
    c1 = t.challenge_bytes(b"c", 32)
    c2 = t.challenge_bytes(b"c", 32)
    assert c1 != c2


def test_challenge_scalar_is_a_reduced_field_element():
    for curve in ("bn254", "bls12_381"):
        r = CURVE_PARAMS[curve]["r"]
        t = Transcript(b"test-e2e")                               # the label of coordinator.rs:411
        t.append_serializable(b"AB-commitment", b"\x01" * 1152)
        x = t.challenge_scalar(b"r-random-fiatshamir", r)
        y = t.challenge_scalar(b"s-random-fiatshamir", r)
        assert 0 <= x < r and 0 <= y < r and x != y


def test_native_permutation_equals_the_python_one():
    """libhekaton's host-side hk_keccak_f1600 (what merlin.py uses when the library is built) and the plain-Python
    permutation are the same function; the transcript KAT above runs on whichever is active."""
    import random
    from hekaton_system_amd import merlin
    if not merlin._native():
        import pytest
        pytest.skip("libhekaton.so not built")
    rnd = random.Random(1)
    for _ in range(20):
        st = bytearray(rnd.getrandbits(8) for _ in range(200))
        a, b = bytearray(st), bytearray(st)
        merlin.keccak_f1600(a)
        merlin.keccak_f1600_py(b)
        assert a == b
