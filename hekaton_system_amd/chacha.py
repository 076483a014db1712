"""rand_chacha 0.3.1 `ChaCha12Rng` and ark-ff 0.4 `Fr::rand`, restated (both crates are absent from
/root/reference; pinned by the published zero-key ChaCha keystreams in tests/test_ark_serialize.py).

The reference derives every commitment randomness as `Fr::rand(&mut ChaCha12Rng::from_seed(com_seed))`
(distributed-prover/src/worker.rs:129-137, cp-groth16/src/committer.rs:85) and re-derives it from the 32-byte
seed for stage 1 (mpi-snark/src/worker.rs:63-66), so a drop-in worker must reproduce the draw bit for bit."""
import struct

import numpy as np

_FR = {   # modulus, byte length (public curve parameters; same constants as cp_groth16.CURVE_PARAMS)
    "bn254": (21888242871839275222246405745257275088548364400416034343698204186575808495617, 32),
    "bls12_381": (52435875175126190479447740508185965837690552500527637822603658699938581184513, 32),
}


def _chacha_block(key_words, counter, stream, rounds):
    c = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574]
    s = c + list(key_words) + [counter & 0xFFFFFFFF, counter >> 32, stream & 0xFFFFFFFF, stream >> 32]
    x = list(s)
    M = 0xFFFFFFFF

    def qr(a, b, c_, d):
        x[a] = (x[a] + x[b]) & M; x[d] ^= x[a]; x[d] = ((x[d] << 16) | (x[d] >> 16)) & M
        x[c_] = (x[c_] + x[d]) & M; x[b] ^= x[c_]; x[b] = ((x[b] << 12) | (x[b] >> 20)) & M
        x[a] = (x[a] + x[b]) & M; x[d] ^= x[a]; x[d] = ((x[d] << 8) | (x[d] >> 24)) & M
        x[c_] = (x[c_] + x[d]) & M; x[b] ^= x[c_]; x[b] = ((x[b] << 7) | (x[b] >> 25)) & M

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & M for a, b in zip(x, s)]


class ChaChaRng:
    """rand_chacha 0.3.1 `ChaCha{8,12,20}Rng::from_seed(seed)`: original (djb) ChaCha with a 64-bit block counter
    starting at 0 and a 64-bit stream id 0; output = the keystream as little-endian u32 words, consumed through
    rand_core's `BlockRng` (a 64-word buffer: `next_u64` takes two consecutive words, low word first)."""

    def __init__(self, seed, rounds=12):
        if len(seed) != 32:
            raise ValueError("seed is [u8; 32]")
        self.key = struct.unpack("<8I", bytes(seed))
        self.rounds = rounds
        self.counter = 0
        self.buf = []
        self.index = 64                                   # empty buffer

    def _refill(self):
        self.buf = []
        for _ in range(4):                                # rand_chacha refills four blocks at a time
            self.buf += _chacha_block(self.key, self.counter, 0, self.rounds)
            self.counter = (self.counter + 1) & (2 ** 64 - 1)
        self.index = 0

    def next_u32(self):
        if self.index >= 64:
            self._refill()
        v = self.buf[self.index]
        self.index += 1
        return v

    def next_u64(self):
        i = self.index
        if i < 63:
            self.index += 2
            return self.buf[i] | (self.buf[i + 1] << 32)
        if i >= 64:
            self._refill()
            self.index = 2
            return self.buf[0] | (self.buf[1] << 32)
        lo = self.buf[63]
        self._refill()
        self.index = 1
        return lo | (self.buf[0] << 32)

    def fill_bytes(self, n):
        out = b""
        while len(out) < n:
            out += struct.pack("<I", self.next_u32())
        return out[:n]

    # the two draws the host mirror's callers make (same duck type as cp_groth16.SeededRng)
    def fr(self, r):
        """`Fr::rand(self)` as a canonical int (the drawn limbs are the Montgomery form: value = limbs / R)."""
        curve = next(c for c, (m, _) in _FR.items() if m == r)
        v = int.from_bytes(fr_rand_mont(curve, self).tobytes(), "little")
        return v * pow(1 << (8 * _FR[curve][1]), -1, r) % r

    def gen_seed(self):
        """`rng.gen::<[u8; 32]>()`: rand 0.8 draws array elements one by one, a u8 being `next_u32() as u8`."""
        return bytes(self.next_u32() & 0xFF for _ in range(32))


def ChaCha12Rng(seed):
    return ChaChaRng(seed, 12)


def fr_rand_mont(curve, rng):
    """ark-ff 0.4 `UniformRand for Fp` (Distribution<Fp> for Standard): draw N u64 limbs, clear the bits above the
    modulus' bit length, accept iff < modulus; the accepted limbs ARE the element's internal (Montgomery) form.
    Returns the 32 ABI bytes, i.e. exactly what hk_commit takes as kappa."""
    r, nb = _FR[curve]
    shave = 8 * nb - r.bit_length()
    while True:
        limbs = [rng.next_u64() for _ in range(nb // 8)]
        limbs[-1] &= (2 ** 64 - 1) >> shave
        v = sum(l << (64 * i) for i, l in enumerate(limbs))
        if v < r:
            return np.frombuffer(v.to_bytes(nb, "little"), dtype=np.uint8).copy()


def commitment_randomness(curve, com_seed):
    """kappa = Fr::rand(&mut ChaCha12Rng::from_seed(com_seed)) (worker.rs:129-137 + committer.rs:85;
    re-derived for stage 1 at mpi-snark/src/worker.rs:63-66)."""
    return fr_rand_mont(curve, ChaCha12Rng(com_seed))
