// ark_serialize.hpp — C++ host-side mirror of the wire formats either side of the hot path:
// ark-serialize 0.4 (uncompressed / compressed short-Weierstrass encoding as ark-bn254 uses it) for
//   Proof, Stage0Response, Stage1Response        cp-groth16/src/data_structures.rs:6-16, distributed-prover/src/worker.rs:20-52
//   Packed (256-byte MPI framing)                mpi-snark/src/lib.rs:68-111
// and the commitment randomness the worker re-derives from com_seed:
//   Fr::rand(&mut ChaCha12Rng::from_seed(com_seed))   distributed-prover/src/worker.rs:129-137, cp-groth16/src/committer.rs:85
// Same restatement as hekaton_system_amd/ark_serialize.py / chacha.py (the crates are third-party and absent from
// the reference tree; PARITY UNPINNED by reference bytes — the two mirrors are checked against each other and against
// the published ChaCha keystream and generator encodings).  Points cross as the ABI's packed-affine Montgomery bytes;
// the Montgomery <-> canonical conversion is hk_field_convert (device).  ArkCodecBn254 is ark's default short-Weierstrass
// encoding; ArkCodecBls381 the zcash format ark-bls12-381 overrides it with (big-endian, Fp2 = c1 || c0, flags in the
// three top bits of the FIRST byte; uncompressed records only - what the worker protocol carries).
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>

#include "cp_groth16.hpp"

namespace hekaton {

struct SerializationError : std::runtime_error {
    explicit SerializationError(const std::string& w) : std::runtime_error(w) {}
};

// ---- short-Weierstrass flags (ark-ec SWFlags): bit 7 = y > -y, bit 6 = infinity, on the LAST byte -------------
class ArkCodecBn254 {
public:
    explicit ArkCodecBn254(const Context& ctx) : ctx_(ctx) {
        if (ctx.sizes().fq != 32) throw SerializationError("ArkCodecBn254 needs a BN254 context");
        // (q - 1) / 2, little-endian
        static const uint8_t q[32] = {0x47, 0xfd, 0x7c, 0xd8, 0x16, 0x8c, 0x20, 0x3c, 0x8d, 0xca, 0x71, 0x68, 0x91, 0x6a, 0x81, 0x97,
                                      0x5d, 0x58, 0x81, 0x81, 0xb6, 0x45, 0x50, 0xb8, 0x29, 0xa0, 0x31, 0xe1, 0x72, 0x4e, 0x64, 0x30};
        unsigned carry = 0;
        for (int i = 31; i >= 0; i--) {                       // (q - 1) >> 1 == q >> 1 (q is odd)
            unsigned v = q[i] | (carry << 8);
            half_[i] = (uint8_t)(v >> 1);
            carry = v & 1;
        }
    }

    // group: 1 = G1 (x, y), 2 = G2 (x.c0, x.c1, y.c0, y.c1); abi = n packed Montgomery points
    Bytes points_to_wire(int group, const Bytes& abi, bool compress = false) const {
        const size_t coords = group == 1 ? 2 : 4, pb = coords * 32, n = abi.size() / pb;
        Bytes canon(abi.size());
        if (n) check(hk_field_convert(ctx_.raw(), 1, abi.data(), canon.data(), n * coords, 0), "hk_field_convert");
        const size_t out_sz = compress ? pb / 2 : pb;
        Bytes out(n * out_sz);
        for (size_t i = 0; i < n; i++) {
            const uint8_t* src = &abi[i * pb];
            const uint8_t* c = &canon[i * pb];
            bool inf = std::all_of(src, src + pb, [](uint8_t b) { return b == 0; });
            uint8_t* o = &out[i * out_sz];
            memcpy(o, c, out_sz);
            uint8_t flag = 0;
            if (inf) flag = 0x40;
            else {
                const uint8_t* y = c + pb / 2;                               // y (G1) or y.c0 (G2)
                bool neg;
                if (group == 1) neg = gt_half(y);
                else {
                    const uint8_t* y1 = y + 32;                              // Fp2 order compares c1 first
                    bool c1_zero = std::all_of(y1, y1 + 32, [](uint8_t b) { return b == 0; });
                    neg = c1_zero ? gt_half(y) : gt_half(y1);
                }
                if (neg) flag = 0x80;
            }
            o[out_sz - 1] |= flag;
        }
        return out;
    }

    // uncompressed only (what the reference reads: deserialize_uncompressed_unchecked) -> ABI bytes
    Bytes points_from_wire(int group, const uint8_t* buf, size_t n) const {
        const size_t coords = group == 1 ? 2 : 4, pb = coords * 32;
        Bytes canon(buf, buf + n * pb);
        std::vector<bool> inf(n);
        for (size_t i = 0; i < n; i++) {
            uint8_t& last = canon[i * pb + pb - 1];
            uint8_t fl = last & 0xC0;
            if (fl == 0xC0) throw SerializationError("UnexpectedFlags");
            last &= 0x3F;
            inf[i] = fl == 0x40;
            if (inf[i] && !std::all_of(&canon[i * pb], &canon[i * pb] + pb, [](uint8_t b) { return b == 0; }))
                throw SerializationError("InvalidData: infinity with non-zero coordinates");
        }
        Bytes abi(n * pb);
        if (n) check(hk_field_convert(ctx_.raw(), 1, canon.data(), abi.data(), n * coords, 1), "hk_field_convert");
        for (size_t i = 0; i < n; i++) if (inf[i]) memset(&abi[i * pb], 0, pb);
        return abi;
    }

    // ---- records ----
    static void put_u64(Bytes& w, uint64_t v) { for (int i = 0; i < 8; i++) w.push_back((uint8_t)(v >> (8 * i))); }
    static uint64_t get_u64(const uint8_t* p) { uint64_t v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | p[i]; return v; }
    void put(Bytes& w, const Bytes& b) const { w.insert(w.end(), b.begin(), b.end()); }

    Bytes proof_to_wire(const Proof& p) const {                                 // data_structures.rs:6-16
        Bytes w;
        put(w, points_to_wire(1, p.a)); put(w, points_to_wire(2, p.b)); put(w, points_to_wire(1, p.c));
        put_u64(w, p.ds.size());                                                // Vec<G1Affine>: u64 length prefix
        for (auto& d : p.ds) put(w, points_to_wire(1, d));
        return w;
    }
    Bytes stage0_response_to_wire(const Stage0Response& r) const {              // worker.rs:20-25, 104 bytes
        Bytes w;
        put_u64(w, r.subcircuit_idx);
        put(w, points_to_wire(1, r.com));
        w.insert(w.end(), r.com_seed.begin(), r.com_seed.end());
        return w;
    }
    Bytes stage1_response_to_wire(const Stage1Response& r) const {              // worker.rs:49-52, 336 bytes with one D
        Bytes w;
        put_u64(w, r.subcircuit_idx);
        put(w, proof_to_wire(r.proof));
        return w;
    }
    Stage0Response stage0_response_from_wire(const Bytes& b) const {
        if (b.size() != 8 + 64 + 32) throw SerializationError("IoError: Stage0Response is 104 bytes");
        Stage0Response r;
        r.subcircuit_idx = get_u64(b.data());
        r.com = points_from_wire(1, b.data() + 8, 1);
        memcpy(r.com_seed.data(), b.data() + 72, 32);
        return r;
    }
    Stage1Response stage1_response_from_wire(const Bytes& b) const {
        if (b.size() < 8 + 64 + 128 + 64 + 8) throw SerializationError("IoError: unexpected end of input");
        Stage1Response r;
        const uint8_t* p = b.data();
        r.subcircuit_idx = get_u64(p); p += 8;
        r.proof.a = points_from_wire(1, p, 1); p += 64;
        r.proof.b = points_from_wire(2, p, 1); p += 128;
        r.proof.c = points_from_wire(1, p, 1); p += 64;
        uint64_t nd = get_u64(p); p += 8;
        if (b.size() != 8 + 64 + 128 + 64 + 8 + nd * 64) throw SerializationError("IoError: wrong length");
        for (uint64_t i = 0; i < nd; i++, p += 64) r.proof.ds.push_back(points_from_wire(1, p, 1));
        return r;
    }
    // serialize_to_packed_vec (mpi-snark/src/lib.rs:74-79): zero-pad to whole 256-byte Packed chunks
    static Bytes to_packed(Bytes b) { b.resize((b.size() + 255) / 256 * 256, 0); return b; }

private:
    bool gt_half(const uint8_t* le) const {                                     // le > (q-1)/2, both little-endian
        for (int i = 31; i >= 0; i--) if (le[i] != half_[i]) return le[i] > half_[i];
        return false;
    }
    const Context& ctx_;
    uint8_t half_[32];
};

// ---- rand_chacha 0.3.1 ChaCha12Rng::from_seed + rand_core BlockRng word discipline ---------------------------------
class ChaCha12Rng {
public:
    explicit ChaCha12Rng(const std::array<uint8_t, 32>& seed) {
        for (int i = 0; i < 8; i++) key_[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) | ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    }
    uint32_t next_u32() { if (index_ >= 64) refill(); return buf_[index_++]; }
    uint64_t next_u64() {
        if (index_ < 63) { uint64_t v = buf_[index_] | ((uint64_t)buf_[index_ + 1] << 32); index_ += 2; return v; }
        if (index_ >= 64) { refill(); index_ = 2; return buf_[0] | ((uint64_t)buf_[1] << 32); }
        uint64_t lo = buf_[63];
        refill();
        index_ = 1;
        return lo | ((uint64_t)buf_[0] << 32);
    }
private:
    static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
    void block(uint64_t counter, uint32_t* out) const {
        uint32_t s[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
        for (int i = 0; i < 8; i++) s[4 + i] = key_[i];
        s[12] = (uint32_t)counter; s[13] = (uint32_t)(counter >> 32); s[14] = 0; s[15] = 0;     // 64-bit counter, stream 0
        uint32_t x[16];
        memcpy(x, s, sizeof x);
        auto qr = [&](int a, int b, int c, int d) {
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16); x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 12);
            x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);  x[c] += x[d]; x[b] = rotl(x[b] ^ x[c], 7);
        };
        for (int r = 0; r < 6; r++) {                                            // 12 rounds
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
        }
        for (int i = 0; i < 16; i++) out[i] = x[i] + s[i];
    }
    void refill() { for (int b = 0; b < 4; b++) block(counter_++, buf_ + 16 * b); index_ = 0; }
    uint32_t key_[8];
    uint64_t counter_ = 0;
    uint32_t buf_[64];
    int index_ = 64;
};

// ark-ff 0.4 `UniformRand for Fp` on BN254 Fr: four u64 limbs, the two bits above the 254-bit modulus cleared, accepted
// iff < r; the accepted limbs ARE the Montgomery form, i.e. the 32 bytes hk_commit takes as kappa.
inline Bytes fr_rand_mont_bn254(ChaCha12Rng& rng) {
    static const uint64_t r[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    for (;;) {
        uint64_t l[4];
        for (int i = 0; i < 4; i++) l[i] = rng.next_u64();
        l[3] &= ~0ull >> 2;
        bool lt = false;
        for (int i = 3; i >= 0; i--) if (l[i] != r[i]) { lt = l[i] < r[i]; break; }
        if (!lt) continue;
        Bytes out(32);
        for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(l[i] >> (8 * b));
        return out;
    }
}
inline Bytes commitment_randomness_bn254(const std::array<uint8_t, 32>& com_seed) {
    ChaCha12Rng rng(com_seed);
    return fr_rand_mont_bn254(rng);
}

// ---- BLS12-381: the zcash point format (ark-bls12-381 `curves/g1.rs`, `g2.rs` serialization overrides) ------------------
class ArkCodecBls381 {
public:
    static constexpr size_t FQ = 48;
    explicit ArkCodecBls381(const Context& ctx) : ctx_(ctx) {
        if (ctx.sizes().fq != FQ) throw SerializationError("ArkCodecBls381 needs a BLS12-381 context");
    }
    // group: 1 = G1 (x, y), 2 = G2 (x.c0, x.c1, y.c0, y.c1 in the ABI; c1 || c0 on the wire); uncompressed
    Bytes points_to_wire(int group, const Bytes& abi) const {
        const size_t coords = group == 1 ? 2 : 4, pb = coords * FQ, n = abi.size() / pb;
        Bytes canon(abi.size());
        if (n) check(hk_field_convert(ctx_.raw(), 1, abi.data(), canon.data(), n * coords, 0), "hk_field_convert");
        Bytes out(n * pb);
        for (size_t i = 0; i < n; i++) {
            const uint8_t* src = &abi[i * pb];
            bool inf = std::all_of(src, src + pb, [](uint8_t b) { return b == 0; });
            uint8_t* o = &out[i * pb];
            for (size_t c = 0; c < coords; c++) {
                size_t from = group == 1 ? c : (c ^ 1);                         // Fp2: c1 first
                const uint8_t* le = &canon[i * pb + from * FQ];
                for (size_t b = 0; b < FQ; b++) o[c * FQ + b] = le[FQ - 1 - b]; // big-endian
            }
            if (inf) o[0] |= 0x40;
        }
        return out;
    }
    Bytes points_from_wire(int group, const uint8_t* buf, size_t n) const {
        const size_t coords = group == 1 ? 2 : 4, pb = coords * FQ;
        Bytes canon(n * pb);
        std::vector<bool> inf(n);
        for (size_t i = 0; i < n; i++) {
            Bytes be(buf + i * pb, buf + (i + 1) * pb);
            uint8_t fl = be[0] & 0xE0;
            if (fl & 0x80) throw SerializationError("UnexpectedFlags: compression bit");
            if (fl & 0x20) throw SerializationError("UnexpectedFlags: sort bit on an uncompressed point");
            be[0] &= 0x1F;
            inf[i] = (fl & 0x40) != 0;
            if (inf[i] && !std::all_of(be.begin(), be.end(), [](uint8_t b) { return b == 0; }))
                throw SerializationError("InvalidData: infinity with non-zero coordinates");
            for (size_t c = 0; c < coords; c++) {
                size_t to = group == 1 ? c : (c ^ 1);
                for (size_t b = 0; b < FQ; b++) canon[i * pb + to * FQ + b] = be[c * FQ + FQ - 1 - b];
            }
        }
        Bytes abi(n * pb);
        if (n) check(hk_field_convert(ctx_.raw(), 1, canon.data(), abi.data(), n * coords, 1), "hk_field_convert");
        for (size_t i = 0; i < n; i++) if (inf[i]) memset(&abi[i * pb], 0, pb);
        return abi;
    }
    static void put_u64(Bytes& w, uint64_t v) { for (int i = 0; i < 8; i++) w.push_back((uint8_t)(v >> (8 * i))); }
    static uint64_t get_u64(const uint8_t* p) { uint64_t v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | p[i]; return v; }
    void put(Bytes& w, const Bytes& b) const { w.insert(w.end(), b.begin(), b.end()); }
    Bytes proof_to_wire(const Proof& p) const {
        Bytes w;
        put(w, points_to_wire(1, p.a)); put(w, points_to_wire(2, p.b)); put(w, points_to_wire(1, p.c));
        put_u64(w, p.ds.size());
        for (auto& d : p.ds) put(w, points_to_wire(1, d));
        return w;
    }
    Bytes stage0_response_to_wire(const Stage0Response& r) const {              // 8 + 96 + 32 = 136 bytes
        Bytes w;
        put_u64(w, r.subcircuit_idx);
        put(w, points_to_wire(1, r.com));
        w.insert(w.end(), r.com_seed.begin(), r.com_seed.end());
        return w;
    }
    Bytes stage1_response_to_wire(const Stage1Response& r) const {              // 8 + 96 + 192 + 96 + 8 + 96 = 496 bytes
        Bytes w;
        put_u64(w, r.subcircuit_idx);
        put(w, proof_to_wire(r.proof));
        return w;
    }
    Stage0Response stage0_response_from_wire(const Bytes& b) const {
        if (b.size() != 8 + 96 + 32) throw SerializationError("IoError: Stage0Response is 136 bytes");
        Stage0Response r;
        r.subcircuit_idx = get_u64(b.data());
        r.com = points_from_wire(1, b.data() + 8, 1);
        memcpy(r.com_seed.data(), b.data() + 104, 32);
        return r;
    }
    Stage1Response stage1_response_from_wire(const Bytes& b) const {
        if (b.size() < 8 + 96 + 192 + 96 + 8) throw SerializationError("IoError: unexpected end of input");
        Stage1Response r;
        const uint8_t* p = b.data();
        r.subcircuit_idx = get_u64(p); p += 8;
        r.proof.a = points_from_wire(1, p, 1); p += 96;
        r.proof.b = points_from_wire(2, p, 1); p += 192;
        r.proof.c = points_from_wire(1, p, 1); p += 96;
        uint64_t nd = get_u64(p); p += 8;
        if (b.size() != 8 + 96 + 192 + 96 + 8 + nd * 96) throw SerializationError("IoError: wrong length");
        for (uint64_t i = 0; i < nd; i++, p += 96) r.proof.ds.push_back(points_from_wire(1, p, 1));
        return r;
    }

private:
    const Context& ctx_;
};

// `Fr::rand` on BLS12-381 Fr: four u64 limbs, the ONE bit above the 255-bit modulus cleared, accepted iff < r
inline Bytes fr_rand_mont_bls12_381(ChaCha12Rng& rng) {
    static const uint64_t r[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
    for (;;) {
        uint64_t l[4];
        for (int i = 0; i < 4; i++) l[i] = rng.next_u64();
        l[3] &= ~0ull >> 1;
        bool lt = false;
        for (int i = 3; i >= 0; i--) if (l[i] != r[i]) { lt = l[i] < r[i]; break; }
        if (!lt) continue;
        Bytes out(32);
        for (int i = 0; i < 4; i++) for (int b = 0; b < 8; b++) out[8 * i + b] = (uint8_t)(l[i] >> (8 * b));
        return out;
    }
}
inline Bytes commitment_randomness_bls12_381(const std::array<uint8_t, 32>& com_seed) {
    ChaCha12Rng rng(com_seed);
    return fr_rand_mont_bls12_381(rng);
}

}  // namespace hekaton
