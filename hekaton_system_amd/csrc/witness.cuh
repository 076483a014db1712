// witness.cuh — witness generation on the GPU for gadget circuits.
//
// Replaces what `circuit.generate_constraints(...)` does for the ASSIGNMENT in the reference's timed region
// (cp-groth16/src/prover.rs:70-75; distributed-prover/src/tree_hash_circuit.rs:313-398: ark-r1cs-std gadgets allocating
// one witness closure per variable): a proving-key class carries, next to its matrices, a WORD PROGRAM - the dataflow
// of its bit gadgets at 32-bit word granularity (XOR / Ch / Maj / modular additions of SHA-256, byte packing), recorded
// once when the class's R1CS is built (hekaton_system_amd/sha_circuit.py) - and a column map (variable -> value, bit).
//   k_word_program   one lane per subcircuit runs the class's program over its 16 .. 54 input words; every lane follows
//                    the same op stream, values live value-major ([value][subcircuit]) so loads / stores coalesce;
//   k_witness_expand one lane per variable: z[i] = bit(value, pos) ? 1 : 0 in Montgomery form, straight into the
//                    buffer hk_prove reads;
//   k_scatter_full   the ~40 full-width variables (instance, portal entries, running evaluations) from the host.
// A subcircuit's assignment never exists on the host and crosses PCIe as its inputs (64 .. 216 B) + ~1.3 KB.
#pragma once
#include "hk_internal.h"

namespace hk {

enum { WOP_INPUT = 0, WOP_CONST, WOP_XOR, WOP_CH, WOP_AND, WOP_MAJ, WOP_ADD, WOP_PACK4,
       // one SHA-256 round / one message-schedule step as ONE entry producing the 11 / 6 values its gadget entries would, in
       // their order (sha_circuit.py OP_SHA_ROUND / OP_SHA_SCHED): the chain pays decode + ring round trips once per round
       WOP_SHA_ROUND, WOP_SHA_SCHED };
constexpr unsigned WOP_ROUND_VALUES = 11, WOP_SCHED_VALUES = 6;

#if defined(__HIPCC__)

// The program is one dependent chain per lane (a SHA-256 round needs the previous round's words), and every link was a
// global store followed by a global load of the same word: ~1.2 us per op.  The last WP_RING values of a lane therefore
// also live in LDS ([slot][lane], conflict-free); a reference reaches back at most ~200 values inside a compression
// (W[t-16]), so nearly every operand comes from there.  Every value still goes to `values` for k_witness_expand.
constexpr u32 WP_RING = 256;

__device__ __forceinline__ u32 wp_word(const u32* __restrict__ values, const u32* ring, u32 id, u32 vid, u32 batch, u32 lane) {
    if (vid - id <= WP_RING) return ring[(id & (WP_RING - 1)) * 64 + threadIdx.x];
    return values[(size_t)id * batch + lane];
}
__device__ __forceinline__ u32 wp_ref(const u32* __restrict__ values, const u32* ring, u32 ref, u32 vid, u32 batch, u32 lane) {
    u32 v = wp_word(values, ring, ref & 0xfffffu, vid, batch, lane);
    u32 rot = (ref >> 20) & 31u, shr = (ref >> 25) & 31u;
    if (shr) return v >> shr;
    return rot ? ((v >> rot) | (v << (32u - rot))) : v;
}

// ops: n_ops x 8 u32 (opcode, a, b, c, imm, -, -, -); inputs: [batch][n_inputs]; values: [n_values][batch]
template <int UNUSED>
__global__ void __launch_bounds__(64)
k_word_program(const u32* __restrict__ ops, u32 n_ops, const u32* __restrict__ refs, const u32* __restrict__ inputs,
               u32 n_inputs, u32 batch, u32* __restrict__ values) {
    __shared__ u32 ring[WP_RING * 64];
    u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= batch) return;
    u32 vid = 0;
    auto put = [&](u32 r) {
        ring[(vid & (WP_RING - 1)) * 64 + threadIdx.x] = r;
        values[(size_t)vid * batch + lane] = r;
        vid++;
    };
    // the op stream is the same for every lane (scalar loads); op k + 1 is requested before op k executes
    uint4 o_next = reinterpret_cast<const uint4*>(ops)[0];
    u32 imm_next = ops[4];
    for (u32 k = 0; k < n_ops; k++) {
        const uint4 o = o_next;
        const u32 imm = imm_next;
        if (k + 1 < n_ops) {
            o_next = reinterpret_cast<const uint4*>(ops)[2 * (k + 1)];
            imm_next = ops[8 * (k + 1) + 4];
        }
        u32 r;
        switch (o.x) {
            case WOP_INPUT: r = inputs[(size_t)lane * n_inputs + imm]; break;
            case WOP_CONST: r = imm; break;
            case WOP_XOR: r = wp_ref(values, ring, o.y, vid, batch, lane) ^ wp_ref(values, ring, o.z, vid, batch, lane); break;
            case WOP_CH: {
                u32 e = wp_ref(values, ring, o.y, vid, batch, lane), f = wp_ref(values, ring, o.z, vid, batch, lane),
                    g = wp_ref(values, ring, o.w, vid, batch, lane);
                r = (e & f) ^ (~e & g);
                break;
            }
            case WOP_AND: r = wp_ref(values, ring, o.y, vid, batch, lane) & wp_ref(values, ring, o.z, vid, batch, lane); break;
            case WOP_MAJ: {
                u32 x = wp_ref(values, ring, o.y, vid, batch, lane), y = wp_ref(values, ring, o.z, vid, batch, lane),
                    z = wp_ref(values, ring, o.w, vid, batch, lane);
                r = (x & y) ^ (x & z) ^ (y & z);
                break;
            }
            case WOP_ADD: {
                u64 tot = imm;
                for (u32 j = 0; j < o.z; j++) tot += wp_ref(values, ring, refs[o.y + j], vid, batch, lane);
                put((u32)tot);
                r = (u32)(tot >> 32);
                break;
            }
            case WOP_SHA_ROUND: {
                const u32 v0 = vid;                               // operands are older than every value of this entry
                u32 in[9];
                for (u32 j = 0; j < 9; j++) in[j] = wp_ref(values, ring, refs[o.y + j], v0, batch, lane);
                const u32 a = in[0], b = in[1], c = in[2], d = in[3], e = in[4], f = in[5], g = in[6], h = in[7], w = in[8];
                auto rotr = [](u32 x, u32 k) { return (x >> k) | (x << (32u - k)); };
                u32 x = rotr(e, 6) ^ rotr(e, 11), s1 = x ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
                u32 y = rotr(a, 2) ^ rotr(a, 13), s0 = y ^ rotr(a, 22), ab = a & b, mj = (a & b) ^ (a & c) ^ (b & c);
                u64 te = (u64)imm + d + h + s1 + ch + w, ta = (u64)imm + h + s1 + ch + w + s0 + mj;
                put(x); put(s1); put(ch); put(y); put(s0); put(ab); put(mj);
                put((u32)te); put((u32)(te >> 32)); put((u32)ta);
                r = (u32)(ta >> 32);
                break;
            }
            case WOP_SHA_SCHED: {
                const u32 v0 = vid;
                u32 w15 = wp_ref(values, ring, refs[o.y], v0, batch, lane), w2 = wp_ref(values, ring, refs[o.y + 1], v0, batch, lane);
                u32 w7 = wp_ref(values, ring, refs[o.y + 2], v0, batch, lane), w16 = wp_ref(values, ring, refs[o.y + 3], v0, batch, lane);
                auto rotr = [](u32 x, u32 k) { return (x >> k) | (x << (32u - k)); };
                u32 x0 = rotr(w15, 7) ^ rotr(w15, 18), s0 = x0 ^ (w15 >> 3), x1 = rotr(w2, 17) ^ rotr(w2, 19), s1 = x1 ^ (w2 >> 10);
                u64 tot = (u64)s1 + w7 + s0 + w16;
                put(x0); put(s0); put(x1); put(s1); put((u32)tot);
                r = (u32)(tot >> 32);
                break;
            }
            default: {   // WOP_PACK4
                u32 p0 = wp_word(values, ring, refs[o.y] & 0xfffffu, vid, batch, lane), p1 = wp_word(values, ring, refs[o.y + 1] & 0xfffffu, vid, batch, lane);
                u32 p2 = wp_word(values, ring, refs[o.y + 2] & 0xfffffu, vid, batch, lane), p3 = wp_word(values, ring, refs[o.y + 3] & 0xfffffu, vid, batch, lane);
                r = (p0 << 24) | (p1 << 16) | (p2 << 8) | p3 | imm;
            }
        }
        put(r);
    }
}

// z_out: [batch][n_v] Fr.  map[i] = value id << 5 | bit, or 0xffffffff (left to k_scatter_full); column 0 = 1.
template <class Fr>
__global__ void k_witness_expand(const u32* __restrict__ map, size_t n_v, const u32* __restrict__ values, u32 batch,
                                 Fr* __restrict__ z_out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    u32 b = blockIdx.y;
    if (i >= n_v) return;
    u32 m = map[i];
    Fr* z = z_out + (size_t)b * n_v;
    if (i == 0) { fr_store(&z[0], Fr::one()); return; }
    if (m == 0xffffffffu) return;
    u32 v = values[(size_t)(m >> 5) * batch + b];
    fr_store(&z[i], ((v >> (m & 31u)) & 1u) ? Fr::one() : Fr::zero());
}

// z_out[b][cols[j]] = vals[b][j]
template <class Fr>
__global__ void k_scatter_full_batch(const u32* __restrict__ cols, const Fr* __restrict__ vals, u32 k, size_t n_v,
                                     Fr* __restrict__ z_out) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    u32 b = blockIdx.y;
    if (j >= k) return;
    u32 c = cols[j];
    if (c < n_v) fr_store(&z_out[(size_t)b * n_v + c], fr_load(&vals[(size_t)b * k + j]));
}

// ---- Poseidon membership block (execution tree, subcircuit_circuit.rs:233-252) --------------------------------------
// The host-side description of one Poseidon instance of the tree (poseidon_util.rs:53-62): width t = rate + 1 (3 or 4),
// S-box exponent 5 or 17, rf full and rp partial rounds; consts = ark[(rf + rp)][t] then mds[t][t], Montgomery.
struct PoseidonDesc { u32 t, alpha, rf, rp, off; };

// Width and S-box exponent are template parameters: the state, the round's S-box outputs and the MDS row sums stay in
// registers (with runtime widths they were runtime-indexed arrays in private memory and the block took 20 ms per step).
template <class Fr, int T, int ALPHA>
__device__ __forceinline__ void poseidon_permute_trace(const Fr* __restrict__ consts, const PoseidonDesc& d, Fr (&s)[T],
                                                       Fr*& w) {
    const Fr* ark = consts + d.off;
    const Fr* mds = ark + (size_t)(d.rf + d.rp) * T;
    const u32 half = d.rf / 2;
    Fr m[T * T];
#pragma unroll
    for (int i = 0; i < T * T; i++) m[i] = fr_load(&mds[i]);
    HK_NOUNROLL for (u32 r = 0; r < d.rf + d.rp; r++) {
        bool full = r < half || r >= half + d.rp;
        Fr y[T];
#pragma unroll
        for (int i = 0; i < T; i++) y[i] = Fr::add(s[i], fr_load(&ark[r * T + i]));
#pragma unroll
        for (int i = 0; i < T; i++) {
            if (i == 0 || full) {
                Fr u = y[i];
                Fr x = Fr::mul(u, u);                       // u^2
                fr_store(w++, x);
#pragma unroll
                for (int k = 0; k < (ALPHA == 5 ? 1 : 3); k++) { x = Fr::mul(x, x); fr_store(w++, x); }   // u^4 (u^8 u^16)
                x = Fr::mul(x, u);                          // u^5 / u^17
                fr_store(w++, x);
                y[i] = x;
            }
        }
#pragma unroll
        for (int i = 0; i < T; i++) {
            Fr acc = Fr::mul(m[i * T], y[0]);
#pragma unroll
            for (int j = 1; j < T; j++) acc = Fr::add(acc, Fr::mul(m[i * T + j], y[j]));
            s[i] = acc;
        }
#pragma unroll
        for (int i = 0; i < T; i++) fr_store(w++, s[i]);
    }
}

// One lane per subcircuit: the witnesses of its membership block, in the order sha_circuit.poseidon_path_trace lists
// them - the leaf hash (sponge of rate 3 over the 4 leaf fields: two permutations), then per level the bit, the sibling,
// the left input and the two-to-one hash - written straight into the assignment at column col0.
// leaf: [batch][4], siblings: [batch][depth], index: [batch]; z: [batch][n_v].
template <class Fr>
__global__ void __launch_bounds__(64)
k_poseidon_path(const Fr* __restrict__ consts, PoseidonDesc leaf_d, PoseidonDesc node_d, const Fr* __restrict__ leaf,
                const Fr* __restrict__ siblings, const u32* __restrict__ index, u32 depth, u32 batch, size_t n_v,
                size_t col0, Fr* __restrict__ z) {
    u32 b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    Fr* w = z + (size_t)b * n_v + col0;
    Fr s[4];
    s[0] = Fr::zero();
    for (u32 i = 0; i < 3; i++) s[1 + i] = fr_load(&leaf[(size_t)b * 4 + i]);
    poseidon_permute_trace<Fr, 4, 5>(consts, leaf_d, s, w);
    s[1] = Fr::add(s[1], fr_load(&leaf[(size_t)b * 4 + 3]));
    poseidon_permute_trace<Fr, 4, 5>(consts, leaf_d, s, w);
    Fr cur = s[1];
    u32 idx = index[b];
    HK_NOUNROLL for (u32 l = 0; l < depth; l++) {
        Fr sib = fr_load(&siblings[(size_t)b * depth + l]);
        bool bit = (idx >> l) & 1u;
        Fr left = bit ? sib : cur, right = bit ? cur : sib;
        fr_store(w++, bit ? Fr::one() : Fr::zero());
        fr_store(w++, sib);
        fr_store(w++, left);
        Fr t3[3];
        t3[0] = Fr::zero(); t3[1] = left; t3[2] = right;
        poseidon_permute_trace<Fr, 3, 17>(consts, node_d, t3, w);
        cur = t3[1];
    }
}

#endif  // __HIPCC__

}  // namespace hk

