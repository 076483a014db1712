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

enum { WOP_INPUT = 0, WOP_CONST, WOP_XOR, WOP_CH, WOP_AND, WOP_MAJ, WOP_ADD, WOP_PACK4 };

#if defined(__HIPCC__)

__device__ __forceinline__ u32 wp_ref(const u32* __restrict__ values, u32 ref, u32 batch, u32 lane) {
    u32 v = values[(size_t)(ref & 0xfffffu) * batch + lane];
    u32 rot = (ref >> 20) & 31u, shr = (ref >> 25) & 31u;
    if (shr) return v >> shr;
    return rot ? ((v >> rot) | (v << (32u - rot))) : v;
}

// ops: n_ops x 8 u32 (opcode, a, b, c, imm, -, -, -); inputs: [batch][n_inputs]; values: [n_values][batch]
template <int UNUSED>
__global__ void __launch_bounds__(64)
k_word_program(const u32* __restrict__ ops, u32 n_ops, const u32* __restrict__ refs, const u32* __restrict__ inputs,
               u32 n_inputs, u32 batch, u32* __restrict__ values) {
    u32 lane = blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= batch) return;
    u32 vid = 0;
    for (u32 k = 0; k < n_ops; k++) {
        const uint4 o = reinterpret_cast<const uint4*>(ops)[2 * k];
        u32 imm = ops[8 * k + 4];
        u32 r;
        switch (o.x) {
            case WOP_INPUT: r = inputs[(size_t)lane * n_inputs + imm]; break;
            case WOP_CONST: r = imm; break;
            case WOP_XOR: r = wp_ref(values, o.y, batch, lane) ^ wp_ref(values, o.z, batch, lane); break;
            case WOP_CH: {
                u32 e = wp_ref(values, o.y, batch, lane), f = wp_ref(values, o.z, batch, lane), g = wp_ref(values, o.w, batch, lane);
                r = (e & f) ^ (~e & g);
                break;
            }
            case WOP_AND: r = wp_ref(values, o.y, batch, lane) & wp_ref(values, o.z, batch, lane); break;
            case WOP_MAJ: {
                u32 x = wp_ref(values, o.y, batch, lane), y = wp_ref(values, o.z, batch, lane), z = wp_ref(values, o.w, batch, lane);
                r = (x & y) ^ (x & z) ^ (y & z);
                break;
            }
            case WOP_ADD: {
                u64 tot = imm;
                for (u32 j = 0; j < o.z; j++) tot += wp_ref(values, refs[o.y + j], batch, lane);
                values[(size_t)vid * batch + lane] = (u32)tot;
                vid++;
                r = (u32)(tot >> 32);
                break;
            }
            default: {   // WOP_PACK4
                u32 p0 = values[(size_t)(refs[o.y] & 0xfffffu) * batch + lane], p1 = values[(size_t)(refs[o.y + 1] & 0xfffffu) * batch + lane];
                u32 p2 = values[(size_t)(refs[o.y + 2] & 0xfffffu) * batch + lane], p3 = values[(size_t)(refs[o.y + 3] & 0xfffffu) * batch + lane];
                r = (p0 << 24) | (p1 << 16) | (p2 << 8) | p3 | imm;
            }
        }
        values[(size_t)vid * batch + lane] = r;
        vid++;
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

#endif  // __HIPCC__

}  // namespace hk

