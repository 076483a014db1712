// ntt.cuh — radix-2 NTT over Fr with ark-poly's conventions, and the R1CS->QAP pointwise kernels.
//
// Replaces ark-poly `Radix2EvaluationDomain::{fft,ifft}_in_place` + `get_coset(F::GENERATOR)` and
// ark-groth16 `LibsnarkReduction::witness_map_from_matrices` as reached from
// cp-groth16/src/prover.rs:123.  Point i of the size-m domain is w_m^i,
// natural order at the API; internally every forward/backward pair is run as
//      DIF (natural -> bit-reversed)  then  DIT (bit-reversed -> natural)
// so the witness map never pays a bit-reversal pass: the coset/1/m scaling between the two reads its
// exponent from the bit-reversed index, and the H-query bases are stored bit-reversed at key upload.
//
// Each pass keeps a tile of 2^8 rows x 8 contiguous elements (64 KiB) in LDS and runs up to 8 butterfly
// stages there, so a 2^21-point transform is 3 HBM round trips (64*m bytes each).
#pragma once
#include "field.cuh"

namespace hk {

constexpr int NTT_TILE_LOG_ROWS = 8;     // stages per pass
constexpr int NTT_TILE_LOG_COLS = 3;     // 8 contiguous elements = 256 B per row
constexpr int NTT_THREADS = 256;         // 512 measured no faster: the passes are bound by field-multiply issue, not latency
constexpr int POW_TABLE_BITS = 10;       // g^j = T0[j & 1023] * T1[(j >> 10) & 1023] * T2[j >> 20]

#if defined(__HIPCC__)

template <class Fr>
__device__ __forceinline__ Fr fr_load(const Fr* p) {
    Fr r;
    const uint4* s = reinterpret_cast<const uint4*>(p);
    uint4 a = s[0], b = s[1];
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
template <class Fr>
__device__ __forceinline__ void fr_store(Fr* p, const Fr& lazy) {
    Fr r = Fr::canon(lazy);                            // memory is always canonical
    uint4* d = reinterpret_cast<uint4*>(p);
    d[0] = make_uint4(r.v[0], r.v[1], r.v[2], r.v[3]);
    d[1] = make_uint4(r.v[4], r.v[5], r.v[6], r.v[7]);
}

// tw[i] = w^i for i < count, from the table sq[k] = w^(2^k)
template <class Fr>
__global__ void k_pow_table(Fr* __restrict__ tw, const Fr* __restrict__ sq, u32 count, u32 nbits) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fr acc = Fr::one();
    for (u32 k = 0; k < nbits; k++)
        if ((i >> k) & 1) acc = Fr::mul(acc, fr_load(&sq[k]));
    fr_store(&tw[i], acc);
}

// One pass of `nst` butterfly stages [lo, lo+nst) on a transform of size 2^logn, in place.
//   DIF (dit == 0): stages run from high to low, butterfly (u, v) -> (u + v, (u - v) * w)
//   DIT (dit == 1): stages run from low to high, butterfly (u, v) -> (u + v*w, u - v*w)
// tw: table of w_M^i, i < M/2, M = 2^log_table.  Batched over blockIdx.y (vectors `stride_vec` apart).
// Optional fused epilogue (post != 0): every element is multiplied by `scale` and, when post == 2, also by
// g^bitrev(index) from the 3x1024 power tables `pw` before it is stored — the "/m and coset shift"
// step that follows a DIF chain in the witness map, so it costs no extra HBM round trip.
template <class Fr>
__global__ void __launch_bounds__(NTT_THREADS)
k_ntt_pass(Fr* __restrict__ data, size_t stride_vec, const Fr* __restrict__ tw, u32 logn, u32 log_table,
           u32 lo, u32 nst, int dit, int post, Fr scale, const Fr* __restrict__ pw) {
    extern __shared__ uint4 lds_raw[];
    Fr* lds = reinterpret_cast<Fr*>(lds_raw);
    Fr* vec = data + (size_t)blockIdx.y * stride_vec;
    u32 cols_bits = lo < (u32)NTT_TILE_LOG_COLS ? lo : (u32)NTT_TILE_LOG_COLS;
    u32 rows = 1u << nst, cols = 1u << cols_bits;
    u32 tile_elems = rows << cols_bits;
    u32 mid_bits = lo - cols_bits;
    u32 t = blockIdx.x;
    u32 mid = t & ((1u << mid_bits) - 1u);
    u32 high = t >> mid_bits;
    size_t base = ((size_t)high << (lo + nst)) | ((size_t)mid << cols_bits);
    // load tile: element (r, c) lives at base | r << lo | c ; LDS index r * cols + c
    for (u32 e = threadIdx.x; e < tile_elems; e += NTT_THREADS) {
        u32 r = e >> cols_bits, c = e & (cols - 1);
        lds[e] = fr_load(&vec[base | ((size_t)r << lo) | c]);
    }
    __syncthreads();
    u32 half_count = tile_elems >> 1;
    for (u32 st = 0; st < nst; st++) {
        u32 ls = dit ? st : (nst - 1 - st);          // local stage (bit of r)
        u32 s = lo + ls;                              // global stage
        for (u32 bidx = threadIdx.x; bidx < half_count; bidx += NTT_THREADS) {
            // butterfly index -> (r without bit ls, c)
            u32 c = bidx & (cols - 1);
            u32 rr = bidx >> cols_bits;
            u32 r0 = ((rr >> ls) << (ls + 1)) | (rr & ((1u << ls) - 1u));
            u32 r1 = r0 | (1u << ls);
            u32 i0 = (r0 << cols_bits) | c, i1 = (r1 << cols_bits) | c;
            size_t g0 = base | ((size_t)r0 << lo) | c;           // global index of the upper element
            u32 j = (u32)(g0 & (((size_t)1 << s) - 1));
            Fr w = fr_load(&tw[(size_t)j << (log_table - s - 1)]);   // issued first: longest latency
            Fr u = lds[i0], v = lds[i1];
            if (dit) {
                v = Fr::mul(v, w);
                lds[i0] = Fr::add(u, v);
                lds[i1] = Fr::sub(u, v);
            } else {
                lds[i0] = Fr::add(u, v);
                lds[i1] = Fr::mul(Fr::sub(u, v), w);
            }
        }
        __syncthreads();
    }
    for (u32 e = threadIdx.x; e < tile_elems; e += NTT_THREADS) {
        u32 r = e >> cols_bits, c = e & (cols - 1);
        size_t gi = base | ((size_t)r << lo) | c;
        Fr x = lds[e];
        if (post) {
            x = Fr::mul(x, scale);
            if (post == 2) {
                u32 j = logn ? (__brev((u32)gi) >> (32 - logn)) : 0u;
                Fr g = fr_load(&pw[j & 1023]);
                if (logn > 10) g = Fr::mul(g, fr_load(&pw[1024 + ((j >> 10) & 1023)]));
                if (logn > 20) g = Fr::mul(g, fr_load(&pw[2048 + (j >> 20)]));
                x = Fr::mul(x, g);
            }
        }
        fr_store(&vec[gi], x);
    }
}

// x[pos] *= scale * g^(idx)  with idx = pos (natural) or bitrev(pos) (after a DIF pass chain).
// pw: three 1024-entry tables of g^(j), g^(1024 j), g^(2^20 j); batched over blockIdx.y.
template <class Fr>
__global__ void k_scale_pow(Fr* __restrict__ data, size_t stride_vec, const Fr* __restrict__ pw, Fr scale,
                            u32 logn, int bitrev_index, int use_pow) {
    size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >> logn) return;
    Fr* vec = data + (size_t)blockIdx.y * stride_vec;
    Fr x = fr_load(&vec[pos]);
    x = Fr::mul(x, scale);
    if (use_pow) {
        u32 j = bitrev_index ? (logn ? (__brev((u32)pos) >> (32 - logn)) : 0u) : (u32)pos;
        Fr g = fr_load(&pw[j & 1023]);
        if (logn > 10) g = Fr::mul(g, fr_load(&pw[1024 + ((j >> 10) & 1023)]));
        if (logn > 20) g = Fr::mul(g, fr_load(&pw[2048 + (j >> 20)]));
        x = Fr::mul(x, g);
    }
    fr_store(&vec[pos], x);
}

// in-place bit-reversal permutation (only the standalone hk_ntt / hk_witness_map need it)
template <class Fr>
__global__ void k_bitrev(Fr* __restrict__ data, u32 logn) {
    size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pos >> logn) return;
    u32 r = logn ? (__brev((u32)pos) >> (32 - logn)) : 0u;
    if (r > pos) {
        Fr a = fr_load(&data[pos]), b = fr_load(&data[r]);
        fr_store(&data[pos], b);
        fr_store(&data[r], a);
    }
}

// ---- R1CS -> QAP --------------------------------------------------------------------------------
// out[row] = <M_row, z>   (ark-groth16 `evaluate_constraint`), one lane per row
template <class Fr>
__global__ void k_spmv(const u64* __restrict__ row_ptr, const u32* __restrict__ col,
                       const Fr* __restrict__ val, const Fr* __restrict__ z, Fr* __restrict__ out,
                       u32 n_rows) {
    u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_rows) return;
    u64 b = row_ptr[row], e = row_ptr[row + 1];
    Fr acc = Fr::zero();
    Fr one = Fr::one();
    for (u64 k = b; k < e; k++) {
        Fr c = fr_load(&val[k]);
        Fr x = fr_load(&z[col[k]]);
        if (!(c == one)) x = Fr::mul(x, c);
        acc = Fr::add(acc, x);
    }
    fr_store(&out[row], acc);
}

// a[n_c + j] = z[j] for j < n_inst  (witness_map_from_matrices: "a[start..end] = full_assignment[..num_inputs]")
template <class Fr>
__global__ void k_copy_inputs(Fr* __restrict__ a, const Fr* __restrict__ z, u32 n_c, u32 n_inst) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_inst) fr_store(&a[n_c + j], fr_load(&z[j]));
}

// ab[i] = (a[i]*b[i] - c[i]) * zinv
template <class Fr>
__global__ void k_qap_combine(Fr* __restrict__ a, const Fr* __restrict__ b, const Fr* __restrict__ c,
                              Fr zinv, size_t m) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    Fr x = Fr::mul(fr_load(&a[i]), fr_load(&b[i]));
    x = Fr::sub(x, fr_load(&c[i]));
    fr_store(&a[i], Fr::mul(x, zinv));
}

#endif  // __HIPCC__

}  // namespace hk
