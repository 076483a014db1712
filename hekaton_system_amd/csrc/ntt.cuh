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
// Each pass keeps a tile of 2^11 elements (64 KiB) in LDS — 2^nst rows x 2^cols_bits contiguous elements —
// and runs up to 11 butterfly stages there, two stages at a time in registers (radix-4 steps), so a
// 2^21-point transform is 2 HBM round trips (64*m bytes each).
#pragma once
#include "field.cuh"

namespace hk {

constexpr int NTT_TILE_LOG = 11;         // elements per LDS tile (2^11 x 32 B = 64 KiB, two tiles per CU)
constexpr int NTT_THREADS = 512;         // one radix-4 step of a full tile = one quad per thread
constexpr int POW_TABLE_BITS = 11;       // g^j = T0[j & 2047] * T1[(j >> 11) & 2047] * T2[j >> 22]
constexpr int POW_TABLE_SIZE = 1 << POW_TABLE_BITS;

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

// g^j from the three-level power tables: T0[j & 2047] * T1[(j >> 11) & 2047] * T2[j >> 22]
template <class Fr>
__device__ __forceinline__ Fr pow_from_tables(const Fr* __restrict__ pw, u32 j, u32 logn) {
    Fr g = fr_load(&pw[j & (POW_TABLE_SIZE - 1)]);
    if (logn > POW_TABLE_BITS) g = Fr::mul(g, fr_load(&pw[POW_TABLE_SIZE + ((j >> POW_TABLE_BITS) & (POW_TABLE_SIZE - 1))]));
    if (logn > 2 * POW_TABLE_BITS) g = Fr::mul(g, fr_load(&pw[2 * POW_TABLE_SIZE + (j >> (2 * POW_TABLE_BITS))]));
    return g;
}

// ---- the pass kernel ----------------------------------------------------------------------------------
// Per-stage twiddle tables: stage s (butterfly span 2^s) owns the 2^s contiguous entries
//   tws[(2^s - 1) + j] = w_{2^(s+1)}^j,  j < 2^s        (gathered from the w_M^i table)
// so the loads of neighbouring butterflies coalesce at every stage and the low stages stay cache-resident.
template <class Fr>
__global__ void k_stage_tables(Fr* __restrict__ tws, const Fr* __restrict__ tw, u32 log_table) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((i + 1) >> log_table) return;
    u32 s = 31 - __clz((u32)(i + 1));
    size_t j = i + 1 - ((size_t)1 << s);
    fr_store(&tws[i], fr_load(&tw[j << (log_table - s - 1)]));
}

// One pass: butterfly stages [lo, lo+nst) of a 2^logn transform, in place, on an LDS tile of 2^nst rows x
// 2^cols_bits contiguous elements (cols_bits <= lo); batched over blockIdx.y (vectors `stride_vec` apart).
//   DIF (DIT == 0): stages run from high to low, butterfly (u, v) -> (u + v, (u - v) * w)
//   DIT (DIT == 1): stages run from low to high, butterfly (u, v) -> (u + v*w, u - v*w)
// Every thread carries FOUR elements through TWO stages in registers (a radix-4 step): half the LDS round
// trips and barriers of a stage-by-stage loop, two independent butterflies in flight per lane, and the three
// twiddles of a step are loaded before the data (their addresses do not depend on it).  An odd stage count
// ends with one radix-2 step.  Measured on 3 x 2^21 (tools/ntt_bench.hip): 78 G butterflies/s against 66 for
// the stage-by-stage kernel it replaced; 103 G/s once the HBM phases are hidden (field-multiply peak 136 G/s).
// Fused epilogue of the LAST pass of a DIF chain (the coset shift / quotient step of the witness map, no
// extra HBM round trip), applied to the vectors blockIdx.y < npost, in this order:
//   post & 2 -> x *= g^bitrev(index)   (power tables `pw`, POW_TABLE_SIZE entries per level)
//   post & 4 -> x -= sub[index] * kc   (`sub`: another vector in the same bit-reversed order)
//   post & 1 -> x *= scale
template <class Fr, int DIT>
__global__ void __launch_bounds__(NTT_THREADS)
k_ntt_pass4(Fr* __restrict__ data, size_t stride_vec, const Fr* __restrict__ tws, u32 logn, u32 lo, u32 nst,
            u32 cols_bits, int post, u32 npost, Fr scale, const Fr* __restrict__ pw, const Fr* __restrict__ sub,
            Fr kc) {
    extern __shared__ uint4 lds_raw[];
    Fr* lds = reinterpret_cast<Fr*>(lds_raw);
    const u32 nthr = blockDim.x;
    if (blockIdx.y >= npost) post = 0;
    Fr* vec = data + (size_t)blockIdx.y * stride_vec;
    const u32 cols = 1u << cols_bits;
    const u32 tile_elems = 1u << (nst + cols_bits);
    const u32 mid_bits = lo - cols_bits;
    const u32 mid = blockIdx.x & ((1u << mid_bits) - 1u);
    const size_t base = ((size_t)(blockIdx.x >> mid_bits) << (lo + nst)) | ((size_t)mid << cols_bits);
    for (u32 e = threadIdx.x; e < tile_elems; e += nthr) {
        u32 r = e >> cols_bits, c = e & (cols - 1);
        lds[e] = fr_load(&vec[base | ((size_t)r << lo) | c]);
    }
    __syncthreads();
    const u32 npair = nst >> 1;
    for (u32 k = 0; k < npair; k++) {
        const u32 l = DIT ? 2 * k : nst - 2 - 2 * k;          // stages lo+l and lo+l+1
        const u32 s = lo + l;
        const Fr* t_lo = tws + (((size_t)1 << s) - 1);         // stage s
        const Fr* t_hi = tws + (((size_t)2 << s) - 1);         // stage s+1
        for (u32 q = threadIdx.x; q < (tile_elems >> 2); q += nthr) {
            u32 c = q & (cols - 1), rr = q >> cols_bits;
            u32 r0 = ((rr >> l) << (l + 2)) | (rr & ((1u << l) - 1u));
            u32 j = (u32)((base | ((size_t)r0 << lo) | c) & (((size_t)1 << s) - 1));
            Fr wc = fr_load(&t_lo[j]);
            Fr wa = fr_load(&t_hi[j]);
            Fr wb = fr_load(&t_hi[j + (1u << s)]);
            u32 i0 = (r0 << cols_bits) | c, st = 1u << (l + cols_bits);
            Fr x0 = lds[i0], x1 = lds[i0 + st], x2 = lds[i0 + 2 * st], x3 = lds[i0 + 3 * st];
            if (DIT) {
                Fr v = Fr::mul(x1, wc), u = Fr::mul(x3, wc);
                x1 = Fr::sub(x0, v); x0 = Fr::add(x0, v);
                x3 = Fr::sub(x2, u); x2 = Fr::add(x2, u);
                v = Fr::mul(x2, wa); u = Fr::mul(x3, wb);
                x2 = Fr::sub(x0, v); x0 = Fr::add(x0, v);
                x3 = Fr::sub(x1, u); x1 = Fr::add(x1, u);
            } else {
                Fr v = Fr::sub(x0, x2), u = Fr::sub(x1, x3);
                x0 = Fr::add(x0, x2); x1 = Fr::add(x1, x3);
                x2 = Fr::mul(v, wa); x3 = Fr::mul(u, wb);
                v = Fr::sub(x0, x1); u = Fr::sub(x2, x3);
                x0 = Fr::add(x0, x1); x2 = Fr::add(x2, x3);
                x1 = Fr::mul(v, wc); x3 = Fr::mul(u, wc);
            }
            lds[i0] = x0; lds[i0 + st] = x1; lds[i0 + 2 * st] = x2; lds[i0 + 3 * st] = x3;
        }
        __syncthreads();
    }
    if (nst & 1) {
        const u32 l = DIT ? nst - 1 : 0;
        const u32 s = lo + l;
        const Fr* t_lo = tws + (((size_t)1 << s) - 1);
        for (u32 q = threadIdx.x; q < (tile_elems >> 1); q += nthr) {
            u32 c = q & (cols - 1), rr = q >> cols_bits;
            u32 r0 = ((rr >> l) << (l + 1)) | (rr & ((1u << l) - 1u));
            u32 j = (u32)((base | ((size_t)r0 << lo) | c) & (((size_t)1 << s) - 1));
            Fr w = fr_load(&t_lo[j]);
            u32 i0 = (r0 << cols_bits) | c, st = 1u << (l + cols_bits);
            Fr x0 = lds[i0], x1 = lds[i0 + st];
            if (DIT) {
                Fr v = Fr::mul(x1, w);
                lds[i0] = Fr::add(x0, v);
                lds[i0 + st] = Fr::sub(x0, v);
            } else {
                lds[i0] = Fr::add(x0, x1);
                lds[i0 + st] = Fr::mul(Fr::sub(x0, x1), w);
            }
        }
        __syncthreads();
    }
    for (u32 e = threadIdx.x; e < tile_elems; e += nthr) {
        u32 r = e >> cols_bits, c = e & (cols - 1);
        size_t gi = base | ((size_t)r << lo) | c;
        Fr x = lds[e];
        if (post & 2) x = Fr::mul(x, pow_from_tables(pw, logn ? (__brev((u32)gi) >> (32 - logn)) : 0u, logn));
        if (post & 4) x = Fr::sub(x, Fr::mul(fr_load(&sub[gi]), kc));
        if (post & 1) x = Fr::mul(x, scale);
        fr_store(&vec[gi], x);
    }
}

// x[pos] *= scale * g^(idx)  with idx = pos (natural) or bitrev(pos) (after a DIF pass chain).
// pw: three POW_TABLE_SIZE-entry tables of g^(j), g^(2^11 j), g^(2^22 j); batched over blockIdx.y.
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
        x = Fr::mul(x, pow_from_tables(pw, j, logn));
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
// out[row] = <M_row, z>   (ark-groth16 `evaluate_constraint`), one lane per row of the WHOLE domain vector: rows past
// the matrix are written too - z[row - n_rows] for the n_copy instance rows of a ("a[start..end] =
// full_assignment[..num_inputs]"), zero for the rest - so the 3 m-element vectors need no memset before the transforms
template <class Fr>
__global__ void k_spmv(const u64* __restrict__ row_ptr, const u32* __restrict__ col,
                       const Fr* __restrict__ val, const Fr* __restrict__ z, Fr* __restrict__ out,
                       u32 n_rows, u32 n_copy, u32 m) {
    u32 row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m) return;
    if (row >= n_rows) {
        fr_store(&out[row], row - n_rows < n_copy ? fr_load(&z[row - n_rows]) : Fr::zero());
        return;
    }
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

// Structural validation of a CSR matrix before any kernel indexes with it: row_ptr non-decreasing and within nnz,
// every column < n_cols.  *bad becomes non-zero on the first violation (grid-stride over rows and non-zeros).
template <int UNUSED>
__global__ void k_csr_check(const u64* __restrict__ row_ptr, const u32* __restrict__ col, u64 n_rows, u64 nnz,
                            u32 n_cols, u32* __restrict__ bad) {
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x, stride = (u64)gridDim.x * blockDim.x;
    u32 f = 0;
    for (u64 i = t; i < n_rows; i += stride) {
        u64 b = row_ptr[i], e = row_ptr[i + 1];
        if (b > e || e > nnz) f = 1;
    }
    for (u64 k = t; k < nnz; k += stride)
        if (col[k] >= n_cols) f = 2;
    if (t == 0 && (row_ptr[0] != 0 || row_ptr[n_rows] != nnz)) f = 3;
    if (f) atomicOr(bad, f);
}

// a[n_c + j] = z[j] for j < n_inst  (witness_map_from_matrices: "a[start..end] = full_assignment[..num_inputs]")
template <class Fr>
__global__ void k_copy_inputs(Fr* __restrict__ a, const Fr* __restrict__ z, u32 n_c, u32 n_inst) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n_inst) fr_store(&a[n_c + j], fr_load(&z[j]));
}

// a[i] *= b[i]   (the product of the coset evaluations; everything else of the quotient step is the epilogue
// of the inverse transform that follows, see QapHost::run)
template <class Fr>
__global__ void k_mul_pointwise(Fr* __restrict__ a, const Fr* __restrict__ b, size_t m) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    fr_store(&a[i], Fr::mul(fr_load(&a[i]), fr_load(&b[i])));
}

#endif  // __HIPCC__

}  // namespace hk
