// fixed_base.cuh — fixed-base batch scalar multiplication  out[i] = scalars[i] * base  on the device.
//
// Replaces ark-ec `FixedBase::{get_window_table, msm}` + `CurveGroup::normalize_batch` as used by the
// reference's trusted setup (cp-groth16/src/generator.rs:126-224: a_g, b_g, b_h, h_g, deltas_abc_g,
// gamma_abc_g, deltas_g are all "scalar vector times one generator").  SURVEY.md §8(f) row 3; it is
// also how tests and bench.py build genuine Groth16 SRSs at full size without touching the oracle.
//
// 8-bit windows: table[w][j] = j * 2^(8w) * base (affine), one mixed add per non-zero byte of the
// scalar; results are normalised with Montgomery's batch-inversion trick, 16 points per lane.
#pragma once
#include "msm.cuh"

namespace hk {

constexpr int FB_WINDOWS = 32;       // 256-bit scalars
constexpr int FB_CHUNK = 16;         // points per lane in the batch normalisation

#if defined(__HIPCC__)

// table[w * 256 + j] = j * 2^(8w) * base ; one lane per entry (setup cost only)
template <class F>
__global__ void __launch_bounds__(64)
k_fb_table(const Affine<F>* __restrict__ base, Affine<F>* __restrict__ table) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= FB_WINDOWS * 256) return;
    u32 w = t >> 8, j = t & 255;
    XYZZ<F> p = XYZZ<F>::from_affine(ld_vec(base));
    for (u32 k = 0; k < 8 * w; k++) p = ec_dbl_ni(p);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (int bit = 7; bit >= 0; bit--) {
        acc = ec_dbl_ni(acc);
        if ((j >> bit) & 1) acc = ec_add_ni(acc, p);
    }
    st_vec(&table[t], ec_to_affine(acc));
}

template <class Fr, class F>
__global__ void __launch_bounds__(64)
k_fb_mul(const Affine<F>* __restrict__ table, const Fr* __restrict__ scalars, int is_mont, u32 n,
         XYZZ<F>* __restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s;
    const uint4* src = reinterpret_cast<const uint4*>(scalars + i);
#pragma unroll
    for (int k = 0; k < Fr::N / 4; k++) {
        uint4 v = src[k];
        s.v[4 * k] = v.x; s.v[4 * k + 1] = v.y; s.v[4 * k + 2] = v.z; s.v[4 * k + 3] = v.w;
    }
    if (is_mont) s = Fr::from_mont(s);
    XYZZ<F> acc = XYZZ<F>::inf();
    for (u32 w = 0; w < (u32)FB_WINDOWS; w++) {
        u32 d = (s.v[w >> 2] >> (8 * (w & 3))) & 255u;
        if (d) acc = ec_madd(acc, ld_vec(&table[w * 256 + d]));
    }
    st_vec(&out[i], acc);
}

// out[i] = sum_j coeffs[j] * vecs[j][i]  (XYZZ), one lane per element, ONE shared doubling chain for the k <= 8 terms
// (Straus): the aggregator's element-wise combinations - `prepared_input = s0 + s1*x0 + s2*x1 + s3*x2`
// (distributed-prover/src/aggregation.rs:192-203) and `left = A + S^s + D^(s^2) + C^(s^3)`, `right = B + H^t + ...`
// (:293-326: three scalar_pairing sweeps with a constant scalar followed by element-wise additions).
constexpr int LINCOMB_MAX = 8;
template <class F> struct LincombVecs { const Affine<F>* v[LINCOMB_MAX]; };
template <class Fr, class F>
__global__ void __launch_bounds__(64)
k_points_lincomb(LincombVecs<F> vecs, const Fr* __restrict__ coeffs_mont, u32 k, u32 n, XYZZ<F>* __restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr c[LINCOMB_MAX];
    int top = -1;
    for (u32 j = 0; j < k; j++) {
        c[j] = Fr::from_mont(ld_vec(&coeffs_mont[j]));
        for (int b = Fr::N * 32 - 1; b > top; b--)
            if ((c[j].v[b >> 5] >> (b & 31)) & 1) { top = b; break; }
    }
    XYZZ<F> acc = XYZZ<F>::inf();
    HK_NOUNROLL for (int b = top; b >= 0; b--) {
        acc = ec_dbl_ni(acc);
        HK_NOUNROLL for (u32 j = 0; j < k; j++)
            if ((c[j].v[b >> 5] >> (b & 31)) & 1) acc = ec_madd_ni(acc, ld_vec(&vecs.v[j][i]));
    }
    st_vec(&out[i], acc);
}

// out[i] = scalars[i] * points[i] (XYZZ), one lane per element, plain double-and-add over the canonical scalar
// (`scalar_pairing`, distributed-prover/src/pairing_ops.rs:32-39; N = #subcircuits <= 1024, latency-bound)
template <class Fr, class F>
__global__ void __launch_bounds__(64)
k_scalar_mul_each(const Affine<F>* __restrict__ pts, const Fr* __restrict__ scalars, u32 n, XYZZ<F>* __restrict__ out) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s = Fr::from_mont(ld_vec(&scalars[i]));
    XYZZ<F> p = XYZZ<F>::from_affine(ld_vec(&pts[i]));
    st_vec(&out[i], ec_mul_limbs(p, s.v));
}

// XYZZ -> affine for n points, `chunk` per lane, one field inversion per lane (Montgomery's trick inside the lane).  The
// caller picks chunk = 1 while that still leaves at most one wave per SIMD (n <= 65536: a lane's serial work is then the
// inversion plus 8 products instead of the inversion plus 8 per chunk element), FB_CHUNK for long vectors.
// scratch: n coordinate-field elements (prefix products of zzz).
template <class F>
__global__ void __launch_bounds__(64)
k_batch_affine(const XYZZ<F>* __restrict__ in, Affine<F>* __restrict__ out, F* __restrict__ scratch, u32 n, u32 chunk) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if ((size_t)t * chunk >= n) return;
    u32 lo = t * chunk;
    u32 hi = min(lo + chunk, n);
    F acc = F::one();
    for (u32 i = lo; i < hi; i++) {
        st_vec(&scratch[i], acc);
        F zzz = ld_vec(&in[i].zzz);
        if (!zzz.is_zero()) acc = f_mul_ni(acc, zzz);
    }
    F inv = fp_inv(acc);
    for (u32 i = hi; i-- > lo;) {
        XYZZ<F> p = ld_vec(&in[i]);
        if (p.is_inf()) { st_vec(&out[i], Affine<F>::inf()); continue; }
        F zzz_inv = f_mul_ni(inv, ld_vec(&scratch[i]));
        inv = f_mul_ni(inv, p.zzz);
        F zz_inv = f_mul_ni(f_mul_ni(zzz_inv, zzz_inv), f_mul_ni(p.zz, p.zz));
        Affine<F> a;
        a.x = f_mul_ni(p.x, zz_inv);
        a.y = f_mul_ni(p.y, zzz_inv);
        st_vec(&out[i], a);
    }
}

#endif

}  // namespace hk
