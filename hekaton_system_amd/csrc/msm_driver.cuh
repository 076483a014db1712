// msm_driver.cuh — host-side launch sequences for the MSM kernels of msm.cuh.
// Declarations only; msm_driver_impl.cuh holds the definitions and is included by the translation
// unit that explicitly instantiates a (scalar field) or (coordinate field) flavour, so the heavy
// kernels are compiled once per flavour and in parallel.
#pragma once
#include "hk_internal.h"
#include "pairing.cuh"

namespace hk {

// private-memory ("scratch") bytes per lane of a kernel, as the loaded code object declares them
static inline size_t hk_private_bytes_of(const void* kernel) {
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, kernel) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return (size_t)fa.localSizeBytes;
}

struct SortBufs {        // device buffers produced by the counting sort
    u32* count;          // [NB]
    u32* start;          // [NB + 1]   start[NB] = number of non-zero digits E
    u32* cursor;         // [NB]
    u32* sorted;         // [n * W]    entry ids grouped by bucket
    short* digits;       // [W][n]     signed window digits, written once by k_msm_hist
};

template <class Fr>
struct MsmSort {
    static hk_status alloc(Lane* L, const MsmPlan& p, SortBufs* out);
    // scalars_d: n field elements on the device (canonical, or Montgomery when is_mont)
    // count_is_zero: the caller cleared sb.count in a kernel of its own that precedes this call in stream order
    static hk_status run(hipStream_t s, const MsmPlan& p, const u32* scalars_d, int is_mont,
                         const SortBufs& sb, bool count_is_zero = false);
};

template <class F>
struct MsmRun {
    struct Bufs {
        XYZZ<F>* buckets;        // [NB]
        u32* pkeys[2];           // boundary-partial keys, ping-pong
        XYZZ<F>* ppts[2];
        XYZZ<F>* red;            // [WP * B / K]
        XYZZ<F>* wsum;           // [WP]
    };
    static hk_status alloc(Lane* L, const MsmPlan& p, Bufs* out);
    // table: F shift groups of n_bases affine points each.  result_d receives one XYZZ point.
    // ev0/ev1 (optional) bracket the bucket-accumulate launches for hk_timings.
    static hk_status run(hipStream_t s, const MsmPlan& p, const Affine<F>* table, u32 n_bases, u32 idx_off,
                         const SortBufs& sb, const Bufs& b, XYZZ<F>* result_d,
                         hipEvent_t ev0, hipEvent_t ev1);
    static hk_status build_tables(hipStream_t s, Affine<F>* table, u32 n, u32 groups, u32 shift_bits);
    static hk_status to_affine(hipStream_t s, const XYZZ<F>* in, Affine<F>* out, u32 n);
    // fixed-base batch scalar multiplication (fixed_base.cuh); all pointers device.
    // table: 32*256 affine scratch, xy: n XYZZ scratch, pref: n field-element scratch
    static hk_status fixed_base(hipStream_t s, const Affine<F>* base, const void* scalars, int is_mont,
                                u32 n, Affine<F>* table, XYZZ<F>* xy, F* pref, Affine<F>* out, bool build_table = true);
    static hk_status batch_affine(hipStream_t s, const XYZZ<F>* in, Affine<F>* out, F* pref, u32 n);
    // out[i] = scalars[i] * points[i] (pairing_ops.rs:32-39); xy / pref: n-element scratch
    // tab: 2^K x n XYZZ scratch (K = 2 in G1, 4 in G2) for the subset sums of the endomorphism images (nullptr: the plain
    // 254-step ladder)
    static hk_status scalar_mul_each(hipStream_t s, const Affine<F>* pts, const void* scalars_mont, u32 n,
                                     XYZZ<F>* xy, F* pref, Affine<F>* out, XYZZ<F>* tab = nullptr);
    // out[i] = sum_j coeffs[j] * vecs[j][i], k <= LINCOMB_MAX (aggregation.rs:192-203,293-326)
    static hk_status lincomb(hipStream_t s, const Affine<F>* const* vecs, const void* coeffs_mont, u32 k, u32 n,
                             XYZZ<F>* xy, F* pref, Affine<F>* out);
    // out[i] = lo[i] + c * hi[i], c given as EndoOf<F>::K magnitudes (Montgomery Fr) and a sign mask (endo.cuh
    // k_points_fold_endo / k_points_mul_split); tab: endo_tab_bytes(n) of scratch
    static hk_status fold_endo(hipStream_t s, u32 k, const Affine<F>* const* lo, const Affine<F>* const* hi, const void* coeffs_mont,
                               u32 neg_mask, u32 n, XYZZ<F>* tab, XYZZ<F>* xy, F* pref, Affine<F>* out);
    // a short one-off MSM without tables (n K <= SPLIT_MAX_LANES): element-wise products + one workgroup's sum; result: 1 XYZZ
    static hk_status small_msm(hipStream_t s, const Affine<F>* bases, const void* scalars, int mont, u32 n, XYZZ<F>* tab,
                               XYZZ<F>* xy, XYZZ<F>* result);
    static hk_status small_msm_rows(hipStream_t s, const Affine<F>* bases, const void* scalars, u32 seg, u32 batch, XYZZ<F>* tab,
                                    XYZZ<F>* xy, XYZZ<F>* result);
    // largest private-memory ("scratch") frame per lane among this flavour's kernels, from the loaded code object
    // (hipFuncGetAttributes): what sizes a hardware queue's scratch ring
    static size_t max_private_bytes();
};

// which (lhs vector, rhs vector) pairs a call multiplies out: n = 0 -> every pair, product p = a * n_r + b; else product p
// pairs lhs vector a[p] with rhs vector b[p] (hk_pairing_pairs: the ten cross terms of a GIPA round out of 6 x 6)
constexpr int PAIR_LIST_MAX = 64;
struct PairList {
    u32 n;
    unsigned char a[PAIR_LIST_MAX], b[PAIR_LIST_MAX];
};

// multi-pairing launch sequence (pairing.cuh); explicit instantiation in hk_<curve>_pair.hip
template <class P>
struct PairRun {
    // g1: n_l vectors of n points, g2: n_r vectors of n points (device).  miller: scratch_bytes(n, n_l*n_r) of scratch,
    // prod: n_l*n_r scratch, out: n_l*n_r results (device), out[a*n_r + b] = prod_i e(g1[a][i], g2[b][i]);
    // with `pairs`: count = pairs->n products, out[p] = prod_i e(g1[pairs->a[p]][i], g2[pairs->b[p]][i]).
    static hk_status run(hipStream_t s, const Affine<Fp<P>>* g1, const Affine<Fp2<P>>* g2, u32 n, u32 n_l, u32 n_r,
                         Fp12<P>* miller, Fp12<P>* prod, Fp12<P>* out, const PairList* pairs = nullptr);
    static size_t scratch_bytes(u32 n, u32 count, u32 n_r);
    static size_t max_private_bytes();   // as MsmRun<F>::max_private_bytes, over the pairing / endomorphism kernels
    static u32 steps();          // line-evaluation steps of the Miller loop (the factor of `count` in the tree launches' grid.y)
    // out[e] = in[e]^scalars[e] (GT powers; device pointers)
    // in_gt: the elements lie in GT (order r) - the exponent is then split along the Frobenius (k_gt_pow_endo)
    static hk_status gt_pow(hipStream_t s, const Fp12<P>* in, const void* scalars_mont, u32 n, Fp12<P>* out, bool in_gt);
    static hk_status gt_prod(hipStream_t s, const Fp12<P>* in, u32 len, u32 groups, Fp12<P>* out);
};

}  // namespace hk
