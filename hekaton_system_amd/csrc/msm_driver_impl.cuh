// msm_driver_impl.cuh — definitions for msm_driver.cuh (see there).
#pragma once
#include <hip/hip_ext.h>
#include "msm_driver.cuh"
#include "fixed_base.cuh"
#include "endo.cuh"

namespace hk {

// HK_DEBUG_SYNC=1: synchronise after every launch and name it on stderr (locates a faulting kernel)
static inline bool hk_dbg_sync() { static const bool on = getenv("HK_DEBUG_SYNC") != nullptr; return on; }
#define HK_DBG(stream, name)                                                              \
    do {                                                                                  \
        if (hk_dbg_sync()) {                                                              \
            fprintf(stderr, "[hk] launched %s\n", name); fflush(stderr);                  \
            hipError_t _e = hipStreamSynchronize(stream);                                 \
            fprintf(stderr, "[hk] finished %s: %s\n", name, hipGetErrorName(_e)); fflush(stderr); \
        }                                                                                 \
    } while (0)

template <class Fr>
hk_status MsmSort<Fr>::alloc(Lane* L, const MsmPlan& p, SortBufs* out) {
    out->count = L->alloc_n<u32>(p.NB);
    out->start = L->alloc_n<u32>(p.NB + 1);
    out->cursor = L->alloc_n<u32>(p.NB);
    out->sorted = L->alloc_n<u32>((size_t)p.n * p.W + 1);
    out->digits = L->alloc_n<short>((size_t)p.n * p.W + 8);
    if (!out->count || !out->start || !out->cursor || !out->sorted || !out->digits) return HK_ERR_NOMEM;
    return HK_OK;
}

template <class Fr>
hk_status MsmSort<Fr>::run(hipStream_t s, const MsmPlan& p, const u32* scalars_d, int is_mont,
                           const SortBufs& sb, bool count_is_zero) {
    if (p.NB > (u32)MSM_LDS_COUNTERS || p.c > 16) return HK_ERR_ARG;       // digits are stored as int16
    if (!count_is_zero) HK_HIP(hipMemsetAsync(sb.count, 0, sizeof(u32) * p.NB, s));
    u32 blocks = (p.n + p.chunk - 1) / p.chunk;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL((k_msm_hist<Fr>), dim3(blocks), dim3(MSM_SORT_THREADS), 0, s,
                       scalars_d, is_mont, p, sb.count, sb.digits);
    HK_DBG(s, "k_msm_hist");
    hipLaunchKernelGGL((k_msm_scan<0>), dim3(1), dim3(1024), 0, s, sb.count, sb.start, sb.cursor, p.NB);
    HK_DBG(s, "k_msm_scan");
    hipLaunchKernelGGL((k_msm_scatter<Fr>), dim3(blocks), dim3(MSM_SORT_THREADS), 0, s,
                       (const short*)sb.digits, p, sb.cursor, sb.sorted);
    HK_DBG(s, "k_msm_scatter");
    HK_HIP(hipGetLastError());
    return HK_OK;
}

// accumulate schedule for this coordinate field: as many lanes as the kernel keeps resident
// (AccumOcc<F>::waves per SIMD x 1024 SIMDs x 64), so the sorted list is consumed in ONE balanced round
template <class F>
static inline MsmPlan msm_lane_plan(const MsmPlan& p0) {
    MsmPlan p = p0;
    msm_set_lanes(p, (u32)AccumOcc<F>::waves * 65536u);
    return p;
}

template <class F>
hk_status MsmRun<F>::alloc(Lane* L, const MsmPlan& p0, Bufs* out) {
    const MsmPlan p = msm_lane_plan<F>(p0);
    out->buckets = L->alloc_n<XYZZ<F>>(p.NB + 1);          // + one slot: the reduction ticket, zeroed with the buckets
    size_t n0 = 2ull * p.T[0];
    size_t n1 = p.n_levels > 1 ? 2ull * p.T[1] : 2;
    out->pkeys[0] = L->alloc_n<u32>(n0);
    out->ppts[0] = L->alloc_n<XYZZ<F>>(n0);
    out->pkeys[1] = L->alloc_n<u32>(n1);
    out->ppts[1] = L->alloc_n<XYZZ<F>>(n1);
    out->red = L->alloc_n<XYZZ<F>>((size_t)p.WP * (p.B / p.K));
    out->wsum = L->alloc_n<XYZZ<F>>(p.WP);
    if (!out->buckets || !out->pkeys[0] || !out->ppts[0] || !out->pkeys[1] || !out->ppts[1] ||
        !out->red || !out->wsum)
        return HK_ERR_NOMEM;
    return HK_OK;
}

template <class F>
hk_status MsmRun<F>::run(hipStream_t s, const MsmPlan& p0, const Affine<F>* table, u32 n_bases, u32 idx_off,
                         const SortBufs& sb, const Bufs& b, XYZZ<F>* result_d,
                         hipEvent_t ev0, hipEvent_t ev1) {
    const MsmPlan p = msm_lane_plan<F>(p0);
    // no memset of the buckets (4 - 8 MB per MSM): k_msm_accum0 writes every bucket that has entries and clears the
    // ticket, the reductions skip the empty ones
    u32* ticket = reinterpret_cast<u32*>(b.buckets + p.NB);
    // with profiling on, ev0/ev1 take the kernel's own start/stop timestamps (hipExtLaunchKernelGGL), so
    // the figure agrees with rocprofv3's kernel trace even when other lanes share the hardware queues
    if (ev0 && ev1)
        hipExtLaunchKernelGGL((k_msm_accum0<F>), dim3((p.T[0] + 63) / 64), dim3(64), 0, s, ev0, ev1, 0,
                              table, n_bases, idx_off, sb.sorted, sb.start, p, b.buckets, b.pkeys[0], b.ppts[0]);
    else
        hipLaunchKernelGGL((k_msm_accum0<F>), dim3((p.T[0] + 63) / 64), dim3(64), 0, s,
                           table, n_bases, idx_off, sb.sorted, sb.start, p, b.buckets, b.pkeys[0], b.ppts[0]);
    HK_DBG(s, "k_msm_accum0");
    u32 k = 1;
    for (; k < p.n_levels && p.T[k] > (u32)MSM_TAIL_THREADS; k++) {
        int in = (k - 1) & 1, out = k & 1;
        hipLaunchKernelGGL((k_msm_accum_lvl<F>), dim3((p.T[k] + 63) / 64), dim3(64), 0, s,
                           (int)k, b.pkeys[in], b.ppts[in], sb.start, p, b.buckets, b.pkeys[out], b.ppts[out]);
        HK_DBG(s, "k_msm_accum_lvl");
    }
    if (k < p.n_levels) {                                   // the remaining levels fit one workgroup each: one launch
        hipLaunchKernelGGL((k_msm_accum_tail<F>), dim3(1), dim3(MSM_TAIL_THREADS), 0, s, (int)k, b.pkeys[0], b.ppts[0],
                           b.pkeys[1], b.ppts[1], sb.start, p, b.buckets);
        HK_DBG(s, "k_msm_accum_tail");
    }
    u32 J = p.B / p.K;
    if (p.WP == 1) {
        u32 blocks = (J + MSM_REDUCE_THREADS - 1) / MSM_REDUCE_THREADS;
        hipLaunchKernelGGL((k_msm_reduce_fused<F>), dim3(blocks), dim3(MSM_REDUCE_THREADS), 0, s, b.buckets, sb.start, p, b.red,
                           ticket, result_d);
        HK_DBG(s, "k_msm_reduce_fused");
    } else {
        hipLaunchKernelGGL((k_msm_bucket_reduce<F>), dim3((p.WP * J + 63) / 64), dim3(64), 0, s,
                           b.buckets, sb.start, p, b.red);
        HK_DBG(s, "k_msm_bucket_reduce");
        hipLaunchKernelGGL((k_msm_window_sum<F>), dim3(p.WP), dim3(MSM_WSUM_THREADS), 0, s, b.red, p, b.wsum);
        HK_DBG(s, "k_msm_window_sum");
        hipLaunchKernelGGL((k_msm_final<F>), dim3(1), dim3(64), 0, s, b.wsum, p, result_d);
        HK_DBG(s, "k_msm_final");
    }
    HK_HIP(hipGetLastError());
    return HK_OK;
}


template <class F>
size_t MsmRun<F>::max_private_bytes() {
    typedef typename ScalarOf<F>::type Fr;
    const void* ks[] = {(const void*)k_msm_accum0<F>, (const void*)k_msm_accum_lvl<F>, (const void*)k_msm_accum_tail<F>,
                        (const void*)k_msm_reduce_fused<F>, (const void*)k_msm_bucket_reduce<F>,
                        (const void*)k_msm_window_sum<F>, (const void*)k_msm_final<F>, (const void*)k_to_affine<F>,
                        (const void*)k_msm_build_tables<F>, (const void*)k_batch_affine<F>, (const void*)k_fb_table<F>,
                        (const void*)k_fb_mul<Fr, F>, (const void*)k_scalar_mul_each<Fr, F>, (const void*)k_scalar_mul_endo<Fr, F>, (const void*)k_points_fold_endo<Fr, F>, (const void*)k_points_mul_split<Fr, F, true>, (const void*)k_points_mul_split<Fr, F, false>, (const void*)k_points_sum<F>, (const void*)k_points_sum_seg<F>,
                        (const void*)k_points_lincomb<Fr, F>};
    size_t m = 0;
    for (const void* k : ks) { size_t b = hk_private_bytes_of(k); if (b > m) m = b; }
    return m;
}

template <class F>
hk_status MsmRun<F>::build_tables(hipStream_t s, Affine<F>* table, u32 n, u32 groups, u32 shift_bits) {
    if (groups <= 1 || n == 0) return HK_OK;
    hipLaunchKernelGGL((k_msm_build_tables<F>), dim3((n + 63) / 64), dim3(64), 0, s, table, n, groups,
                       shift_bits);
    HK_HIP(hipGetLastError());
    return HK_OK;
}

template <class F>
hk_status MsmRun<F>::to_affine(hipStream_t s, const XYZZ<F>* in, Affine<F>* out, u32 n) {
    hipLaunchKernelGGL((k_to_affine<F>), dim3((n + 63) / 64), dim3(64), 0, s, in, out, n);
    HK_DBG(s, "k_to_affine");
    HK_HIP(hipGetLastError());
    return HK_OK;
}

template <class F>
hk_status MsmRun<F>::batch_affine(hipStream_t s, const XYZZ<F>* in, Affine<F>* out, F* pref, u32 n) {
    if (n == 0) return HK_OK;
    u32 chunk = (n + 65535) / 65536;                       // 1 up to one wave per SIMD, then longer serial runs per lane
    if (chunk > (u32)FB_CHUNK) chunk = FB_CHUNK;
    u32 lanes = (n + chunk - 1) / chunk;
    hipLaunchKernelGGL((k_batch_affine<F>), dim3((lanes + 63) / 64), dim3(64), 0, s, in, out, pref, n, chunk);
    HK_HIP(hipGetLastError());
    return HK_OK;
}

// the K-lane sweep (endo.cuh): for G2 with few elements every Fq2 value on a quad of lanes (Fp2Q), else one lane per value
template <class Fr, class P, bool UNIFORM>
static inline void launch_mul_split(hipStream_t s, const SplitVecs<Fp2<P>>& v, u32 k, const Fr* scalars, u32 neg_mask, u32 n,
                                    const EndoSplit<4>& E, Jac<Fp2<P>>* tab, XYZZ<Fp2<P>>* xy, int scalars_mont = 1) {
    const bool no_quad = getenv("HK_ENDO_NO_QUAD") != nullptr;
    u32 lanes = n * 4;
    if (!no_quad && (size_t)lanes * 4 <= SPLIT_MAX_LANES) {
        typedef Fp2Q<P> Q;                                    // same memory layout as Fp2<P>
        SplitVecs<Q> vq;
        for (int y = 0; y < FOLD_MAX; y++) { vq.lo[y] = (const Affine<Q>*)v.lo[y]; vq.pts[y] = (const Affine<Q>*)v.pts[y]; }
        vq.pts_mod = v.pts_mod;
        hipLaunchKernelGGL((k_points_mul_split<Fr, Q, UNIFORM>), dim3((lanes * 4 + 63) / 64, k), dim3(64), 0, s, vq, scalars,
                           neg_mask, n, E, (Jac<Q>*)tab, (XYZZ<Q>*)xy, scalars_mont);
    } else {
        hipLaunchKernelGGL((k_points_mul_split<Fr, Fp2<P>, UNIFORM>), dim3((lanes + 63) / 64, k), dim3(64), 0, s, v, scalars,
                           neg_mask, n, E, tab, xy, scalars_mont);
    }
}
template <class Fr, class P, bool UNIFORM>
static inline void launch_mul_split(hipStream_t s, const SplitVecs<Fp<P>>& v, u32 k, const Fr* scalars, u32 neg_mask, u32 n,
                                    const EndoSplit<2>& E, Jac<Fp<P>>* tab, XYZZ<Fp<P>>* xy, int scalars_mont = 1) {
    hipLaunchKernelGGL((k_points_mul_split<Fr, Fp<P>, UNIFORM>), dim3((n * 2 + 63) / 64, k), dim3(64), 0, s, v, scalars,
                       neg_mask, n, E, tab, xy, scalars_mont);
}

template <class F>
hk_status MsmRun<F>::scalar_mul_each(hipStream_t s, const Affine<F>* pts, const void* scalars_mont, u32 n,
                                     XYZZ<F>* xy, F* pref, Affine<F>* out, XYZZ<F>* tab) {
    typedef typename ScalarOf<F>::type Fr;
    if (n == 0) return HK_OK;
    static const bool plain = getenv("HK_SCALAR_MUL_PLAIN") != nullptr;      // the 254-step ladder (A/B, debugging)
    const bool one_lane = getenv("HK_ENDO_ONE_LANE") != nullptr;             // never K lanes per element (A/B, tests)
    if (tab && !plain && !one_lane && (size_t)n * EndoOf<F>::K <= SPLIT_MAX_LANES) {
        // short vector: K lanes per element, 4-bit windows, Jacobian chain (endo.cuh)
        static const auto E = EndoOf<F>::split();
        SplitVecs<F> v = {};
        v.pts[0] = pts;
        launch_mul_split<Fr, typename F::Params, false>(s, v, 1u, (const Fr*)scalars_mont, 0u, n, E, reinterpret_cast<Jac<F>*>(tab), xy);
    } else if (tab && !plain) {
        // scalars split on the device along phi / psi: one shared chain of 131 (G1) / 68 (G2) doublings (endo.cuh)
        static const auto E = EndoOf<F>::split();
        hipLaunchKernelGGL((k_scalar_mul_endo<Fr, F>), dim3((n + 63) / 64), dim3(64), 0, s, pts, (const Fr*)scalars_mont, n, E,
                           tab, xy);
    } else {
        hipLaunchKernelGGL((k_scalar_mul_each<Fr, F>), dim3((n + 63) / 64), dim3(64), 0, s, pts, (const Fr*)scalars_mont,
                           n, xy);
    }
    HK_HIP(hipGetLastError());
    return batch_affine(s, xy, out, pref, n);
}

// sum_i s_i P_i for a SHORT vector without tables: n element-wise products over the endomorphism (k_points_mul_split),
// then one workgroup's sum (k_points_sum).  scalars: Montgomery (mont != 0) or canonical integers.
template <class F>
hk_status MsmRun<F>::small_msm(hipStream_t s, const Affine<F>* bases, const void* scalars, int mont, u32 n, XYZZ<F>* tab,
                               XYZZ<F>* xy, XYZZ<F>* result) {
    typedef typename ScalarOf<F>::type Fr;
    if (n == 0 || (size_t)n * EndoOf<F>::K > SPLIT_MAX_LANES) return HK_ERR_ARG;
    static const auto E = EndoOf<F>::split();
    SplitVecs<F> v = {};
    v.pts[0] = bases;
    launch_mul_split<Fr, typename F::Params, false>(s, v, 1u, (const Fr*)scalars, 0u, n, E, reinterpret_cast<Jac<F>*>(tab), xy, mont ? 1 : 0);
    hipLaunchKernelGGL((k_points_sum<F>), dim3(1), dim3(PointsSum<F>::THREADS), 0, s, (const XYZZ<F>*)xy, n, result);
    HK_HIP(hipGetLastError());
    return HK_OK;
}

// batch x seg element-wise products over ONE base set of seg points (scalars: [batch][seg] Montgomery), summed per row:
// result[b] = sum_t scalars[b][t] * bases[t] - the stage commitments of every subcircuit of a key class in one launch
template <class F>
hk_status MsmRun<F>::small_msm_rows(hipStream_t s, const Affine<F>* bases, const void* scalars, u32 seg, u32 batch, XYZZ<F>* tab,
                                    XYZZ<F>* xy, XYZZ<F>* result) {
    typedef typename ScalarOf<F>::type Fr;
    size_t n = (size_t)seg * batch;
    if (n == 0 || n * EndoOf<F>::K > SPLIT_MAX_LANES) return HK_ERR_ARG;
    static const auto E = EndoOf<F>::split();
    SplitVecs<F> v = {};
    v.pts[0] = bases;
    v.pts_mod = seg;
    launch_mul_split<Fr, typename F::Params, false>(s, v, 1u, (const Fr*)scalars, 0u, (u32)n, E, reinterpret_cast<Jac<F>*>(tab), xy, 1);
    hipLaunchKernelGGL((k_points_sum_seg<F>), dim3(batch), dim3(64), 0, s, (const XYZZ<F>*)xy, seg, result);
    HK_HIP(hipGetLastError());
    return HK_OK;
}

template <class F>
hk_status MsmRun<F>::fold_endo(hipStream_t s, u32 k, const Affine<F>* const* lo, const Affine<F>* const* hi, const void* coeffs_mont,
                               u32 neg_mask, u32 n, XYZZ<F>* tab, XYZZ<F>* xy, F* pref, Affine<F>* out) {
    typedef typename ScalarOf<F>::type Fr;
    if (n == 0 || k == 0) return HK_OK;
    if (k > (u32)FOLD_MAX) return HK_ERR_ARG;
    const bool one_lane = getenv("HK_ENDO_ONE_LANE") != nullptr;
    if (!one_lane && (size_t)n * EndoOf<F>::K <= SPLIT_MAX_LANES) {
        static const auto E = EndoOf<F>::split();                                // unused by the uniform form
        SplitVecs<F> v = {};
        for (u32 y = 0; y < k; y++) { v.lo[y] = lo[y]; v.pts[y] = hi[y]; }
        launch_mul_split<Fr, typename F::Params, true>(s, v, k, (const Fr*)coeffs_mont, neg_mask, n, E, reinterpret_cast<Jac<F>*>(tab), xy);
    } else {
        for (u32 y = 0; y < k; y++)                                              // long vectors: throughput-bound, one after the other
            hipLaunchKernelGGL((k_points_fold_endo<Fr, F>), dim3((n + 63) / 64), dim3(64), 0, s, lo[y], hi[y], (const Fr*)coeffs_mont,
                               neg_mask, n, tab, xy + (size_t)y * n);
    }
    HK_HIP(hipGetLastError());
    return batch_affine(s, xy, out, pref, k * n);
}

template <class F>
hk_status MsmRun<F>::lincomb(hipStream_t s, const Affine<F>* const* vecs, const void* coeffs_mont, u32 k, u32 n,
                             XYZZ<F>* xy, F* pref, Affine<F>* out) {
    typedef typename ScalarOf<F>::type Fr;
    if (n == 0) return HK_OK;
    if (k == 0 || k > (u32)LINCOMB_MAX) return HK_ERR_ARG;
    LincombVecs<F> lv;
    for (u32 j = 0; j < (u32)LINCOMB_MAX; j++) lv.v[j] = j < k ? vecs[j] : nullptr;
    hipLaunchKernelGGL((k_points_lincomb<Fr, F>), dim3((n + 63) / 64), dim3(64), 0, s, lv, (const Fr*)coeffs_mont, k, n, xy);
    HK_HIP(hipGetLastError());
    return batch_affine(s, xy, out, pref, n);
}

template <class F>
hk_status MsmRun<F>::fixed_base(hipStream_t s, const Affine<F>* base, const void* scalars, int is_mont,
                                u32 n, Affine<F>* table, XYZZ<F>* xy, F* pref, Affine<F>* out, bool build_table) {
    typedef typename ScalarOf<F>::type Fr;
    if (n == 0) return HK_OK;
    if (build_table) hipLaunchKernelGGL((k_fb_table<F>), dim3(FB_WINDOWS * 256 / 64), dim3(64), 0, s, base, table);
    hipLaunchKernelGGL((k_fb_mul<Fr, F>), dim3((n + 63) / 64), dim3(64), 0, s, table, (const Fr*)scalars,
                       is_mont, n, xy);
    HK_HIP(hipGetLastError());
    return batch_affine(s, xy, out, pref, n);
}

}  // namespace hk
