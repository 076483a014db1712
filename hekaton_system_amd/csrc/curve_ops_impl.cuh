// curve_ops_impl.cuh — per-curve host orchestration (templated on the curve traits); included by
// hk_<curve>_ops.hip.  Heavy kernels are instantiated in their own translation units
// (hk_<curve>_{g1,g2,fr}.hip) and referenced here through extern templates.
#pragma once
#include "msm_driver.cuh"
#include "endo.cuh"

namespace hk {

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

inline size_t msm_sort_bytes(const MsmPlan& p) {
    return al256(4ull * p.NB) * 2 + al256(4ull * (p.NB + 1)) + al256(4ull * ((size_t)p.n * p.W + 1)) +
           al256(2ull * ((size_t)p.n * p.W + 8)) + 1024;
}
template <class F>
inline size_t msm_run_bytes(const MsmPlan& p0) {
    MsmPlan p = p0;
    msm_set_lanes(p, 4u * 65536u);             // upper bound over the per-flavour lane schedules
    size_t n0 = 2ull * p.T[0], n1 = p.n_levels > 1 ? 2ull * p.T[1] : 2;
    return al256(sizeof(XYZZ<F>) * (p.NB + 1)) + al256(4 * n0) + al256(sizeof(XYZZ<F>) * n0) + al256(4 * n1) +
           al256(sizeof(XYZZ<F>) * n1) + al256(sizeof(XYZZ<F>) * p.WP * (p.B / p.K)) +
           al256(sizeof(XYZZ<F>) * p.WP) + 2048;
}

// window size for an MSM over caller-supplied bases (no shift tables: all W windows keep their own
// buckets, W * 2^(c-1) counters must fit the 128 KiB LDS histogram)
inline u32 msm_pick_c_plain(size_t n, u32 fr_bits) {
    u32 best = 4;
    double best_cost = 1e300;
    for (u32 c = 3; c <= 12; c++) {
        u32 W = (fr_bits + 2 + c - 1) / c;
        u64 NB = (u64)W << (c - 1);
        if (NB > (u64)MSM_LDS_COUNTERS) continue;
        double cost = (double)n * W + 4.0 * (double)NB;
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}
// window size when shift tables make every window share one bucket set (WP = 1)
inline u32 msm_pick_c_tables(size_t n, u32 fr_bits) {
    u32 best = 6;
    double best_cost = 1e300;
    for (u32 c = 5; c <= 16; c++) {
        u32 W = (fr_bits + 2 + c - 1) / c;
        double cost = (double)n * W + 6.0 * (double)(1u << (c - 1));
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

template <class C>
struct Ops {
    typedef typename C::Fr Fr;
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;

    template <class F>
    static hk_status msm_plain(hk_ctx* ctx, const void* bases, size_t n_bases, const void* scalars,
                               size_t n_scalars, int mont, int checked, void* out) {
        if (checked && n_bases != n_scalars) return HK_ERR_LEN;     // ark `msm`: Err(min_len)
        size_t n = n_bases < n_scalars ? n_bases : n_scalars;      // ark `msm_unchecked`: zip
        if (n == 0) { memset(out, 0, sizeof(Affine<F>)); return HK_OK; }
        if (!bases || !scalars) return HK_ERR_ARG;
        if (n >= (1u << 26)) return HK_ERR_ARG;
        LaneGuard g(ctx);
        Lane* L = g.lane;
        if (!L) return HK_ERR_DEVICE;
        if (n * EndoOf<F>::K <= SPLIT_MAX_LANES && !getenv("HK_MSM_NO_SMALL")) {
            // short vector, no tables: a Pippenger pass would end in ~254 serial doublings (3 / 8.5 ms whatever n)
            size_t need = al256(n * sizeof(Fr)) + al256(n * sizeof(Affine<F>)) + al256(n * sizeof(XYZZ<F>)) +
                          al256(endo_tab_bytes<F>(n)) + al256(sizeof(XYZZ<F>)) + al256(sizeof(Affine<F>)) + 4096;
            HK_TRY(L->reserve(need));
            const void *sc_d, *b_d;
            HK_TRY(to_device(L, scalars, n * sizeof(Fr), &sc_d));
            HK_TRY(to_device(L, bases, n * sizeof(Affine<F>), &b_d));
            XYZZ<F>* xy = L->alloc_n<XYZZ<F>>(n);
            XYZZ<F>* tab = (XYZZ<F>*)L->alloc_n<unsigned char>(endo_tab_bytes<F>(n));
            XYZZ<F>* res = L->alloc_n<XYZZ<F>>(1);
            Affine<F>* aff = L->alloc_n<Affine<F>>(1);
            if (!xy || !tab || !res || !aff) return HK_ERR_NOMEM;
            HK_TRY(MsmRun<F>::small_msm(L->stream, (const Affine<F>*)b_d, sc_d, mont, (u32)n, tab, xy, res));
            HK_TRY(MsmRun<F>::to_affine(L->stream, res, aff, 1));
            HK_HIP(hipMemcpyAsync(out, aff, sizeof(Affine<F>), hipMemcpyDeviceToHost, L->stream));
            HK_HIP(hipStreamSynchronize(L->stream));
            return HK_OK;
        }
        u32 c = msm_pick_c_plain(n, C::FR_BITS);
        MsmPlan p = msm_make_plan((u32)n, C::FR_BITS, c, 0xffffffffu, ctx->max_lanes0, C::Fr::Params::MOD, C::Fr::Params::N);
        size_t need = al256(n * sizeof(Fr)) + al256(n * sizeof(Affine<F>)) + msm_sort_bytes(p) +
                      msm_run_bytes<F>(p) + al256(sizeof(XYZZ<F>)) + al256(sizeof(Affine<F>)) + 4096;
        HK_TRY(L->reserve(need));
        const void *sc_d, *b_d;
        HK_TRY(to_device(L, scalars, n * sizeof(Fr), &sc_d));
        HK_TRY(to_device(L, bases, n * sizeof(Affine<F>), &b_d));
        SortBufs sb;
        HK_TRY(MsmSort<Fr>::alloc(L, p, &sb));
        typename MsmRun<F>::Bufs rb;
        HK_TRY(MsmRun<F>::alloc(L, p, &rb));
        XYZZ<F>* res = L->alloc_n<XYZZ<F>>(1);
        Affine<F>* aff = L->alloc_n<Affine<F>>(1);
        if (!res || !aff) return HK_ERR_NOMEM;
        HK_TRY(MsmSort<Fr>::run(L->stream, p, (const u32*)sc_d, mont, sb));
        HK_TRY(MsmRun<F>::run(L->stream, p, (const Affine<F>*)b_d, (u32)n, 0, sb, rb, res, nullptr, nullptr));
        HK_TRY(MsmRun<F>::to_affine(L->stream, res, aff, 1));
        HK_HIP(hipMemcpyAsync(out, aff, sizeof(Affine<F>), hipMemcpyDeviceToHost, L->stream));
        HK_HIP(hipStreamSynchronize(L->stream));
        return HK_OK;
    }

    static hk_status msm(hk_ctx* ctx, int group, const void* bases, size_t n_bases, const void* scalars,
                         size_t n_scalars, int mont, int checked, void* out) {
        if (group == 1) return msm_plain<Fq>(ctx, bases, n_bases, scalars, n_scalars, mont, checked, out);
        return msm_plain<Fq2>(ctx, bases, n_bases, scalars, n_scalars, mont, checked, out);
    }

    // ---- filled in by later includes (ntt / qap / prove) -------------------------------------------
    static hk_status ntt(hk_ctx*, void*, unsigned, int, int);
    static hk_status witness_map(hk_ctx*, const hk_csr*, const hk_csr*, const hk_csr*, size_t, size_t,
                                 const void*, size_t, void*, size_t, size_t*);
    static hk_status pk_upload(hk_ctx*, const hk_pk_desc*, hk_pk**);
    static void pk_free(hk_pk*);
    static hk_status commit(hk_ctx*, const hk_pk*, size_t, const void*, size_t, const void*, void*);
    static hk_status commit_batch(hk_ctx*, const hk_pk*, size_t, const void*, size_t, const void*, size_t, void*);
    static hk_status prove(hk_ctx*, const hk_pk*, const void*, size_t, const void*, const void*,
                           const void*, size_t, void*, void*, void*);
    static void ctx_release(hk_ctx*);
    static hk_status fixed_base(hk_ctx*, int, const void*, const void*, size_t, int, void*);
    static hk_status scalar_pairing(hk_ctx*, int, const void*, const void*, size_t, void*);
    static hk_status field_convert(hk_ctx*, int, const void*, void*, size_t, int);
    static hk_status bases_upload(hk_ctx*, int, const void*, size_t, hk_bases**);
    static void bases_free(hk_bases*);
    static hk_status msm_bases(hk_ctx*, const hk_bases*, const void*, size_t, int, int, void*);
    static hk_status pairing_products(hk_ctx*, const void* const*, size_t, const void* const*, size_t, size_t, void*);
    static hk_status points_lincomb(hk_ctx*, int, const void* const*, const void*, size_t, size_t, void*);
    template <class F>
    static hk_status points_fold(hk_ctx*, size_t, const void* const*, const void* const*, const void*, unsigned, size_t, void* const*);
    static hk_status assignment_scatter(hk_ctx*, const uint32_t*, const void*, size_t, size_t, size_t, void*);
    static hk_status pairing_pairs(hk_ctx*, const void* const*, size_t, const void* const*, size_t, const uint32_t*, const uint32_t*, size_t,
                                   size_t, void*);
    static hk_status points_fold_many(hk_ctx*, int, size_t, const void* const*, const void* const*, const void*, unsigned, size_t, void* const*);
    static hk_status points_fold_g2(hk_ctx*, const void*, const void*, const void*, unsigned, size_t, void*);
    static hk_status points_fold_g1(hk_ctx*, const void*, const void*, const void*, unsigned, size_t, void*);
    static hk_status assignment_from_bits(hk_ctx*, const void*, size_t, const uint32_t*, const void*, size_t, void*);
    static hk_status wprog_upload(hk_ctx*, const uint32_t*, size_t, const uint32_t*, size_t, const uint32_t*, size_t, size_t,
                                  size_t, hk_wprog**);
    static void wprog_free(hk_wprog*);
    static hk_status gt_pow(hk_ctx*, const void*, const void*, size_t, void*, int, size_t);
    static hk_status wprog_run(hk_ctx*, const hk_wprog*, const uint32_t*, size_t, const uint32_t*, const void*, size_t, void*);

    static size_t max_private_bytes() {
        size_t m = MsmRun<Fq>::max_private_bytes();
        size_t b = MsmRun<Fq2>::max_private_bytes();
        if (b > m) m = b;
        b = PairRun<typename Fq::Params>::max_private_bytes();
        if (b > m) m = b;
        b = finish_private_bytes();
        return b > m ? b : m;
    }
    static size_t finish_private_bytes();      // prove_impl.cuh (k_finish)
    static hk_status poseidon_path(hk_ctx*, const void*, size_t, const hk_poseidon_desc*, const hk_poseidon_desc*, const void*,
                                   const void*, const uint32_t*, size_t, size_t, size_t, size_t, void*);

    static const CurveOps* table() {
        static const CurveOps t = {sizeof(Fr), sizeof(Fq), sizeof(Affine<Fq>), sizeof(Affine<Fq2>),
                                   &msm, &ntt, &witness_map, &pk_upload, &pk_free, &commit, &prove,
                                   &ctx_release, &fixed_base, &scalar_pairing, &field_convert, &bases_upload,
                                   &bases_free, &msm_bases, &pairing_products,
                                   sizeof(Fp12<typename Fq::Params>), &points_lincomb, &points_fold_g2, &points_fold_g1, &assignment_from_bits, &wprog_upload, &wprog_free, &wprog_run, &gt_pow,
                                   &max_private_bytes, &poseidon_path, &points_fold_many, &pairing_pairs, &assignment_scatter, &commit_batch};
        return &t;
    }
};

}  // namespace hk
