// pairing_driver_impl.cuh — launch sequence of the multi-pairing kernels (see pairing.cuh).
#pragma once
#include "msm_driver_impl.cuh"
#include "pairing_wave.cuh"

namespace hk {

template <class P> struct PairLoopOf;
template <> struct PairLoopOf<Bn254FqP> { static PairLoop get() { return pair_loop_bn254(); } };
template <> struct PairLoopOf<Bls381FqP> { static PairLoop get() { return pair_loop_bls381(); } };

template <class P> struct ScalarOfQ;
template <> struct ScalarOfQ<Bn254FqP> { typedef Fp<Bn254FrP> type; };
template <> struct ScalarOfQ<Bls381FqP> { typedef Fp<Bls381FrP> type; };

template <class P>
hk_status PairRun<P>::gt_pow(hipStream_t s, const Fp12<P>* in, const void* scalars_mont, u32 n, Fp12<P>* out, bool in_gt) {
    typedef typename ScalarOfQ<P>::type Fr;
    if (n == 0) return HK_OK;
    const bool plain = !in_gt || getenv("HK_GT_POW_PLAIN") != nullptr;   // the 254-step chain: any Fq12 element (or A/B)
    if (plain) {
        size_t lds = sizeof(WaveArea<P>) + 2 * WV_SLOT * sizeof(Fp<P>);
        hipLaunchKernelGGL((k_gt_pow<P, Fr>), dim3(n), dim3(64), lds, s, in, (const Fr*)scalars_mont, n, out);
    } else {
        static const EndoSplit<4> E = endo_split_g2((const P*)nullptr);
        size_t lds = sizeof(WaveArea<P>) + 16 * WV_SLOT * sizeof(Fp<P>);
        hipLaunchKernelGGL((k_gt_pow_endo<P, Fr>), dim3(n), dim3(64), lds, s, in, (const Fr*)scalars_mont, n, E, out);
    }
    HK_DBG(s, "k_gt_pow");
    HK_HIP(hipGetLastError());
    return HK_OK;
}

// out[g] = prod_{j < len} in[g * len + j]: one wave per group on the wave multiplier (the tree kernel with one group per row)
template <class P>
hk_status PairRun<P>::gt_prod(hipStream_t s, const Fp12<P>* in, u32 len, u32 groups, Fp12<P>* out) {
    if (groups == 0 || len == 0) return HK_OK;
    size_t lds_tree = sizeof(WaveArea<P>) + 2 * WV_SLOT * sizeof(Fp<P>);
    hipLaunchKernelGGL((k_pair_tree<P>), dim3(1, groups), dim3(64), lds_tree, s, in, len, len, out);
    HK_DBG(s, "k_pair_tree (gt_prod)");
    HK_HIP(hipGetLastError());
    return HK_OK;
}

template <class P>
size_t PairRun<P>::max_private_bytes() {
    const void* ks[] = {(const void*)k_pair_lines<Fp2<P>>, (const void*)k_pair_lines<Fp2Q<P>>, (const void*)k_pair_tree_lines<P>, (const void*)k_pair_tree<P>,
                        (const void*)k_pair_horner<P>,
                        (const void*)k_gt_pow<P, typename ScalarOfQ<P>::type>, (const void*)k_gt_pow_endo<P, typename ScalarOfQ<P>::type>};
    const void* serial[] = {(const void*)k_pair_miller<P>, (const void*)k_f12_product<P>, (const void*)k_final_exp<P>};
    size_t m = 0;
    for (const void* k : ks) { size_t b = hk_private_bytes_of(k); if (b > m) m = b; }
    if (getenv("HK_PAIR_SERIAL"))            // the one-lane-per-pair debugging path carries the deepest frames (5 KB)
        for (const void* k : serial) { size_t b = hk_private_bytes_of(k); if (b > m) m = b; }
    return m;
}

template <class P>
u32 PairRun<P>::steps() {
    return (u32)pair_steps(PairLoopOf<P>::get(), TowerParams<P>::TWIST_IS_D).n;
}

template <class P>
size_t PairRun<P>::scratch_bytes(u32 n, u32 count, u32 n_r) {
    PairSteps st = pair_steps(PairLoopOf<P>::get(), TowerParams<P>::TWIST_IS_D);
    size_t S = (size_t)st.n, g0 = (n + 15) / 16;
    size_t lines = ((size_t)n_r * S * n * sizeof(Line6<P>) + 255) & ~(size_t)255;
    size_t pipeline = lines + 2 * count * S * g0 * sizeof(Fp12<P>);
    size_t serial = (size_t)n * count * sizeof(Fp12<P>);
    return (pipeline > serial ? pipeline : serial) + 4096;
}

template <class P>
hk_status PairRun<P>::run(hipStream_t s, const Affine<Fp<P>>* g1, const Affine<Fp2<P>>* g2, u32 n, u32 n_l, u32 n_r,
                          Fp12<P>* miller, Fp12<P>* prod, Fp12<P>* out, const PairList* pairs) {
    u32 count = pairs ? pairs->n : n_l * n_r;
    PairList pl;
    pl.n = 0;
    if (pairs) {
        if (pairs->n == 0 || pairs->n > (u32)PAIR_LIST_MAX) return HK_ERR_ARG;
        for (u32 k = 0; k < pairs->n; k++) if (pairs->a[k] >= n_l || pairs->b[k] >= n_r) return HK_ERR_ARG;
        pl = *pairs;
    }
    size_t total = (size_t)n * count;
    if (count == 0 || n == 0) return HK_ERR_ARG;
    PairLoop loop = PairLoopOf<P>::get();
    static const bool serial = getenv("HK_PAIR_SERIAL") != nullptr;      // one lane per pair / per product (A/B, debugging)
    size_t lds_tree = sizeof(WaveArea<P>) + 2 * WV_SLOT * sizeof(Fp<P>);
    size_t lds_fin = sizeof(WaveArea<P>) + WV_FINISH_SLOTS * WV_SLOT * sizeof(Fp<P>);
    if (serial && !pairs) {
        hipLaunchKernelGGL((k_pair_miller<P>), dim3((u32)((total + 63) / 64)), dim3(64), 0, s, g1, g2, n, n_l, n_r, loop, miller);
        HK_DBG(s, "k_pair_miller");
        hipLaunchKernelGGL((k_f12_product<P>), dim3(count), dim3(PAIR_TREE_THREADS), sizeof(Fp12<P>) * PAIR_TREE_THREADS, s,
                           (const Fp12<P>*)miller, n, prod);
        HK_DBG(s, "k_f12_product");
        hipLaunchKernelGGL((k_final_exp<P>), dim3(count), dim3(64), sizeof(Fp12<P>) * PAIR_FEXP_WORDS, s,
                           (const Fp12<P>*)prod, count, out);
        HK_DBG(s, "k_final_exp");
    } else {
        // pipeline (pairing_wave.cuh): lines -> per-step product trees (wave multiplier) -> Horner + final exponentiation
        PairSteps st = pair_steps(loop, TowerParams<P>::TWIST_IS_D);
        u32 S = (u32)st.n;
        Line6<P>* lines = reinterpret_cast<Line6<P>*>(miller);
        // buffers behind the lines: two ping-pong arrays of at most count * S * ceil(n / 16) Fq12
        size_t lines_bytes = ((size_t)n_r * S * n * sizeof(Line6<P>) + 255) & ~(size_t)255;
        u32 g0 = (n + 15) / 16;
        Fp12<P>* pp[2];
        pp[0] = reinterpret_cast<Fp12<P>*>(reinterpret_cast<char*>(miller) + lines_bytes);
        pp[1] = pp[0] + (size_t)count * S * g0;
        // few points: a quad of lanes per point (4 n n_r lanes still at most one wave per SIMD)
        const bool quad = (size_t)n * n_r * 4 <= SPLIT_MAX_LANES && getenv("HK_ENDO_NO_QUAD") == nullptr;
        if (quad)
            hipLaunchKernelGGL((k_pair_lines<Fp2Q<P>>), dim3((4 * n + 63) / 64, n_r), dim3(64), 0, s,
                               (const Affine<Fp2Q<P>>*)g2, n, n_r, loop, S, lines);
        else
            hipLaunchKernelGGL((k_pair_lines<Fp2<P>>), dim3((n + 63) / 64, n_r), dim3(64), 0, s, g2, n, n_r, loop, S, lines);
        HK_DBG(s, "k_pair_lines");
        hipLaunchKernelGGL((k_pair_tree_lines<P>), dim3(g0, count * S), dim3(64), lds_tree, s, (const Line6<P>*)lines, g1, n, 16u, n_r, S, pl, pp[0]);
        HK_DBG(s, "k_pair_tree_lines");
        u32 m = g0;
        int cur = 0;
        while (m > 1) {
            u32 groups = (m + 15) / 16;
            hipLaunchKernelGGL((k_pair_tree<P>), dim3(groups, count * S), dim3(64), lds_tree, s, (const Fp12<P>*)pp[cur], m, 16u,
                               pp[cur ^ 1]);
            HK_DBG(s, "k_pair_tree");
            cur ^= 1;
            m = groups;
        }
        hipLaunchKernelGGL((k_pair_horner<P>), dim3(count), dim3(64), lds_fin, s, (const Fp12<P>*)pp[cur], st, out);
        HK_DBG(s, "k_pair_horner");
    }
    HK_HIP(hipGetLastError());
    return HK_OK;
}

}  // namespace hk
