// pairing_driver_impl.cuh — launch sequence of the multi-pairing kernels (see pairing.cuh).
#pragma once
#include "msm_driver_impl.cuh"
#include "pairing_wave.cuh"

namespace hk {

template <class P> struct PairLoopOf;
template <> struct PairLoopOf<Bn254FqP> { static PairLoop get() { return pair_loop_bn254(); } };
template <> struct PairLoopOf<Bls381FqP> { static PairLoop get() { return pair_loop_bls381(); } };

template <class P>
hk_status PairRun<P>::run(hipStream_t s, const Affine<Fp<P>>* g1, const Affine<Fp2<P>>* g2, u32 n, u32 n_l, u32 n_r,
                          Fp12<P>* miller, Fp12<P>* prod, Fp12<P>* out) {
    u32 count = n_l * n_r;
    size_t total = (size_t)n * count;
    if (count == 0 || n == 0) return HK_ERR_ARG;
    PairLoop loop = PairLoopOf<P>::get();
    hipLaunchKernelGGL((k_pair_miller<P>), dim3((u32)((total + 63) / 64)), dim3(64), 0, s, g1, g2, n, n_l, n_r, loop, miller);
    HK_DBG(s, "k_pair_miller");
    static const bool serial = getenv("HK_PAIR_SERIAL") != nullptr;      // the one-lane-per-product tail (A/B, debugging)
    if (serial) {
        hipLaunchKernelGGL((k_f12_product<P>), dim3(count), dim3(PAIR_TREE_THREADS), sizeof(Fp12<P>) * PAIR_TREE_THREADS, s,
                           (const Fp12<P>*)miller, n, prod);
        HK_DBG(s, "k_f12_product");
        hipLaunchKernelGGL((k_final_exp<P>), dim3(count), dim3(64), sizeof(Fp12<P>) * PAIR_FEXP_WORDS, s,
                           (const Fp12<P>*)prod, count, out);
        HK_DBG(s, "k_final_exp");
    } else {
        // wave-parallel tail: tree levels of 16 (one wave per group), then one wave per product finishes the product
        // and runs the final exponentiation.  The two halves of `miller` (sized 2 x n x count) are the ping-pong
        // buffers of the tree.
        size_t lds_tree = sizeof(WaveArea<P>) + 2 * WV_SLOT * sizeof(Fp<P>);
        size_t lds_fin = sizeof(WaveArea<P>) + WV_FINISH_SLOTS * WV_SLOT * sizeof(Fp<P>);
        const Fp12<P>* cur = miller;
        u32 m = n;
        while (m > 32) {
            u32 groups = (m + 15) / 16;
            Fp12<P>* dst = miller + (cur == miller ? (size_t)n * count : 0);
            hipLaunchKernelGGL((k_pair_tree<P>), dim3(groups, count), dim3(64), lds_tree, s, cur, m, 16u, dst);
            HK_DBG(s, "k_pair_tree");
            cur = dst;
            m = groups;
        }
        hipLaunchKernelGGL((k_pair_finish<P>), dim3(count), dim3(64), lds_fin, s, cur, m, out);
        HK_DBG(s, "k_pair_finish");
    }
    HK_HIP(hipGetLastError());
    return HK_OK;
}

}  // namespace hk
