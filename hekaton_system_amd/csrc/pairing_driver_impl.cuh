// pairing_driver_impl.cuh — launch sequence of the multi-pairing kernels (see pairing.cuh).
#pragma once
#include "msm_driver_impl.cuh"

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
    hipLaunchKernelGGL((k_f12_product<P>), dim3(count), dim3(PAIR_TREE_THREADS), sizeof(Fp12<P>) * PAIR_TREE_THREADS, s,
                       (const Fp12<P>*)miller, n, prod);
    HK_DBG(s, "k_f12_product");
    hipLaunchKernelGGL((k_final_exp<P>), dim3(count), dim3(64), sizeof(Fp12<P>) * PAIR_FEXP_WORDS, s,
                       (const Fp12<P>*)prod, count, out);
    HK_DBG(s, "k_final_exp");
    HK_HIP(hipGetLastError());
    return HK_OK;
}

}  // namespace hk
