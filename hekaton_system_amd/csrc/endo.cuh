// endo.cuh — element-wise scalar multiplication out[i] = s_i * P_i over the curve endomorphisms, with the scalar split ON
// THE DEVICE (`scalar_pairing`, distributed-prover/src/pairing_ops.rs:32-39: 11 sweeps per aggregation,
// aggregation.rs:236-242,289-310, N = #subcircuits elements each - one lane per element, latency-bound).
//
// A 254-bit double-and-add chain is 254 dependent doublings per lane.  G1 has phi(x, y) = (BETA x, y) = [lambda] and G2
// has psi = twist . Frobenius . untwist = [q mod r]:  s = sum_j k_j lambda^j  with K = 2 parts of <= 130 bits (G1) or
// K = 4 parts of <= 67 bits (G2), so  s P = sum_j (+-)|k_j| endo^j(P)  runs on ONE shared chain of 131 / 68 doublings
// (Straus).  The split is Babai's nearest-plane step with precomputed reciprocals (gen_tower_params.py emits them from
// the LLL-reduced bases of hekaton_system_amd/endo.py):  t_j = sign_j ((s G_j) >> 256),  k_i = s [i = 0] - sum_j t_j B[j][i],
// evaluated modulo 2^192 - exact for ANY integers t_j because every basis row is a relation sum_i B[j][i] lambda^i = 0
// (mod r); truncating instead of rounding costs at most one bit of length.
#pragma once
#include "pairing.cuh"

namespace hk {

template <int K>
struct EndoSplit {
    u32 g[K][8];          // floor(2^256 |(B^-1)[0][j]|)
    u32 g_neg;            // bit j: (B^-1)[0][j] < 0
    u32 b[K][K][6];       // |B[j][i]|
    u32 b_neg;            // bit j * K + i: B[j][i] < 0
};

#define HK_DEFINE_ENDO(FQP, PRE)                                                                                   \
    inline EndoSplit<2> endo_split_g1(const FQP*) {                                                                \
        EndoSplit<2> e = {PRE##_PHI2_G, PRE##_PHI2_G_NEG, PRE##_PHI2_B, PRE##_PHI2_B_NEG};                          \
        return e;                                                                                                  \
    }                                                                                                              \
    inline EndoSplit<4> endo_split_g2(const FQP*) {                                                                \
        EndoSplit<4> e = {PRE##_PSI4_G, PRE##_PSI4_G_NEG, PRE##_PSI4_B, PRE##_PSI4_B_NEG};                          \
        return e;                                                                                                  \
    }
HK_DEFINE_ENDO(Bn254FqP, HK_BN254_TW)
HK_DEFINE_ENDO(Bls381FqP, HK_BLS12_381_TW)

// the endomorphism and its order, by coordinate field
template <class F> struct EndoOf;
template <class P> struct EndoOf<Fp<P>> {
    static constexpr int K = 2, STEPS = 131;
    static EndoSplit<2> split() { return endo_split_g1((const P*)nullptr); }
    HK_HD static Affine<Fp<P>> apply(const Affine<Fp<P>>& p) {
        Fp<P> beta;
        HK_UNROLL for (int k = 0; k < P::N; k++) beta.v[k] = TowerParams<P>::BETA[k];
        Affine<Fp<P>> r;
        r.x = Fp<P>::mul(p.x, beta);
        r.y = p.y;
        return r;
    }
};
template <class P> struct EndoOf<Fp2<P>> {
    static constexpr int K = 4, STEPS = 68;
    static EndoSplit<4> split() { return endo_split_g2((const P*)nullptr); }
    HK_HD static Affine<Fp2<P>> apply(const Affine<Fp2<P>>& p) { return g2_psi(p); }
};

// mag[i] = |k_i| (6 limbs), bit i of the result = k_i < 0.   c: canonical scalar, 8 limbs.
template <int K>
HK_HD u32 endo_decompose(const u32 (&c)[8], const EndoSplit<K>& E, u32 (&mag)[K][6]) {
    u32 t[K][6];
    for (int j = 0; j < K; j++) {
        // (c * g_j) >> 256, low 6 limbs: schoolbook with a 64-bit column accumulator (+ overflow word)
        u64 acc = 0;
        u32 over = 0;
        for (int col = 0; col < 14; col++) {
            int lo = col > 7 ? col - 7 : 0, hi = col < 7 ? col : 7;
            for (int a = lo; a <= hi; a++) {
                u64 p = (u64)c[a] * E.g[j][col - a];
                acc += p;
                if (acc < p) over++;
            }
            if (col >= 8) t[j][col - 8] = (u32)acc;
            acc = (acc >> 32) | ((u64)over << 32);
            over = 0;
        }
    }
    u32 neg = 0;
    for (int i = 0; i < K; i++) {
        u32 k[6];
        for (int l = 0; l < 6; l++) k[l] = i == 0 ? c[l] : 0u;
        for (int j = 0; j < K; j++) {
            // term = t_j * |B[j][i]| mod 2^192
            u32 term[6];
            u64 acc = 0;
            u32 over = 0;
            for (int col = 0; col < 6; col++) {
                for (int a = 0; a <= col; a++) {
                    u64 p = (u64)t[j][a] * E.b[j][i][col - a];
                    acc += p;
                    if (acc < p) over++;
                }
                term[col] = (u32)acc;
                acc = (acc >> 32) | ((u64)over << 32);
                over = 0;
            }
            bool minus = (((E.g_neg >> j) ^ (E.b_neg >> (j * K + i))) & 1u) == 0;     // k -= t_j B[j][i]
            u64 carry = minus ? 1 : 0;
            for (int l = 0; l < 6; l++) {
                carry += (u64)k[l] + (minus ? ~term[l] : term[l]);
                k[l] = (u32)carry;
                carry >>= 32;
            }
        }
        bool n_ = k[5] >> 31;
        if (n_) {
            u64 carry = 1;
            for (int l = 0; l < 6; l++) { carry += (u64)(~k[l]); k[l] = (u32)carry; carry >>= 32; }
            neg |= 1u << i;
        }
        for (int l = 0; l < 6; l++) mag[i][l] = k[l];
    }
    return neg;
}

#if defined(__HIPCC__)
// tab: 2^K x n XYZZ scratch.  The lanes of a wavefront hold DIFFERENT scalars, so a "madd when bit j is set" executes for
// the whole wave whenever any lane has the bit: with four images that was four mixed adds per step for every lane
// (68 x (dbl + 4 madd): 7.5 ms in G2).  Each lane therefore first tabulates the 2^K - 1 subset sums of its images
// (tab[mask][i] = sum of +-endo^j(P_i) over the bits of mask) and the chain adds ONE table entry per step, picked by the
// column of the K sub-scalars: 68 x (dbl + add).
template <class Fr, class F>
__global__ void __launch_bounds__(64)
k_scalar_mul_endo(const Affine<F>* __restrict__ pts, const Fr* __restrict__ scalars, u32 n,
                  EndoSplit<EndoOf<F>::K> E, XYZZ<F>* __restrict__ tab, XYZZ<F>* __restrict__ out) {
    constexpr int K = EndoOf<F>::K;
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr s = Fr::from_mont(ld_vec(&scalars[i]));
    u32 c[8];
    HK_UNROLL for (int k = 0; k < 8; k++) c[k] = s.v[k];
    u32 mag[K][6];
    u32 neg = endo_decompose<K>(c, E, mag);
    Affine<F> q = ld_vec(&pts[i]);
    if (q.is_inf()) { st_vec(&out[i], XYZZ<F>::inf()); return; }
    HK_NOUNROLL for (int j = 0; j < K; j++) {
        if (j) q = EndoOf<F>::apply(q);
        Affine<F> w = q;
        if ((neg >> j) & 1u) w.y = F::neg(w.y);
        // every mask whose TOP bit is j: the entry without that bit (already there) plus this image
        st_vec(&tab[((size_t)1 << j) * n + i], XYZZ<F>::from_affine(w));
        HK_NOUNROLL for (u32 m = 1; m < (1u << j); m++)
            st_vec(&tab[(((size_t)1 << j) | m) * n + i], ec_madd_ni(ld_vec(&tab[(size_t)m * n + i]), w));
    }
    XYZZ<F> acc = XYZZ<F>::inf();
    HK_NOUNROLL for (int b = EndoOf<F>::STEPS - 1; b >= 0; b--) {
        acc = ec_dbl_ni(acc);
        u32 m = 0;
        HK_UNROLL for (int j = 0; j < K; j++) m |= ((mag[j][b >> 5] >> (b & 31)) & 1u) << j;
        if (m) acc = ec_add_ni(acc, ld_vec(&tab[(size_t)m * n + i]));
    }
    st_vec(&out[i], acc);
}

// out[i] = lo[i] + c * hi[i] with ONE scalar c for the whole vector, already split on the host (hekaton_system_amd/endo.py)
// into K magnitudes and a sign mask: the fold of a TIPA round (hk_points_fold_g1 / _g2).  Same table-driven chain as
// k_scalar_mul_endo - the endomorphism images and their subset sums are built per element inside this launch (no
// separate image kernel, no image vectors in HBM), one table add per step - then + lo.
// coeffs: K Montgomery Fr (the magnitudes); steps: bit length of the longest magnitude.
template <class Fr, class F>
__global__ void __launch_bounds__(64)
k_points_fold_endo(const Affine<F>* __restrict__ lo, const Affine<F>* __restrict__ hi, const Fr* __restrict__ coeffs, u32 neg,
                   u32 n, XYZZ<F>* __restrict__ tab, XYZZ<F>* __restrict__ out) {
    constexpr int K = EndoOf<F>::K;
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr mag[K];
    int top = -1;
    HK_UNROLL for (int j = 0; j < K; j++) {
        mag[j] = Fr::from_mont(ld_vec(&coeffs[j]));
        for (int b = Fr::N * 32 - 1; b > top; b--)
            if ((mag[j].v[b >> 5] >> (b & 31)) & 1u) { top = b; break; }
    }
    Affine<F> l = ld_vec(&lo[i]);
    Affine<F> q = ld_vec(&hi[i]);
    XYZZ<F> acc = XYZZ<F>::inf();
    if (!q.is_inf()) {
        HK_NOUNROLL for (int j = 0; j < K; j++) {
            if (j) q = EndoOf<F>::apply(q);
            Affine<F> w = q;
            if ((neg >> j) & 1u) w.y = F::neg(w.y);
            st_vec(&tab[((size_t)1 << j) * n + i], XYZZ<F>::from_affine(w));
            HK_NOUNROLL for (u32 m = 1; m < (1u << j); m++)
                st_vec(&tab[(((size_t)1 << j) | m) * n + i], ec_madd_ni(ld_vec(&tab[(size_t)m * n + i]), w));
        }
        HK_NOUNROLL for (int b = top; b >= 0; b--) {
            acc = ec_dbl_ni(acc);
            u32 m = 0;
            HK_UNROLL for (int j = 0; j < K; j++) m |= ((mag[j].v[b >> 5] >> (b & 31)) & 1u) << j;
            if (m) acc = ec_add_ni(acc, ld_vec(&tab[(size_t)m * n + i]));
        }
    }
    st_vec(&out[i], ec_madd_ni(acc, l));
}
#endif

}  // namespace hk
