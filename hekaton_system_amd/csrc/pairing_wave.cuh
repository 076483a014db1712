// pairing_wave.cuh — WAVE-PARALLEL Fq12 arithmetic: one 64-lane wavefront multiplies two Fq12 values.
//
// A pairing's critical path (Miller accumulation, final exponentiation) is a chain of ~500 dependent Fq12 products;
// on one lane each costs 54 field products back to back (~35 us).  Here the three-level Karatsuba of the tower is laid
// across the wave instead: lane l < 54 forms its two operands (sums of <= 8 coefficients, PRE tables), does ONE field
// product, and the 12 output coefficients are integer combinations of the 54 products (POST tables: multipliers
// 1, 8, 9, 10 on BN254, 1, 2 on BLS12-381, evaluated Horner-wise over their bits; each output split over 4 lanes).
// Values live in LDS as 13 Fq (12 coefficients, ark order, + a zero the operand tables pad with), lazily reduced;
// a workgroup IS one wave, so the barriers between the phases are wave-local.
// Tables: gen_tower_params.py (derived symbolically from the tower formulas and self-checked numerically).
#pragma once
#include "pairing.cuh"
#include "hk_wave_f12.h"

namespace hk {

#if defined(__HIPCC__)

template <class P> struct WaveTab;
#define HK_DEFINE_WAVETAB(FQP, PRE)                                                                      \
    static __device__ const unsigned char PRE##_pre_cnt[54] = PRE##_PRE_CNT;                             \
    static __device__ const unsigned char PRE##_pre_idx[54][8] = PRE##_PRE_IDX;                          \
    static __device__ const unsigned char PRE##_postf[48][PRE##_POSTF_LEN] = PRE##_POSTF;                \
    static __device__ const u32 PRE##_frob[3][6][2][FQP::N] = PRE##_FROB;                                \
    static __device__ const unsigned char PRE##_cyc_pre_idx[27][4] = PRE##_CYC_PRE_IDX;                  \
    static __device__ const unsigned char PRE##_cyc_postf[48][PRE##_CYC_POSTF_LEN] = PRE##_CYC_POSTF;    \
    template <> struct WaveTab<FQP> {                                                                    \
        static constexpr int NLEVELS = PRE##_NLEVELS;                                                    \
        static constexpr int POSTF_LEN = PRE##_POSTF_LEN;                                                \
        static constexpr int n_add(int l) { constexpr int t[PRE##_NLEVELS] = PRE##_POSTF_ADD; return t[l]; } \
        static constexpr int n_sub(int l) { constexpr int t[PRE##_NLEVELS] = PRE##_POSTF_SUB; return t[l]; } \
        static __device__ __forceinline__ u32 pre_cnt(u32 l) { return PRE##_pre_cnt[l]; }               \
        static __device__ __forceinline__ const unsigned char* pre_idx(u32 l) { return PRE##_pre_idx[l]; } \
        static __device__ __forceinline__ const unsigned char* postf(u32 l) { return PRE##_postf[l]; }   \
        static __device__ __forceinline__ const u32* frob(int k, u32 j, u32 c) { return PRE##_frob[k - 1][j][c]; } \
        static constexpr int CYC_NLEVELS = PRE##_CYC_NLEVELS;                                            \
        static constexpr int CYC_POSTF_LEN = PRE##_CYC_POSTF_LEN;                                        \
        static constexpr int cyc_n_add(int l) { constexpr int t[PRE##_CYC_NLEVELS] = PRE##_CYC_POSTF_ADD; return t[l]; } \
        static constexpr int cyc_n_sub(int l) { constexpr int t[PRE##_CYC_NLEVELS] = PRE##_CYC_POSTF_SUB; return t[l]; } \
        static __device__ __forceinline__ const unsigned char* cyc_pre_idx(u32 l) { return PRE##_cyc_pre_idx[l]; } \
        static __device__ __forceinline__ const unsigned char* cyc_postf(u32 l) { return PRE##_cyc_postf[l]; } \
    };
HK_DEFINE_WAVETAB(Bn254FqP, HK_BN254_WV)
HK_DEFINE_WAVETAB(Bls381FqP, HK_BLS12_381_WV)

constexpr int WV_SLOT = 13;            // Fq per value: 12 coefficients + one zero

// LDS work area of one wave.  prod[54..63] stay zero: the fixed-length recombination lists are padded with slot 54.
// The lane's rows of the operand / recombination tables are copied here once per kernel (WaveF12::init).
template <class P>
struct alignas(16) WaveArea {
    Fp<P> prod[64];
    alignas(8) unsigned char pre_idx[54][8];
    alignas(4) unsigned char postf[48][WaveTab<P>::POSTF_LEN];
    alignas(4) unsigned char cyc_pre_idx[28][4];                                   // squaring in the cyclotomic subgroup
    alignas(4) unsigned char cyc_postf[48][WaveTab<P>::CYC_POSTF_LEN];
};

template <class P>
struct WaveF12 {
    typedef Fp<P> Fq;
    typedef WaveTab<P> T;

    static __device__ __forceinline__ void sync() { __syncthreads(); }

    // once per kernel, before the first product: this lane's table rows -> LDS, the padding products -> 0
    static __device__ __forceinline__ void init(WaveArea<P>* w) {
        u32 lane = threadIdx.x;
        if (lane < 54) {
            const unsigned char* ix = T::pre_idx(lane);
            for (int j = 0; j < 8; j++) w->pre_idx[lane][j] = ix[j];
        }
        if (lane < 48) {
            const unsigned char* ps = T::postf(lane);
            for (int j = 0; j < T::POSTF_LEN; j++) w->postf[lane][j] = ps[j];
            const unsigned char* cs = T::cyc_postf(lane);
            for (int j = 0; j < T::CYC_POSTF_LEN; j++) w->cyc_postf[lane][j] = cs[j];
        }
        if (lane < 27) {
            const unsigned char* ci = T::cyc_pre_idx(lane);
            for (int j = 0; j < 4; j++) w->cyc_pre_idx[lane][j] = ci[j];
        }
        if (lane >= 54) w->prod[lane] = Fq::zero();
        sync();
    }

    // sum of a value over the four lanes of a quad, left in all four (two DPP quad_perm exchanges; no LDS, no barrier)
    template <int CTRL>
    static __device__ __forceinline__ Fq quad_perm(const Fq& v) {
        Fq r;
        HK_UNROLL for (int i = 0; i < P::N; i++) r.v[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)v.v[i], CTRL, 0xf, 0xf, true);
        return r;
    }
    static __device__ __forceinline__ Fq quad_sum(const Fq& v) {
        Fq t = Fq::add(v, quad_perm<0xB1>(v));                   // quad_perm [1, 0, 3, 2]
        return Fq::add(t, quad_perm<0x4E>(t));                   // quad_perm [2, 3, 0, 1]
    }

    // dst = a * b   (dst may alias a or b).  Straight-line code: every lane sums 8 (zero-padded) coefficients per
    // operand and, per bit level of the multipliers, a fixed number of products to add and to subtract (padded with a
    // zero product) - the trip counts the SIMD ran anyway, without the ~60 dependent LDS round trips of parsing a list.
    static __device__ __noinline__ void mul(Fq* dst, const Fq* a, const Fq* b, WaveArea<P>* w) {
        u32 lane = threadIdx.x;
        if (lane < 54) {
            const u32* ixw = reinterpret_cast<const u32*>(w->pre_idx[lane]);
            u32 i0 = ixw[0], i1 = ixw[1];
            Fq x = a[i0 & 0xff], y = b[i0 & 0xff];
            HK_UNROLL for (int j = 1; j < 8; j++) {
                u32 ix = ((j < 4 ? i0 : i1) >> (8 * (j & 3))) & 0xff;
                x = Fq::add(x, a[ix]);
                y = Fq::add(y, b[ix]);
            }
            w->prod[lane] = Fq::mul(x, y);
        }
        sync();
        Fq part = Fq::zero();
        if (lane < 48) {
            u32 codes[T::POSTF_LEN / 4];
            const u32* row = reinterpret_cast<const u32*>(w->postf[lane]);
            HK_UNROLL for (int k = 0; k < T::POSTF_LEN / 4; k++) codes[k] = row[k];
            Fq acc = Fq::zero();
            int pos = 0;
            HK_UNROLL for (int lev = 0; lev < T::NLEVELS; lev++) {
                if (lev) acc = Fq::dbl(acc);
                HK_UNROLL for (int e = 0; e < T::n_add(lev); e++, pos++)
                    acc = Fq::add(acc, w->prod[(codes[pos >> 2] >> (8 * (pos & 3))) & 0xff]);
                HK_UNROLL for (int e = 0; e < T::n_sub(lev); e++, pos++)
                    acc = Fq::sub(acc, w->prod[(codes[pos >> 2] >> (8 * (pos & 3))) & 0xff]);
            }
            part = acc;
        }
        // the four partial sums of an output sit on the lanes of one quad: summed by DPP, written by the quad's first lane
        // (no lane reads a or b after the barrier above, so dst may alias them)
        Fq r = quad_sum(part);
        if (lane < 48 && (lane & 3u) == 0) dst[lane >> 2] = r;
        if (lane == 48) dst[12] = Fq::zero();
        sync();
    }
    static __device__ __forceinline__ void sqr(Fq* dst, const Fq* a, WaveArea<P>* w) { mul(dst, a, a, w); }

    // dst = a^2 for a IN THE CYCLOTOMIC SUBGROUP (everything after the easy part of the final exponentiation, every
    // element of GT): Granger-Scott, alpha^2 = (3 a^2 - 2 conj(a)) + (3 s c^2 + 2 conj(b)) t + (3 b^2 - 2 conj(c)) t^2 over
    // Fq4, written with squares only - 27 lanes square the sum of <= 4 coefficients (tables generated and checked against
    // the plain square on a cyclotomic element by gen_tower_params.py: wave_cyc_forms / wave_cyc_check), the recombination
    // has 20 (BN254) / 12 (BLS12-381) entries per lane instead of 27 / 16, and there is ONE operand sum of <= 4 terms
    // instead of two of 8; every output takes -+2 of its own input coefficient (- on the c0 half, + on the c1 half).
    static __device__ __noinline__ void cyc_sqr(Fq* dst, const Fq* a, WaveArea<P>* w) {
        u32 lane = threadIdx.x;
        if (lane < 27) {
            u32 i0 = *reinterpret_cast<const u32*>(w->cyc_pre_idx[lane]);
            Fq x = a[i0 & 0xff];
            HK_UNROLL for (int j = 1; j < 4; j++) x = Fq::add(x, a[(i0 >> (8 * j)) & 0xff]);
            w->prod[lane] = Fq::mul(x, x);
        }
        sync();
        Fq part = Fq::zero();
        if (lane < 48) {
            u32 codes[T::CYC_POSTF_LEN / 4];
            const u32* row = reinterpret_cast<const u32*>(w->cyc_postf[lane]);
            HK_UNROLL for (int k = 0; k < T::CYC_POSTF_LEN / 4; k++) codes[k] = row[k];
            Fq acc = Fq::zero();
            int pos = 0;
            HK_UNROLL for (int lev = 0; lev < T::CYC_NLEVELS; lev++) {
                if (lev) acc = Fq::dbl(acc);
                HK_UNROLL for (int e = 0; e < T::cyc_n_add(lev); e++, pos++)
                    acc = Fq::add(acc, w->prod[(codes[pos >> 2] >> (8 * (pos & 3))) & 0xff]);
                HK_UNROLL for (int e = 0; e < T::cyc_n_sub(lev); e++, pos++)
                    acc = Fq::sub(acc, w->prod[(codes[pos >> 2] >> (8 * (pos & 3))) & 0xff]);
            }
            part = acc;
        }
        Fq r = quad_sum(part);
        if (lane < 48 && (lane & 3u) == 0) {                     // output k = lane / 4: only this lane touches slot k from here on
            u32 k = lane >> 2;
            Fq two_a = Fq::dbl(a[k]);
            dst[k] = k < 6 ? Fq::sub(r, two_a) : Fq::add(r, two_a);
        }
        if (lane == 48) dst[12] = Fq::zero();
        sync();
    }

    static __device__ __forceinline__ void copy(Fq* dst, const Fq* a) {
        u32 lane = threadIdx.x;
        Fq v;
        if (lane < 13) v = a[lane];
        sync();
        if (lane < 13) dst[lane] = v;
        sync();
    }
    static __device__ __forceinline__ void set_one(Fq* dst) {
        u32 lane = threadIdx.x;
        if (lane < 13) dst[lane] = lane == 0 ? Fq::one() : Fq::zero();
        sync();
    }
    // dst = conj(a) = a^(q^6)
    static __device__ __forceinline__ void conj(Fq* dst, const Fq* a) {
        u32 lane = threadIdx.x;
        Fq v;
        if (lane < 13) { v = a[lane]; if (lane >= 6 && lane < 12) v = Fq::neg(v); }
        sync();
        if (lane < 13) dst[lane] = v;
        sync();
    }
    // dst = a^(q^K), K = 1..3: lane 2j + c computes component c of (Fq2 coefficient j, conjugated when K is odd) * const_j
    template <int K>
    static __device__ __noinline__ void frob(Fq* dst, const Fq* a) {
        u32 lane = threadIdx.x;
        Fq r;
        if (lane < 12) {
            u32 j = lane >> 1, c = lane & 1;
            Fq x0 = a[2 * j], x1 = a[2 * j + 1];
            if (K & 1) x1 = Fq::neg(x1);
            Fq k0, k1;
            const u32* p0 = T::frob(K, j, 0);
            const u32* p1 = T::frob(K, j, 1);
            for (int i = 0; i < P::N; i++) { k0.v[i] = p0[i]; k1.v[i] = p1[i]; }
            r = c == 0 ? Fq::sub(Fq::mul(x0, k0), Fq::mul(x1, k1)) : Fq::add(Fq::mul(x0, k1), Fq::mul(x1, k0));
        }
        sync();
        if (lane < 12) dst[lane] = r;
        if (lane == 12) dst[12] = Fq::zero();
        sync();
    }
    // slot <-> tower struct (lane 0 only; used for the one inversion of the final exponentiation)
    static __device__ __forceinline__ Fp12<P> load_tower(const Fq* a) {
        Fp12<P> f;
        f.c0.c0.c0 = a[0]; f.c0.c0.c1 = a[1]; f.c0.c1.c0 = a[2]; f.c0.c1.c1 = a[3]; f.c0.c2.c0 = a[4]; f.c0.c2.c1 = a[5];
        f.c1.c0.c0 = a[6]; f.c1.c0.c1 = a[7]; f.c1.c1.c0 = a[8]; f.c1.c1.c1 = a[9]; f.c1.c2.c0 = a[10]; f.c1.c2.c1 = a[11];
        return f;
    }
    static __device__ __forceinline__ void store_tower(Fq* a, const Fp12<P>& f) {
        a[0] = f.c0.c0.c0; a[1] = f.c0.c0.c1; a[2] = f.c0.c1.c0; a[3] = f.c0.c1.c1; a[4] = f.c0.c2.c0; a[5] = f.c0.c2.c1;
        a[6] = f.c1.c0.c0; a[7] = f.c1.c0.c1; a[8] = f.c1.c1.c0; a[9] = f.c1.c1.c1; a[10] = f.c1.c2.c0; a[11] = f.c1.c2.c1;
        a[12] = Fq::zero();
    }
    // dst = a^-1 = conj(a) * (a * conj(a))^-1.  The norm a * conj(a) lies in Fq6 (its w-part is zero), so the part that
    // runs on ONE lane inverts a 6-coefficient value: the by-value Fq12 temporaries of f12_inv (3.1 KB of private memory
    // per lane on BLS12-381 - the deepest frame of the default path, and the frame sizes the scratch ring of every
    // hardware queue) become two more wave products.  tmp: one scratch slot; dst, a, tmp distinct.
    static __device__ __noinline__ void inv(Fq* dst, const Fq* a, Fq* tmp, WaveArea<P>* w) {
        conj(tmp, a);
        mul(dst, a, tmp, w);                     // the norm: coefficients 0..5
        if (threadIdx.x == 0) {
            Fp6<P> n;
            n.c0.c0 = dst[0]; n.c0.c1 = dst[1]; n.c1.c0 = dst[2]; n.c1.c1 = dst[3]; n.c2.c0 = dst[4]; n.c2.c1 = dst[5];
            Fp6<P> ni = f6_inv(n);
            dst[0] = ni.c0.c0; dst[1] = ni.c0.c1; dst[2] = ni.c1.c0; dst[3] = ni.c1.c1; dst[4] = ni.c2.c0; dst[5] = ni.c2.c1;
            for (int k = 6; k < 13; k++) dst[k] = Fq::zero();
        }
        sync();
        mul(dst, tmp, dst, w);                   // conj(a) * norm^-1
    }
    // dst = a^X (X = the curve parameter); t: one scratch slot.  dst, a, t distinct.
    static __device__ __noinline__ void pow_x(Fq* dst, const Fq* a, WaveArea<P>* w) {
        const u64 x = TowerParams<P>::X;
        copy(dst, a);
        int top = 63;
        while (!((x >> top) & 1)) top--;
        for (int bit = top - 1; bit >= 0; bit--) {                 // only ever called on the cyclotomic subgroup (hard part)
            cyc_sqr(dst, dst, w);
            if ((x >> bit) & 1) mul(dst, dst, a, w);
        }
    }
    static __device__ __forceinline__ void exp_by_x(Fq* dst, const Fq* a, WaveArea<P>* w) {          // bls12 exp_by_x
        pow_x(dst, a, w);
        if (TowerParams<P>::X_IS_NEGATIVE) conj(dst, dst);
    }
    static __device__ __forceinline__ void exp_by_neg_x(Fq* dst, const Fq* a, WaveArea<P>* w) {      // bn exp_by_neg_x
        pow_x(dst, a, w);
        if (!TowerParams<P>::X_IS_NEGATIVE) conj(dst, dst);
    }

    // s: >= 10 slots; s[0] holds f on entry and the result on exit (same chains as pair_final_exp)
    static __device__ __noinline__ void final_exp(Fq* s, WaveArea<P>* w) {
        typedef TowerParams<P> TP;
        auto S = [&](int i) { return s + i * WV_SLOT; };
        Fq *r = S(0), *t = S(7), *u = S(8);
        inv(t, r, u, w);                 // f^-1
        conj(u, r);
        mul(r, u, t, w);                 // f^(q^6 - 1)
        frob<2>(t, r);
        mul(r, t, r, w);                 // r = f^((q^6 - 1)(q^2 + 1))
        if (TP::TWIST_IS_D) {
            Fq *y1 = S(1), *y3 = S(2), *y4 = S(3), *y6 = S(4), *y8 = S(5), *y9 = S(6);
            exp_by_neg_x(t, r, w);       // y0
            sqr(y1, t, w);
            sqr(t, y1, w);               // y2
            mul(y3, t, y1, w);
            exp_by_neg_x(y4, y3, w);
            sqr(t, y4, w);               // y5
            exp_by_neg_x(y6, t, w);
            conj(y3, y3);
            conj(y6, y6);
            mul(t, y6, y4, w);           // y7
            mul(y8, t, y3, w);
            mul(y9, y8, y1, w);
            mul(t, y8, y4, w);           // y10
            mul(t, t, r, w);             // y11
            frob<1>(u, y9);              // y12
            mul(t, u, t, w);             // y13
            frob<2>(y8, y8);
            mul(t, y8, t, w);            // y14
            conj(r, r);
            mul(u, r, y9, w);
            frob<3>(u, u);               // y15
            mul(r, u, t, w);
        } else {
            Fq *y0 = S(1), *y1 = S(2), *y2 = S(3);
            sqr(y0, r, w);
            exp_by_x(y1, r, w);
            conj(y2, r);
            mul(y1, y1, y2, w);
            exp_by_x(y2, y1, w);
            conj(y1, y1);
            mul(y1, y1, y2, w);
            exp_by_x(y2, y1, w);
            frob<1>(y1, y1);
            mul(y1, y1, y2, w);
            mul(r, r, y0, w);
            exp_by_x(y0, y1, w);
            exp_by_x(y2, y0, w);
            frob<2>(y0, y1);
            conj(y1, y1);
            mul(y1, y1, y2, w);
            mul(y1, y1, y0, w);
            mul(r, r, y1, w);
        }
    }
    // global (canonical, 12 Fq) <-> slot
    static __device__ __forceinline__ void load(Fq* dst, const Fp12<P>* g) {
        u32 lane = threadIdx.x;
        const Fq* src = reinterpret_cast<const Fq*>(g);
        if (lane < 12) dst[lane] = ld_vec(&src[lane]);
        if (lane == 12) dst[12] = Fq::zero();
        sync();
    }
    static __device__ __forceinline__ void store(Fp12<P>* g, const Fq* a) {
        u32 lane = threadIdx.x;
        Fq* d = reinterpret_cast<Fq*>(g);
        if (lane < 12) st_vec(&d[lane], a[lane]);        // canonicalises
        sync();
    }
};

constexpr int WV_FINISH_SLOTS = 12;

// ---- the Miller loop as a pipeline -------------------------------------------------------------------------------
// prod_i f_i with f_i = Miller(P_i, Q_i) is regrouped by loop step: f = prod_s L_s^(2^(squarings after s)),
// L_s = prod_i line_{i,s}(P_i).  (1) k_pair_lines: one lane per G2 point walks ark's projective steps ONCE and writes
// every line's coefficients (the step arithmetic is shared by all lhs vectors);
// (2) k_pair_tree_lines / k_pair_tree: the lines evaluated at each lhs vector's G1 points, then L_s by product trees on the
// wave multiplier, all steps and products in parallel;
// (3) k_pair_horner: one wave per product folds the L_s with the loop's squarings and runs the final exponentiation.
struct PairSteps {
    int n;                       // number of line steps S
    unsigned char sq[100];       // sq[s] != 0: the accumulator is squared before line s is multiplied in
};
inline PairSteps pair_steps(const PairLoop& loop, bool is_bn) {
    PairSteps st;
    st.n = 0;
    for (int i = loop.len - 1; i >= 1; i--) {
        st.sq[st.n++] = i != loop.len - 1;
        if (loop.digits[i - 1] != 0) st.sq[st.n++] = 0;
    }
    if (is_bn) { st.sq[st.n++] = 0; st.sq[st.n++] = 0; }
    return st;
}

template <class P> struct Line6 { Fp2<P> c0, c1, c2; };


// RAW lines (the step's coefficients, not yet evaluated at a G1 point): lines[(b * S + s) * n + i]; a G2 point at infinity
// writes all-zero coefficients (no real step does: its c0 / c2 carries 2yz resp. lambda, non-zero on the prime-order
// subgroup), which the tree kernel reads as "this pair contributes 1".  The evaluation at the lhs points (4 Fq products per
// line and lhs vector) is NOT part of this serial chain: with four lhs vectors it was 40 % of it.
// F: the Fq2 representation - Fp2<P> (one lane per G2 point) or, for short vectors, Fp2Q<P> (a quad of lanes per point:
// every Fq2 product of the chain is ONE base-field product per lane, endo.cuh).  grid: (ceil(n L / 64), n_r), L lanes per point.
template <class F>
__global__ void __launch_bounds__(64)
k_pair_lines(const Affine<F>* __restrict__ g2, u32 n, u32 n_r, PairLoop loop, u32 S, Line6<typename F::Params>* __restrict__ lines) {
    typedef typename F::Params P;
    typedef TowerParams<P> T;
    constexpr int L = LanesPerValue<F>::value;
    u32 i = (blockIdx.x * blockDim.x + threadIdx.x) / L, b = blockIdx.y;
    if (i >= n) return;
    Affine<F> q = ld_vec(&g2[(size_t)b * n + i]);
    bool q_inf = q.is_inf();
    G2ProjF<F> r;
    r.x = q.x; r.y = q.y; r.z = F::one();
    Affine<F> nq = q;
    nq.y = F::neg(q.y);
    Affine<F> q1 = q, q2 = q;
    if (T::TWIST_IS_D && !q_inf) {
        q1 = pair_mul_by_char(q);
        q2 = pair_mul_by_char(q1);
        q2.y = F::neg(q2.y);
    }
    u32 s = 0;
    auto emit = [&](const LineCoeffsF<F>& l) {
        F* dst = reinterpret_cast<F*>(&lines[((size_t)b * S + s) * n + i]);          // Line6: three Fq2, same layout
        st_vec(&dst[0], l.c0); st_vec(&dst[1], l.c1); st_vec(&dst[2], l.c2);
        s++;
    };
    LineCoeffsF<F> dummy; dummy.c0 = F::zero(); dummy.c1 = F::zero(); dummy.c2 = F::zero();
    HK_NOUNROLL for (int k = loop.len - 1; k >= 1; k--) {
        emit(q_inf ? dummy : pair_doubling_step(r));
        int d = loop.digits[k - 1];
        if (d == 1) emit(q_inf ? dummy : pair_addition_step(r, q));
        else if (d == -1) emit(q_inf ? dummy : pair_addition_step(r, nq));
    }
    if (T::TWIST_IS_D) {
        emit(q_inf ? dummy : pair_addition_step(r, q1));
        emit(q_inf ? dummy : pair_addition_step(r, q2));
    }
}

// first tree level over SPARSE lines: wave g of group y = (a * n_r + b) * S + s evaluates the raw lines (b, s, g*c ..) at
// the points of lhs vector a - the lanes that hold a scaled coefficient multiply it by the point's x or y, side by side -
// and multiplies them into one full Fq12 (product p = blockIdx.y / S pairs lhs vector a with rhs vector b: PairList)
template <class P>
__global__ void __launch_bounds__(64)
k_pair_tree_lines(const Line6<P>* __restrict__ lines, const Affine<Fp<P>>* __restrict__ g1, u32 n, u32 c, u32 n_r, u32 S,
                  PairList pl, Fp12<P>* __restrict__ out) {
    extern __shared__ unsigned char pair_lds[];
    typedef WaveF12<P> W;
    typedef Fp<P> Fq;
    typedef TowerParams<P> T;
    WaveArea<P>* w = reinterpret_cast<WaveArea<P>*>(pair_lds);
    Fq* s = reinterpret_cast<Fq*>(pair_lds + sizeof(WaveArea<P>));
    WaveF12<P>::init(w);
    Fq *acc = s, *cur = s + WV_SLOT;
    u32 lo = blockIdx.x * c, hi = min(lo + c, n);
    u32 p = blockIdx.y / S, st = blockIdx.y % S;
    u32 a = pl.n ? pl.a[p] : p / n_r, b = pl.n ? pl.b[p] : p % n_r;
    const Line6<P>* src = lines + ((size_t)b * S + st) * n;
    const Affine<Fq>* pts = g1 + (size_t)a * n;
    u32 lane = threadIdx.x;
    // sparse positions (ark mul_by_034 / mul_by_014): D: c0 -> 0,1  c1 -> 6,7  c2 -> 8,9;  M: c0 -> 0,1  c1 -> 2,3  c2 -> 8,9
    // evaluation (ark `ell`): D: c0 * p.y, c1 * p.x;  M: c1 * p.x, c2 * p.y
    int src_idx = -1, scale = 0;                    // scale: 0 none, 1 by p.x, 2 by p.y
    if (lane < 2) { src_idx = lane; scale = T::TWIST_IS_D ? 2 : 0; }
    else if (T::TWIST_IS_D && lane >= 6 && lane < 8) { src_idx = 2 + (lane - 6); scale = 1; }
    else if (!T::TWIST_IS_D && lane >= 2 && lane < 4) { src_idx = 2 + (lane - 2); scale = 1; }
    else if (lane >= 8 && lane < 10) { src_idx = 4 + (lane - 8); scale = T::TWIST_IS_D ? 0 : 2; }
    auto load_line = [&](Fq* dst, u32 i) {
        Fq v = Fq::zero();
        bool raw_nz = false, pt_nz = false;
        if (src_idx >= 0) {
            v = ld_vec(&reinterpret_cast<const Fq*>(&src[i])[src_idx]);
            raw_nz = !v.is_zero();
            if (scale) {
                Fq k = ld_vec(scale == 1 ? &pts[i].x : &pts[i].y);
                pt_nz = !k.is_zero();
                v = Fq::mul(v, k);
            }
        }
        // the pair contributes 1 when either point is at infinity (all-zero raw line / (0, 0) lhs point)
        bool one = __ballot(raw_nz) == 0 || __ballot(pt_nz) == 0;
        if (lane < 13) dst[lane] = one ? (lane == 0 ? Fq::one() : Fq::zero()) : v;
        W::sync();
    };
    load_line(acc, lo);
    for (u32 i = lo + 1; i < hi; i++) {
        load_line(cur, i);
        W::mul(acc, acc, cur, w);
    }
    W::store(&out[(size_t)blockIdx.y * gridDim.x + blockIdx.x], acc);
}

// one wave per product: f = Horner over the step products L[prod][s], conjugate when x < 0, final exponentiation
template <class P>
__global__ void __launch_bounds__(64)
k_pair_horner(const Fp12<P>* __restrict__ L, PairSteps st, Fp12<P>* __restrict__ out) {
    extern __shared__ unsigned char pair_lds[];
    typedef WaveF12<P> W;
    typedef Fp<P> Fq;
    WaveArea<P>* w = reinterpret_cast<WaveArea<P>*>(pair_lds);
    Fq* s = reinterpret_cast<Fq*>(pair_lds + sizeof(WaveArea<P>));
    WaveF12<P>::init(w);
    Fq* acc = s;
    Fq* cur = s + 9 * WV_SLOT;
    const Fp12<P>* src = L + (size_t)blockIdx.x * st.n;
    W::load(acc, &src[0]);
    for (int k = 1; k < st.n; k++) {
        if (st.sq[k]) W::sqr(acc, acc, w);
        W::load(cur, &src[k]);
        W::mul(acc, acc, cur, w);
    }
    if (TowerParams<P>::X_IS_NEGATIVE) W::conj(acc, acc);
    W::final_exp(s, w);
    W::store(&out[blockIdx.x], acc);
}

// One wave per product: the product of its n Miller values (each lane-group of the wave takes a strided share through
// the wave multiplier: acc *= in[i]), then the final exponentiation.  in: [count][n] canonical Fq12; out: [count].
template <class P>
__global__ void __launch_bounds__(64)
k_pair_finish(const Fp12<P>* __restrict__ in, u32 n, Fp12<P>* __restrict__ out) {
    extern __shared__ unsigned char pair_lds[];
    typedef WaveF12<P> W;
    typedef Fp<P> Fq;
    WaveArea<P>* w = reinterpret_cast<WaveArea<P>*>(pair_lds);
    Fq* s = reinterpret_cast<Fq*>(pair_lds + sizeof(WaveArea<P>));
    WaveF12<P>::init(w);
    const Fp12<P>* src = in + (size_t)blockIdx.x * n;
    Fq* acc = s;
    Fq* cur = s + 9 * WV_SLOT;
    W::load(acc, &src[0]);
    for (u32 i = 1; i < n; i++) {
        W::load(cur, &src[i]);
        W::mul(acc, acc, cur, w);
    }
    W::final_exp(s, w);
    W::store(&out[blockIdx.x], acc);
}

// Tree level of the Miller-value product: wave g multiplies in[g*c .. min((g+1)*c, n)) of its product -> out[g].
// grid = (groups, count); in: [count][n], out: [count][groups].
template <class P>
__global__ void __launch_bounds__(64)
k_pair_tree(const Fp12<P>* __restrict__ in, u32 n, u32 c, Fp12<P>* __restrict__ out) {
    extern __shared__ unsigned char pair_lds[];
    typedef WaveF12<P> W;
    typedef Fp<P> Fq;
    WaveArea<P>* w = reinterpret_cast<WaveArea<P>*>(pair_lds);
    Fq* s = reinterpret_cast<Fq*>(pair_lds + sizeof(WaveArea<P>));
    WaveF12<P>::init(w);
    Fq *acc = s, *cur = s + WV_SLOT;
    u32 lo = blockIdx.x * c, hi = min(lo + c, n);
    const Fp12<P>* src = in + (size_t)blockIdx.y * n;
    W::load(acc, &src[lo]);
    for (u32 i = lo + 1; i < hi; i++) {
        W::load(cur, &src[i]);
        W::mul(acc, acc, cur, w);
    }
    W::store(&out[(size_t)blockIdx.y * gridDim.x + blockIdx.x], acc);
}

// out[e] = in[e]^scalars[e] in Fq12 (GT): one wave per element, square-and-multiply on the wave multiplier.
// (`Commitment * scalar` / `PairingOutput * scalar`: distributed-prover/src/aggregation.rs:171-174,328-332 and the
// verifier side of TIPA - six GT powers per GIPA round.)
template <class P, class Fr>
__global__ void __launch_bounds__(64)
k_gt_pow(const Fp12<P>* __restrict__ in, const Fr* __restrict__ scalars_mont, u32 n, Fp12<P>* __restrict__ out) {
    extern __shared__ unsigned char pair_lds[];
    typedef WaveF12<P> W;
    typedef Fp<P> Fq;
    __shared__ u32 kbits[Fr::N];
    WaveArea<P>* w = reinterpret_cast<WaveArea<P>*>(pair_lds);
    Fq* s = reinterpret_cast<Fq*>(pair_lds + sizeof(WaveArea<P>));
    WaveF12<P>::init(w);
    Fq *base = s, *acc = s + WV_SLOT;
    if (blockIdx.x >= n) return;
    if (threadIdx.x == 0) {
        Fr k = Fr::from_mont(ld_vec(&scalars_mont[blockIdx.x]));
        for (int i = 0; i < Fr::N; i++) kbits[i] = k.v[i];
    }
    W::load(base, &in[blockIdx.x]);
    W::set_one(acc);
    int top = -1;
    for (int b = Fr::N * 32 - 1; b >= 0; b--)
        if ((kbits[b >> 5] >> (b & 31)) & 1) { top = b; break; }
    for (int b = top; b >= 0; b--) {
        W::sqr(acc, acc, w);
        if ((kbits[b >> 5] >> (b & 31)) & 1) W::mul(acc, acc, base, w);
    }
    W::store(&out[blockIdx.x], acc);
}

// The same power over the Frobenius: an element of GT has order r and pi(z) = z^q, so z^c = prod_j pi^j(z)^(+-|k_j|) for
// the split c = sum_j k_j (q mod r)^j of the G2 folds (endo.cuh: four parts <= 67 bits; the inverse of a GT element is its
// conjugate).  One joint chain of <= 68 squarings with ONE product per step from the table of the 15 subset products:
// ~150 dependent wave products instead of ~380.  Only for elements of GT (pairing values and their products).
template <class P, class Fr>
__global__ void __launch_bounds__(64)
k_gt_pow_endo(const Fp12<P>* __restrict__ in, const Fr* __restrict__ scalars_mont, u32 n, EndoSplit<4> E,
              Fp12<P>* __restrict__ out) {
    extern __shared__ unsigned char pair_lds[];
    typedef WaveF12<P> W;
    typedef Fp<P> Fq;
    __shared__ u32 mags[4][6];
    __shared__ u32 negs;
    WaveArea<P>* w = reinterpret_cast<WaveArea<P>*>(pair_lds);
    Fq* tab = reinterpret_cast<Fq*>(pair_lds + sizeof(WaveArea<P>));       // tab[m]: slot m, m = 1 .. 15; slot 0: acc
    WaveF12<P>::init(w);
    if (blockIdx.x >= n) return;
    if (threadIdx.x == 0) {
        Fr k = Fr::from_mont(ld_vec(&scalars_mont[blockIdx.x]));
        u32 c[8], mg[4][6];
        for (int i = 0; i < 8; i++) c[i] = i < Fr::N ? k.v[i] : 0u;
        negs = endo_decompose<4>(c, E, mg);
        for (int j = 0; j < 4; j++) for (int l = 0; l < 6; l++) mags[j][l] = mg[j][l];
    }
    W::sync();
    Fq* acc = tab;
    W::load(tab + 1 * WV_SLOT, &in[blockIdx.x]);
    W::template frob<1>(tab + 2 * WV_SLOT, tab + 1 * WV_SLOT);
    W::template frob<2>(tab + 4 * WV_SLOT, tab + 1 * WV_SLOT);
    W::template frob<3>(tab + 8 * WV_SLOT, tab + 1 * WV_SLOT);
    for (int j = 0; j < 4; j++)
        if ((negs >> j) & 1u) W::conj(tab + (1 << j) * WV_SLOT, tab + (1 << j) * WV_SLOT);
    for (int j = 1; j < 4; j++)
        for (int m = 1; m < (1 << j); m++) W::mul(tab + ((1 << j) | m) * WV_SLOT, tab + m * WV_SLOT, tab + (1 << j) * WV_SLOT, w);
    int top = -1;
    for (int b = 6 * 32 - 1; b >= 0 && top < 0; b--)
        for (int j = 0; j < 4; j++)
            if ((mags[j][b >> 5] >> (b & 31)) & 1u) top = b;
    W::set_one(acc);
    for (int b = top; b >= 0; b--) {
        W::cyc_sqr(acc, acc, w);                                 // GT lies in the cyclotomic subgroup
        u32 m = 0;
        for (int j = 0; j < 4; j++) m |= ((mags[j][b >> 5] >> (b & 31)) & 1u) << j;
        if (m) W::mul(acc, acc, tab + m * WV_SLOT, w);
    }
    W::store(&out[blockIdx.x], acc);
}

#endif  // __HIPCC__

}  // namespace hk
