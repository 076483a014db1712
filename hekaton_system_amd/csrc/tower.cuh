// tower.cuh — Fq6 / Fq12 extension-tower arithmetic over the Fp / Fp2 of field.cuh, for the pairings of the
// aggregation path.  Same tower ark-ff instantiates for ark-bn254 / ark-bls12-381
// (`Fp6 = Fp2[v]/(v^3 - xi)`, `Fp12 = Fp6[w]/(w^2 - v)`), so a GT element crosses the C ABI as the 12 Fq limbs
// arrays of ark's `Fp12 { c0: Fp6 { c0, c1, c2: Fp2 { c0, c1 } }, c1 }`, Montgomery form.
// The reference reaches this arithmetic through `E::multi_miller_loop` / `E::final_exponentiation`
// (distributed-prover/src/pairing_ops.rs:9-29).  Compiles as host C++ too (tests/host_shim), like field.cuh.
//
// Code-size rule: every out-of-line function here stays far below the 128 KiB reach of s_cbranch —
// the Fp2 product is the inlining boundary (f_mul_ni<Fp2>), Fp6 / Fp12 products are calls built from it.
#pragma once
#include "ec.cuh"
#include "hk_tower_params.h"

namespace hk {

template <class P> struct TowerParams;
#define HK_DEFINE_TOWER(FQP, PRE)                                                                       \
    template <> struct TowerParams<FQP> {                                                               \
        static constexpr int XI_C0 = PRE##_XI_C0;                                                       \
        static constexpr bool TWIST_IS_D = PRE##_TWIST_IS_D != 0;                                       \
        static constexpr u64 X = PRE##_X;                                                               \
        static constexpr bool X_IS_NEGATIVE = PRE##_X_IS_NEGATIVE != 0;                                 \
        static constexpr int LOOP_LEN = PRE##_LOOP_LEN;                                                 \
        static constexpr u32 TWO_INV[FQP::N] = PRE##_TWO_INV;                                           \
        static constexpr u32 B_TWIST[2][FQP::N] = PRE##_B_TWIST;                                        \
        static constexpr bool B_TWIST_IS_4_4 = PRE##_B_TWIST_IS_4_4 != 0;                               \
        static constexpr u32 FROB6_C1[3][2][FQP::N] = {PRE##_FROB6_C1_1, PRE##_FROB6_C1_2, PRE##_FROB6_C1_3};    \
        static constexpr u32 FROB6_C2[3][2][FQP::N] = {PRE##_FROB6_C2_1, PRE##_FROB6_C2_2, PRE##_FROB6_C2_3};    \
        static constexpr u32 FROB12_C1[3][2][FQP::N] = {PRE##_FROB12_C1_1, PRE##_FROB12_C1_2, PRE##_FROB12_C1_3}; \
        static constexpr u32 MUL_BY_Q_X[2][FQP::N] = PRE##_MUL_BY_Q_X;                                  \
        static constexpr u32 MUL_BY_Q_Y[2][FQP::N] = PRE##_MUL_BY_Q_Y;                                  \
        static constexpr u32 BETA[FQP::N] = PRE##_BETA;                                                 \
        static constexpr u32 PSI_X[2][FQP::N] = PRE##_PSI_X;                                            \
        static constexpr u32 PSI_Y[2][FQP::N] = PRE##_PSI_Y;                                            \
    };
HK_DEFINE_TOWER(Bn254FqP, HK_BN254_TW)
HK_DEFINE_TOWER(Bls381FqP, HK_BLS12_381_TW)

// Miller-loop digits travel as a kernel argument (dynamic indexing of a constexpr member array is not portable
// device code); filled on the host from the generated tables
struct PairLoop {
    int len;
    signed char digits[72];
};
inline PairLoop pair_loop_bn254() { PairLoop l = {HK_BN254_TW_LOOP_LEN, HK_BN254_TW_LOOP_DIGITS}; return l; }
inline PairLoop pair_loop_bls381() { PairLoop l = {HK_BLS12_381_TW_LOOP_LEN, HK_BLS12_381_TW_LOOP_DIGITS}; return l; }

template <class P> struct Fp6 { Fp2<P> c0, c1, c2; };
template <class P> struct Fp12 { Fp6<P> c0, c1; };

// ---- constants -----------------------------------------------------------------------------------------
template <class P>
HK_HD Fp<P> fp_const(const u32 (&l)[P::N]) {
    Fp<P> r;
    HK_UNROLL for (int i = 0; i < P::N; i++) r.v[i] = l[i];
    return r;
}
template <class P>
HK_HD Fp2<P> fp2_const(const u32 (&l)[2][P::N]) {
    Fp2<P> r;
    HK_UNROLL for (int i = 0; i < P::N; i++) { r.c0.v[i] = l[0][i]; r.c1.v[i] = l[1][i]; }
    return r;
}

// ---- Fp2 extras ------------------------------------------------------------------------------------------
template <class P>
HK_HD Fp2<P> f2_conj(const Fp2<P>& a) { Fp2<P> r; r.c0 = a.c0; r.c1 = Fp<P>::neg(a.c1); return r; }
template <class P>
HK_HD Fp2<P> f2_scale(const Fp2<P>& a, const Fp<P>& k) {        // Fp2 x Fp
    Fp2<P> r; r.c0 = Fp<P>::mul(a.c0, k); r.c1 = Fp<P>::mul(a.c1, k); return r;
}
// a * xi, xi = XI_C0 + u  (9 + u / 1 + u): adds only
template <class P>
HK_HD Fp2<P> f2_mul_xi(const Fp2<P>& a) {
    typedef Fp<P> B;
    Fp2<P> r;
    if constexpr (TowerParams<P>::XI_C0 == 1) {
        r.c0 = B::sub(a.c0, a.c1);
        r.c1 = B::add(a.c0, a.c1);
    } else {
        static_assert(TowerParams<P>::XI_C0 == 9 || TowerParams<P>::XI_C0 == 1, "xi = 9 + u or 1 + u");
        B t0 = B::add(B::dbl(B::dbl(B::dbl(a.c0))), a.c0);      // 9 a0
        B t1 = B::add(B::dbl(B::dbl(B::dbl(a.c1))), a.c1);      // 9 a1
        r.c0 = B::sub(t0, a.c1);
        r.c1 = B::add(t1, a.c0);
    }
    return r;
}
template <class P, int K>
HK_HD Fp2<P> f2_frob(const Fp2<P>& a) { if constexpr (K & 1) return f2_conj(a); else return a; }

// the out-of-line Fp2 product every larger product is built from
template <class P>
HK_HD Fp2<P> f2m(const Fp2<P>& a, const Fp2<P>& b) { return f_mul_ni<Fp2<P>>(a, b); }
template <class P>
HK_RARE Fp2<P> f2_sqr_ni(const Fp2<P>& a) { return Fp2<P>::sqr(a); }
template <class P>
HK_HD Fp2<P> f2s(const Fp2<P>& a) { return f2_sqr_ni<P>(a); }

// ---- Fp6 ----------------------------------------------------------------------------------------------
template <class P> HK_HD Fp6<P> f6_zero() { Fp6<P> r; r.c0 = Fp2<P>::zero(); r.c1 = Fp2<P>::zero(); r.c2 = Fp2<P>::zero(); return r; }
template <class P> HK_HD Fp6<P> f6_one() { Fp6<P> r = f6_zero<P>(); r.c0 = Fp2<P>::one(); return r; }
template <class P> HK_HD Fp6<P> f6_add(const Fp6<P>& a, const Fp6<P>& b) {
    Fp6<P> r; r.c0 = Fp2<P>::add(a.c0, b.c0); r.c1 = Fp2<P>::add(a.c1, b.c1); r.c2 = Fp2<P>::add(a.c2, b.c2); return r;
}
template <class P> HK_HD Fp6<P> f6_sub(const Fp6<P>& a, const Fp6<P>& b) {
    Fp6<P> r; r.c0 = Fp2<P>::sub(a.c0, b.c0); r.c1 = Fp2<P>::sub(a.c1, b.c1); r.c2 = Fp2<P>::sub(a.c2, b.c2); return r;
}
template <class P> HK_HD Fp6<P> f6_neg(const Fp6<P>& a) {
    Fp6<P> r; r.c0 = Fp2<P>::neg(a.c0); r.c1 = Fp2<P>::neg(a.c1); r.c2 = Fp2<P>::neg(a.c2); return r;
}
template <class P> HK_HD Fp6<P> f6_mul_by_v(const Fp6<P>& a) {      // a * v: (xi a2, a0, a1)
    Fp6<P> r; r.c0 = f2_mul_xi(a.c2); r.c1 = a.c0; r.c2 = a.c1; return r;
}
// Karatsuba, 6 Fp2 products
template <class P>
HK_RARE Fp6<P> f6_mul(const Fp6<P>& a, const Fp6<P>& b) {
    typedef Fp2<P> F;
    F v0 = f2m(a.c0, b.c0), v1 = f2m(a.c1, b.c1), v2 = f2m(a.c2, b.c2);
    Fp6<P> r;
    r.c0 = F::add(v0, f2_mul_xi(F::sub(F::sub(f2m(F::add(a.c1, a.c2), F::add(b.c1, b.c2)), v1), v2)));
    r.c1 = F::add(F::sub(F::sub(f2m(F::add(a.c0, a.c1), F::add(b.c0, b.c1)), v0), v1), f2_mul_xi(v2));
    r.c2 = F::add(F::sub(F::sub(f2m(F::add(a.c0, a.c2), F::add(b.c0, b.c2)), v0), v2), v1);
    return r;
}
template <class P>
HK_RARE Fp6<P> f6_inv(const Fp6<P>& a) {
    typedef Fp2<P> F;
    F t0 = F::sub(f2s(a.c0), f2_mul_xi(f2m(a.c1, a.c2)));
    F t1 = F::sub(f2_mul_xi(f2s(a.c2)), f2m(a.c0, a.c1));
    F t2 = F::sub(f2s(a.c1), f2m(a.c0, a.c2));
    F d = F::add(f2m(a.c0, t0), f2_mul_xi(F::add(f2m(a.c2, t1), f2m(a.c1, t2))));
    F di = fp_inv(d);
    Fp6<P> r; r.c0 = f2m(t0, di); r.c1 = f2m(t1, di); r.c2 = f2m(t2, di);
    return r;
}
template <class P, int K>
HK_HD Fp6<P> f6_frob(const Fp6<P>& a) {
    typedef TowerParams<P> T;
    Fp6<P> r;
    r.c0 = f2_frob<P, K>(a.c0);
    r.c1 = f2m(f2_frob<P, K>(a.c1), fp2_const<P>(T::FROB6_C1[K - 1]));
    r.c2 = f2m(f2_frob<P, K>(a.c2), fp2_const<P>(T::FROB6_C2[K - 1]));
    return r;
}

// ---- Fp12 ---------------------------------------------------------------------------------------------
template <class P> HK_HD Fp12<P> f12_one() { Fp12<P> r; r.c0 = f6_one<P>(); r.c1 = f6_zero<P>(); return r; }
template <class P>
HK_RARE Fp12<P> f12_mul(const Fp12<P>& a, const Fp12<P>& b) {          // Karatsuba, 3 Fp6 products
    Fp6<P> v0 = f6_mul(a.c0, b.c0), v1 = f6_mul(a.c1, b.c1);
    Fp12<P> r;
    r.c1 = f6_sub(f6_sub(f6_mul(f6_add(a.c0, a.c1), f6_add(b.c0, b.c1)), v0), v1);
    r.c0 = f6_add(v0, f6_mul_by_v(v1));
    return r;
}
template <class P>
HK_RARE Fp12<P> f12_sqr(const Fp12<P>& a) {                            // complex squaring, 2 Fp6 products
    Fp6<P> ab = f6_mul(a.c0, a.c1);
    Fp6<P> t = f6_mul(f6_add(a.c0, a.c1), f6_add(a.c0, f6_mul_by_v(a.c1)));
    Fp12<P> r;
    r.c0 = f6_sub(f6_sub(t, ab), f6_mul_by_v(ab));
    r.c1 = f6_add(ab, ab);
    return r;
}
template <class P> HK_HD Fp12<P> f12_conj(const Fp12<P>& a) { Fp12<P> r; r.c0 = a.c0; r.c1 = f6_neg(a.c1); return r; }
template <class P>
HK_RARE Fp12<P> f12_inv(const Fp12<P>& a) {
    Fp6<P> d = f6_sub(f6_mul(a.c0, a.c0), f6_mul_by_v(f6_mul(a.c1, a.c1)));
    Fp6<P> di = f6_inv(d);
    Fp12<P> r; r.c0 = f6_mul(a.c0, di); r.c1 = f6_neg(f6_mul(a.c1, di));
    return r;
}
template <class P, int K>
HK_RARE Fp12<P> f12_frob(const Fp12<P>& a) {
    typedef TowerParams<P> T;
    Fp12<P> r;
    r.c0 = f6_frob<P, K>(a.c0);
    Fp6<P> c1 = f6_frob<P, K>(a.c1);
    Fp2<P> co = fp2_const<P>(T::FROB12_C1[K - 1]);
    r.c1.c0 = f2m(c1.c0, co); r.c1.c1 = f2m(c1.c1, co); r.c1.c2 = f2m(c1.c2, co);
    return r;
}
template <class P>
HK_HD bool f12_is_one(const Fp12<P>& a) {
    Fp12<P> o = f12_one<P>();
    return a.c0.c0 == o.c0.c0 && a.c0.c1.is_zero() && a.c0.c2.is_zero() && a.c1.c0.is_zero() && a.c1.c1.is_zero() &&
           a.c1.c2.is_zero();
}
// canonical representative of every coordinate (memory form)
template <class P>
HK_HD Fp12<P> f12_canon(const Fp12<P>& a) {
    Fp12<P> r;
    r.c0.c0 = Fp2<P>::canon(a.c0.c0); r.c0.c1 = Fp2<P>::canon(a.c0.c1); r.c0.c2 = Fp2<P>::canon(a.c0.c2);
    r.c1.c0 = Fp2<P>::canon(a.c1.c0); r.c1.c1 = Fp2<P>::canon(a.c1.c1); r.c1.c2 = Fp2<P>::canon(a.c1.c2);
    return r;
}
// a^X for the curve parameter X (plain square-and-multiply, MSB first)
template <class P>
HK_RARE Fp12<P> f12_pow_x(const Fp12<P>& a) {
    const u64 x = TowerParams<P>::X;
    Fp12<P> r = a;
    int top = 63;
    while (!((x >> top) & 1)) top--;
    HK_NOUNROLL for (int bit = top - 1; bit >= 0; bit--) {
        r = f12_sqr(r);
        if ((x >> bit) & 1) r = f12_mul(r, a);
    }
    return r;
}

}  // namespace hk
