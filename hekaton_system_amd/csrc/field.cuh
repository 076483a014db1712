// field.cuh — Montgomery prime-field and quadratic-extension arithmetic on 32-bit limbs,
// written for CDNA4 VALU (v_mad_u64_u32 + v_add_co/v_addc chains); all loops fully unrolled
// so the modulus limbs fold into instruction immediates and every element lives in VGPRs.
//
// Memory form == ark-ff `Fp<MontBackend<_,N64>>` (R = 2^(64*N64) = 2^(32*N)), so buffers cross
// the C ABI without conversion.  Also compiles as plain host C++ (tests build it with g++ to
// unit-test the arithmetic against tests/golden without a GPU; the library itself never runs it
// on the CPU).
#pragma once
#include <stdint.h>
#include "hk_params.h"
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HK_NO_ASM_MUL)
#include "mont_asm.h"
#define HK_USE_ASM_MUL 1
#endif

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HK_HD __host__ __device__ __forceinline__
#define HK_CALL __host__ __device__ __noinline__
#define HK_UNROLL _Pragma("unroll")
#else
#define HK_HD inline
#define HK_CALL inline
#define HK_UNROLL
#endif

// Lazy reduction (device code only): values of fields with two spare bits (4p <= R) live in [0, 2p) while
// they are in registers — the Montgomery product of two such values is again < 2p when 4p <= R, so the
// conditional subtraction after every product disappears (ubench: +12 % mults/s, +8 % mixed adds/s).
// Everything written to memory is canonical [0, p): see canon() and the st_vec / fr_store helpers.
// Host code keeps the canonical form throughout.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HK_NO_LAZY)
#define HK_LAZY_DEVICE 1
#else
#define HK_LAZY_DEVICE 0
#endif

namespace hk {

typedef uint32_t u32;
typedef uint64_t u64;

// ---- parameter packs ---------------------------------------------------------------------
#define HK_DEFINE_FIELD(NAME, PREFIX, ASMID)                                 \
    struct NAME {                                                            \
        static constexpr int N = PREFIX##_N;                                 \
        static constexpr int ASM_ID = ASMID;                                 \
        static constexpr u32 MOD[PREFIX##_N] = PREFIX##_MOD;                 \
        static constexpr u32 MOD2[PREFIX##_N] = PREFIX##_MOD2;               \
        static constexpr bool LAZY = HK_LAZY_DEVICE && PREFIX##_LAZY_OK;                     \
        static constexpr u32 ONE[PREFIX##_N] = PREFIX##_ONE;                 \
        static constexpr u32 R2[PREFIX##_N] = PREFIX##_R2;                   \
        static constexpr u32 INV = PREFIX##_INV32;                           \
    };

HK_DEFINE_FIELD(Bn254FrP, HK_BN254_FR, 1)
HK_DEFINE_FIELD(Bn254FqP, HK_BN254_FQ, 2)
HK_DEFINE_FIELD(Bls381FrP, HK_BLS12_381_FR, 3)
HK_DEFINE_FIELD(Bls381FqP, HK_BLS12_381_FQ, 4)      // 12 limbs: tied-operand asm form

// ---- prime field ---------------------------------------------------------------------------
template <class P>
struct Fp {
    static constexpr int N = P::N;
    typedef P Params;
    u32 v[N];

    HK_HD static Fp zero() {
        Fp r;
        HK_UNROLL for (int i = 0; i < N; i++) r.v[i] = 0;
        return r;
    }
    HK_HD static Fp one() {
        Fp r;
        HK_UNROLL for (int i = 0; i < N; i++) r.v[i] = P::ONE[i];
        return r;
    }
    HK_HD static Fp r2() {
        Fp r;
        HK_UNROLL for (int i = 0; i < N; i++) r.v[i] = P::R2[i];
        return r;
    }
    HK_HD bool is_zero() const {                      // 0, or p in the lazy form
        u32 acc = 0;
        HK_UNROLL for (int i = 0; i < N; i++) acc |= v[i];
        if constexpr (P::LAZY) {
            u32 accp = 0;
            HK_UNROLL for (int i = 0; i < N; i++) accp |= (v[i] ^ P::MOD[i]);
            return acc == 0 || accp == 0;
        }
        return acc == 0;
    }
    HK_HD bool operator==(const Fp& o) const {
        Fp a = canon(*this), b = canon(o);
        u32 acc = 0;
        HK_UNROLL for (int i = 0; i < N; i++) acc |= (a.v[i] ^ b.v[i]);
        return acc == 0;
    }
    // canonical representative in [0, p) (identity when the field is not lazy)
    HK_HD static Fp canon(const Fp& a) {
        if constexpr (P::LAZY) return reduce_once(a);
        return a;
    }
    HK_HD bool operator!=(const Fp& o) const { return !(*this == o); }

    // r = a - MOD if a >= MOD (a < 2*MOD, carry-free because MOD has a spare top bit)
    HK_HD static Fp reduce_once(const Fp& a) {
#if defined(HK_USE_ASM_MUL)
        if constexpr (P::LAZY && P::ASM_ID == 1) { Fp r; HK_CANON_ASM_BN254_FR(r, a); return r; }
        if constexpr (P::LAZY && P::ASM_ID == 2) { Fp r; HK_CANON_ASM_BN254_FQ(r, a); return r; }
        if constexpr (P::LAZY && P::ASM_ID == 4) { Fp r; HK_CANON_ASM_BLS12_381_FQ(r, a); return r; }
#endif
        Fp s;
        u64 borrow = 0;
        HK_UNROLL for (int i = 0; i < N; i++) {
            u64 d = (u64)a.v[i] - P::MOD[i] - borrow;
            s.v[i] = (u32)d;
            borrow = (d >> 32) & 1;
        }
        Fp r;
        HK_UNROLL for (int i = 0; i < N; i++) r.v[i] = borrow ? a.v[i] : s.v[i];
        return r;
    }

    // r = a - 2*MOD if a >= 2*MOD  (lazy form: a < 4*MOD <= 2^(32N))
    HK_HD static Fp reduce_once_2p(const Fp& a) {
        Fp s;
        u64 borrow = 0;
        HK_UNROLL for (int i = 0; i < N; i++) {
            u64 d = (u64)a.v[i] - P::MOD2[i] - borrow;
            s.v[i] = (u32)d;
            borrow = (d >> 32) & 1;
        }
        Fp r;
        HK_UNROLL for (int i = 0; i < N; i++) r.v[i] = borrow ? a.v[i] : s.v[i];
        return r;
    }
    HK_HD static Fp add(const Fp& a, const Fp& b) {
#if defined(HK_USE_ASM_MUL)
        // lazy fields: carry chain in VCC, 3N VALU instructions (gen_mont_asm.py gen_addsub)
        if constexpr (P::LAZY && P::ASM_ID == 1) { Fp r; HK_ADD_ASM_BN254_FR(r, a, b); return r; }
        if constexpr (P::LAZY && P::ASM_ID == 2) { Fp r; HK_ADD_ASM_BN254_FQ(r, a, b); return r; }
        if constexpr (!P::LAZY && P::ASM_ID == 3) { Fp r; HK_ADD_ASM_BLS12_381_FR(r, a, b); return r; }   // canonical
        if constexpr (P::LAZY && P::ASM_ID == 4) { Fp r; HK_ADD_ASM_BLS12_381_FQ(r, a, b); return r; }
#endif
        Fp t;
        u64 c = 0;
        HK_UNROLL for (int i = 0; i < N; i++) {
            c += (u64)a.v[i] + b.v[i];
            t.v[i] = (u32)c;
            c >>= 32;
        }
        if constexpr (P::LAZY) return reduce_once_2p(t);   // a + b < 4*MOD
        return reduce_once(t);   // a + b < 2*MOD < 2^(32N)
    }
    HK_HD static Fp dbl(const Fp& a) {
#if defined(HK_USE_ASM_MUL)
        if constexpr (P::LAZY && P::ASM_ID == 1) { Fp r; HK_DBL_ASM_BN254_FR(r, a); return r; }
        if constexpr (P::LAZY && P::ASM_ID == 2) { Fp r; HK_DBL_ASM_BN254_FQ(r, a); return r; }
        if constexpr (!P::LAZY && P::ASM_ID == 3) { Fp r; HK_DBL_ASM_BLS12_381_FR(r, a); return r; }
        if constexpr (P::LAZY && P::ASM_ID == 4) { Fp r; HK_DBL_ASM_BLS12_381_FQ(r, a); return r; }
#endif
        return add(a, a);
    }

    HK_HD static Fp sub(const Fp& a, const Fp& b) {
#if defined(HK_USE_ASM_MUL)
        if constexpr (P::LAZY && P::ASM_ID == 1) { Fp r; HK_SUB_ASM_BN254_FR(r, a, b); return r; }
        if constexpr (P::LAZY && P::ASM_ID == 2) { Fp r; HK_SUB_ASM_BN254_FQ(r, a, b); return r; }
        if constexpr (!P::LAZY && P::ASM_ID == 3) { Fp r; HK_SUB_ASM_BLS12_381_FR(r, a, b); return r; }
        if constexpr (P::LAZY && P::ASM_ID == 4) { Fp r; HK_SUB_ASM_BLS12_381_FQ(r, a, b); return r; }
#endif
        Fp t;
        u64 borrow = 0;
        HK_UNROLL for (int i = 0; i < N; i++) {
            u64 d = (u64)a.v[i] - b.v[i] - borrow;
            t.v[i] = (u32)d;
            borrow = (d >> 32) & 1;
        }
        // add MOD (2*MOD in the lazy form) back when we borrowed
        u32 mask = (u32)0 - (u32)borrow;
        u64 c = 0;
        HK_UNROLL for (int i = 0; i < N; i++) {
            c += (u64)t.v[i] + ((P::LAZY ? P::MOD2[i] : P::MOD[i]) & mask);
            t.v[i] = (u32)c;
            c >>= 32;
        }
        return t;
    }
    HK_HD static Fp neg(const Fp& a) {
        // branch-free: (2)p - a, masked to 0 when a is 0 (-0 = 0 keeps the result below the bound).  No early return: a
        // data-dependent exit here diverges inside every loop that negates per lane.
        u32 any = 0;
        HK_UNROLL for (int i = 0; i < N; i++) any |= a.v[i];
        const u32 mask = any ? 0xffffffffu : 0u;
        Fp t;
        u64 borrow = 0;
        HK_UNROLL for (int i = 0; i < N; i++) {
            u64 d = (u64)(P::LAZY ? P::MOD2[i] : P::MOD[i]) - a.v[i] - borrow;
            t.v[i] = (u32)d & mask;
            borrow = (d >> 32) & 1;
        }
        return t;
    }

    // CIOS Montgomery product, one 32-bit word of b per round.  MOD < 2^(32N-1) keeps the
    // running value below 2*MOD, so N+1 words suffice and the top word is 0 after each round.
    HK_HD static Fp mul(const Fp& a, const Fp& b) {
#if defined(HK_USE_ASM_MUL)
        // hand-scheduled product-scanning form (gen_mont_asm.py): 128 mad+addc pairs, no pair shuffles
        if constexpr (P::ASM_ID == 1) { Fp r; HK_MONT_ASM_BN254_FR(r, a, b); if constexpr (P::LAZY) return r; else return reduce_once(r); }
        if constexpr (P::ASM_ID == 2) { Fp r; HK_MONT_ASM_BN254_FQ(r, a, b); if constexpr (P::LAZY) return r; else return reduce_once(r); }
        if constexpr (P::ASM_ID == 3) { Fp t, r; HK_MONT_ASM_BLS12_381_FR(t, a, b); HK_RED_ASM_BLS12_381_FR(r, t); return r; }
        if constexpr (P::ASM_ID == 4) { Fp r; HK_MONT_ASM_BLS12_381_FQ(r, a, b); if constexpr (P::LAZY) return r; else return reduce_once(r); }
#endif
        u32 t[N + 1];
        HK_UNROLL for (int i = 0; i <= N; i++) t[i] = 0;
        HK_UNROLL for (int i = 0; i < N; i++) {
            u64 c = 0;
            HK_UNROLL for (int j = 0; j < N; j++) {
                c = (u64)a.v[j] * b.v[i] + t[j] + c;
                t[j] = (u32)c;
                c >>= 32;
            }
            c += t[N];
            t[N] = (u32)c;
            u32 m = t[0] * P::INV;
            c = (u64)m * P::MOD[0] + t[0];
            c >>= 32;
            HK_UNROLL for (int j = 1; j < N; j++) {
                c = (u64)m * P::MOD[j] + t[j] + c;
                t[j - 1] = (u32)c;
                c >>= 32;
            }
            c += t[N];
            t[N - 1] = (u32)c;
            t[N] = (u32)(c >> 32);
        }
        Fp r;
        HK_UNROLL for (int i = 0; i < N; i++) r.v[i] = t[i];
        if constexpr (P::LAZY) return r;             // < 2*MOD for inputs < 2*MOD (4*MOD <= R)
        return reduce_once(r);
    }
    HK_HD static Fp sqr(const Fp& a) { return mul(a, a); }
    // a / 2: (a odd ? a + p : a) >> 1 - a Montgomery value halves like its integer.  Lazy input < 2p gives < 1.5p.
    HK_HD static Fp halve(const Fp& a) {
        u32 odd = 0u - (a.v[0] & 1u);
        u32 t[N];
        u64 c = 0;
        HK_UNROLL for (int i = 0; i < N; i++) {
            c += (u64)a.v[i] + (P::MOD[i] & odd);
            t[i] = (u32)c;
            c >>= 32;
        }
        Fp r;
        HK_UNROLL for (int i = 0; i < N; i++) r.v[i] = (t[i] >> 1) | ((i + 1 < N ? t[i + 1] : (u32)c) << 31);
        return r;
    }
    // out-of-line product with by-value (register) arguments: lets big callers (Fp2 / G2 code) keep
    // their state in registers around a compact callee instead of inlining 500 instructions per use
    HK_CALL static Fp mul_call(Fp a, Fp b) { return mul(a, b); }

    // canonical integer -> Montgomery, Montgomery -> canonical
    HK_HD static Fp to_mont(const Fp& a) { return mul(a, r2()); }
    HK_HD static Fp from_mont(const Fp& a) {        // canonical integer: its limbs get read as bits
        Fp o = zero();
        o.v[0] = 1;
        return canon(mul(a, o));
    }
};

// ---- Fq2 = Fq[u]/(u^2+1) (both BN254 and BLS12-381 towers) -------------------------------------
template <class P>
struct Fp2 {
    typedef Fp<P> B;
    typedef P Params;
    static constexpr int N = 2 * P::N;
    B c0, c1;

    HK_HD static Fp2 zero() { Fp2 r; r.c0 = B::zero(); r.c1 = B::zero(); return r; }
    HK_HD static Fp2 one() { Fp2 r; r.c0 = B::one(); r.c1 = B::zero(); return r; }
    HK_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    HK_HD bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
    HK_HD bool operator!=(const Fp2& o) const { return !(*this == o); }
    HK_HD static Fp2 canon(const Fp2& a) { Fp2 r; r.c0 = B::canon(a.c0); r.c1 = B::canon(a.c1); return r; }
    HK_HD static Fp2 add(const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = B::add(a.c0, b.c0); r.c1 = B::add(a.c1, b.c1); return r; }
    HK_HD static Fp2 sub(const Fp2& a, const Fp2& b) { Fp2 r; r.c0 = B::sub(a.c0, b.c0); r.c1 = B::sub(a.c1, b.c1); return r; }
    HK_HD static Fp2 dbl(const Fp2& a) { return add(a, a); }
    HK_HD static Fp2 halve(const Fp2& a) { Fp2 r; r.c0 = B::halve(a.c0); r.c1 = B::halve(a.c1); return r; }
    HK_HD static Fp2 neg(const Fp2& a) { Fp2 r; r.c0 = B::neg(a.c0); r.c1 = B::neg(a.c1); return r; }
    // base-field product used by the extension: inline (asm block) for 8-limb fields, an out-of-line
    // call with register arguments for 12-limb fields (keeps BLS12-381 code size and scratch small)
    HK_HD static B bmul(const B& x, const B& y) {
        if constexpr (P::N <= 8) return B::mul(x, y);
        else return B::mul_call(x, y);
    }
    HK_HD static Fp2 mul(const Fp2& a, const Fp2& b) {       // Karatsuba, 3 base muls
        B v0 = bmul(a.c0, b.c0);
        B v1 = bmul(a.c1, b.c1);
        B s = bmul(B::add(a.c0, a.c1), B::add(b.c0, b.c1));
        Fp2 r;
        r.c0 = B::sub(v0, v1);
        r.c1 = B::sub(B::sub(s, v0), v1);
        return r;
    }
    HK_HD static Fp2 sqr(const Fp2& a) {                     // (a0+a1)(a0-a1), 2 a0 a1
        B t = bmul(a.c0, a.c1);
        Fp2 r;
        r.c0 = bmul(B::add(a.c0, a.c1), B::sub(a.c0, a.c1));
        r.c1 = B::dbl(t);
        return r;
    }
};

}  // namespace hk
