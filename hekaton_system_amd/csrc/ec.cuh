// ec.cuh — short-Weierstrass (a = 0) group arithmetic in extended-Jacobian "XYZZ"
// coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), templated over the coordinate field so the
// same code serves G1 (Fp) and G2 (Fp2).  Bucket accumulation uses the 8M+2S mixed add; the
// group law is the one ark-ec `short_weierstrass` implements for the reference
// (cp-groth16/src/prover.rs:86-147), so any sum normalised to affine is bit-identical.
#pragma once
#include "field.cuh"

#if defined(__HIPCC__)
#define HK_RARE __host__ __device__ __noinline__
#else
#define HK_RARE inline
#endif

namespace hk {

template <class F>
struct Affine {
    F x, y;
    HK_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }   // (0,0) encodes infinity
    HK_HD static Affine inf() { Affine a; a.x = F::zero(); a.y = F::zero(); return a; }
};

template <class F>
struct XYZZ {
    F x, y, zz, zzz;
    HK_HD bool is_inf() const { return zz.is_zero(); }
    HK_HD static XYZZ inf() {
        XYZZ r; r.x = F::zero(); r.y = F::zero(); r.zz = F::zero(); r.zzz = F::zero();
        return r;
    }
    HK_HD static XYZZ from_affine(const Affine<F>& p) {
        if (p.is_inf()) return inf();
        XYZZ r; r.x = p.x; r.y = p.y; r.zz = F::one(); r.zzz = F::one();
        return r;
    }
};

// 2*(x,y) for an affine, non-infinity point ("mdbl-2008-s-1", a = 0); only reached from the
// P == Q corner of ec_madd, so kept out of line
template <class F>
HK_RARE XYZZ<F> ec_dbl_affine(const Affine<F>& p) {
    F u = F::dbl(p.y);
    F v = F::sqr(u);
    F w = F::mul(u, v);
    F s = F::mul(p.x, v);
    F xx = F::sqr(p.x);
    F m = F::add(F::dbl(xx), xx);
    XYZZ<F> r;
    r.x = F::sub(F::sqr(m), F::dbl(s));
    r.y = F::sub(F::mul(m, F::sub(s, r.x)), F::mul(w, p.y));
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2*P ("dbl-2008-s-1", a = 0)
template <class F>
HK_HD XYZZ<F> ec_dbl(const XYZZ<F>& p) {
    if (p.is_inf()) return p;
    F u = F::dbl(p.y);
    F v = F::sqr(u);
    F w = F::mul(u, v);
    F s = F::mul(p.x, v);
    F xx = F::sqr(p.x);
    F m = F::add(F::dbl(xx), xx);
    XYZZ<F> r;
    r.x = F::sub(F::sqr(m), F::dbl(s));
    r.y = F::sub(F::mul(m, F::sub(s, r.x)), F::mul(w, p.y));
    r.zz = F::mul(v, p.zz);
    r.zzz = F::mul(w, p.zzz);
    return r;
}

// The P == Q corner of ec_add stays OUT of line, for a reason that is a toolchain limit, not taste: with it inlined
// the out-of-line ec_add_ni of a 12-limb field becomes a LEAF function of 188 KB, more than the +-128 KiB reach of
// s_cbranch, and hipcc (ROCm 7.2) then materialises its long jumps as `s_getpc_b64 s[30:31]` ... `s_setpc_b64 s[30:31]`
// - through the function's own return address, which a leaf never saves.  The `if (b.is_inf()) return a;` path of
// that build "returned" into itself and faulted on a flat load from {lo = sret+48, hi = sret+64} = 0x280_0000_0xxx:
// the two GPU faults of round 1 (profiles/r02_bls_fault_root_cause.txt).  `make` fails the build when any
// device function outgrows the branch reach or such a long branch appears (tools/kernel_meta.py --check).
template <class F>
HK_RARE XYZZ<F> ec_dbl_rare(const XYZZ<F>& p) { return ec_dbl(p); }

// acc + affine q  ("madd-2008-s"), all exceptional cases handled.  Everything is inlined, including
// the P == Q corner: an out-of-line call there takes the addresses of `a`/`q`, which forces hipcc to
// keep copies of them in scratch memory on EVERY iteration of the accumulate loop (rocprofv3 showed
// 6.3 GB of WRITE_SIZE per 2^21-point MSM from it).
template <class F, bool INLINE_CORNER = (F::Params::N <= 8)>
HK_HD XYZZ<F> ec_madd(const XYZZ<F>& a, const Affine<F>& q) {
    if (q.is_inf()) return a;
    if (a.is_inf()) return XYZZ<F>::from_affine(q);
    F u2 = F::mul(q.x, a.zz);
    F s2 = F::mul(q.y, a.zzz);
    F p = F::sub(u2, a.x);
    F r = F::sub(s2, a.y);
    XYZZ<F> o;
    if (p.is_zero()) {
        if (!r.is_zero()) return XYZZ<F>::inf();
        // 12-limb coordinate fields keep the out-of-line form by default (code size: see the note at ec_dbl_rare);
        // the G1 accumulate loop asks for the inline form explicitly (145 VGPRs and no scratch instead of 187 VGPRs
        // and 304 B of scratch per lane)
        if constexpr (!INLINE_CORNER) return ec_dbl_affine(q);
        // doubling of the affine point ("mdbl-2008-s-1", a = 0)
        F u = F::dbl(q.y);
        F v = F::sqr(u);
        F w = F::mul(u, v);
        F s = F::mul(q.x, v);
        F xx = F::sqr(q.x);
        F m = F::add(F::dbl(xx), xx);
        o.x = F::sub(F::sqr(m), F::dbl(s));
        o.y = F::sub(F::mul(m, F::sub(s, o.x)), F::mul(w, q.y));
        o.zz = v;
        o.zzz = w;
        return o;
    }
    F pp = F::sqr(p);
    F ppp = F::mul(p, pp);
    F qq = F::mul(a.x, pp);
    o.x = F::sub(F::sub(F::sqr(r), ppp), F::dbl(qq));
    o.y = F::sub(F::mul(r, F::sub(qq, o.x)), F::mul(a.y, ppp));
    o.zz = F::mul(a.zz, pp);
    o.zzz = F::mul(a.zzz, ppp);
    return o;
}

// a + b, both XYZZ ("add-2008-s")
template <class F>
HK_HD XYZZ<F> ec_add(const XYZZ<F>& a, const XYZZ<F>& b) {
    if (b.is_inf()) return a;
    if (a.is_inf()) return b;
    F u1 = F::mul(a.x, b.zz);
    F u2 = F::mul(b.x, a.zz);
    F s1 = F::mul(a.y, b.zzz);
    F s2 = F::mul(b.y, a.zzz);
    F p = F::sub(u2, u1);
    F r = F::sub(s2, s1);
    if (p.is_zero()) {
        if (r.is_zero()) return ec_dbl_rare(a);
        return XYZZ<F>::inf();
    }
    F pp = F::sqr(p);
    F ppp = F::mul(p, pp);
    F qq = F::mul(u1, pp);
    XYZZ<F> o;
    o.x = F::sub(F::sub(F::sqr(r), ppp), F::dbl(qq));
    o.y = F::sub(F::mul(r, F::sub(qq, o.x)), F::mul(s1, ppp));
    o.zz = F::mul(F::mul(a.zz, b.zz), pp);
    o.zzz = F::mul(F::mul(a.zzz, b.zzz), ppp);
    return o;
}

// out-of-line forms for the latency-bound tail kernels (keeps code size and compile time down)
template <class F>
HK_RARE XYZZ<F> ec_add_ni(const XYZZ<F>& a, const XYZZ<F>& b) { return ec_add(a, b); }
template <class F>
HK_RARE XYZZ<F> ec_dbl_ni(const XYZZ<F>& a) { return ec_dbl(a); }
template <class F>
HK_RARE XYZZ<F> ec_madd_ni(const XYZZ<F>& a, const Affine<F>& b) { return ec_madd(a, b); }
template <class F>
HK_RARE F f_mul_ni(const F& a, const F& b) { return F::mul(a, b); }

#if defined(__HIPCC__)
#define HK_NOUNROLL _Pragma("unroll 1")
#else
#define HK_NOUNROLL
#endif

template <class F>
HK_HD Affine<F> ec_neg(const Affine<F>& p) {
    Affine<F> r; r.x = p.x; r.y = F::neg(p.y);
    return r;
}
template <class F>
HK_HD XYZZ<F> ec_neg(const XYZZ<F>& p) {
    XYZZ<F> r = p; r.y = F::neg(p.y);
    return r;
}

// k * P for a small unsigned k (bucket-reduction weights)
template <class F>
HK_HD XYZZ<F> ec_mul_small(const XYZZ<F>& p, u32 k) {
    XYZZ<F> acc = XYZZ<F>::inf();
    HK_NOUNROLL for (int bit = 31; bit >= 0; bit--) {
        acc = ec_dbl_ni(acc);
        if ((k >> bit) & 1) acc = ec_add_ni(acc, p);
    }
    return acc;
}

// k * P for a canonical (non-Montgomery) multi-limb scalar, MSB-first double-and-add
template <class F, int NL>
HK_HD XYZZ<F> ec_mul_limbs(const XYZZ<F>& p, const u32 (&k)[NL]) {
    XYZZ<F> acc = XYZZ<F>::inf();
    HK_NOUNROLL for (int i = NL - 1; i >= 0; i--) {
        HK_NOUNROLL for (int bit = 31; bit >= 0; bit--) {
            acc = ec_dbl_ni(acc);
            if ((k[i] >> bit) & 1) acc = ec_add_ni(acc, p);
        }
    }
    return acc;
}

// ---- inversion and normalisation to affine ---------------------------------------------------
// a^(p-2): the first form, kept as the cross-check of the host tests (tests/test_capi_cpu.py)
template <class P>
HK_HD Fp<P> fp_inv_fermat(const Fp<P>& a) {
    // exponent limbs = modulus - 2 with borrow propagation
    u32 ex[P::N];
    u32 borrow = 2;
    for (int i = 0; i < P::N; i++) {
        u32 m = P::MOD[i];
        ex[i] = m - borrow;
        borrow = (m < borrow) ? 1u : 0u;
    }
    Fp<P> result = Fp<P>::one();
    HK_NOUNROLL for (int i = P::N - 1; i >= 0; i--) {
        u32 e = ex[i];
        HK_NOUNROLL for (int bit = 31; bit >= 0; bit--) {
            result = f_mul_ni(result, result);
            if ((e >> bit) & 1) result = f_mul_ni(result, a);
        }
    }
    return result;
}

// Binary extended Euclid without data-dependent branches inside an iteration (the plain form of Pornin, "Optimized Binary
// GCD for Modular Inversion", algorithm 1): with b odd,
//     if a is odd: { if a < b: (a, u, b, v) <- (b, v, a, u);  a <- a - b;  u <- u - v mod p }   a <- a / 2;  u <- u / 2 mod p
// keeps a = u y, b = v y (mod p) and ends with a = 0, b = gcd = 1, v = 1 / y after at most 2 len(p) - 1 iterations
// (~1.4 len(p) typically; a lane leaves the loop when ITS a is 0).  An iteration is ~14 N word instructions against the
// ~2 N^2 + of a Montgomery product, and there are fewer of them than the 1.5 len(p) products of a^(p-2): the inversion
// is the serial tail of every normalisation to affine (k_batch_affine after each fold / sweep, k_to_affine, k_finish) and
// of the final exponentiation - 0.35 ms per call on BN254 and 1.7 ms on BLS12-381 as a^(p-2).
// Works on the plain integer y = a R mod p of the Montgomery value; 1 / y = a^-1 R^-1, times R^3 (one Montgomery product
// by R^2 R^2 / R) gives a^-1 R.  0 maps to 0, as a^(p-2) does.
template <class P>
HK_RARE Fp<P> fp_inv(const Fp<P>& a_in) {
    constexpr int N = P::N;
    Fp<P> y = Fp<P>::canon(a_in);
    if (y.is_zero()) return Fp<P>::zero();
    u32 a[N], b[N], u[N], v[N];
    HK_UNROLL for (int i = 0; i < N; i++) { a[i] = y.v[i]; b[i] = P::MOD[i]; u[i] = i == 0; v[i] = 0; }
    for (;;) {
        u32 nz = 0;
        HK_UNROLL for (int i = 0; i < N; i++) nz |= a[i];
        if (!nz) break;
        u32 odd = 0u - (a[0] & 1u);                      // all-ones when a is odd
        // d = a - b, borrow -> a < b
        u32 d[N];
        u64 br = 0;
        HK_UNROLL for (int i = 0; i < N; i++) {
            u64 t = (u64)a[i] - b[i] - br;
            d[i] = (u32)t;
            br = (t >> 32) & 1u;
        }
        u32 lt = (0u - (u32)br) & odd;                   // swap: a odd and a < b
        u32 ge = odd & ~lt;
        // a odd, a >= b: a <- d;   a odd, a < b: b <- a, a <- -d (= b - a);   a even: unchanged
        u64 c = 1;
        HK_UNROLL for (int i = 0; i < N; i++) {
            c += (u64)(u32)~d[i];
            u32 nd = (u32)c;
            c >>= 32;
            u32 ai = a[i];
            a[i] = (d[i] & ge) | (nd & lt) | (ai & ~odd);
            b[i] = (ai & lt) | (b[i] & ~lt);
        }
        // (u, v) <- (v, u) on a swap, then u <- u - v mod p when a was odd
        u64 sb = 0;
        u32 w[N];
        HK_UNROLL for (int i = 0; i < N; i++) {
            u32 ui = u[i], vi = v[i];
            u32 un = (vi & lt) | (ui & ~lt);
            v[i] = (ui & lt) | (vi & ~lt);
            u64 t = (u64)un - (v[i] & odd) - sb;
            w[i] = (u32)t;
            sb = (t >> 32) & 1u;
        }
        u32 fix = 0u - (u32)sb;                          // negative: add p back
        u64 cc = 0;
        HK_UNROLL for (int i = 0; i < N; i++) {
            cc += (u64)w[i] + (P::MOD[i] & fix);
            w[i] = (u32)cc;
            cc >>= 32;
        }
        // a <- a / 2;  u <- u / 2 mod p = (u odd ? u + p : u) >> 1, the sum kept with its carry bit
        u32 uo = 0u - (w[0] & 1u);
        u64 hc = 0;
        HK_UNROLL for (int i = 0; i < N; i++) {
            hc += (u64)w[i] + (P::MOD[i] & uo);
            w[i] = (u32)hc;
            hc >>= 32;
        }
        HK_UNROLL for (int i = 0; i < N; i++) {
            u32 hi_u = i + 1 < N ? w[i + 1] : (u32)hc;
            u32 hi_a = i + 1 < N ? a[i + 1] : 0u;
            u[i] = (w[i] >> 1) | (hi_u << 31);
            a[i] = (a[i] >> 1) | (hi_a << 31);
        }
    }
    Fp<P> r;
    HK_UNROLL for (int i = 0; i < N; i++) r.v[i] = v[i];
    Fp<P> r3 = f_mul_ni(Fp<P>::r2(), Fp<P>::r2());
    return f_mul_ni(r, r3);
}
template <class P>
HK_HD Fp2<P> fp_inv(const Fp2<P>& a) {
    typedef Fp<P> B;
    B n = B::add(f_mul_ni(a.c0, a.c0), f_mul_ni(a.c1, a.c1));
    B ni = fp_inv(n);
    Fp2<P> r;
    r.c0 = f_mul_ni(a.c0, ni);
    r.c1 = B::neg(f_mul_ni(a.c1, ni));
    return r;
}

template <class F>
HK_HD Affine<F> ec_to_affine(const XYZZ<F>& p) {
    if (p.is_inf()) return Affine<F>::inf();
    // 1/zzz, then 1/zz = zzz^-2 * zz^2  (zz^3 = zzz^2  =>  zz^-1 = zz^2 / zzz^2)
    F zzz_inv = fp_inv(p.zzz);
    F zz_inv = f_mul_ni(f_mul_ni(zzz_inv, zzz_inv), f_mul_ni(p.zz, p.zz));
    Affine<F> r;
    r.x = f_mul_ni(p.x, zz_inv);
    r.y = f_mul_ni(p.y, zzz_inv);
    return r;
}

}  // namespace hk
