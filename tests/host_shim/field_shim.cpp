// Host-compiled view of the PRODUCT's field.cuh / ec.cuh (g++), so the arithmetic the HIP kernels
// run can be unit-tested against the oracle without a GPU.  Test-only; never part of libhekaton.
#include "../../hekaton_system_amd/csrc/ec.cuh"
#include "../../hekaton_system_amd/csrc/pairing.cuh"
#include "../../hekaton_system_amd/csrc/msm.cuh"
#include "../../hekaton_system_amd/csrc/endo.cuh"
#include <string.h>
using namespace hk;

template <class F> static void ld(F& f, const void* p) { memcpy(&f, p, sizeof(F)); }
template <class F> static void st(void* p, const F& f) { memcpy(p, &f, sizeof(F)); }

template <class P> static Fp<P> inv_fermat(const Fp<P>& x) { return fp_inv_fermat(x); }
template <class P> static Fp2<P> inv_fermat(const Fp2<P>& x) { return fp_inv(x); }

template <class F>
static void field_op(int op, const void* a, const void* b, void* out) {
    F x, y, r; ld(x, a); ld(y, b);
    switch (op) {
        case 0: r = F::add(x, y); break;
        case 1: r = F::sub(x, y); break;
        case 2: r = F::mul(x, y); break;
        case 3: r = F::neg(x); break;
        case 4: r = fp_inv(x); break;
        case 5: r = F::sqr(x); break;
        case 6: r = inv_fermat(x); break;
        default: r = F::zero();
    }
    st(out, r);
}
template <class F>
static void group_op(int op, const void* a, const void* b, void* out) {
    // a, b: affine; out: affine.  op 0: madd path, 1: full add path, 2: dbl, 3: small mul by *(u32*)b
    Affine<F> p, q; ld(p, a);
    XYZZ<F> r;
    if (op == 0) { ld(q, b); r = ec_madd(XYZZ<F>::from_affine(p), q); }
    else if (op == 1) {
        ld(q, b);
        // de-normalise p first so the full-add formulas see non-trivial zz/zzz
        XYZZ<F> pp = ec_dbl(XYZZ<F>::from_affine(p));
        pp = ec_madd(pp, ec_neg(p));
        r = ec_add(pp, ec_dbl(XYZZ<F>::from_affine(q)));
        r = ec_madd(r, ec_neg(q));
    }
    else if (op == 2) r = ec_dbl(ec_dbl(XYZZ<F>::from_affine(p)));
    else { u32 k; memcpy(&k, b, 4); r = ec_mul_small(XYZZ<F>::from_affine(p), k); }
    st(out, ec_to_affine(r));
}

// Fq12 ops on ark-ordered coordinates.  op 0 mul, 1 sqr, 2 inv, 3 conj, 4 frob1, 5 frob2, 6 frob3, 7 pow_x, 8 final exp
template <class P>
static void f12_op(int op, const void* a, const void* b, void* out) {
    Fp12<P> x, y, r; ld(x, a); if (b) ld(y, b);
    static Fp12<P> w[PAIR_FEXP_WORDS];
    switch (op) {
        case 0: r = f12_mul(x, y); break;
        case 1: r = f12_sqr(x); break;
        case 2: r = f12_inv(x); break;
        case 3: r = f12_conj(x); break;
        case 4: r = f12_frob<P, 1>(x); break;
        case 5: r = f12_frob<P, 2>(x); break;
        case 6: r = f12_frob<P, 3>(x); break;
        case 7: r = f12_pow_x(x); break;
        default: r = pair_final_exp<P>(x, w);
    }
    st(out, f12_canon(r));
}
// prod_i e(g1[i], g2[i]) exactly as the kernels compute it: per-pair Miller values, their product, final exponentiation
template <class P>
static void multi_pairing(const void* g1, const void* g2, size_t n, const PairLoop& loop, void* out) {
    static Fp12<P> w[PAIR_FEXP_WORDS];
    Fp12<P> acc = f12_one<P>();
    for (size_t i = 0; i < n; i++) {
        Affine<Fp<P>> p; Affine<Fp2<P>> q;
        ld(p, (const char*)g1 + i * sizeof(p)); ld(q, (const char*)g2 + i * sizeof(q));
        acc = f12_mul(acc, pair_miller_one<P>(p, q, loop));
    }
    st(out, f12_canon(pair_final_exp<P>(acc, w)));
}

// |k| * P exactly as one lane of k_points_mul_split computes it (csrc/endo.cuh): table 1P .. 8P by doublings and mixed adds,
// the magnitude biased by 0x88..8, signed 4-bit digits from the top, four Jacobian doublings and one table add per digit
template <class F>
static void split_mul(const void* pt, const unsigned* mag, void* out) {
    constexpr int ND = SplitDigits<F>::ND;
    Affine<F> q; ld(q, pt);
    u32 m[6];
    for (int l = 0; l < 6; l++) m[l] = mag[l];
    split_bias<ND>(m);
    Jac<F> tab[SPLIT_TABLE];
    Jac<F> acc = Jac<F>::inf();
    if (!q.is_inf()) {
        tab[0] = Jac<F>::from_affine(q);
        for (int k = 2; k <= SPLIT_TABLE; k++) tab[k - 1] = (k & 1) ? jac_madd_ni(tab[k - 2], q) : jac_dbl_ni(tab[k / 2 - 1]);
        for (int d = ND - 1; d >= 0; d--) {
            int dig = split_digit(m, d);
            for (int r = 0; r < 4; r++) acc = jac_dbl_ni(acc);
            if (dig != 0) {
                Jac<F> e = tab[(dig < 0 ? -dig : dig) - 1];
                if (dig < 0) e.y = F::neg(e.y);
                acc = jac_add_ni(acc, e);
            }
        }
    }
    XYZZ<F> o = XYZZ<F>::inf();
    if (!acc.is_inf()) { o.x = acc.x; o.y = acc.y; o.zz = F::sqr(acc.z); o.zzz = F::mul(o.zz, acc.z); }
    st(out, ec_to_affine(o));
}

extern "C" {
// group: 0 bn254 G1, 1 bn254 G2, 2 bls G1, 3 bls G2; mag: 6 limbs
void shim_split_mul(int group, const void* pt, const unsigned* mag, void* out) {
    switch (group) {
        case 0: split_mul<Fp<Bn254FqP>>(pt, mag, out); break;
        case 1: split_mul<Fp2<Bn254FqP>>(pt, mag, out); break;
        case 2: split_mul<Fp<Bls381FqP>>(pt, mag, out); break;
        case 3: split_mul<Fp2<Bls381FqP>>(pt, mag, out); break;
    }
}
// curve: 0 bn254, 1 bls12-381
void shim_f12_op(int curve, int op, const void* a, const void* b, void* out) {
    if (curve == 0) f12_op<Bn254FqP>(op, a, b, out); else f12_op<Bls381FqP>(op, a, b, out);
}
void shim_multi_pairing(int curve, const void* g1, const void* g2, size_t n, void* out) {
    if (curve == 0) multi_pairing<Bn254FqP>(g1, g2, n, pair_loop_bn254(), out);
    else multi_pairing<Bls381FqP>(g1, g2, n, pair_loop_bls381(), out);
}
// field: 0 bn254 Fr, 1 bn254 Fq, 2 bls Fr, 3 bls Fq, 4 bn254 Fq2, 5 bls Fq2
void shim_field_op(int field, int op, const void* a, const void* b, void* out) {
    switch (field) {
        case 0: field_op<Fp<Bn254FrP>>(op, a, b, out); break;
        case 1: field_op<Fp<Bn254FqP>>(op, a, b, out); break;
        case 2: field_op<Fp<Bls381FrP>>(op, a, b, out); break;
        case 3: field_op<Fp<Bls381FqP>>(op, a, b, out); break;
        case 4: field_op<Fp2<Bn254FqP>>(op, a, b, out); break;
        case 5: field_op<Fp2<Bls381FqP>>(op, a, b, out); break;
    }
}
// group: 0 bn254 G1, 1 bn254 G2, 2 bls G1, 3 bls G2
void shim_group_op(int group, int op, const void* a, const void* b, void* out) {
    switch (group) {
        case 0: group_op<Fp<Bn254FqP>>(op, a, b, out); break;
        case 1: group_op<Fp2<Bn254FqP>>(op, a, b, out); break;
        case 2: group_op<Fp<Bls381FqP>>(op, a, b, out); break;
        case 3: group_op<Fp2<Bls381FqP>>(op, a, b, out); break;
    }
}
// the device-side scalar split of csrc/endo.cuh.  group: 1 (phi, 2 parts) or 2 (psi, 4 parts); c: canonical scalar, 8 limbs;
// mag: parts x 6 limbs; returns the sign mask
unsigned shim_endo_decompose(int curve, int group, const unsigned* c, unsigned* mag) {
    u32 cc[8];
    for (int i = 0; i < 8; i++) cc[i] = c[i];
    if (group == 1) {
        EndoSplit<2> E = curve == 0 ? endo_split_g1((const Bn254FqP*)nullptr) : endo_split_g1((const Bls381FqP*)nullptr);
        u32 m[2][6];
        u32 neg = endo_decompose<2>(cc, E, m);
        memcpy(mag, m, sizeof(m));
        return neg;
    }
    EndoSplit<4> E = curve == 0 ? endo_split_g2((const Bn254FqP*)nullptr) : endo_split_g2((const Bls381FqP*)nullptr);
    u32 m[4][6];
    u32 neg = endo_decompose<4>(cc, E, m);
    memcpy(mag, m, sizeof(m));
    return neg;
}
// the MSM schedule (host code of csrc/msm.cuh): windows, buckets, level lanes of a plan
// out[0..5] = W, F, NB, n_levels, T[0], chunk; curve: 0 bn254, 1 bls12-381
void shim_msm_plan(int curve, unsigned n, unsigned c, unsigned WP, unsigned max_lanes0, unsigned* out) {
    MsmPlan p = curve == 0 ? msm_make_plan(n, 254, c, WP, max_lanes0, Bn254FrP::MOD, Bn254FrP::N)
                           : msm_make_plan(n, 255, c, WP, max_lanes0, Bls381FrP::MOD, Bls381FrP::N);
    out[0] = p.W; out[1] = p.F; out[2] = p.NB; out[3] = p.n_levels; out[4] = p.T[0]; out[5] = p.chunk;
    for (int i = 0; i < 10; i++) out[6 + i] = p.kconst[i];
}
}
