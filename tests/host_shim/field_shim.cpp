// Host-compiled view of the PRODUCT's field.cuh / ec.cuh (g++), so the arithmetic the HIP kernels
// run can be unit-tested against the oracle without a GPU.  Test-only; never part of libhekaton.
#include "../../hekaton_system_amd/csrc/ec.cuh"
#include <string.h>
using namespace hk;

template <class F> static void ld(F& f, const void* p) { memcpy(&f, p, sizeof(F)); }
template <class F> static void st(void* p, const F& f) { memcpy(p, &f, sizeof(F)); }

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

extern "C" {
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
}
