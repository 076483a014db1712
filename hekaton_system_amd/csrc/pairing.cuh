// pairing.cuh — optimal-ate multi-pairing for BN254 / BLS12-381 on CDNA4.
//
// Replaces `pairing(left, right)` of distributed-prover/src/pairing_ops.rs:9-29
// (`E::multi_miller_loop(G1Prepared, G2Prepared)` + `E::final_exponentiation`), which the aggregator calls 16 times
// per job for the cross terms (aggregation.rs:255-263) and inside every IPP commitment.
// GT values are unique field elements: the product of Miller values is raised to ark's exponent
// (lambda * (q^12 - 1)/r with lambda = 2x(6x^2+3x+1) on BN254, 3 on BLS12-381 — the addition chains of ark-ec
// `bn::final_exponentiation` / `bls12::final_exponentiation`), so any schedule that multiplies the same lines gives
// ark's bytes.  The schedule here is for the GPU, not ark's shared-squaring loop:
//   k_pair_miller   one lane per (G1, G2) pair runs that pair's whole Miller loop (projective G2 steps as in ark's
//                   G2Prepared, D-/M-type sparse line products) — N independent lanes, no shared accumulator;
//   k_f12_product   one workgroup per product: strided serial products, then an LDS tree -> one Fq12;
//   k_final_exp     one lane per product: easy part (one Fq12 inversion), hard part by ark's chain.
// Several products are batched in one call (hk_pairing_products: every lhs vector x every rhs vector).
#pragma once
#include "tower.cuh"

namespace hk {

// ---- G2 line steps, homogeneous projective (X, Y, Z) (ark-ec bn/g2.rs, bls12/g2.rs) -------------------------
// Generic over the Fq2 representation F (Fp2<P>: one lane per value; Fp2Q<P>, endo.cuh: a quad of lanes per value - the
// products go through the f2m / f2s overloads of F).
template <class F> struct G2ProjF { F x, y, z; };
template <class F> struct LineCoeffsF { F c0, c1, c2; };
template <class P> using G2Proj = G2ProjF<Fp2<P>>;
template <class P> using LineCoeffs = LineCoeffsF<Fp2<P>>;
template <class F>
HK_HD F f2_const(const u32 (&l)[2][F::Params::N]) {
    F r;
    HK_UNROLL for (int i = 0; i < F::Params::N; i++) { r.c0.v[i] = l[0][i]; r.c1.v[i] = l[1][i]; }
    return r;
}

template <class F>
HK_RARE LineCoeffsF<F> pair_doubling_step(G2ProjF<F>& r) {
    typedef typename F::Params P;
    typedef TowerParams<P> T;
    // the two divisions by 2 are halvings (add p if odd, shift), not products by 1/2: 4 of the step's 28 base-field products
    F a = F::halve(f2m(r.x, r.y));
    F b = f2s(r.y), c = f2s(r.z);
    F c3 = F::add(F::add(c, c), c);
    F e;
    if constexpr (T::B_TWIST_IS_4_4) {                       // b' = 4 (1 + i): (u + v i) b' = 4 (u - v) + 4 (u + v) i
        F t;
        t.c0 = Fp<P>::sub(c3.c0, c3.c1);
        t.c1 = Fp<P>::add(c3.c0, c3.c1);
        e = F::dbl(F::dbl(t));
    } else {
        e = f2m(f2_const<F>(T::B_TWIST), c3);
    }
    F f = F::add(F::add(e, e), e);
    F g = F::halve(F::add(b, f));
    F h = F::sub(f2s(F::add(r.y, r.z)), F::add(b, c));
    F i = F::sub(e, b);
    F j = f2s(r.x);
    F e_sq = f2s(e);
    r.x = f2m(a, F::sub(b, f));
    r.y = F::sub(f2s(g), F::add(F::add(e_sq, e_sq), e_sq));
    r.z = f2m(b, h);
    F j3 = F::add(F::add(j, j), j);
    LineCoeffsF<F> l;
    if constexpr (T::TWIST_IS_D) { l.c0 = F::neg(h); l.c1 = j3; l.c2 = i; }
    else { l.c0 = i; l.c1 = j3; l.c2 = F::neg(h); }
    return l;
}

template <class F>
HK_RARE LineCoeffsF<F> pair_addition_step(G2ProjF<F>& r, const Affine<F>& q) {
    typedef typename F::Params P;
    typedef TowerParams<P> T;
    F theta = F::sub(r.y, f2m(q.y, r.z));
    F lambda = F::sub(r.x, f2m(q.x, r.z));
    F c = f2s(theta), d = f2s(lambda);
    F e = f2m(lambda, d);
    F f = f2m(r.z, c);
    F g = f2m(r.x, d);
    F h = F::sub(F::add(e, f), F::add(g, g));
    F ny = F::sub(f2m(theta, F::sub(g, h)), f2m(e, r.y));
    r.x = f2m(lambda, h);
    r.y = ny;
    r.z = f2m(r.z, e);
    F j = F::sub(f2m(theta, q.x), f2m(lambda, q.y));
    LineCoeffsF<F> l;
    if constexpr (T::TWIST_IS_D) { l.c0 = lambda; l.c1 = F::neg(theta); l.c2 = j; }
    else { l.c0 = j; l.c1 = F::neg(theta); l.c2 = lambda; }
    return l;
}

// f * line(P): the line as a sparse Fq12.  D-type (ark mul_by_034): c0 + (c1 + c2 v) w with c0 *= P.y, c1 *= P.x;
// M-type (ark mul_by_014): (c0 + c1 v) + (c2 v) w with c2 *= P.y, c1 *= P.x.
template <class P>
HK_RARE Fp12<P> pair_ell(const Fp12<P>& f, const LineCoeffs<P>& l, const Affine<Fp<P>>& p) {
    typedef TowerParams<P> T;
    Fp12<P> s;
    Fp2<P> z = Fp2<P>::zero();
    if constexpr (T::TWIST_IS_D) {
        s.c0.c0 = f2_scale(l.c0, p.y); s.c0.c1 = z; s.c0.c2 = z;
        s.c1.c0 = f2_scale(l.c1, p.x); s.c1.c1 = l.c2; s.c1.c2 = z;
    } else {
        s.c0.c0 = l.c0; s.c0.c1 = f2_scale(l.c1, p.x); s.c0.c2 = z;
        s.c1.c0 = z; s.c1.c1 = f2_scale(l.c2, p.y); s.c1.c2 = z;
    }
    return f12_mul(f, s);
}

template <class F>
HK_HD Affine<F> pair_mul_by_char(const Affine<F>& q) {       // ark bn/g2.rs mul_by_char
    typedef TowerParams<typename F::Params> T;
    Affine<F> r;
    r.x = f2m(f2_conj(q.x), f2_const<F>(T::MUL_BY_Q_X));
    r.y = f2m(f2_conj(q.y), f2_const<F>(T::MUL_BY_Q_Y));
    return r;
}

// psi(Q) = [q mod r] Q on G2 (the untwist-Frobenius-twist endomorphism; eigenvalue 6x^2 on BN254, x on BLS12-381):
// what lets a scalar multiplication in G2 run over four ~64-bit sub-scalars (hk_points_fold_g2).  (0, 0) -> (0, 0).
template <class P>
HK_HD Affine<Fp2<P>> g2_psi(const Affine<Fp2<P>>& q) {
    typedef TowerParams<P> T;
    Affine<Fp2<P>> r;
    r.x = f2m(f2_conj(q.x), fp2_const<P>(T::PSI_X));
    r.y = f2m(f2_conj(q.y), fp2_const<P>(T::PSI_Y));
    return r;
}

// (Round 2 applied psi / phi in kernels of their own, k_points_psi4 / k_points_phi2; the first form of k_points_psi4 is the
// kernel hipcc miscompiled - DESIGN.md section 3b, tests/golden/isa/.  The images are now built inside the fold kernels of
// endo.cuh, which call g2_psi / EndoOf<F>::apply per element.)

// Miller value of ONE pair (1 when either member is infinity: ark's multi_miller_loop skips such pairs)
template <class P>
HK_RARE Fp12<P> pair_miller_one(const Affine<Fp<P>>& p, const Affine<Fp2<P>>& q, const PairLoop& loop) {
    typedef TowerParams<P> T;
    Fp12<P> f = f12_one<P>();
    if (p.is_inf() || q.is_inf()) return f;
    G2Proj<P> r;
    r.x = q.x; r.y = q.y; r.z = Fp2<P>::one();
    Affine<Fp2<P>> nq = q;
    nq.y = Fp2<P>::neg(q.y);
    HK_NOUNROLL for (int i = loop.len - 1; i >= 1; i--) {
        if (i != loop.len - 1) f = f12_sqr(f);
        f = pair_ell(f, pair_doubling_step(r), p);
        int d = loop.digits[i - 1];
        if (d == 1) f = pair_ell(f, pair_addition_step(r, q), p);
        else if (d == -1) f = pair_ell(f, pair_addition_step(r, nq), p);
    }
    if constexpr (T::X_IS_NEGATIVE) f = f12_conj(f);
    if constexpr (T::TWIST_IS_D) {                      // BN: the two Frobenius line steps
        Affine<Fp2<P>> q1 = pair_mul_by_char(q);
        Affine<Fp2<P>> q2 = pair_mul_by_char(q1);
        q2.y = Fp2<P>::neg(q2.y);
        f = pair_ell(f, pair_addition_step(r, q1), p);
        f = pair_ell(f, pair_addition_step(r, q2), p);
    }
    return f;
}

// ---- final exponentiation (ark's addition chains; see oracle/pyref/pairing.py for the closed forms) -----------
template <class P>
HK_HD Fp12<P> pair_exp_by_x(const Fp12<P>& a) {              // bls12 `exp_by_x`: a^X, conjugated when X < 0
    Fp12<P> r = f12_pow_x(a);
    if constexpr (TowerParams<P>::X_IS_NEGATIVE) r = f12_conj(r);
    return r;
}
template <class P>
HK_HD Fp12<P> pair_exp_by_neg_x(const Fp12<P>& a) {          // bn `exp_by_neg_x`: a^X, conjugated when X > 0
    Fp12<P> r = f12_pow_x(a);
    if constexpr (!TowerParams<P>::X_IS_NEGATIVE) r = f12_conj(r);
    return r;
}

// `w`: workspace of PAIR_FEXP_WORDS Fq12 values (LDS in the kernel, so a lane's private memory holds only the
// temporaries of one product at a time)
constexpr int PAIR_FEXP_WORDS = 10;
template <class P>
HK_RARE Fp12<P> pair_final_exp(const Fp12<P>& f, Fp12<P>* w) {
    typedef TowerParams<P> T;
    // easy part: r = f^((q^6 - 1)(q^2 + 1))
    w[0] = f12_inv(f);
    w[0] = f12_mul(f12_conj(f), w[0]);
    w[0] = f12_mul(f12_frob<P, 2>(w[0]), w[0]);
    Fp12<P>& r = w[0];
    if constexpr (T::TWIST_IS_D) {
        // BN hard part (Fuentes-Castaneda et al.), ark-ec bn/mod.rs; y_k names as there
        Fp12<P>&y1 = w[1], &y3 = w[2], &y4 = w[3], &y6 = w[4], &y8 = w[5], &y9 = w[6], &t = w[7], &u = w[8];
        t = pair_exp_by_neg_x(r);            // y0
        y1 = f12_sqr(t);
        t = f12_sqr(y1);                     // y2
        y3 = f12_mul(t, y1);
        y4 = pair_exp_by_neg_x(y3);
        t = f12_sqr(y4);                     // y5
        y6 = pair_exp_by_neg_x(t);
        y3 = f12_conj(y3);
        y6 = f12_conj(y6);
        t = f12_mul(y6, y4);                 // y7
        y8 = f12_mul(t, y3);
        y9 = f12_mul(y8, y1);
        t = f12_mul(y8, y4);                 // y10
        t = f12_mul(t, r);                   // y11
        u = f12_frob<P, 1>(y9);              // y12
        t = f12_mul(u, t);                   // y13
        y8 = f12_frob<P, 2>(y8);
        t = f12_mul(y8, t);                  // y14
        r = f12_conj(r);
        u = f12_mul(r, y9);
        u = f12_frob<P, 3>(u);               // y15
        return f12_mul(u, t);
    } else {
        // BLS12 hard part (Hayashida-Hayasaka-Teruya, eprint 2020/875), ark-ec bls12/mod.rs
        Fp12<P>&y0 = w[1], &y1 = w[2], &y2 = w[3];
        y0 = f12_sqr(r);
        y1 = pair_exp_by_x(r);
        y2 = f12_conj(r);
        y1 = f12_mul(y1, y2);
        y2 = pair_exp_by_x(y1);
        y1 = f12_conj(y1);
        y1 = f12_mul(y1, y2);
        y2 = pair_exp_by_x(y1);
        y1 = f12_frob<P, 1>(y1);
        y1 = f12_mul(y1, y2);
        r = f12_mul(r, y0);
        y0 = pair_exp_by_x(y1);
        y2 = pair_exp_by_x(y0);
        y0 = f12_frob<P, 2>(y1);
        y1 = f12_conj(y1);
        y1 = f12_mul(y1, y2);
        y1 = f12_mul(y1, y0);
        return f12_mul(r, y1);
    }
}

#if defined(__HIPCC__)

constexpr int PAIR_TREE_THREADS = 128;

// g1: n_l vectors of n affine points, g2: n_r vectors of n; pair t = ((a * n_r + b) * n + i) multiplies into product
// (a, b).  out[t] = Miller value of (g1[a][i], g2[b][i]).
template <class P>
__global__ void __launch_bounds__(64)
k_pair_miller(const Affine<Fp<P>>* __restrict__ g1, const Affine<Fp2<P>>* __restrict__ g2, u32 n, u32 n_l, u32 n_r,
              PairLoop loop, Fp12<P>* __restrict__ out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)n * n_l * n_r;
    if (t >= total) return;
    u32 i = (u32)(t % n);
    u32 ab = (u32)(t / n);
    u32 a = ab / n_r, b = ab % n_r;
    Affine<Fp<P>> p = ld_vec(&g1[(size_t)a * n + i]);
    Affine<Fp2<P>> q = ld_vec(&g2[(size_t)b * n + i]);
    Fp12<P> f = pair_miller_one<P>(p, q, loop);
    st_vec(&out[t], f12_canon(f));
}

// one workgroup per product: out[prod] = prod_i in[prod * n + i]
template <class P>
__global__ void __launch_bounds__(PAIR_TREE_THREADS)
k_f12_product(const Fp12<P>* __restrict__ in, u32 n, Fp12<P>* __restrict__ out) {
    extern __shared__ unsigned char pair_lds[];
    Fp12<P>* sh = reinterpret_cast<Fp12<P>*>(pair_lds);
    const Fp12<P>* src = in + (size_t)blockIdx.x * n;
    Fp12<P> acc = f12_one<P>();
    bool first = true;
    for (u32 i = threadIdx.x; i < n; i += PAIR_TREE_THREADS) {
        Fp12<P> v = ld_vec(&src[i]);
        if (first) { acc = v; first = false; }
        else acc = f12_mul(acc, v);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (u32 off = PAIR_TREE_THREADS / 2; off >= 1; off >>= 1) {
        if (threadIdx.x < off) {
            Fp12<P> x = sh[threadIdx.x], y = sh[threadIdx.x + off];
            sh[threadIdx.x] = f12_mul(x, y);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) st_vec(&out[blockIdx.x], f12_canon(sh[0]));
}

// one workgroup (one active lane) per product; the chain's named temporaries live in LDS
template <class P>
__global__ void __launch_bounds__(64)
k_final_exp(const Fp12<P>* __restrict__ in, u32 count, Fp12<P>* __restrict__ out) {
    extern __shared__ unsigned char pair_lds[];
    Fp12<P>* w = reinterpret_cast<Fp12<P>*>(pair_lds);
    if (threadIdx.x != 0 || blockIdx.x >= count) return;
    Fp12<P> f = ld_vec(&in[blockIdx.x]);
    st_vec(&out[blockIdx.x], f12_canon(pair_final_exp<P>(f, w)));
}

#endif  // __HIPCC__

}  // namespace hk
