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
#endif  // __HIPCC__ (the one-lane kernels)

// ---- K lanes per element ------------------------------------------------------------------------------------------------
// A sweep over a few hundred elements is ONE wave per SIMD at most: its time is the length of one lane's dependent chain,
// and a chain of 68 / 131 x (double + add) cannot be shortened on one lane.  So for short vectors (n K <= SPLIT_MAX_LANES)
// the K sub-scalars of an element go to K neighbouring lanes: lane j multiplies endo^j(P) by |k_j| alone, which lets it use
// a 4-bit signed window (table 1P .. 8P of its own image: 4 doublings + 3 mixed adds; then one add per FOUR doublings
// instead of one per doubling) and Jacobian coordinates (a doubling is 2M + 5S instead of XYZZ's 6M + 3S - the chain is
// four fifths doublings now).  Per lane, in base-field products for G2: 151 (table) + 18 x (4 x 16 + 43) against
// 68 x (24 + 40) + 440 on one lane; the K partial products meet in LDS.  Total work per element is HIGHER (K tables), so
// long vectors keep the one-lane kernels above.
template <class F>
struct Jac {
    F x, y, z;                                   // (X / Z^2, Y / Z^3); Z = 0: infinity
    HK_HD bool is_inf() const { return z.is_zero(); }
    HK_HD static Jac inf() { Jac r; r.x = F::one(); r.y = F::one(); r.z = F::zero(); return r; }
    HK_HD static Jac from_affine(const Affine<F>& p) {
        if (p.is_inf()) return inf();
        Jac r; r.x = p.x; r.y = p.y; r.z = F::one();
        return r;
    }
};

// "dbl-2009-l" (a = 0): 2M + 5S; infinity maps to infinity (Z3 = 2 Y Z)
template <class F>
HK_RARE Jac<F> jac_dbl_ni(const Jac<F>& p) {
    F a = F::sqr(p.x), b = F::sqr(p.y), c = F::sqr(b);
    F d = F::dbl(F::sub(F::sub(F::sqr(F::add(p.x, b)), a), c));
    F e = F::add(F::dbl(a), a);
    Jac<F> r;
    r.x = F::sub(F::sqr(e), F::dbl(d));
    F c8 = F::dbl(F::dbl(F::dbl(c)));
    r.y = F::sub(F::mul(e, F::sub(d, r.x)), c8);
    r.z = F::dbl(F::mul(p.y, p.z));
    return r;
}

// "add-2007-bl", all exceptional cases handled
template <class F>
HK_RARE Jac<F> jac_add_ni(const Jac<F>& p, const Jac<F>& q) {
    if (q.is_inf()) return p;
    if (p.is_inf()) return q;
    F z1z1 = F::sqr(p.z), z2z2 = F::sqr(q.z);
    F u1 = F::mul(p.x, z2z2), u2 = F::mul(q.x, z1z1);
    F s1 = F::mul(F::mul(p.y, q.z), z2z2), s2 = F::mul(F::mul(q.y, p.z), z1z1);
    F h = F::sub(u2, u1), rr = F::sub(s2, s1);
    if (h.is_zero()) {
        if (rr.is_zero()) return jac_dbl_ni(p);
        return Jac<F>::inf();
    }
    rr = F::dbl(rr);
    F i = F::sqr(F::dbl(h));
    F j = F::mul(h, i);
    F v = F::mul(u1, i);
    Jac<F> r;
    r.x = F::sub(F::sub(F::sqr(rr), j), F::dbl(v));
    r.y = F::sub(F::mul(rr, F::sub(v, r.x)), F::dbl(F::mul(s1, j)));
    r.z = F::mul(F::sub(F::sub(F::sqr(F::add(p.z, q.z)), z1z1), z2z2), h);
    return r;
}

// "madd-2007-bl" with an affine, non-infinity q; all exceptional cases handled
template <class F>
HK_RARE Jac<F> jac_madd_ni(const Jac<F>& p, const Affine<F>& q) {
    if (p.is_inf()) return Jac<F>::from_affine(q);
    F z1z1 = F::sqr(p.z);
    F u2 = F::mul(q.x, z1z1);
    F s2 = F::mul(F::mul(q.y, p.z), z1z1);
    F h = F::sub(u2, p.x), rr = F::sub(s2, p.y);
    if (h.is_zero()) {
        if (rr.is_zero()) return jac_dbl_ni(p);
        return Jac<F>::inf();
    }
    rr = F::dbl(rr);
    F hh = F::sqr(h);
    F i = F::dbl(F::dbl(hh));
    F j = F::mul(h, i);
    F v = F::mul(p.x, i);
    Jac<F> r;
    r.x = F::sub(F::sub(F::sqr(rr), j), F::dbl(v));
    r.y = F::sub(F::mul(rr, F::sub(v, r.x)), F::dbl(F::mul(p.y, j)));
    r.z = F::sub(F::sub(F::sqr(F::add(p.z, h)), z1z1), hh);
    return r;
}

constexpr u32 SPLIT_MAX_LANES = 65536;           // n K (x 4 in the quad form) up to one wave on every SIMD of the chip
constexpr int SPLIT_TABLE = 8;                   // 1P .. 8P

// bytes of the table scratch of k_points_mul_split for n elements
template <class F> inline size_t split_tab_bytes(size_t n) { return (size_t)SPLIT_TABLE * EndoOf<F>::K * n * sizeof(Jac<F>); }
// bytes of the table scratch either form (one lane / K lanes per element) may ask for
template <class F> inline size_t endo_tab_bytes(size_t n, size_t vectors = 1) {
    size_t one = ((size_t)1 << EndoOf<F>::K) * n * sizeof(XYZZ<F>);
    size_t split = n * EndoOf<F>::K <= SPLIT_MAX_LANES ? vectors * split_tab_bytes<F>(n) : 0;
    return (one > split ? one : split) + 256;
}

// Signed 4-bit digits without a carry chain at run time: m' = m + 0x88..8 over ND nibbles (ND = ceil(bits / 4) + 1, so the
// top nibble of m' is 8 or 9), digit d = nibble_d(m') - 8 in -8 .. 7, and m = sum_d digit_d 16^d.
template <int ND>
HK_HD void split_bias(u32 (&m)[6]) {
    u64 carry = 0;
    for (int l = 0; l < 6; l++) {
        int nib = ND - 8 * l;                                // nibbles of the bias inside limb l
        u32 bias = nib >= 8 ? 0x88888888u : nib <= 0 ? 0u : (0x88888888u & ((1u << (4 * nib)) - 1u));
        carry += (u64)m[l] + bias;
        m[l] = (u32)carry;
        carry >>= 32;
    }
}
HK_HD int split_digit(const u32 (&m)[6], int d) { return (int)((m[d >> 3] >> (4 * (d & 7))) & 15u) - 8; }
template <class F> struct SplitDigits { static constexpr int ND = (EndoOf<F>::STEPS + 3) / 4 + 1; };

#if defined(__HIPCC__)
// ---- Fq2 on FOUR lanes -------------------------------------------------------------------------------------------------
// The lanes of a quad (4 t .. 4 t + 3) hold the SAME Fq2 value; a product is ONE base-field product per lane - lane 0:
// a0 b0, lane 1: a1 b1, lanes 2, 3: (a0 + a1)(b0 + b1) - quad-broadcast (DPP quad_perm, 3 N moves) and recombined by every
// lane: the 3 (product) / 2 (square) sequential base-field products of Fq2 become one.  Additions run redundantly.  Same
// memory layout as Fp2<P>; every control decision depends on values all four lanes share, so a quad never diverges.
// Only for chains whose time is one lane's latency (few hundred elements): total work is 4 / 1.7 times the one-lane form's.
template <class P>
struct Fp2Q {
    typedef Fp<P> B;
    typedef P Params;
    static constexpr int N = 2 * P::N;
    B c0, c1;

    __device__ __forceinline__ static Fp2Q zero() { Fp2Q r; r.c0 = B::zero(); r.c1 = B::zero(); return r; }
    __device__ __forceinline__ static Fp2Q one() { Fp2Q r; r.c0 = B::one(); r.c1 = B::zero(); return r; }
    __device__ __forceinline__ bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    __device__ __forceinline__ static Fp2Q canon(const Fp2Q& a) { Fp2Q r; r.c0 = B::canon(a.c0); r.c1 = B::canon(a.c1); return r; }
    __device__ __forceinline__ static Fp2Q add(const Fp2Q& a, const Fp2Q& b) { Fp2Q r; r.c0 = B::add(a.c0, b.c0); r.c1 = B::add(a.c1, b.c1); return r; }
    __device__ __forceinline__ static Fp2Q sub(const Fp2Q& a, const Fp2Q& b) { Fp2Q r; r.c0 = B::sub(a.c0, b.c0); r.c1 = B::sub(a.c1, b.c1); return r; }
    __device__ __forceinline__ static Fp2Q dbl(const Fp2Q& a) { return add(a, a); }
    __device__ __forceinline__ static Fp2Q neg(const Fp2Q& a) { Fp2Q r; r.c0 = B::neg(a.c0); r.c1 = B::neg(a.c1); return r; }
    __device__ __forceinline__ static Fp2Q conj(const Fp2Q& a) { Fp2Q r; r.c0 = a.c0; r.c1 = B::neg(a.c1); return r; }
    __device__ __forceinline__ static Fp2Q halve(const Fp2Q& a) { Fp2Q r; r.c0 = B::halve(a.c0); r.c1 = B::halve(a.c1); return r; }
    template <int CTRL>
    __device__ __forceinline__ static B quad_bcast(const B& v) {
        B r;
        HK_UNROLL for (int i = 0; i < P::N; i++)
            r.v[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)v.v[i], CTRL, 0xf, 0xf, true);
        return r;
    }
    __device__ __forceinline__ static B pick(u32 role, const B& r0, const B& r1, const B& r2) {
        B r;
        HK_UNROLL for (int i = 0; i < P::N; i++) r.v[i] = role == 0 ? r0.v[i] : role == 1 ? r1.v[i] : r2.v[i];
        return r;
    }
    __device__ __forceinline__ static Fp2Q mul(const Fp2Q& a, const Fp2Q& b) {
        u32 role = threadIdx.x & 3u;
        B x = pick(role, a.c0, a.c1, B::add(a.c0, a.c1));
        B y = pick(role, b.c0, b.c1, B::add(b.c0, b.c1));
        B v = B::mul(x, y);
        B v0 = quad_bcast<0x00>(v), v1 = quad_bcast<0x55>(v), v2 = quad_bcast<0xAA>(v);
        Fp2Q r;
        r.c0 = B::sub(v0, v1);
        r.c1 = B::sub(B::sub(v2, v0), v1);
        return r;
    }
    __device__ __forceinline__ static Fp2Q sqr(const Fp2Q& a) {       // lanes 0, 2: (a0 + a1)(a0 - a1); lanes 1, 3: a0 a1
        bool odd = threadIdx.x & 1u;
        B s = B::add(a.c0, a.c1), d = B::sub(a.c0, a.c1);
        B x = pick(odd ? 1u : 0u, s, a.c0, s);
        B y = pick(odd ? 1u : 0u, d, a.c1, d);
        B v = B::mul(x, y);
        Fp2Q r;
        r.c0 = quad_bcast<0x00>(v);
        r.c1 = B::dbl(quad_bcast<0x55>(v));
        return r;
    }
};
// the names the generic line steps of pairing.cuh call
template <class P> __device__ __forceinline__ Fp2Q<P> f2m(const Fp2Q<P>& a, const Fp2Q<P>& b) { return Fp2Q<P>::mul(a, b); }
template <class P> __device__ __forceinline__ Fp2Q<P> f2s(const Fp2Q<P>& a) { return Fp2Q<P>::sqr(a); }
template <class P> __device__ __forceinline__ Fp2Q<P> f2_conj(const Fp2Q<P>& a) { return Fp2Q<P>::conj(a); }
template <class P>
__device__ __forceinline__ void st_vec(Fp2Q<P>* p, const Fp2Q<P>& v) {
    st_vec(&p->c0, v.c0);
    st_vec(&p->c1, v.c1);
}
template <class P> struct EndoOf<Fp2Q<P>> {
    static constexpr int K = 4, STEPS = 68;
    __device__ static Affine<Fp2Q<P>> apply(const Affine<Fp2Q<P>>& q) {          // g2_psi (pairing.cuh) on quad values
        typedef TowerParams<P> T;
        typedef Fp2Q<P> F;
        F kx, ky;
        HK_UNROLL for (int i = 0; i < P::N; i++) {
            kx.c0.v[i] = T::PSI_X[0][i]; kx.c1.v[i] = T::PSI_X[1][i];
            ky.c0.v[i] = T::PSI_Y[0][i]; ky.c1.v[i] = T::PSI_Y[1][i];
        }
        Affine<F> r;
        r.x = F::mul(F::conj(q.x), kx);
        r.y = F::mul(F::conj(q.y), ky);
        return r;
    }
};
// lanes that carry one value of the field: 1, or 4 for the quad form
template <class F> struct LanesPerValue { static constexpr int value = 1; };
template <class P> struct LanesPerValue<Fp2Q<P>> { static constexpr int value = 4; };

// out[i] = (lo ? lo[i] : 0) + s_i * pts[i].   UNIFORM: one scalar for the whole vector, split on the host: `scalars` = K
// Montgomery magnitudes, neg_all = their sign mask (the fold of a TIPA round);  else scalars[i], split here by every lane
// of the element (the split is ~300 integer products: cheaper than passing it between lanes).
// grid: (ceil(n K L / 64), vectors) blocks of 64 lanes (L = 1, or 4 with the quad field Fp2Q: t counts quads); lane / quad t:
// element t / K, part t % K; vector y = blockIdx.y reads
// v.lo[y] / v.pts[y] and writes out[y n ..) (UNIFORM: the same scalar for every vector - the folds of one TIPA round).
// tab: vectors x split_tab_bytes(n).
constexpr int FOLD_MAX = 4;
// pts_mod != 0: element i reads pts[i % pts_mod] - one short base set under many scalar vectors (hk_commit_batch)
template <class F> struct SplitVecs { const Affine<F>* lo[FOLD_MAX]; const Affine<F>* pts[FOLD_MAX]; u32 pts_mod; };

template <class Fr, class F, bool UNIFORM>
__global__ void __launch_bounds__(64)
k_points_mul_split(SplitVecs<F> v, const Fr* __restrict__ scalars, u32 neg_all, u32 n, EndoSplit<EndoOf<F>::K> E,
                   Jac<F>* __restrict__ tab, XYZZ<F>* __restrict__ out, int scalars_mont) {
    constexpr int K = EndoOf<F>::K;
    constexpr int ND = SplitDigits<F>::ND;                        // nibbles of m'
    constexpr int L = LanesPerValue<F>::value;                    // 4: every value on a quad of lanes (Fp2Q), t = the quad
    __shared__ Jac<F> sh[64 / L];
    const Affine<F>* __restrict__ lo = v.lo[blockIdx.y];
    const Affine<F>* __restrict__ pts = v.pts[blockIdx.y];
    const u32 tl = threadIdx.x / L;
    u32 t = blockIdx.x * (64 / L) + tl;
    u32 i = t / K, j = t % K;
    bool valid = i < n;
    size_t lanes = (size_t)n * K;
    tab += (size_t)blockIdx.y * SPLIT_TABLE * lanes;
    out += (size_t)blockIdx.y * n;
    Jac<F> acc = Jac<F>::inf();
    if (valid) {
        u32 m[6];
        bool negate;
        if (UNIFORM) {
            Fr c = Fr::from_mont(ld_vec(&scalars[j]));
            HK_UNROLL for (int l = 0; l < 6; l++) m[l] = c.v[l];
            negate = (neg_all >> j) & 1u;
        } else {
            Fr s = ld_vec(&scalars[i]);
            if (scalars_mont) s = Fr::from_mont(s);                  // else: canonical integers < r (hk_msm with mont = 0)
            u32 c[8];
            HK_UNROLL for (int l = 0; l < 8; l++) c[l] = s.v[l];
            u32 mag[K][6];
            u32 neg = endo_decompose<K>(c, E, mag);
            HK_UNROLL for (int l = 0; l < 6; l++) {
                u32 v = 0;
                HK_UNROLL for (int k = 0; k < K; k++) v = j == (u32)k ? mag[k][l] : v;
                m[l] = v;
            }
            negate = (neg >> j) & 1u;
        }
        split_bias<ND>(m);
        Affine<F> q = ld_vec(&pts[v.pts_mod ? i % v.pts_mod : i]);
        if (!q.is_inf()) {
            HK_NOUNROLL for (u32 k = 0; k < j; k++) q = EndoOf<F>::apply(q);
            if (negate) q.y = F::neg(q.y);
            // 1P .. 8P: 2 = dbl 1, 3 = 2 + P, 4 = dbl 2, 5 = 4 + P, 6 = dbl 3, 7 = 6 + P, 8 = dbl 4
            Jac<F> e = Jac<F>::from_affine(q);
            st_vec(&tab[0 * lanes + t], e);
            HK_NOUNROLL for (int k = 2; k <= SPLIT_TABLE; k++) {
                if (k & 1) e = jac_madd_ni(ld_vec(&tab[(size_t)(k - 2) * lanes + t]), q);
                else e = jac_dbl_ni(ld_vec(&tab[(size_t)(k / 2 - 1) * lanes + t]));
                st_vec(&tab[(size_t)(k - 1) * lanes + t], e);
            }
        }
        bool q_inf = q.is_inf();
        HK_NOUNROLL for (int d = ND - 1; d >= 0; d--) {
            int dig = split_digit(m, d);
            bool idle = q_inf || (dig == 0 && acc.is_inf());
            if (__all(idle)) continue;                           // leading zero digits of the whole wave
            HK_NOUNROLL for (int r = 0; r < 4; r++) acc = jac_dbl_ni(acc);
            if (dig != 0 && !q_inf) {
                u32 a = dig < 0 ? (u32)(-dig) : (u32)dig;
                Jac<F> e = ld_vec(&tab[(size_t)(a - 1) * lanes + t]);
                if (dig < 0) e.y = F::neg(e.y);
                acc = jac_add_ni(acc, e);
            }
        }
    }
    sh[tl] = acc;
    __syncthreads();
    if (valid && j == 0) {
        HK_NOUNROLL for (int k = 1; k < K; k++) acc = jac_add_ni(acc, sh[tl + k]);
        XYZZ<F> o = XYZZ<F>::inf();
        if (!acc.is_inf()) {
            o.x = acc.x; o.y = acc.y;
            o.zz = F::sqr(acc.z);
            o.zzz = F::mul(o.zz, acc.z);
        }
        if (lo) o = ec_madd_ni(o, ld_vec(&lo[i]));
        st_vec(&out[i], o);
    }
}
// out[0] = sum of n XYZZ points: one workgroup, strided partial sums, LDS tree (the tail of a SMALL one-off MSM:
// hk_msm_g1 / _g2 over caller-supplied bases with n K <= SPLIT_MAX_LANES run as n element-wise products + this sum -
// without shift tables a Pippenger pass ends in ~254 serial doublings, 3 ms in G1 and 8.5 ms in G2 whatever n)
template <class F> struct PointsSum { static constexpr int THREADS = sizeof(XYZZ<F>) > 256 ? 128 : 256; };   // LDS <= 64 KiB
template <class F>
__global__ void __launch_bounds__(PointsSum<F>::THREADS)
k_points_sum(const XYZZ<F>* __restrict__ in, u32 n, XYZZ<F>* __restrict__ out) {
    constexpr u32 T = PointsSum<F>::THREADS;
    __shared__ XYZZ<F> sh[T];
    XYZZ<F> acc = XYZZ<F>::inf();
    HK_NOUNROLL for (u32 i = threadIdx.x; i < n; i += T) acc = ec_add_ni(acc, ld_vec(&in[i]));
    sh[threadIdx.x] = acc;
    __syncthreads();
    HK_NOUNROLL for (u32 off = T / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) { XYZZ<F> t = ec_add_ni(sh[threadIdx.x], sh[threadIdx.x + off]); sh[threadIdx.x] = t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) st_vec(out, sh[0]);
}
// out[g] = sum of in[g * seg .. (g + 1) * seg): one 64-lane workgroup per segment (the commitments of hk_commit_batch)
template <class F>
__global__ void __launch_bounds__(64)
k_points_sum_seg(const XYZZ<F>* __restrict__ in, u32 seg, XYZZ<F>* __restrict__ out) {
    __shared__ XYZZ<F> sh[64];
    const XYZZ<F>* src = in + (size_t)blockIdx.x * seg;
    XYZZ<F> acc = XYZZ<F>::inf();
    HK_NOUNROLL for (u32 i = threadIdx.x; i < seg; i += 64) acc = ec_add_ni(acc, ld_vec(&src[i]));
    sh[threadIdx.x] = acc;
    __syncthreads();
    HK_NOUNROLL for (u32 off = 32; off > 0; off >>= 1) {
        if (threadIdx.x < off) { XYZZ<F> t = ec_add_ni(sh[threadIdx.x], sh[threadIdx.x + off]); sh[threadIdx.x] = t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) st_vec(&out[blockIdx.x], sh[0]);
}
#endif


}  // namespace hk
