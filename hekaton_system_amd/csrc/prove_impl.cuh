// prove_impl.cuh — per-curve host orchestration of the CP-Groth16 hot path on one GPU:
// NTT / witness map, proving-key residency (shift tables), hk_commit and hk_prove.
//
// Reference path restated (all arithmetic on the device; this file only sequences launches):
//   CPGroth16::prove_last_stage            cp-groth16/src/prover.rs:78-155
//   CommitmentBuilder::{commit,prove}      cp-groth16/src/committer.rs:87-91, 112-114
//   LibsnarkReduction::witness_map         ark-groth16 0.4, called at prover.rs:123
//
// Fusion specific to this design: the O(1) fixed-base scalar multiplications of the reference
// (r*delta, s*delta, r*s*delta, kappa_i*delta_i — prover.rs:86,96,106,135; committer.rs:113) ride the
// big MSMs as extra (base, scalar) pairs appended to the assignment ("ext" slots), so only the two
// genuinely variable-base products s*A and r*B1 remain for the finish kernel.
#pragma once
#include <string>
#include "curve_ops_impl.cuh"
#include "ntt.cuh"
#include "fixed_base.cuh"
#include "witness.cuh"
#include "endo.cuh"

namespace hk {

// ---- small host helpers on Montgomery values (setup constants only) -------------------------------
template <class Fr>
static Fr host_halve(const Fr& a) {          // a/2 in the field (works on Montgomery residues too)
    u32 t[Fr::N + 1];
    u64 c = 0;
    bool odd = a.v[0] & 1;
    for (int i = 0; i < Fr::N; i++) {
        c += (u64)a.v[i] + (odd ? Fr::Params::MOD[i] : 0u);
        t[i] = (u32)c;
        c >>= 32;
    }
    t[Fr::N] = (u32)c;
    Fr r;
    for (int i = 0; i < Fr::N; i++) r.v[i] = (t[i] >> 1) | (t[i + 1] << 31);
    return r;
}
template <class Fr>
static Fr host_from_limbs(const u32* l) {
    Fr r;
    for (int i = 0; i < Fr::N; i++) r.v[i] = l[i];
    return r;
}

// ---- twiddle / coset tables (one set per context) ---------------------------------------------------
struct NttTables {
    std::mutex mu;
    u32 log_table = 0;
    void* tw_fwd = nullptr;      // per-stage tables of w (k_stage_tables), 2^log_table - 1 entries; stage s always uses the
                                 // 2^(s+1)-th roots, so every transform size <= 2^log_table shares them
    void* tw_inv = nullptr;      // same for w^-1
    void* pw_g = nullptr;        // 3 x POW_TABLE_SIZE powers of F::GENERATOR
    void* pw_ginv = nullptr;
    std::vector<void*> retired;  // superseded tables stay alive until the context dies
};

template <class C>
struct NttHost {
    typedef typename C::Fr Fr;

    static hk_status ensure(hk_ctx* ctx, u32 log_m, NttTables** out) {
        if (log_m > C::TWO_ADICITY) return HK_ERR_DOMAIN_TOO_LARGE;
        std::unique_lock<std::mutex> lk(ctx->mu);
        if (!ctx->ntt) ctx->ntt = new NttTables();
        NttTables* T = ctx->ntt;
        lk.unlock();
        std::unique_lock<std::mutex> tl(T->mu);
        *out = T;
        if (T->log_table >= log_m && T->tw_fwd) return HK_OK;
        u32 L = log_m < 16 ? 16 : log_m;
        if (L > C::TWO_ADICITY) L = C::TWO_ADICITY;
        HK_HIP(hipSetDevice(ctx->device));
        // host: w_M = ROOT^(2^(s-L)); sq[k] = w_M^(2^k); w_M^-1 = prod_k sq[k]
        std::vector<Fr> sq(32), sqi(32);
        Fr w = host_from_limbs<Fr>(C::ROOT);
        for (u32 k = 0; k < C::TWO_ADICITY - L; k++) w = Fr::sqr(w);
        Fr winv = Fr::one();
        for (u32 k = 0; k < L; k++) {
            sq[k] = w;
            winv = Fr::mul(winv, w);
            w = Fr::sqr(w);
        }
        Fr t = winv;
        for (u32 k = 0; k < L; k++) { sqi[k] = t; t = Fr::sqr(t); }
        const u32 NG = 3 * POW_TABLE_BITS;
        std::vector<Fr> gs(NG), gis(NG);
        Fr g = host_from_limbs<Fr>(C::GEN), gi = host_from_limbs<Fr>(C::GEN_INV);
        for (u32 k = 0; k < NG; k++) { gs[k] = g; gis[k] = gi; g = Fr::sqr(g); gi = Fr::sqr(gi); }
        Fr *d_sq = nullptr, *tmp = nullptr, *tf = nullptr, *ti = nullptr, *pg = nullptr, *pgi = nullptr;
        size_t half = (size_t)1 << (L - 1), full = (size_t)1 << L;
        HK_HIP(hipMalloc((void**)&d_sq, sizeof(Fr) * (64 + 2 * NG)));
        HK_HIP(hipMalloc((void**)&tmp, sizeof(Fr) * half));
        HK_HIP(hipMalloc((void**)&tf, sizeof(Fr) * full));
        HK_HIP(hipMalloc((void**)&ti, sizeof(Fr) * full));
        HK_HIP(hipMalloc((void**)&pg, sizeof(Fr) * 3 * POW_TABLE_SIZE));
        HK_HIP(hipMalloc((void**)&pgi, sizeof(Fr) * 3 * POW_TABLE_SIZE));
        HK_HIP(hipMemcpy(d_sq, sq.data(), sizeof(Fr) * 32, hipMemcpyHostToDevice));
        HK_HIP(hipMemcpy(d_sq + 32, sqi.data(), sizeof(Fr) * 32, hipMemcpyHostToDevice));
        HK_HIP(hipMemcpy(d_sq + 64, gs.data(), sizeof(Fr) * NG, hipMemcpyHostToDevice));
        HK_HIP(hipMemcpy(d_sq + 64 + NG, gis.data(), sizeof(Fr) * NG, hipMemcpyHostToDevice));
        u32 blocks = (u32)((half + 255) / 256), blocks_full = (u32)((full + 255) / 256);
        // w_M^i for i < M/2 (scratch), regrouped into one contiguous table per butterfly stage
        hipLaunchKernelGGL((k_pow_table<Fr>), dim3(blocks), dim3(256), 0, 0, tmp, d_sq, (u32)half, L - 1);
        hipLaunchKernelGGL((k_stage_tables<Fr>), dim3(blocks_full), dim3(256), 0, 0, tf, tmp, L);
        hipLaunchKernelGGL((k_pow_table<Fr>), dim3(blocks), dim3(256), 0, 0, tmp, d_sq + 32, (u32)half, L - 1);
        hipLaunchKernelGGL((k_stage_tables<Fr>), dim3(blocks_full), dim3(256), 0, 0, ti, tmp, L);
        for (u32 lvl = 0; lvl < 3; lvl++) {
            hipLaunchKernelGGL((k_pow_table<Fr>), dim3(POW_TABLE_SIZE / 256), dim3(256), 0, 0, pg + POW_TABLE_SIZE * lvl,
                               d_sq + 64 + POW_TABLE_BITS * lvl, (u32)POW_TABLE_SIZE, (u32)POW_TABLE_BITS);
            hipLaunchKernelGGL((k_pow_table<Fr>), dim3(POW_TABLE_SIZE / 256), dim3(256), 0, 0, pgi + POW_TABLE_SIZE * lvl,
                               d_sq + 64 + NG + POW_TABLE_BITS * lvl, (u32)POW_TABLE_SIZE, (u32)POW_TABLE_BITS);
        }
        HK_HIP(hipGetLastError());
        HK_HIP(hipDeviceSynchronize());
        HK_HIP(hipFree(d_sq));
        HK_HIP(hipFree(tmp));
        for (void* p : {T->tw_fwd, T->tw_inv, T->pw_g, T->pw_ginv})
            if (p) T->retired.push_back(p);
        T->tw_fwd = tf; T->tw_inv = ti; T->pw_g = pg; T->pw_ginv = pgi;
        T->log_table = L;
        return HK_OK;
    }

    static Fr size_inv(u32 log_m) {               // (2^log_m)^-1, Montgomery
        Fr x = Fr::one();
        for (u32 k = 0; k < log_m; k++) x = host_halve(x);
        return x;
    }
    static Fr vanishing_inv_on_coset(u32 log_m) {  // (g^m - 1)^-1  (SURVEY.md A.1 `zinv`)
        Fr g = host_from_limbs<Fr>(C::GEN);
        for (u32 k = 0; k < log_m; k++) g = Fr::sqr(g);
        return fp_inv(Fr::sub(g, Fr::one()));
    }

    // fused epilogue of the last pass (k_ntt_pass4): which steps, on how many of the batched vectors, operands
    struct Post {
        int post = 0;
        u32 nvec = 0xffffffffu;
        const Fr* scale = nullptr;    // post & 1
        const Fr* pw = nullptr;       // post & 2
        const Fr* sub = nullptr;      // post & 4
        const Fr* kc = nullptr;
    };

    // all butterfly stages of a size-2^logn transform, `batch` vectors `stride` elements apart.
    // tws: per-stage twiddle tables.
    static hk_status passes(hipStream_t s, Fr* data, size_t stride, u32 batch, u32 logn, const Fr* tws, int dit,
                            const Post& ep = Post()) {
        if (logn == 0) return HK_OK;
        // bottom pass: the low min(logn, 11) stages on contiguous tiles; the rest in passes of at most
        // `upper_max` stages whose tiles are 2^nst rows of 2^(11 - nst) contiguous elements
        static const u32 upper_max = [] {
            const char* e = getenv("HK_NTT_UPPER_MAX");
            u32 v = e ? (u32)atoi(e) : 6u;      // 2^21: 11+5+5, 2^22: 11+6+5 (a single 10-stage upper pass with
                                                // 64-B rows measured the same alone and less steady under load)
            return v < 1 ? 1u : (v > 10 ? 10u : v);
        }();
        static const u32 tile_log = [] {
            const char* e = getenv("HK_NTT_TILE_LOG");
            u32 v = e ? (u32)atoi(e) : (u32)NTT_TILE_LOG;
            return v < 8 ? 8u : (v > (u32)NTT_TILE_LOG ? (u32)NTT_TILE_LOG : v);
        }();
        const u32 threads = 1u << (tile_log - 2);                    // one radix-4 quad per thread
        u32 bottom = logn < tile_log ? logn : tile_log;
        u32 rest = logn - bottom;
        u32 npass = (rest + upper_max - 1) / upper_max;
        struct P { u32 lo, nst, cols_bits; } ps[34];
        int np = 0;
        ps[np++] = {0, bottom, 0};
        u32 lo = bottom;
        for (u32 i = 0; i < npass; i++) {
            u32 nst = (rest - (lo - bottom) + (npass - i) - 1) / (npass - i);
            ps[np++] = {lo, nst, tile_log - nst};
            lo += nst;
        }
        Fr one = Fr::one();
        for (int k = 0; k < np; k++) {
            const P& p = dit ? ps[k] : ps[np - 1 - k];
            u32 tile_log = p.nst + p.cols_bits;
            dim3 grid(1u << (logn - tile_log), batch);
            size_t lds = sizeof(Fr) << tile_log;
            bool last = k == np - 1;
            int pp = last ? ep.post : 0;
            const Fr& sc = (pp & 1) ? *ep.scale : one;
            const Fr& kc = (pp & 4) ? *ep.kc : one;
            if (dit)
                hipLaunchKernelGGL((k_ntt_pass4<Fr, 1>), grid, dim3(threads), lds, s, data, stride, tws, logn, p.lo,
                                   p.nst, p.cols_bits, pp, ep.nvec, sc, ep.pw, ep.sub, kc);
            else
                hipLaunchKernelGGL((k_ntt_pass4<Fr, 0>), grid, dim3(threads), lds, s, data, stride, tws, logn, p.lo,
                                   p.nst, p.cols_bits, pp, ep.nvec, sc, ep.pw, ep.sub, kc);
        }
        HK_HIP(hipGetLastError());
        return HK_OK;
    }

    static hk_status scale(hipStream_t s, Fr* data, size_t stride, u32 batch, u32 logn, const Fr* pw,
                           const Fr& sc, int bitrev_index, int use_pow) {
        size_t n = (size_t)1 << logn;
        hipLaunchKernelGGL((k_scale_pow<Fr>), dim3((u32)((n + 255) / 256), batch), dim3(256), 0, s, data,
                           stride, pw, sc, logn, bitrev_index, use_pow);
        HK_HIP(hipGetLastError());
        return HK_OK;
    }
    static hk_status bitrev(hipStream_t s, Fr* data, u32 logn) {
        size_t n = (size_t)1 << logn;
        hipLaunchKernelGGL((k_bitrev<Fr>), dim3((u32)((n + 255) / 256)), dim3(256), 0, s, data, logn);
        HK_HIP(hipGetLastError());
        return HK_OK;
    }
};

template <class C>
hk_status Ops<C>::ntt(hk_ctx* ctx, void* data, unsigned log_m, int inverse, int coset) {
    typedef NttHost<C> N;
    NttTables* T;
    HK_TRY(N::ensure(ctx, log_m, &T));
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    size_t n = (size_t)1 << log_m;
    HK_TRY(L->reserve(n * sizeof(Fr) + 4096));
    bool dev = is_device_ptr(data);
    Fr* d = (Fr*)data;
    if (!dev) {
        d = L->alloc_n<Fr>(n);
        if (!d) return HK_ERR_NOMEM;
        HK_HIP(hipMemcpyAsync(d, data, n * sizeof(Fr), hipMemcpyHostToDevice, L->stream));
    }
    hipStream_t s = L->stream;
    if (!inverse) {
        // coset FFT: coeff j *= g^j, then FFT (A.2).  DIF then un-permute.
        if (coset) HK_TRY(N::scale(s, d, n, 1, log_m, (const Fr*)T->pw_g, Fr::one(), 0, 1));
        HK_TRY(N::passes(s, d, n, 1, log_m, (const Fr*)T->tw_fwd, 0));
        HK_TRY(N::bitrev(s, d, log_m));
    } else {
        // iFFT: DIF with w^-1, scale by 1/m (and g^-j for the coset form), un-permute
        Fr minv = N::size_inv(log_m);
        typename N::Post ep;
        ep.post = coset ? 3 : 1;
        ep.scale = &minv;
        ep.pw = (const Fr*)T->pw_ginv;
        HK_TRY(N::passes(s, d, n, 1, log_m, (const Fr*)T->tw_inv, 0, ep));
        HK_TRY(N::bitrev(s, d, log_m));
    }
    if (!dev) HK_HIP(hipMemcpyAsync(data, d, n * sizeof(Fr), hipMemcpyDeviceToHost, s));
    HK_HIP(hipStreamSynchronize(s));
    return HK_OK;
}

// ---- witness map on device buffers --------------------------------------------------------------------
struct CsrDev { const u64* row_ptr; const u32* col; const void* val; size_t n_rows, nnz; };

// HK_ERR_ARG unless the (device-resident) matrix is structurally sound for n_cols variables: a malformed matrix
// must come back as an error (the reference returns an ark error), never as an out-of-bounds device read.
// `flag`: one u32 of device scratch.  Synchronises `s`.
static hk_status csr_validate(hipStream_t s, const CsrDev& M, size_t n_cols, u32* flag) {
    if (n_cols >= ((size_t)1 << 32)) return HK_ERR_ARG;
    HK_HIP(hipMemsetAsync(flag, 0, sizeof(u32), s));
    size_t work = M.n_rows > M.nnz ? M.n_rows : M.nnz;
    u32 blocks = (u32)std::min<size_t>((work + 255) / 256, 2048);
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL((k_csr_check<0>), dim3(blocks), dim3(256), 0, s, M.row_ptr, M.col, (u64)M.n_rows, (u64)M.nnz,
                       (u32)n_cols, flag);
    u32 h = 0;
    HK_HIP(hipMemcpyAsync(&h, flag, sizeof(u32), hipMemcpyDeviceToHost, s));
    HK_HIP(hipStreamSynchronize(s));
    return h ? HK_ERR_ARG : HK_OK;
}

template <class C>
struct QapHost {
    typedef typename C::Fr Fr;
    typedef NttHost<C> N;

    static u32 domain_log(size_t n_c, size_t n_inst) {
        size_t need = n_c + n_inst;
        u32 lg = 0;
        while (((size_t)1 << lg) < need) lg++;
        return lg;
    }
    // abc: 3*m Fr scratch (a | b | c).  On return a[0..m) = h in BIT-REVERSED order.
    static hk_status run(hipStream_t s, NttTables* T, const CsrDev& A, const CsrDev& B, const CsrDev& Cm,
                         size_t n_inst, size_t n_c, const Fr* z, Fr* abc, u32 log_m) {
        size_t m = (size_t)1 << log_m;
        const CsrDev* Ms[3] = {&A, &B, &Cm};
        for (int k = 0; k < 3; k++) {
            // every row of the m-element vector is written: matrix rows, the instance copy behind them (a only:
            // a[n_c + j] = z[j]), zeros - no memset of the 3 m x 32 B (a 201 MB fill at m = 2^21)
            hipLaunchKernelGGL((k_spmv<Fr>), dim3((u32)((m + 255) / 256)), dim3(256), 0, s,
                               Ms[k]->row_ptr, Ms[k]->col, (const Fr*)Ms[k]->val, z, abc + k * m,
                               (u32)n_c, k == 0 ? (u32)n_inst : 0u, (u32)m);
        }
        // With Z constant on the coset (Z(g w^i) = g^m - 1) and the transforms linear,
        //     h = zinv * (coset_ifft(a_coset o b_coset) - ifft(c))
        // which is bit for bit what A.1 computes with its seventh transform (c's coset fft) left out.
        // Every inverse transform here is UNSCALED (m times too large); the powers of 1/m are folded into k, kc.
        const Fr* tinv = (const Fr*)T->tw_inv;
        const Fr* tfwd = (const Fr*)T->tw_fwd;
        typename N::Post e1;                                   // ifft (DIF) of a, b, c; "* g^j" on a and b only
        e1.post = 2;
        e1.nvec = 2;
        e1.pw = (const Fr*)T->pw_g;
        HK_TRY(N::passes(s, abc, m, 3, log_m, tinv, 0, e1));
        HK_TRY(N::passes(s, abc, m, 2, log_m, tfwd, 1));       // coset fft (DIT) of a, b
        hipLaunchKernelGGL((k_mul_pointwise<Fr>), dim3((u32)((m + 255) / 256)), dim3(256), 0, s, abc, abc + m, m);
        Fr minv = N::size_inv(log_m);
        Fr mm = fp_inv(Fr::mul(minv, minv));                                                    // m^2
        Fr k = Fr::mul(N::vanishing_inv_on_coset(log_m), Fr::mul(minv, Fr::mul(minv, minv)));   // zinv / m^3
        typename N::Post e2;                                   // coset ifft (DIF): (x * g^-j - c' * m^2) * zinv/m^3
        e2.post = 2 | 4 | 1;
        e2.pw = (const Fr*)T->pw_ginv;
        e2.sub = abc + 2 * m;
        e2.kc = &mm;
        e2.scale = &k;
        HK_TRY(N::passes(s, abc, m, 1, log_m, tinv, 0, e2));
        HK_HIP(hipGetLastError());
        return HK_OK;
    }
};

template <class C>
hk_status Ops<C>::witness_map(hk_ctx* ctx, const hk_csr* A, const hk_csr* B, const hk_csr* Cm, size_t n_inst,
                              size_t n_c, const void* z, size_t n_v, void* h_out, size_t h_cap,
                              size_t* m_out) {
    typedef QapHost<C> Q;
    if (A->n_rows != n_c || B->n_rows != n_c || Cm->n_rows != n_c || n_inst > n_v || n_inst < 1) return HK_ERR_ARG;
    for (auto M : {A, B, Cm})
        if (!M->row_ptr || (M->nnz && (!M->col || !M->val_mont))) return HK_ERR_ARG;
    u32 log_m = Q::domain_log(n_c, n_inst);
    if (log_m > C::TWO_ADICITY) return HK_ERR_DOMAIN_TOO_LARGE;
    size_t m = (size_t)1 << log_m;
    if (m_out) *m_out = m;
    if (h_cap < m) return HK_ERR_LEN;
    NttTables* T;
    HK_TRY(NttHost<C>::ensure(ctx, log_m, &T));
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    size_t need = 3 * m * sizeof(Fr) + n_v * sizeof(Fr) + 8192 + 256;
    const hk_csr* Ms[3] = {A, B, Cm};
    for (auto M : Ms) need += al256(8 * (M->n_rows + 1)) + al256(4 * M->nnz) + al256(sizeof(Fr) * M->nnz);
    HK_TRY(L->reserve(need));
    CsrDev D[3];
    for (int k = 0; k < 3; k++) {
        const void *rp, *cl, *vl;
        HK_TRY(to_device(L, Ms[k]->row_ptr, 8 * (Ms[k]->n_rows + 1), &rp));
        HK_TRY(to_device(L, Ms[k]->col, 4 * Ms[k]->nnz, &cl));
        HK_TRY(to_device(L, Ms[k]->val_mont, sizeof(Fr) * Ms[k]->nnz, &vl));
        D[k] = {(const u64*)rp, (const u32*)cl, vl, Ms[k]->n_rows, Ms[k]->nnz};
    }
    u32* flag = L->alloc_n<u32>(1);
    if (!flag) return HK_ERR_NOMEM;
    for (int k = 0; k < 3; k++) HK_TRY(csr_validate(L->stream, D[k], n_v, flag));
    const void* zd;
    HK_TRY(to_device(L, z, n_v * sizeof(Fr), &zd));
    Fr* abc = L->alloc_n<Fr>(3 * m);
    if (!abc) return HK_ERR_NOMEM;
    HK_TRY(Q::run(L->stream, T, D[0], D[1], D[2], n_inst, n_c, (const Fr*)zd, abc, log_m));
    HK_TRY(NttHost<C>::bitrev(L->stream, abc, log_m));         // API returns natural order
    HK_HIP(hipMemcpyAsync(h_out, abc, m * sizeof(Fr),
                          is_device_ptr(h_out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, L->stream));
    HK_HIP(hipStreamSynchronize(L->stream));
    return HK_OK;
}

// ---- device-resident proving key ------------------------------------------------------------------------
template <class C>
struct PkImpl {
    typedef typename C::Fr Fr;
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;
    u32 n_v = 0, n_inst = 0, n_c = 0, n_stages = 0, n_ext = 0, n_extra = 0;
    MsmPlan plan_z;                        // shared by the A / B1 / B2 / L queries (same scalar vector)
    Affine<Fq>* a_tab = nullptr;           // [F][n_ext]   a_g[1..] | delta_g | inf ...
    Affine<Fq>* b1_tab = nullptr;          // [F][n_ext]   b_g[1..] | inf | delta_g | inf ...
    Affine<Fq2>* b2_tab = nullptr;         // [F][n_ext]   b_h[1..] | inf | delta_h | inf ...
    // B-query density (bellman's DensityTracker idea): b_g[i] and b_h[i] are infinity for every variable that
    // never occurs in B.  When enough of them are, B1 and B2 run over the compacted list b_idx (ext indices of
    // the non-infinity bases, then every ext slot) with their own digit sort; b1_tab / b2_tab then hold
    // [F][b_n] entries and plan_b replaces plan_z for them.
    bool b_compact = false;
    u32 b_n = 0;
    u32* b_idx = nullptr;
    MsmPlan plan_b;
    Affine<Fq>* l_tab = nullptr;           // [F][l_n]     ck_last | inf | inf | -delta_g | -delta_i ...
    u32 l_n = 0, l_off = 0;
    bool has_qap = false;
    u32 log_m = 0;
    MsmPlan plan_h;
    Affine<Fq>* h_tab = nullptr;           // [F][m]  h_g in bit-reversed order (slot m-1 = inf)
    std::vector<MsmPlan> plan_ck;
    std::vector<Affine<Fq>*> ck_tab;       // [F][ck_len + 1]  ck[stage] | last_delta_g
    std::vector<u32> ck_n;
    Affine<Fq>* consts_g1 = nullptr;       // a_g[0], alpha_g, b_g[0], beta_g
    Affine<Fq2>* consts_g2 = nullptr;      // b_h[0], beta_h
    CsrDev csr[3];
    std::vector<void*> owned;              // every hipMalloc of this key
    size_t bytes = 0;
};

template <class F>
static hk_status pk_alloc_table(std::vector<void*>& owned, size_t& total, size_t groups, size_t n,
                                Affine<F>** out) {
    size_t b = groups * n * sizeof(Affine<F>);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, b ? b : 16);
    if (e != hipSuccess) { (void)hipGetLastError(); return HK_ERR_NOMEM; }
    owned.push_back(p);
    total += b;
    *out = (Affine<F>*)p;
    return HK_OK;
}

static inline hipMemcpyKind h2d_kind(const void* src) {
    return is_device_ptr(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
}

// scatter h_g into bit-reversed order on the device: tab[bitrev(j)] = h_g[j], j < h_len; others inf
template <class F>
__global__ void k_pk_bitrev_copy(Affine<F>* __restrict__ tab, const Affine<F>* __restrict__ src, u32 h_len,
                                 u32 log_m) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >> log_m) return;
    u32 r = log_m ? (__brev(j) >> (32 - log_m)) : 0u;
    Affine<F> p = j < h_len ? ld_vec(&src[j]) : Affine<F>::inf();
    st_vec(&tab[r], p);
}

// flags[i] = 1 iff pts[i] is not the point at infinity
template <class F>
__global__ void k_mark_noninf(const Affine<F>* __restrict__ pts, u32* __restrict__ flags, u32 n) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = ld_vec(&pts[i]).is_inf() ? 0u : 1u;
}
// dst[k] = src[idx[k]] for idx[k] < n_src (ext slots beyond the source stay as they are)
template <class T>
__global__ void k_gather(T* __restrict__ dst, const T* __restrict__ src, const u32* __restrict__ idx, u32 n,
                         u32 n_src) {
    u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    u32 i = idx[k];
    if (i < n_src) st_vec(&dst[k], ld_vec(&src[i]));
}

// error exits of pk_upload: release everything allocated so far (fail()) and tell out-of-memory from other faults
#define PK_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            (void)hipGetLastError();                                                          \
            fprintf(stderr, "[hekaton] HIP error %s at %s:%d: %s\n", hipGetErrorName(_e), __FILE__, __LINE__, #expr); \
            return fail(_e == hipErrorOutOfMemory ? HK_ERR_NOMEM : HK_ERR_DEVICE);            \
        }                                                                                     \
    } while (0)
#define PK_TRY(expr)                               \
    do {                                           \
        hk_status _s = (expr);                     \
        if (_s != HK_OK) return fail(_s);          \
    } while (0)

template <class C>
hk_status Ops<C>::pk_upload(hk_ctx* ctx, const hk_pk_desc* d, hk_pk** out) {
    typedef PkImpl<C> PK;
    *out = nullptr;
    if (!d->a_g || !d->b_g || !d->b_h || !d->h_g || !d->deltas_g || !d->last_delta_h || !d->alpha_g ||
        !d->beta_g || !d->beta_h || d->n_stages == 0 || !d->ck_stage || !d->ck_len)
        return HK_ERR_ARG;
    size_t n_v = d->a_len;
    if (n_v < 1 || d->b_g_len != n_v || d->b_h_len != n_v) return HK_ERR_LEN;
    if (n_v + d->n_stages + 4 >= ((size_t)1 << MSM_ENTRY_GROUP_SHIFT)) return HK_ERR_ARG;   // sorted-entry index field
    size_t n_wit = 0;
    for (size_t s = 0; s < d->n_stages; s++) n_wit += d->ck_len[s];
    if (d->n_inst < 1 || d->n_inst + n_wit != n_v) return HK_ERR_LEN;   // instance || stage witnesses
    HK_HIP(hipSetDevice(ctx->device));
    PK* pk = new PK();
    hk_pk* h = new hk_pk{ctx->ops, ctx, pk};
    void* staging = nullptr;                          // transient device copy of h_g (freed on every exit)
    auto fail = [&](hk_status st) { if (staging) (void)hipFree(staging); Ops<C>::pk_free(h); return st; };
    pk->n_v = (u32)n_v; pk->n_inst = (u32)d->n_inst; pk->n_c = (u32)d->n_constraints;
    pk->n_stages = (u32)d->n_stages;
    u32 k = pk->n_stages - 1;
    pk->n_extra = 3 + k;                               // r, s, r*s, kappa_0..kappa_{k-1}
    pk->n_ext = (u32)(n_v - 1) + pk->n_extra;
    const char* wp_env = getenv("HK_MSM_WP");
    u32 WP = wp_env && atoi(wp_env) > 0 ? (u32)atoi(wp_env) : 1u;
    auto make_plan = [&](size_t n) {
        u32 c = msm_pick_c_tables(n, C::FR_BITS);
        if (WP > 1) { while (c > 5 && ((u64)WP << (c - 1)) > (u64)MSM_LDS_COUNTERS) c--; }
        return msm_make_plan((u32)n, C::FR_BITS, c, WP, ctx->max_lanes0, C::Fr::Params::MOD, C::Fr::Params::N);
    };
    pk->plan_z = make_plan(pk->n_ext);
    const MsmPlan& pz = pk->plan_z;
    hipStream_t s0 = 0;
    size_t g1 = sizeof(Affine<Fq>), g2 = sizeof(Affine<Fq2>);
    size_t nq = n_v - 1;                                // query[1..]
    Affine<Fq> inf1 = Affine<Fq>::inf();
    // --- A / B1 / B2 tables
    hk_status st;
    const char* a_g = (const char*)d->a_g; const char* b_g = (const char*)d->b_g; const char* b_h = (const char*)d->b_h;
    const char* deltas = (const char*)d->deltas_g;
    const char* delta_last_g = deltas + g1 * k;
    u32 shift = pz.c * pz.WP;
    if ((st = pk_alloc_table(pk->owned, pk->bytes, pz.F, pk->n_ext, &pk->a_tab)) != HK_OK) return fail(st);
    PK_HIP(hipMemset(pk->a_tab, 0, g1 * pk->n_ext));
    if (nq) PK_HIP(hipMemcpy(pk->a_tab, a_g + g1, g1 * nq, h2d_kind(a_g)));
    PK_HIP(hipMemcpy(pk->a_tab + nq + 0, delta_last_g, g1, h2d_kind(deltas)));          // r * delta_g
    PK_TRY(MsmRun<Fq>::build_tables(s0, pk->a_tab, pk->n_ext, pz.F, shift));
    {
        // B-query: full-size staging copies on the device, then either used as the tables' first group
        // or compacted to the non-infinity bases
        Affine<Fq>* sb1 = nullptr; Affine<Fq2>* sb2 = nullptr; u32* flags = nullptr;
        std::vector<void*> tmp;
        auto cleanup = [&]() { for (void* q : tmp) (void)hipFree(q); tmp.clear(); };
        auto tfail = [&](hk_status e) { cleanup(); return fail(e); };
        auto talloc = [&](void** q, size_t b) { if (hipMalloc(q, b ? b : 16) != hipSuccess) { (void)hipGetLastError(); return false; } tmp.push_back(*q); return true; };
        if (!talloc((void**)&sb1, g1 * (nq + 1)) || !talloc((void**)&sb2, g2 * (nq + 1)) || !talloc((void**)&flags, 4 * (nq + 1)))
            return tfail(HK_ERR_NOMEM);
        std::vector<u32> idx;
        if (nq) {
            if (hipMemcpy(sb1, b_g + g1, g1 * nq, h2d_kind(b_g)) != hipSuccess) return tfail(HK_ERR_DEVICE);
            if (hipMemcpy(sb2, b_h + g2, g2 * nq, h2d_kind(b_h)) != hipSuccess) return tfail(HK_ERR_DEVICE);
            hipLaunchKernelGGL((k_mark_noninf<Fq>), dim3((u32)((nq + 255) / 256)), dim3(256), 0, s0, sb1, flags, (u32)nq);
            std::vector<u32> hf(nq);
            if (hipMemcpy(hf.data(), flags, 4 * nq, hipMemcpyDeviceToHost) != hipSuccess) return tfail(HK_ERR_DEVICE);
            for (size_t i = 0; i < nq; i++) if (hf[i]) idx.push_back((u32)i);
        }
        static const char* dens_env = getenv("HK_B_COMPACT_BELOW");     // density threshold in percent; 0 disables
        double thr = dens_env ? atof(dens_env) / 100.0 : 0.75;
        pk->b_compact = nq >= 4096 && (double)idx.size() < thr * (double)nq;
        if (pk->b_compact) {
            for (u32 e = 0; e < pk->n_extra; e++) idx.push_back((u32)nq + e);
            pk->b_n = (u32)idx.size();
            pk->plan_b = make_plan(pk->b_n);
            void* di = nullptr;
            if (hipMalloc(&di, 4 * (size_t)pk->b_n) != hipSuccess) { (void)hipGetLastError(); return tfail(HK_ERR_NOMEM); }
            pk->owned.push_back(di);
            pk->b_idx = (u32*)di;
            pk->bytes += 4 * (size_t)pk->b_n;
            if (hipMemcpy(di, idx.data(), 4 * (size_t)pk->b_n, hipMemcpyHostToDevice) != hipSuccess) return tfail(HK_ERR_DEVICE);
        } else {
            pk->b_n = pk->n_ext;
            pk->plan_b = pz;
        }
        const MsmPlan& pb = pk->plan_b;
        if ((st = pk_alloc_table(pk->owned, pk->bytes, pb.F, pk->b_n, &pk->b1_tab)) != HK_OK) return tfail(st);
        if ((st = pk_alloc_table(pk->owned, pk->bytes, pb.F, pk->b_n, &pk->b2_tab)) != HK_OK) return tfail(st);
        if (hipMemset(pk->b1_tab, 0, g1 * pk->b_n) != hipSuccess || hipMemset(pk->b2_tab, 0, g2 * pk->b_n) != hipSuccess)
            return tfail(HK_ERR_DEVICE);
        if (pk->b_compact) {
            u32 blocks = (pk->b_n + 255) / 256;
            hipLaunchKernelGGL((k_gather<Affine<Fq>>), dim3(blocks), dim3(256), 0, s0, pk->b1_tab, (const Affine<Fq>*)sb1, pk->b_idx, pk->b_n, (u32)nq);
            hipLaunchKernelGGL((k_gather<Affine<Fq2>>), dim3(blocks), dim3(256), 0, s0, pk->b2_tab, (const Affine<Fq2>*)sb2, pk->b_idx, pk->b_n, (u32)nq);
        } else if (nq) {
            if (hipMemcpy(pk->b1_tab, sb1, g1 * nq, hipMemcpyDeviceToDevice) != hipSuccess) return tfail(HK_ERR_DEVICE);
            if (hipMemcpy(pk->b2_tab, sb2, g2 * nq, hipMemcpyDeviceToDevice) != hipSuccess) return tfail(HK_ERR_DEVICE);
        }
        u32 s_slot = pk->b_n - pk->n_extra + 1;                                              // ext slot of s
        if (hipMemcpy(pk->b1_tab + s_slot, delta_last_g, g1, h2d_kind(deltas)) != hipSuccess) return tfail(HK_ERR_DEVICE);     // s * delta_g
        if (hipMemcpy(pk->b2_tab + s_slot, d->last_delta_h, g2, h2d_kind(d->last_delta_h)) != hipSuccess) return tfail(HK_ERR_DEVICE);   // s * delta_h
        if (hipDeviceSynchronize() != hipSuccess) return tfail(HK_ERR_DEVICE);
        cleanup();
        u32 shift_b = pb.c * pb.WP;
        PK_TRY(MsmRun<Fq>::build_tables(s0, pk->b1_tab, pk->b_n, pb.F, shift_b));
        PK_TRY(MsmRun<Fq2>::build_tables(s0, pk->b2_tab, pk->b_n, pb.F, shift_b));
    }
    // --- L table: last-stage committer key, then the negated deltas that fold -rs*delta and -kappa_i*delta_i
    size_t n1 = d->ck_len[k];
    pk->l_n = (u32)n1 + pk->n_extra;
    pk->l_off = (u32)(n_v - 1 - n1);                    // ext index of the first last-stage witness
    if ((st = pk_alloc_table(pk->owned, pk->bytes, pz.F, pk->l_n, &pk->l_tab)) != HK_OK) return fail(st);
    PK_HIP(hipMemset(pk->l_tab, 0, g1 * pk->l_n));
    if (n1) PK_HIP(hipMemcpy(pk->l_tab, d->ck_stage[k], g1 * n1, h2d_kind(d->ck_stage[k])));
    {
        std::vector<Affine<Fq>> dh(k + 1);
        PK_HIP(hipMemcpy(dh.data(), deltas, g1 * (k + 1), is_device_ptr(deltas) ? hipMemcpyDeviceToHost : hipMemcpyHostToHost));
        std::vector<Affine<Fq>> neg(1 + k);
        neg[0] = dh[k].is_inf() ? dh[k] : ec_neg(dh[k]);                       // -delta_g  (scalar r*s)
        for (u32 i = 0; i < k; i++) neg[1 + i] = dh[i].is_inf() ? dh[i] : ec_neg(dh[i]);   // -delta_i (kappa_i)
        PK_HIP(hipMemcpy(pk->l_tab + n1 + 2, neg.data(), g1 * (1 + k), hipMemcpyHostToDevice));
    }
    PK_TRY(MsmRun<Fq>::build_tables(s0, pk->l_tab, pk->l_n, pz.F, shift));
    // --- per-stage commitment tables: ck[stage] | last_delta_g (scalar kappa)
    for (u32 sidx = 0; sidx < pk->n_stages; sidx++) {
        size_t n = d->ck_len[sidx] + 1;
        MsmPlan p = make_plan(n);
        Affine<Fq>* tab;
        if ((st = pk_alloc_table(pk->owned, pk->bytes, p.F, n, &tab)) != HK_OK) return fail(st);
        if (n > 1) PK_HIP(hipMemcpy(tab, d->ck_stage[sidx], g1 * (n - 1), h2d_kind(d->ck_stage[sidx])));
        PK_HIP(hipMemcpy(tab + n - 1, delta_last_g, g1, h2d_kind(deltas)));
        PK_TRY(MsmRun<Fq>::build_tables(s0, tab, (u32)n, p.F, p.c * p.WP));
        pk->plan_ck.push_back(p);
        pk->ck_tab.push_back(tab);
        pk->ck_n.push_back((u32)n);
    }
    // --- constants for the finish kernel
    {
        void* p1; void* p2;
        PK_HIP(hipMalloc(&p1, g1 * 4)); pk->owned.push_back(p1);
        PK_HIP(hipMalloc(&p2, g2 * 2)); pk->owned.push_back(p2);
        pk->consts_g1 = (Affine<Fq>*)p1; pk->consts_g2 = (Affine<Fq2>*)p2;
        PK_HIP(hipMemcpy(pk->consts_g1 + 0, a_g, g1, h2d_kind(a_g)));
        PK_HIP(hipMemcpy(pk->consts_g1 + 1, d->alpha_g, g1, h2d_kind(d->alpha_g)));
        PK_HIP(hipMemcpy(pk->consts_g1 + 2, b_g, g1, h2d_kind(b_g)));
        PK_HIP(hipMemcpy(pk->consts_g1 + 3, d->beta_g, g1, h2d_kind(d->beta_g)));
        PK_HIP(hipMemcpy(pk->consts_g2 + 0, b_h, g2, h2d_kind(b_h)));
        PK_HIP(hipMemcpy(pk->consts_g2 + 1, d->beta_h, g2, h2d_kind(d->beta_h)));
    }
    // --- QAP: matrices + H-query in bit-reversed order
    if (d->A && d->B && d->C) {
        if (d->A->n_rows != d->n_constraints || d->B->n_rows != d->n_constraints ||
            d->C->n_rows != d->n_constraints)
            return fail(HK_ERR_LEN);
        pk->log_m = QapHost<C>::domain_log(d->n_constraints, d->n_inst);
        if (pk->log_m > C::TWO_ADICITY || pk->log_m > (u32)MSM_ENTRY_GROUP_SHIFT) return fail(HK_ERR_DOMAIN_TOO_LARGE);
        size_t m = (size_t)1 << pk->log_m;
        if (d->h_len + 1 != m) return fail(HK_ERR_LEN);                 // prover.rs:128 assert
        const hk_csr* Ms[3] = {d->A, d->B, d->C};
        for (int i = 0; i < 3; i++) {
            void *rp, *cl, *vl;
            PK_HIP(hipMalloc(&rp, 8 * (Ms[i]->n_rows + 1))); pk->owned.push_back(rp);
            PK_HIP(hipMalloc(&cl, 4 * Ms[i]->nnz + 16)); pk->owned.push_back(cl);
            PK_HIP(hipMalloc(&vl, sizeof(Fr) * Ms[i]->nnz + 16)); pk->owned.push_back(vl);
            PK_HIP(hipMemcpy(rp, Ms[i]->row_ptr, 8 * (Ms[i]->n_rows + 1), h2d_kind(Ms[i]->row_ptr)));
            if (!Ms[i]->row_ptr || (Ms[i]->nnz && (!Ms[i]->col || !Ms[i]->val_mont))) return fail(HK_ERR_ARG);
            if (Ms[i]->nnz) {
                PK_HIP(hipMemcpy(cl, Ms[i]->col, 4 * Ms[i]->nnz, h2d_kind(Ms[i]->col)));
                PK_HIP(hipMemcpy(vl, Ms[i]->val_mont, sizeof(Fr) * Ms[i]->nnz, h2d_kind(Ms[i]->val_mont)));
            }
            pk->csr[i] = {(const u64*)rp, (const u32*)cl, vl, Ms[i]->n_rows, Ms[i]->nnz};
            pk->bytes += 8 * (Ms[i]->n_rows + 1) + (4 + sizeof(Fr)) * Ms[i]->nnz;
        }
        {
            // a malformed matrix (column >= n_v, row_ptr not monotone / not ending at nnz) is HK_ERR_ARG here,
            // not an out-of-bounds read in every later hk_prove
            void* flag = nullptr;
            PK_HIP(hipMalloc(&flag, 256)); pk->owned.push_back(flag);
            for (int i = 0; i < 3; i++) PK_TRY(csr_validate(s0, pk->csr[i], n_v, (u32*)flag));
        }
        pk->plan_h = make_plan(m);
        if ((st = pk_alloc_table(pk->owned, pk->bytes, pk->plan_h.F, m, &pk->h_tab)) != HK_OK) return fail(st);
        const Affine<Fq>* src = (const Affine<Fq>*)d->h_g;
        if (!is_device_ptr(d->h_g)) {
            PK_HIP(hipMalloc(&staging, g1 * (d->h_len ? d->h_len : 1)));
            PK_HIP(hipMemcpy(staging, d->h_g, g1 * d->h_len, hipMemcpyHostToDevice));
            src = (const Affine<Fq>*)staging;
        }
        hipLaunchKernelGGL((k_pk_bitrev_copy<Fq>), dim3((u32)((m + 255) / 256)), dim3(256), 0, s0, pk->h_tab,
                           src, (u32)d->h_len, pk->log_m);
        PK_HIP(hipDeviceSynchronize());
        if (staging) { (void)hipFree(staging); staging = nullptr; }
        PK_TRY(MsmRun<Fq>::build_tables(s0, pk->h_tab, (u32)m, pk->plan_h.F, pk->plan_h.c * pk->plan_h.WP));
        NttTables* T;
        PK_TRY(NttHost<C>::ensure(ctx, pk->log_m, &T));
        pk->has_qap = true;
    }
    PK_HIP(hipDeviceSynchronize());
    *out = h;
    return HK_OK;
}

#undef PK_HIP
#undef PK_TRY

template <class C>
void Ops<C>::pk_free(hk_pk* h) {
    if (!h) return;
    PkImpl<C>* pk = (PkImpl<C>*)h->impl;
    (void)hipSetDevice(h->ctx->device);
    (void)hipDeviceSynchronize();
    for (void* p : pk->owned) (void)hipFree(p);
    delete pk;
    delete h;
}

// ---- MSM over a resident base set (hk_bases_*) -------------------------------------------------------------
struct BasesImpl {
    int group = 1;
    u32 n = 0;
    MsmPlan plan;
    void* tab = nullptr;        // [F][n] Affine<Fq> or Affine<Fq2>; [1][n] when the set never needs its shift tables
    size_t bytes = 0;
    bool has_tables = true;     // false: a short G2 set - every MSM over it runs as MsmRun::small_msm
};

template <class C>
hk_status Ops<C>::bases_upload(hk_ctx* ctx, int group, const void* bases, size_t n, hk_bases** out) {
    *out = nullptr;
    if (n >= ((size_t)1 << MSM_ENTRY_GROUP_SHIFT)) return HK_ERR_ARG;
    HK_HIP(hipSetDevice(ctx->device));
    BasesImpl* b = new BasesImpl();
    b->group = group;
    b->n = (u32)n;
    hk_bases* h = new hk_bases{ctx->ops, ctx, b};
    if (n == 0) { *out = h; return HK_OK; }
    b->plan = msm_make_plan((u32)n, C::FR_BITS, msm_pick_c_tables(n, C::FR_BITS), 1u, ctx->max_lanes0, C::Fr::Params::MOD, C::Fr::Params::N);
    auto build = [&](auto ftag) -> hk_status {
        typedef decltype(ftag) F;
        // a short set goes without shift tables (msm_bases then runs n element-wise endomorphism products + one sum): their
        // construction is 15 x (16 doublings + one inversion) per base - 6 ms per G1 set, 10 ms per G2 set, most of
        // `tipa.setup`, whose four sets are multiplied ONCE per aggregation - against 0.3 - 0.5 ms saved per G1 product
        // (1.0 - 1.3 ms with tables, 1.4 - 1.8 ms without; G2 is quicker without).  HK_BASES_TABLES=1: tables for G1 sets of
        // any length, for a caller that multiplies one set many times
        const bool g2 = sizeof(F) > sizeof(Fq);
        const bool short_set = g2 ? n <= 2048 : (n <= 8192 && !getenv("HK_BASES_TABLES"));
        b->has_tables = !(short_set && !getenv("HK_MSM_NO_SMALL"));
        size_t bytes = (size_t)(b->has_tables ? b->plan.F : 1u) * n * sizeof(Affine<F>);
        if (hipMalloc(&b->tab, bytes) != hipSuccess) { (void)hipGetLastError(); return HK_ERR_NOMEM; }
        b->bytes = bytes;
        HK_HIP(hipMemcpy(b->tab, bases, n * sizeof(Affine<F>), h2d_kind(bases)));
        if (!b->has_tables) return HK_OK;
        return MsmRun<F>::build_tables(0, (Affine<F>*)b->tab, (u32)n, b->plan.F, b->plan.c * b->plan.WP);
    };
    hk_status st = group == 1 ? build(Fq()) : build(Fq2());
    if (st == HK_OK && hipDeviceSynchronize() != hipSuccess) st = HK_ERR_DEVICE;
    if (st != HK_OK) { Ops<C>::bases_free(h); return st; }
    *out = h;
    return HK_OK;
}

template <class C>
void Ops<C>::bases_free(hk_bases* h) {
    if (!h) return;
    BasesImpl* b = (BasesImpl*)h->impl;
    (void)hipSetDevice(h->ctx->device);
    (void)hipDeviceSynchronize();
    if (b->tab) (void)hipFree(b->tab);
    delete b;
    delete h;
}

template <class C>
hk_status Ops<C>::msm_bases(hk_ctx* ctx, const hk_bases* h, const void* scalars, size_t n_scalars, int mont,
                            int checked, void* out) {
    const BasesImpl* b = (const BasesImpl*)h->impl;
    if (checked && n_scalars != b->n) return HK_ERR_LEN;            // ark `msm`: Err(min_len)
    size_t n = n_scalars < b->n ? n_scalars : b->n;                 // ark `msm_unchecked`: zip
    auto run = [&](auto ftag) -> hk_status {
        typedef decltype(ftag) F;
        if (n == 0) { memset(out, 0, sizeof(Affine<F>)); return HK_OK; }
        if (!scalars) return HK_ERR_ARG;
        LaneGuard g(ctx);
        Lane* L = g.lane;
        if (!L) return HK_ERR_DEVICE;
        if (!b->has_tables || (sizeof(F) > sizeof(Fq) && n <= 2048 && !getenv("HK_MSM_NO_SMALL"))) {
            // a short G2 MSM: even with the tables' bucket pass free of a Horner tail, n element-wise products over psi +
            // one sum are quicker (1.9 - 2.4 ms against 2.3 - 3.0; G1 stays with the tables: 1.0 - 1.2 ms against 1.4 - 1.5)
            HK_TRY(L->reserve(al256(n * sizeof(Fr)) + al256(n * sizeof(XYZZ<F>)) + al256(endo_tab_bytes<F>(n)) +
                              al256(sizeof(XYZZ<F>)) + al256(sizeof(Affine<F>)) + 4096));
            const void* sc_d;
            HK_TRY(to_device(L, scalars, n * sizeof(Fr), &sc_d));
            XYZZ<F>* xy = L->alloc_n<XYZZ<F>>(n);
            XYZZ<F>* tab = (XYZZ<F>*)L->alloc_n<unsigned char>(endo_tab_bytes<F>(n));
            XYZZ<F>* res = L->alloc_n<XYZZ<F>>(1);
            Affine<F>* aff = L->alloc_n<Affine<F>>(1);
            if (!xy || !tab || !res || !aff) return HK_ERR_NOMEM;
            HK_TRY(MsmRun<F>::small_msm(L->stream, (const Affine<F>*)b->tab, sc_d, mont, (u32)n, tab, xy, res));   // group 0 of the table = the bases
            HK_TRY(MsmRun<F>::to_affine(L->stream, res, aff, 1));
            HK_HIP(hipMemcpyAsync(out, aff, sizeof(Affine<F>), is_device_ptr(out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, L->stream));
            HK_HIP(hipStreamSynchronize(L->stream));
            return HK_OK;
        }
        const MsmPlan& p = b->plan;                                  // planned for b->n scalars; the tail reads zeros
        size_t need = al256(b->n * sizeof(Fr)) + msm_sort_bytes(p) + msm_run_bytes<F>(p) + 8192;
        HK_TRY(L->reserve(need));
        hipStream_t s = L->stream;
        Fr* sc = L->alloc_n<Fr>(b->n);
        if (!sc) return HK_ERR_NOMEM;
        HK_HIP(hipMemcpyAsync(sc, scalars, n * sizeof(Fr), h2d_kind(scalars), s));
        if (n < b->n) HK_HIP(hipMemsetAsync(sc + n, 0, (b->n - n) * sizeof(Fr), s));
        SortBufs sb;
        HK_TRY(MsmSort<Fr>::alloc(L, p, &sb));
        typename MsmRun<F>::Bufs rb;
        HK_TRY(MsmRun<F>::alloc(L, p, &rb));
        XYZZ<F>* res = L->alloc_n<XYZZ<F>>(1);
        Affine<F>* aff = L->alloc_n<Affine<F>>(1);
        if (!res || !aff) return HK_ERR_NOMEM;
        HK_TRY(MsmSort<Fr>::run(s, p, (const u32*)sc, mont, sb));
        HK_TRY(MsmRun<F>::run(s, p, (const Affine<F>*)b->tab, b->n, 0, sb, rb, res, nullptr, nullptr));
        HK_TRY(MsmRun<F>::to_affine(s, res, aff, 1));
        HK_HIP(hipMemcpyAsync(out, aff, sizeof(Affine<F>), is_device_ptr(out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
        HK_HIP(hipStreamSynchronize(s));
        return HK_OK;
    };
    return b->group == 1 ? run(Fq()) : run(Fq2());
}

template <class C>
hk_status Ops<C>::fixed_base(hk_ctx* ctx, int group, const void* base, const void* scalars, size_t n, int mont,
                             void* out) {
    if (n == 0) return HK_OK;
    if (n >= (1u << 30)) return HK_ERR_ARG;
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    auto run = [&](auto ftag) -> hk_status {
        typedef decltype(ftag) F;
        size_t need = al256(sizeof(Affine<F>)) + al256(n * sizeof(Fr)) + al256(sizeof(Affine<F>) * FB_WINDOWS * 256) +
                      al256(n * sizeof(XYZZ<F>)) + al256(n * sizeof(F)) + al256(n * sizeof(Affine<F>)) + 8192;
        HK_TRY(L->reserve(need));
        // the base's window table: from the context's cache when this base has been multiplied before (host bases only: the
        // key is the base's bytes), else built now - into a cache slot when one is free, into the lane's scratch otherwise
        const size_t tbytes = sizeof(Affine<F>) * FB_WINDOWS * 256;
        Affine<F>* table = nullptr;
        bool build = true;
        int slot = -1;
        struct SlotGuard {                                       // a call that fails after claiming a slot retires it: the
            hk_ctx* c;                                           // entry never matches again (its base may claim another)
            int* slot;
            bool done;
            ~SlotGuard() {
                if (*slot < 0 || done) return;
                std::lock_guard<std::mutex> lk(c->mu);
                c->fb_cache[*slot].group = -1;
            }
        } claimed{ctx, &slot, false};
        if (!is_device_ptr(base) && !getenv("HK_FB_NO_CACHE")) {
            std::string key((const char*)base, sizeof(Affine<F>));
            std::lock_guard<std::mutex> lk(ctx->mu);
            for (auto& e : ctx->fb_cache)
                if (e.group == group && e.base == key) {
                    if (e.ready) { table = (Affine<F>*)e.table; build = false; }
                    slot = -2;                                   // present (ready, or being built by another call)
                    break;
                }
            if (slot == -1 && ctx->fb_cache.size() < (size_t)hk_ctx::FB_CACHE_MAX) {
                void* t = nullptr;
                if (hipMalloc(&t, tbytes) == hipSuccess) {
                    ctx->fb_cache.push_back({group, key, t, false});
                    slot = (int)ctx->fb_cache.size() - 1;
                    table = (Affine<F>*)t;
                } else {
                    (void)hipGetLastError();
                }
            }
        }
        const void *bd, *sd;
        HK_TRY(to_device(L, base, sizeof(Affine<F>), &bd));
        HK_TRY(to_device(L, scalars, n * sizeof(Fr), &sd));
        if (!table) table = L->alloc_n<Affine<F>>(FB_WINDOWS * 256);
        XYZZ<F>* xy = L->alloc_n<XYZZ<F>>(n);
        F* pref = L->alloc_n<F>(n);
        bool out_dev = is_device_ptr(out);
        Affine<F>* od = out_dev ? (Affine<F>*)out : L->alloc_n<Affine<F>>(n);
        if (!table || !xy || !pref || !od) return HK_ERR_NOMEM;
        HK_TRY(MsmRun<F>::fixed_base(L->stream, (const Affine<F>*)bd, sd, mont, (u32)n, table, xy, pref, od, build));
        if (!out_dev) HK_HIP(hipMemcpyAsync(out, od, n * sizeof(Affine<F>), hipMemcpyDeviceToHost, L->stream));
        HK_HIP(hipStreamSynchronize(L->stream));
        if (slot >= 0) {                                         // the table is complete: later calls may read it
            std::lock_guard<std::mutex> lk(ctx->mu);
            ctx->fb_cache[slot].ready = true;
        }
        claimed.done = true;
        return HK_OK;
    };
    return group == 1 ? run(Fq()) : run(Fq2());
}

template <class C>
hk_status Ops<C>::scalar_pairing(hk_ctx* ctx, int group, const void* points, const void* scalars, size_t n,
                                 void* out) {
    if (n == 0) return HK_OK;
    if (n >= (1u << 28)) return HK_ERR_ARG;
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    auto run = [&](auto ftag) -> hk_status {
        typedef decltype(ftag) F;
        size_t need = al256(n * sizeof(Affine<F>)) * 2 + al256(n * sizeof(Fr)) + al256(n * sizeof(XYZZ<F>)) +
                      al256(endo_tab_bytes<F>(n)) + al256(n * sizeof(F)) + 8192;
        HK_TRY(L->reserve(need));
        const void *pd, *sd;
        HK_TRY(to_device(L, points, n * sizeof(Affine<F>), &pd));
        HK_TRY(to_device(L, scalars, n * sizeof(Fr), &sd));
        XYZZ<F>* xy = L->alloc_n<XYZZ<F>>(n);
        F* pref = L->alloc_n<F>(n);
        XYZZ<F>* tab = (XYZZ<F>*)L->alloc_n<unsigned char>(endo_tab_bytes<F>(n));   // the chains' tables (endo.cuh)
        bool out_dev = is_device_ptr(out);
        Affine<F>* od = out_dev ? (Affine<F>*)out : L->alloc_n<Affine<F>>(n);
        if (!xy || !pref || !od || !tab) return HK_ERR_NOMEM;
        HK_TRY(MsmRun<F>::scalar_mul_each(L->stream, (const Affine<F>*)pd, sd, (u32)n, xy, pref, od, tab));
        if (!out_dev) HK_HIP(hipMemcpyAsync(out, od, n * sizeof(Affine<F>), hipMemcpyDeviceToHost, L->stream));
        HK_HIP(hipStreamSynchronize(L->stream));
        return HK_OK;
    };
    return group == 1 ? run(Fq()) : run(Fq2());
}

template <class C>
hk_status Ops<C>::points_lincomb(hk_ctx* ctx, int group, const void* const* vecs, const void* coeffs, size_t k,
                                 size_t n, void* out) {
    if (n == 0) return HK_OK;
    if (k == 0 || k > (size_t)LINCOMB_MAX || n >= (1u << 28)) return HK_ERR_ARG;
    for (size_t j = 0; j < k; j++) if (!vecs[j]) return HK_ERR_ARG;
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    auto run = [&](auto ftag) -> hk_status {
        typedef decltype(ftag) F;
        size_t need = (k + 1) * al256(n * sizeof(Affine<F>)) + al256(k * sizeof(Fr)) + al256(n * sizeof(XYZZ<F>)) +
                      al256(n * sizeof(F)) + 8192;
        HK_TRY(L->reserve(need));
        const Affine<F>* dv[LINCOMB_MAX];
        for (size_t j = 0; j < k; j++) {
            const void* d;
            HK_TRY(to_device(L, vecs[j], n * sizeof(Affine<F>), &d));
            dv[j] = (const Affine<F>*)d;
        }
        const void* cd;
        HK_TRY(to_device(L, coeffs, k * sizeof(Fr), &cd));
        XYZZ<F>* xy = L->alloc_n<XYZZ<F>>(n);
        F* pref = L->alloc_n<F>(n);
        bool out_dev = is_device_ptr(out);
        Affine<F>* od = out_dev ? (Affine<F>*)out : L->alloc_n<Affine<F>>(n);
        if (!xy || !pref || !od) return HK_ERR_NOMEM;
        HK_TRY(MsmRun<F>::lincomb(L->stream, dv, cd, (u32)k, (u32)n, xy, pref, od));
        if (!out_dev) HK_HIP(hipMemcpyAsync(out, od, n * sizeof(Affine<F>), hipMemcpyDeviceToHost, L->stream));
        HK_HIP(hipStreamSynchronize(L->stream));
        return HK_OK;
    };
    return group == 1 ? run(Fq()) : run(Fq2());
}

// device-resident input vectors are packed next to each other by ONE launch (a round of the aggregator's recursion hands
// over twelve windows of its arena: twelve 5 us copies in a row, and their twelve API calls, were 0.1 ms of a 4 ms call)
struct GatherRows {
    enum { MAX = 32 };
    const uint4* src[MAX];
    uint4* dst[MAX];
    u32 vecs[MAX];
};
template <class Tag>
__global__ void k_gather_rows(GatherRows g) {
    u32 r = blockIdx.y;
    const uint4* s = g.src[r];
    uint4* d = g.dst[r];
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < g.vecs[r]; i += gridDim.x * blockDim.x) d[i] = s[i];
}

// out_y[i] = lo_y[i] + c * hi_y[i] for k <= FOLD_MAX vector pairs and ONE scalar c, split by the caller along the group's
// endomorphism into K magnitudes and a sign mask (G2: four ~64-bit parts along psi, G1: two ~128-bit parts along phi): the
// folds of one TIPA round that share a challenge go out as one launch and one normalisation
template <class C>
template <class F>
hk_status Ops<C>::points_fold(hk_ctx* ctx, size_t k, const void* const* lo, const void* const* hi, const void* coeffs,
                              unsigned neg_mask, size_t n, void* const* out) {
    constexpr int K = EndoOf<F>::K;
    if (n == 0 || k == 0) return HK_OK;
    if (!lo || !hi || !coeffs || !out || k > (size_t)FOLD_MAX || n >= (1u << 28) / FOLD_MAX || neg_mask >= (1u << K)) return HK_ERR_ARG;
    for (size_t y = 0; y < k; y++) if (!lo[y] || !hi[y] || !out[y]) return HK_ERR_ARG;
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    size_t need = 3 * k * al256(n * sizeof(Affine<F>)) + al256(K * sizeof(Fr)) + al256(k * n * sizeof(XYZZ<F>)) +
                  al256(endo_tab_bytes<F>(n, k)) + al256(k * n * sizeof(F)) + 8192;
    HK_TRY(L->reserve(need));
    const Affine<F>*lod[FOLD_MAX], *hid[FOLD_MAX];
    for (size_t y = 0; y < k; y++) {
        const void* d;
        HK_TRY(to_device(L, lo[y], n * sizeof(Affine<F>), &d));
        lod[y] = (const Affine<F>*)d;
        HK_TRY(to_device(L, hi[y], n * sizeof(Affine<F>), &d));
        hid[y] = (const Affine<F>*)d;
    }
    Fr* cd = L->alloc_n<Fr>(K);
    if (!cd) return HK_ERR_NOMEM;
    HK_HIP(hipMemcpyAsync(cd, coeffs, K * sizeof(Fr), is_device_ptr(coeffs) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                          L->stream));
    if (!is_device_ptr(coeffs)) HK_HIP(hipStreamSynchronize(L->stream));      // a pageable caller buffer: done with it now
    XYZZ<F>* tab = (XYZZ<F>*)L->alloc_n<unsigned char>(endo_tab_bytes<F>(n, k));
    XYZZ<F>* xy = L->alloc_n<XYZZ<F>>(k * n);
    F* pref = L->alloc_n<F>(k * n);
    // one vector into a device buffer is normalised in place; otherwise into one array that is then handed out
    bool direct = k == 1 && is_device_ptr(out[0]);
    Affine<F>* od = direct ? (Affine<F>*)out[0] : L->alloc_n<Affine<F>>(k * n);
    if (!tab || !xy || !pref || !od) return HK_ERR_NOMEM;
    HK_TRY(MsmRun<F>::fold_endo(L->stream, (u32)k, lod, hid, cd, neg_mask, (u32)n, tab, xy, pref, od));
    if (!direct) {
        GatherRows gr;
        bool ok = n * sizeof(Affine<F>) < ((size_t)1 << 32);
        for (size_t y = 0; ok && y < k; y++) {
            ok = ((uintptr_t)out[y] & 15) == 0 && is_device_ptr(out[y]);
            gr.src[y] = (const uint4*)(od + y * n);
            gr.dst[y] = (uint4*)out[y];
            gr.vecs[y] = (u32)(n * sizeof(Affine<F>) / 16);
        }
        if (ok) {                                                   // the folded vectors go to their windows in one launch
            u32 gx = (gr.vecs[0] + 255) / 256;
            hipLaunchKernelGGL((k_gather_rows<Fr>), dim3(gx > 1024 ? 1024 : gx, (u32)k), dim3(256), 0, L->stream, gr);
            HK_HIP(hipGetLastError());
        } else {
            for (size_t y = 0; y < k; y++)
                HK_HIP(hipMemcpyAsync(out[y], od + y * n, n * sizeof(Affine<F>),
                                      is_device_ptr(out[y]) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, L->stream));
        }
    }
    HK_HIP(hipStreamSynchronize(L->stream));
    return HK_OK;
}

// out[i] = lo[i] + sum_{j<4} (+-) coeffs4[j] * psi^j(hi[i]) in G2: the fold lo + c * hi of a TIPA round with c split into
// four ~64-bit parts on the host (c = sum +-coeffs4[j] lambda^j mod r, lambda = psi's eigenvalue), so the shared doubling
// chain of the element-wise combination is ~66 steps instead of 254; one table add per step (endo.cuh)
template <class C>
hk_status Ops<C>::points_fold_g2(hk_ctx* ctx, const void* lo, const void* hi, const void* coeffs4, unsigned neg_mask,
                                 size_t n, void* out) {
    return points_fold<Fq2>(ctx, 1, &lo, &hi, coeffs4, neg_mask, n, &out);
}

// out[i] = lo[i] + (+-) coeffs2[0] * hi[i] + (+-) coeffs2[1] * phi(hi[i]) in G1: the G1 fold lo + c * hi with c split along the
// GLV endomorphism phi(x, y) = (BETA x, y) into two ~128-bit parts on the host (128 doubling steps instead of 254)
template <class C>
hk_status Ops<C>::points_fold_g1(hk_ctx* ctx, const void* lo, const void* hi, const void* coeffs2, unsigned neg_mask,
                                 size_t n, void* out) {
    return points_fold<Fq>(ctx, 1, &lo, &hi, coeffs2, neg_mask, n, &out);
}

template <class C>
hk_status Ops<C>::points_fold_many(hk_ctx* ctx, int group, size_t k, const void* const* lo, const void* const* hi,
                                   const void* coeffs, unsigned neg_mask, size_t n, void* const* out) {
    return group == 1 ? points_fold<Fq>(ctx, k, lo, hi, coeffs, neg_mask, n, out)
                      : points_fold<Fq2>(ctx, k, lo, hi, coeffs, neg_mask, n, out);
}

// z[i] = bits[i] ? 1 : 0 (Montgomery), then z[full_cols[k]] = full_vals[k]
template <class Fr>
__global__ void k_expand_bits(const unsigned char* __restrict__ bits, size_t n, Fr* __restrict__ z) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fr_store(&z[i], bits[i] ? Fr::one() : Fr::zero());
}
template <class Fr>
__global__ void k_scatter_full(const u32* __restrict__ cols, const Fr* __restrict__ vals, u32 k, size_t n, Fr* __restrict__ z) {
    u32 j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    u32 c = cols[j];
    if (c < n) fr_store(&z[c], fr_load(&vals[j]));
}

template <class C>
hk_status Ops<C>::assignment_from_bits(hk_ctx* ctx, const void* bits, size_t n_v, const uint32_t* full_cols,
                                       const void* full_vals, size_t n_full, void* z_out) {
    if (n_v == 0) return HK_OK;
    if (!is_device_ptr(z_out)) return HK_ERR_ARG;
    for (size_t k = 0; k < n_full; k++) if (!is_device_ptr(full_cols) && full_cols[k] >= n_v) return HK_ERR_ARG;
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    HK_TRY(L->reserve(al256(n_v) + al256(4 * n_full) + al256(sizeof(Fr) * n_full) + 4096));
    const void *bd, *cd = nullptr, *vd = nullptr;
    HK_TRY(to_device(L, bits, n_v, &bd));
    if (n_full) {
        HK_TRY(to_device(L, full_cols, 4 * n_full, &cd));
        HK_TRY(to_device(L, full_vals, sizeof(Fr) * n_full, &vd));
    }
    hipLaunchKernelGGL((k_expand_bits<Fr>), dim3((u32)((n_v + 255) / 256)), dim3(256), 0, L->stream, (const unsigned char*)bd, n_v,
                       (Fr*)z_out);
    if (n_full)
        hipLaunchKernelGGL((k_scatter_full<Fr>), dim3((u32)((n_full + 63) / 64)), dim3(64), 0, L->stream, (const u32*)cd,
                           (const Fr*)vd, (u32)n_full, n_v, (Fr*)z_out);
    HK_HIP(hipGetLastError());
    HK_HIP(hipStreamSynchronize(L->stream));
    return HK_OK;
}

// ---- word programs (witness.cuh) ---------------------------------------------------------------------------------
template <class C>
hk_status Ops<C>::wprog_upload(hk_ctx* ctx, const uint32_t* ops, size_t n_ops, const uint32_t* refs, size_t n_refs,
                               const uint32_t* map, size_t n_v, size_t n_values, size_t n_inputs, hk_wprog** out) {
    *out = nullptr;
    if (n_values >= (1u << 20) || n_ops == 0 || n_v == 0 || n_v >= ((size_t)1 << 32)) return HK_ERR_ARG;
    // validate on the host what the interpreter will index with: operand references, operand tables, the column map
    size_t vid = 0;
    auto ref_ok = [&](uint32_t r) { return (r & 0xfffffu) < vid; };
    for (size_t k = 0; k < n_ops; k++) {
        const uint32_t* o = ops + 8 * k;
        bool ok = true;
        switch (o[0]) {
            case WOP_INPUT: ok = o[4] < n_inputs; break;
            case WOP_CONST: break;
            case WOP_XOR: case WOP_AND: ok = ref_ok(o[1]) && ref_ok(o[2]); break;
            case WOP_CH: case WOP_MAJ: ok = ref_ok(o[1]) && ref_ok(o[2]) && ref_ok(o[3]); break;
            case WOP_ADD:
                ok = (size_t)o[1] + o[2] <= n_refs && o[2] <= 16;
                for (uint32_t j = 0; ok && j < o[2]; j++) ok = ref_ok(refs[o[1] + j]);
                vid++;
                break;
            case WOP_PACK4:
                ok = (size_t)o[1] + 4 <= n_refs;
                for (uint32_t j = 0; ok && j < 4; j++) ok = ref_ok(refs[o[1] + j]);
                break;
            case WOP_SHA_ROUND: case WOP_SHA_SCHED: {
                const uint32_t nr = o[0] == WOP_SHA_ROUND ? 9u : 4u;
                ok = (size_t)o[1] + nr <= n_refs;
                for (uint32_t j = 0; ok && j < nr; j++) ok = ref_ok(refs[o[1] + j]);
                vid += (o[0] == WOP_SHA_ROUND ? WOP_ROUND_VALUES : WOP_SCHED_VALUES) - 1;
                break;
            }
            default: ok = false;
        }
        if (!ok) return HK_ERR_ARG;
        vid++;
    }
    if (vid != n_values) return HK_ERR_ARG;
    for (size_t i = 0; i < n_v; i++)
        if (map[i] != 0xffffffffu && (map[i] >> 5) >= n_values) return HK_ERR_ARG;
    HK_HIP(hipSetDevice(ctx->device));
    WprogImpl* w = new WprogImpl();
    hk_wprog* h = new hk_wprog{ctx->ops, ctx, w};
    auto fail = [&](hk_status st) { Ops<C>::wprog_free(h); return st; };
    auto up = [&](u32** dst, const uint32_t* src, size_t n) -> bool {
        if (hipMalloc((void**)dst, 4 * (n ? n : 1)) != hipSuccess) { (void)hipGetLastError(); return false; }
        return n == 0 || hipMemcpy(*dst, src, 4 * n, hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(&w->ops, ops, 8 * n_ops) || !up(&w->refs, refs, n_refs) || !up(&w->map, map, n_v)) return fail(HK_ERR_NOMEM);
    w->n_ops = (u32)n_ops; w->n_refs = (u32)n_refs; w->n_values = (u32)n_values; w->n_inputs = (u32)n_inputs; w->n_v = n_v;
    *out = h;
    return HK_OK;
}

template <class C>
void Ops<C>::wprog_free(hk_wprog* h) {
    if (!h) return;
    (void)hipSetDevice(h->ctx->device);
    (void)hipDeviceSynchronize();
    for (u32* p : {h->impl->ops, h->impl->refs, h->impl->map}) if (p) (void)hipFree(p);
    delete h->impl;
    delete h;
}

template <class C>
hk_status Ops<C>::wprog_run(hk_ctx* ctx, const hk_wprog* h, const uint32_t* inputs, size_t batch,
                            const uint32_t* full_cols, const void* full_vals, size_t n_full, void* z_out) {
    const WprogImpl* w = h->impl;
    if (batch == 0) return HK_OK;
    if (batch >= (1u << 16) || !is_device_ptr(z_out)) return HK_ERR_ARG;
    if (!is_device_ptr(full_cols))
        for (size_t k = 0; k < n_full; k++) if (full_cols[k] >= w->n_v) return HK_ERR_ARG;
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    size_t need = al256(4 * batch * w->n_inputs) + al256(4 * (size_t)w->n_values * batch) + al256(4 * n_full) +
                  al256(sizeof(Fr) * n_full * batch) + 8192;
    HK_TRY(L->reserve(need));
    hipStream_t s = L->stream;
    const void *in_d, *cd = nullptr, *vd = nullptr;
    HK_TRY(to_device(L, inputs, 4 * batch * w->n_inputs, &in_d));
    if (n_full) {
        HK_TRY(to_device(L, full_cols, 4 * n_full, &cd));
        HK_TRY(to_device(L, full_vals, sizeof(Fr) * n_full * batch, &vd));
    }
    u32* values = L->alloc_n<u32>((size_t)w->n_values * batch);
    if (!values) return HK_ERR_NOMEM;
    hipLaunchKernelGGL((k_word_program<0>), dim3((u32)((batch + 63) / 64)), dim3(64), 0, s, w->ops, w->n_ops, w->refs,
                       (const u32*)in_d, w->n_inputs, (u32)batch, values);
    hipLaunchKernelGGL((k_witness_expand<Fr>), dim3((u32)((w->n_v + 255) / 256), (u32)batch), dim3(256), 0, s, w->map, w->n_v,
                       (const u32*)values, (u32)batch, (Fr*)z_out);
    if (n_full)
        hipLaunchKernelGGL((k_scatter_full_batch<Fr>), dim3((u32)((n_full + 63) / 64), (u32)batch), dim3(64), 0, s,
                           (const u32*)cd, (const Fr*)vd, (u32)n_full, w->n_v, (Fr*)z_out);
    HK_HIP(hipGetLastError());
    HK_HIP(hipStreamSynchronize(s));
    return HK_OK;
}

// z_out[b][full_cols[j]] = full_vals[b][j]: the full-width values alone, for a caller that ran the class's word program
// earlier (hk_wprog_run with n_full = 0) and learns the values that depend on the round's challenges later
template <class C>
hk_status Ops<C>::assignment_scatter(hk_ctx* ctx, const uint32_t* full_cols, const void* full_vals, size_t n_full, size_t batch,
                                     size_t n_v, void* z_out) {
    if (batch == 0 || n_full == 0) return HK_OK;
    if (batch >= (1u << 16) || !is_device_ptr(z_out)) return HK_ERR_ARG;
    if (!is_device_ptr(full_cols))
        for (size_t k = 0; k < n_full; k++) if (full_cols[k] >= n_v) return HK_ERR_ARG;
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    HK_TRY(L->reserve(al256(4 * n_full) + al256(sizeof(Fr) * n_full * batch) + 4096));
    const void *cd, *vd;
    HK_TRY(to_device(L, full_cols, 4 * n_full, &cd));
    HK_TRY(to_device(L, full_vals, sizeof(Fr) * n_full * batch, &vd));
    hipLaunchKernelGGL((k_scatter_full_batch<Fr>), dim3((u32)((n_full + 63) / 64), (u32)batch), dim3(64), 0, L->stream,
                       (const u32*)cd, (const Fr*)vd, (u32)n_full, n_v, (Fr*)z_out);
    HK_HIP(hipGetLastError());
    HK_HIP(hipStreamSynchronize(L->stream));
    return HK_OK;
}

template <class C>
hk_status Ops<C>::poseidon_path(hk_ctx* ctx, const void* consts, size_t n_consts, const hk_poseidon_desc* lh,
                                const hk_poseidon_desc* nh, const void* leaf, const void* siblings, const uint32_t* index,
                                size_t depth, size_t batch, size_t n_v, size_t col0, void* z_out) {
    if (batch == 0) return HK_OK;
    if (!consts || !lh || !nh || !leaf || !index || (depth && !siblings) || !is_device_ptr(z_out)) return HK_ERR_ARG;
    if (batch >= (1u << 20) || depth > 32) return HK_ERR_ARG;
    auto per_perm = [](const hk_poseidon_desc* d) -> size_t {          // witnesses of one permutation
        size_t chain = d->alpha == 5 ? 3 : 5;
        return (size_t)d->full_rounds * (d->t * chain + d->t) + (size_t)d->partial_rounds * (chain + d->t);
    };
    for (const hk_poseidon_desc* d : {lh, nh}) {
        if (d->t < 2 || d->t > 4 || (d->alpha != 5 && d->alpha != 17) || (d->full_rounds & 1) ||
            (size_t)d->consts_offset + (size_t)(d->full_rounds + d->partial_rounds) * d->t + (size_t)d->t * d->t > n_consts)
            return HK_ERR_ARG;
    }
    // the kernel is compiled for the reference's two instances (poseidon_util.rs:53-62): rate 3 / x^5 over the 4 leaf
    // fields, rate 2 / x^17 for two-to-one; round counts and constants stay run-time data
    if (lh->t != 4 || nh->t != 3 || lh->alpha != 5 || nh->alpha != 17) return HK_ERR_ARG;
    size_t block = 2 * per_perm(lh) + depth * (3 + per_perm(nh));
    if (col0 > n_v || block > n_v - col0) return HK_ERR_ARG;
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    HK_TRY(L->reserve(al256(n_consts * sizeof(Fr)) + al256(batch * 4 * sizeof(Fr)) + al256(batch * depth * sizeof(Fr)) +
                      al256(4 * batch) + 4096));
    const void *cd, *ld, *sd = nullptr, *id;
    HK_TRY(to_device(L, consts, n_consts * sizeof(Fr), &cd));
    HK_TRY(to_device(L, leaf, batch * 4 * sizeof(Fr), &ld));
    if (depth) HK_TRY(to_device(L, siblings, batch * depth * sizeof(Fr), &sd));
    HK_TRY(to_device(L, index, 4 * batch, &id));
    PoseidonDesc a{lh->t, lh->alpha, lh->full_rounds, lh->partial_rounds, lh->consts_offset};
    PoseidonDesc b{nh->t, nh->alpha, nh->full_rounds, nh->partial_rounds, nh->consts_offset};
    hipLaunchKernelGGL((k_poseidon_path<Fr>), dim3((u32)((batch + 63) / 64)), dim3(64), 0, L->stream, (const Fr*)cd, a, b,
                       (const Fr*)ld, (const Fr*)sd, (const u32*)id, (u32)depth, (u32)batch, n_v, col0, (Fr*)z_out);
    HK_HIP(hipGetLastError());
    HK_HIP(hipStreamSynchronize(L->stream));
    return HK_OK;
}

// ---- multi-pairings (pairing.cuh) ------------------------------------------------------------------------------
template <class C>
hk_status Ops<C>::pairing_products(hk_ctx* ctx, const void* const* lhs, size_t n_lhs, const void* const* rhs,
                                   size_t n_rhs, size_t n, void* out) {
    return pairing_pairs(ctx, lhs, n_lhs, rhs, n_rhs, nullptr, nullptr, 0, n, out);
}

// pair_lhs == nullptr: every (lhs, rhs) pair, out[a * n_rhs + b]; else out[p] for the n_pairs listed pairs
template <class C>
hk_status Ops<C>::pairing_pairs(hk_ctx* ctx, const void* const* lhs, size_t n_lhs, const void* const* rhs, size_t n_rhs,
                                const uint32_t* pair_lhs, const uint32_t* pair_rhs, size_t n_pairs, size_t n, void* out) {
    typedef typename Fq::Params P;
    typedef Fp12<P> GT;
    PairList pl;
    pl.n = 0;
    if (pair_lhs || pair_rhs) {
        if (!pair_lhs || !pair_rhs || n_pairs == 0 || n_pairs > (size_t)PAIR_LIST_MAX || n_lhs > 255 || n_rhs > 255) return HK_ERR_ARG;
        for (size_t k = 0; k < n_pairs; k++) {
            if (pair_lhs[k] >= n_lhs || pair_rhs[k] >= n_rhs) return HK_ERR_ARG;
            pl.a[k] = (unsigned char)pair_lhs[k];
            pl.b[k] = (unsigned char)pair_rhs[k];
        }
        pl.n = (u32)n_pairs;
    }
    size_t count = pl.n ? pl.n : n_lhs * n_rhs;
    if (count == 0 || count > 4096 || n_lhs == 0 || n_rhs == 0 || !out) return HK_ERR_ARG;
    if (n * count >= ((size_t)1 << 31)) return HK_ERR_ARG;
    // the per-step product trees run as grid (groups, count * steps): grid.y is a 16-bit quantity
    if (count * PairRun<P>::steps() > 65535 || n_rhs > 65535) return HK_ERR_ARG;
    if (n == 0) {                                                  // empty product: 1 (final_exponentiation(1) = 1)
        GT one = f12_one<P>();
        HK_HIP(hipSetDevice(ctx->device));
        for (size_t k = 0; k < count; k++)                        // `out` may be a device pointer, as on the n > 0 path
            HK_HIP(hipMemcpy((char*)out + k * sizeof(GT), &one, sizeof(GT), is_device_ptr(out) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
        return HK_OK;
    }
    for (size_t a = 0; a < n_lhs; a++) if (!lhs[a]) return HK_ERR_ARG;
    for (size_t b = 0; b < n_rhs; b++) if (!rhs[b]) return HK_ERR_ARG;
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    size_t g1b = sizeof(Affine<Fq>), g2b = sizeof(Affine<Fq2>);
    size_t mbytes = PairRun<P>::scratch_bytes((u32)n, (u32)count, (u32)n_rhs);
    size_t need = al256(n_lhs * n * g1b) + al256(n_rhs * n * g2b) + al256(mbytes) +
                  2 * al256(count * sizeof(GT)) + 8192;
    HK_TRY(L->reserve(need));
    hipStream_t s = L->stream;
    Affine<Fq>* d1 = L->alloc_n<Affine<Fq>>(n_lhs * n);
    Affine<Fq2>* d2 = L->alloc_n<Affine<Fq2>>(n_rhs * n);
    GT* miller = (GT*)L->alloc(mbytes);                   // lines + per-step tree buffers (or the serial path's Miller values)
    GT* prod = L->alloc_n<GT>(count);
    GT* res = L->alloc_n<GT>(count);
    if (!d1 || !d2 || !miller || !prod || !res) return HK_ERR_NOMEM;
    bool packed = false;
    if (n_lhs + n_rhs <= (size_t)GatherRows::MAX && n * g2b < ((size_t)1 << 32)) {
        GatherRows gr;
        u32 most = 0;
        bool ok = true;
        for (size_t k = 0; ok && k < n_lhs + n_rhs; k++) {
            const void* src = k < n_lhs ? lhs[k] : rhs[k - n_lhs];
            ok = ((uintptr_t)src & 15) == 0 && is_device_ptr(src);
            gr.src[k] = (const uint4*)src;
            gr.dst[k] = k < n_lhs ? (uint4*)(d1 + k * n) : (uint4*)(d2 + (k - n_lhs) * n);
            gr.vecs[k] = (u32)(n * (k < n_lhs ? g1b : g2b) / 16);
            if (gr.vecs[k] > most) most = gr.vecs[k];
        }
        if (ok) {
            u32 gx = (most + 255) / 256;
            if (gx > 1024) gx = 1024;
            hipLaunchKernelGGL((k_gather_rows<Fr>), dim3(gx, (u32)(n_lhs + n_rhs)), dim3(256), 0, s, gr);
            HK_HIP(hipGetLastError());
            packed = true;
        }
    }
    if (!packed) {
        for (size_t a = 0; a < n_lhs; a++) HK_HIP(hipMemcpyAsync(d1 + a * n, lhs[a], n * g1b, h2d_kind(lhs[a]), s));
        for (size_t b = 0; b < n_rhs; b++) HK_HIP(hipMemcpyAsync(d2 + b * n, rhs[b], n * g2b, h2d_kind(rhs[b]), s));
    }
    HK_TRY(PairRun<P>::run(s, d1, d2, (u32)n, (u32)n_lhs, (u32)n_rhs, miller, prod, res, pl.n ? &pl : nullptr));
    HK_HIP(hipMemcpyAsync(out, res, count * sizeof(GT), is_device_ptr(out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    HK_HIP(hipStreamSynchronize(s));
    return HK_OK;
}

template <class C>
hk_status Ops<C>::gt_pow(hk_ctx* ctx, const void* gt_in, const void* scalars, size_t n, void* gt_out, int in_gt, size_t group_len) {
    typedef typename Fq::Params P;
    typedef Fp12<P> GT;
    if (n == 0) return HK_OK;
    if (n >= (1u << 20)) return HK_ERR_ARG;
    // group_len > 1: out[g] = prod_{j < group_len} in[g * group_len + j]^scalars[...] (a verifier's multi-exponentiations)
    if (group_len == 0 || n % group_len != 0 || n / group_len > 65535) return HK_ERR_ARG;
    size_t n_out = n / group_len;
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    HK_TRY(L->reserve(3 * al256(n * sizeof(GT)) + al256(n * sizeof(Fr)) + 4096));
    const void *ind, *sd;
    HK_TRY(to_device(L, gt_in, n * sizeof(GT), &ind));
    HK_TRY(to_device(L, scalars, n * sizeof(Fr), &sd));
    bool out_dev = is_device_ptr(gt_out);
    GT* pw = (out_dev && group_len == 1) ? (GT*)gt_out : L->alloc_n<GT>(n);
    if (!pw) return HK_ERR_NOMEM;
    HK_TRY(PairRun<P>::gt_pow(L->stream, (const GT*)ind, sd, (u32)n, pw, in_gt != 0));
    GT* od = pw;
    if (group_len > 1) {
        od = out_dev ? (GT*)gt_out : L->alloc_n<GT>(n_out);
        if (!od) return HK_ERR_NOMEM;
        HK_TRY(PairRun<P>::gt_prod(L->stream, pw, (u32)group_len, (u32)n_out, od));
    }
    if (!out_dev) HK_HIP(hipMemcpyAsync(gt_out, od, n_out * sizeof(GT), hipMemcpyDeviceToHost, L->stream));
    HK_HIP(hipStreamSynchronize(L->stream));
    return HK_OK;
}

// out[i] = in[i] * R (to_mont) or in[i] / R; memory canonical either way
template <class F>
__global__ void k_field_convert(const F* __restrict__ in, F* __restrict__ out, size_t n, int to_mont) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    F x = ld_vec(&in[i]);
    st_vec(&out[i], to_mont ? F::to_mont(x) : F::from_mont(x));
}

template <class C>
hk_status Ops<C>::field_convert(hk_ctx* ctx, int which, const void* in, void* out, size_t n, int to_mont) {
    if (n == 0) return HK_OK;
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    auto run = [&](auto ftag) -> hk_status {
        typedef decltype(ftag) F;
        const size_t CH = (size_t)1 << 24;                       // host buffers go through the lane in chunks
        bool in_dev = is_device_ptr(in), out_dev = is_device_ptr(out);
        HK_TRY(L->reserve(2 * al256(std::min(n, CH) * sizeof(F)) + 4096));
        for (size_t off = 0; off < n; off += CH) {
            size_t k = std::min(CH, n - off);
            L->arena_off = 0;
            const F* src = (const F*)in + off;
            F* dst = (F*)out + off;
            const F* sd = src;
            if (!in_dev) {
                F* t = L->alloc_n<F>(k);
                if (!t) return HK_ERR_NOMEM;
                HK_HIP(hipMemcpyAsync(t, src, k * sizeof(F), hipMemcpyHostToDevice, L->stream));
                sd = t;
            }
            F* dd = out_dev ? dst : L->alloc_n<F>(k);
            if (!dd) return HK_ERR_NOMEM;
            hipLaunchKernelGGL(k_field_convert<F>, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, L->stream, sd, dd, k, to_mont);
            HK_HIP(hipGetLastError());
            if (!out_dev) HK_HIP(hipMemcpyAsync(dst, dd, k * sizeof(F), hipMemcpyDeviceToHost, L->stream));
            HK_HIP(hipStreamSynchronize(L->stream));
        }
        return HK_OK;
    };
    return which == 0 ? run(Fr()) : run(Fq());
}

template <class C>
void Ops<C>::ctx_release(hk_ctx* ctx) {
    for (auto& e : ctx->fb_cache)
        if (e.table) (void)hipFree(e.table);
    ctx->fb_cache.clear();
    if (!ctx->ntt) return;
    NttTables* T = ctx->ntt;
    for (void* p : {T->tw_fwd, T->tw_inv, T->pw_g, T->pw_ginv})
        if (p) (void)hipFree(p);
    for (void* p : T->retired) (void)hipFree(p);
    delete T;
    ctx->ntt = nullptr;
}

// ---- small device helpers for the fused calls -------------------------------------------------------------
// ext[0] = r, ext[1] = s, ext[2] = r*s, ext[3+i] = kappa_i   (all Montgomery)
// Also clears the bucket counters of the proof's digit sorts (up to three arrays of `nb` u32 each; every workgroup takes a
// share): they are accumulated with atomics by k_msm_hist, and this kernel precedes every sort of the proof in stream
// order (the side streams wait for the event recorded behind it) - three memset launches less per proof.
template <class Fr>
__global__ void k_prep_ext(Fr* __restrict__ ext, const Fr* __restrict__ rs_kappas, u32 n_kappas, u32* __restrict__ c0,
                           u32* __restrict__ c1, u32* __restrict__ c2, u32 nb0, u32 nb1, u32 nb2) {
    u32 gt = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (u32 i = gt; i < nb0; i += stride) c0[i] = 0;
    for (u32 i = gt; i < nb1; i += stride) c1[i] = 0;
    for (u32 i = gt; i < nb2; i += stride) c2[i] = 0;
    if (blockIdx.x || threadIdx.x) return;
    Fr r = fr_load(&rs_kappas[0]), s = fr_load(&rs_kappas[1]);
    fr_store(&ext[0], r);
    fr_store(&ext[1], s);
    fr_store(&ext[2], Fr::mul(r, s));
    for (u32 i = 0; i < n_kappas; i++) fr_store(&ext[3 + i], fr_load(&rs_kappas[2 + i]));
}

// Finish: A = MA + a_g[0] + alpha_g ; B = MB2 + b_h[0] + beta_h ; B1 = MB1 + b_g[0] + beta_g ;
//         C = s*A + r*B1 + ML' + MH   with ML' = L - rs*delta_g - sum kappa_i*delta_i   (see file header)
// (prover.rs:135-155 "Finish C" + into_affine, committer.rs:112-114).  One lane per output point.
template <class Fr, class Fq, class Fq2>
__global__ void k_finish(const XYZZ<Fq>* __restrict__ res_g1,   // MA, MB1, ML, MH
                         const XYZZ<Fq2>* __restrict__ res_g2,  // MB2
                         const Affine<Fq>* __restrict__ c1, const Affine<Fq2>* __restrict__ c2,
                         const Fr* __restrict__ rs,              // r, s (Montgomery)
                         Affine<Fq>* __restrict__ out_a, Affine<Fq2>* __restrict__ out_b,
                         Affine<Fq>* __restrict__ out_c, EndoSplit<2> E) {
    if (blockIdx.x < 2 && threadIdx.x) return;
    if (blockIdx.x == 0) {
        XYZZ<Fq> A = ec_madd_ni(ec_madd_ni(ld_vec(&res_g1[0]), ld_vec(&c1[0])), ld_vec(&c1[1]));
        st_vec(out_a, ec_to_affine(A));
    } else if (blockIdx.x == 1) {
        XYZZ<Fq2> B = ec_madd_ni(ec_madd_ni(ld_vec(&res_g2[0]), ld_vec(&c2[0])), ld_vec(&c2[1]));
        st_vec(out_b, ec_to_affine(B));
    } else {
        // C = s*A + r*B1 + ML' + MH.  This kernel is the tail of every proof's latency, and its two variable-base products
        // were one lane's chain of 256 doublings + <= 128 additions (Straus over a joint table: 3 ms).  They now run as
        // the short element-wise sweeps do (endo.cuh): A and B1 are normalised by lanes 0 and 1, then FOUR lanes take one
        // GLV half each (s = s0 + s1 lambda on A, r = r0 + r1 lambda on B1: <= 131 bits) with a signed 4-bit window over
        // 1P .. 8P and a Jacobian chain - 34 digits x (4 doublings + 1 add) - and lane 0 joins the four partial products.
        __shared__ Affine<Fq> base[2];
        __shared__ Jac<Fq> tab[4][SPLIT_TABLE];
        __shared__ Jac<Fq> part[4];
        const u32 t = threadIdx.x;
        if (t < 2) {
            XYZZ<Fq> X = ec_madd_ni(ec_madd_ni(ld_vec(&res_g1[t]), ld_vec(&c1[2 * t])), ld_vec(&c1[2 * t + 1]));   // A | B1
            base[t] = ec_to_affine(X);
        }
        __syncthreads();
        if (t < 4) {
            constexpr int ND = SplitDigits<Fq>::ND;
            Fr k = Fr::from_mont(fr_load(&rs[(t >> 1) == 0 ? 1 : 0]));          // lanes 0, 1: s (on A);  lanes 2, 3: r (on B1)
            u32 c[8], mag[2][6];
            HK_UNROLL for (int l = 0; l < 8; l++) c[l] = l < Fr::N ? k.v[l] : 0u;
            u32 neg = endo_decompose<2>(c, E, mag);
            u32 m[6];
            HK_UNROLL for (int l = 0; l < 6; l++) m[l] = (t & 1u) ? mag[1][l] : mag[0][l];
            split_bias<ND>(m);
            Affine<Fq> q = base[t >> 1];
            Jac<Fq> acc = Jac<Fq>::inf();
            const bool q_inf = q.is_inf();
            if (!q_inf) {
                if (t & 1u) q = EndoOf<Fq>::apply(q);
                if ((neg >> (t & 1u)) & 1u) q.y = Fq::neg(q.y);
                Jac<Fq> e = Jac<Fq>::from_affine(q);
                tab[t][0] = e;
                HK_NOUNROLL for (int i = 2; i <= SPLIT_TABLE; i++) {
                    Jac<Fq> prev = (i & 1) ? tab[t][i - 2] : tab[t][i / 2 - 1];
                    e = (i & 1) ? jac_madd_ni(prev, q) : jac_dbl_ni(prev);
                    tab[t][i - 1] = e;
                }
                HK_NOUNROLL for (int d = ND - 1; d >= 0; d--) {
                    int dig = split_digit(m, d);
                    if (dig == 0 && acc.is_inf()) continue;
                    HK_NOUNROLL for (int r4 = 0; r4 < 4; r4++) acc = jac_dbl_ni(acc);
                    if (dig != 0) {
                        Jac<Fq> e2 = tab[t][(dig < 0 ? -dig : dig) - 1];
                        if (dig < 0) e2.y = Fq::neg(e2.y);
                        acc = jac_add_ni(acc, e2);
                    }
                }
            }
            part[t] = acc;
        }
        __syncthreads();
        if (t == 0) {
            Jac<Fq> sum = part[0];
            HK_NOUNROLL for (int i = 1; i < 4; i++) sum = jac_add_ni(sum, part[i]);
            XYZZ<Fq> Cc = XYZZ<Fq>::inf();
            if (!sum.is_inf()) {
                Cc.x = sum.x; Cc.y = sum.y;
                Cc.zz = Fq::sqr(sum.z);
                Cc.zzz = Fq::mul(Cc.zz, sum.z);
            }
            Cc = ec_add_ni(Cc, ld_vec(&res_g1[2]));
            Cc = ec_add_ni(Cc, ld_vec(&res_g1[3]));
            st_vec(out_c, ec_to_affine(Cc));
        }
    }
}

template <class C>
size_t Ops<C>::finish_private_bytes() {
    return hk_private_bytes_of((const void*)k_finish<Fr, Fq, Fq2>);
}

static inline float ev_ms(hipEvent_t a, hipEvent_t b) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) { (void)hipGetLastError(); return 0.f; }
    return ms;
}

template <class C>
hk_status Ops<C>::commit(hk_ctx* ctx, const hk_pk* h, size_t stage, const void* w, size_t n,
                         const void* kappa, void* out) {
    if (h->ctx != ctx) return HK_ERR_ARG;                  // a key lives on the context (device) that uploaded it
    PkImpl<C>* pk = (PkImpl<C>*)h->impl;
    if (stage >= pk->n_stages) return HK_ERR_ARG;          // "no more values left in committing key"
    if (n + 1 != pk->ck_n[stage]) return HK_ERR_LEN;       // committer.rs:83
    if (n && !w) return HK_ERR_ARG;
    // a short stage (the 16 stage-0 witnesses of a big-merkle subcircuit) takes no bucket pass: the one-row form of
    // hk_commit_batch - 17 element-wise products over the endomorphism and one sum, 1.4 ms instead of 2.2
    if ((n + 1) * EndoOf<Fq>::K <= SPLIT_MAX_LANES && !is_device_ptr(out) && !getenv("HK_MSM_NO_SMALL"))
        return commit_batch(ctx, h, stage, w, n, kappa, 1, out);
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    const MsmPlan& p = pk->plan_ck[stage];
    size_t need = al256((n + 1) * sizeof(Fr)) + msm_sort_bytes(p) + msm_run_bytes<Fq>(p) + 4096;
    HK_TRY(L->reserve(need));
    hipStream_t s = L->stream;
    bool prof = ctx->profiling;
    if (prof) HK_HIP(hipEventRecord(L->ev[0], s));
    Fr* sc = L->alloc_n<Fr>(n + 1);
    if (!sc) return HK_ERR_NOMEM;
    if (n) HK_HIP(hipMemcpyAsync(sc, w, n * sizeof(Fr), h2d_kind(w), s));
    HK_HIP(hipMemcpyAsync(sc + n, kappa, sizeof(Fr), hipMemcpyHostToDevice, s));
    SortBufs sb;
    HK_TRY(MsmSort<Fr>::alloc(L, p, &sb));
    typename MsmRun<Fq>::Bufs rb;
    HK_TRY(MsmRun<Fq>::alloc(L, p, &rb));
    XYZZ<Fq>* res = L->alloc_n<XYZZ<Fq>>(1);
    Affine<Fq>* aff = L->alloc_n<Affine<Fq>>(1);
    if (!res || !aff) return HK_ERR_NOMEM;
    HK_TRY(MsmSort<Fr>::run(s, p, (const u32*)sc, 1, sb));
    HK_TRY(MsmRun<Fq>::run(s, p, pk->ck_tab[stage], (u32)(n + 1), 0, sb, rb, res, prof ? L->ev[20] : nullptr,
                           prof ? L->ev[21] : nullptr));
    HK_TRY(MsmRun<Fq>::to_affine(s, res, aff, 1));
    HK_HIP(hipMemcpyAsync(out, aff, sizeof(Affine<Fq>), hipMemcpyDeviceToHost, s));
    if (prof) HK_HIP(hipEventRecord(L->ev[1], s));
    HK_HIP(hipStreamSynchronize(s));
    if (prof) {
        memset(&L->timings, 0, sizeof(L->timings));
        L->timings.total_ms = ev_ms(L->ev[0], L->ev[1]);
        L->timings.accum_kernel_ms = ev_ms(L->ev[20], L->ev[21]);
        L->timings.accum_kernel_launches = 1;
    }
    return HK_OK;
}

// `batch` commitments under one key and stage.  Short stages (batch (n + 1) K <= SPLIT_MAX_LANES: the 16 stage-0 witnesses
// of a big-merkle subcircuit) run as ONE set of launches - every term an element-wise product over the endomorphism on its
// own lanes, one workgroup's sum per commitment, one normalisation - instead of `batch` bucket passes of ten tiny launches
// each; longer stages fall back to hk_commit per row.
template <class C>
hk_status Ops<C>::commit_batch(hk_ctx* ctx, const hk_pk* h, size_t stage, const void* w, size_t n, const void* kappas,
                               size_t batch, void* out) {
    if (h->ctx != ctx) return HK_ERR_ARG;
    PkImpl<C>* pk = (PkImpl<C>*)h->impl;
    if (stage >= pk->n_stages) return HK_ERR_ARG;
    if (n + 1 != pk->ck_n[stage]) return HK_ERR_LEN;
    if (batch == 0) return HK_OK;
    if ((n && !w) || !kappas || !out) return HK_ERR_ARG;
    const size_t seg = n + 1, tot = seg * batch;
    if (tot * EndoOf<Fq>::K > SPLIT_MAX_LANES || is_device_ptr(kappas) || getenv("HK_MSM_NO_SMALL")) {
        if (is_device_ptr(kappas) || is_device_ptr(out)) return HK_ERR_ARG;
        for (size_t b = 0; b < batch; b++)
            HK_TRY(commit(ctx, h, stage, n ? (const char*)w + b * n * sizeof(Fr) : nullptr, n, (const char*)kappas + b * sizeof(Fr),
                          (char*)out + b * sizeof(Affine<Fq>)));
        return HK_OK;
    }
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    size_t need = al256(tot * sizeof(Fr)) + al256(tot * sizeof(XYZZ<Fq>)) + al256(endo_tab_bytes<Fq>(tot)) +
                  al256(batch * sizeof(XYZZ<Fq>)) + al256(batch * sizeof(Fq)) + al256(batch * sizeof(Affine<Fq>)) + 8192;
    HK_TRY(L->reserve(need));
    hipStream_t s = L->stream;
    bool prof = ctx->profiling;
    if (prof) HK_HIP(hipEventRecord(L->ev[0], s));
    Fr* sc = L->alloc_n<Fr>(tot);                              // [batch][n + 1]: a row's witnesses, then its kappa
    XYZZ<Fq>* xy = L->alloc_n<XYZZ<Fq>>(tot);
    XYZZ<Fq>* tab = (XYZZ<Fq>*)L->alloc_n<unsigned char>(endo_tab_bytes<Fq>(tot));
    XYZZ<Fq>* res = L->alloc_n<XYZZ<Fq>>(batch);
    Fq* pref = L->alloc_n<Fq>(batch);
    Affine<Fq>* aff = L->alloc_n<Affine<Fq>>(batch);
    if (!sc || !xy || !tab || !res || !pref || !aff) return HK_ERR_NOMEM;
    if (n) HK_HIP(hipMemcpy2DAsync(sc, seg * sizeof(Fr), w, n * sizeof(Fr), n * sizeof(Fr), batch, h2d_kind(w), s));
    HK_HIP(hipMemcpy2DAsync(sc + n, seg * sizeof(Fr), kappas, sizeof(Fr), sizeof(Fr), batch, hipMemcpyHostToDevice, s));
    HK_TRY(MsmRun<Fq>::small_msm_rows(s, pk->ck_tab[stage], sc, (u32)seg, (u32)batch, tab, xy, res));   // group 0 of the table
    HK_TRY(MsmRun<Fq>::batch_affine(s, res, aff, pref, (u32)batch));
    HK_HIP(hipMemcpyAsync(out, aff, batch * sizeof(Affine<Fq>), is_device_ptr(out) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    if (prof) HK_HIP(hipEventRecord(L->ev[1], s));
    HK_HIP(hipStreamSynchronize(s));
    if (prof) {
        memset(&L->timings, 0, sizeof(L->timings));
        L->timings.total_ms = ev_ms(L->ev[0], L->ev[1]);
    }
    return HK_OK;
}

template <class C>
hk_status Ops<C>::prove(hk_ctx* ctx, const hk_pk* h, const void* z, size_t n_v, const void* r_m,
                        const void* s_m, const void* kappas, size_t n_kappas, void* out_a, void* out_b,
                        void* out_c) {
    typedef QapHost<C> Q;
    if (h->ctx != ctx) return HK_ERR_ARG;
    PkImpl<C>* pk = (PkImpl<C>*)h->impl;
    if (!pk->has_qap) return HK_ERR_ARG;
    if (n_v != pk->n_v) return HK_ERR_LEN;
    if (n_kappas + 1 != pk->n_stages) return HK_ERR_LEN;   // committer.rs:112 assert
    if (n_kappas && !kappas) return HK_ERR_ARG;
    NttTables* T;
    HK_TRY(NttHost<C>::ensure(ctx, pk->log_m, &T));
    LaneGuard g(ctx);
    Lane* L = g.lane;
    if (!L) return HK_ERR_DEVICE;
    const MsmPlan &pz = pk->plan_z, &ph = pk->plan_h;
    size_t m = (size_t)1 << pk->log_m;
    const MsmPlan& pb = pk->plan_b;
    size_t need = al256(sizeof(Fr) * pk->n_ext) + al256(sizeof(Fr) * n_v) + msm_sort_bytes(pz) +
                  msm_sort_bytes(ph) + 2 * msm_run_bytes<Fq>(pz) + msm_run_bytes<Fq>(pb) + msm_run_bytes<Fq>(ph) +
                  msm_run_bytes<Fq2>(pb) + al256(3 * m * sizeof(Fr)) + 16384;
    if (pk->b_compact) need += msm_sort_bytes(pb) + al256(sizeof(Fr) * pk->b_n);
    HK_TRY(L->reserve(need));
    hipStream_t s = L->stream;
    bool prof = ctx->profiling;
    hipEvent_t* ev = L->ev;
    int e = 0;
    auto mark = [&]() -> hk_status { if (prof) HK_HIP(hipEventRecord(ev[e], s)); e++; return HK_OK; };
    HK_TRY(mark());                                                            // ev0
    // --- extended scalar vector: z[1..] | r | s | rs | kappas
    Fr* zext = L->alloc_n<Fr>(pk->n_ext);
    Fr* small = L->alloc_n<Fr>(2 + n_kappas);
    if (!zext || !small) return HK_ERR_NOMEM;
    const Fr* zd;
    if (is_device_ptr(z)) zd = (const Fr*)z;
    else {
        Fr* t = L->alloc_n<Fr>(n_v);
        if (!t) return HK_ERR_NOMEM;
        HK_HIP(hipMemcpyAsync(t, z, n_v * sizeof(Fr), hipMemcpyHostToDevice, s));
        zd = t;
    }
    if (n_v > 1) HK_HIP(hipMemcpyAsync(zext, zd + 1, (n_v - 1) * sizeof(Fr), hipMemcpyDeviceToDevice, s));
    HK_HIP(hipMemcpyAsync(small, r_m, sizeof(Fr), hipMemcpyHostToDevice, s));
    HK_HIP(hipMemcpyAsync(small + 1, s_m, sizeof(Fr), hipMemcpyHostToDevice, s));
    if (n_kappas) HK_HIP(hipMemcpyAsync(small + 2, kappas, n_kappas * sizeof(Fr), hipMemcpyHostToDevice, s));
    // --- one digit sort shared by the four assignment-indexed queries
    SortBufs sb, sbh, sbb;
    sbb.count = nullptr;
    HK_TRY(MsmSort<Fr>::alloc(L, pz, &sb));
    HK_TRY(MsmSort<Fr>::alloc(L, ph, &sbh));
    Fr* zb = nullptr;
    if (pk->b_compact) {
        HK_TRY(MsmSort<Fr>::alloc(L, pb, &sbb));
        zb = L->alloc_n<Fr>(pk->b_n);
        if (!zb) return HK_ERR_NOMEM;
    }
    hipLaunchKernelGGL((k_prep_ext<Fr>), dim3(64), dim3(256), 0, s, zext + (n_v - 1), small, (u32)n_kappas, sb.count,
                       sbh.count, sbb.count, pz.NB, ph.NB, pk->b_compact ? pb.NB : 0u);
    typename MsmRun<Fq>::Bufs rbA, rbB1, rbL, rbh;
    typename MsmRun<Fq2>::Bufs rb2;
    HK_TRY(MsmRun<Fq>::alloc(L, pz, &rbA));
    HK_TRY(MsmRun<Fq>::alloc(L, pb, &rbB1));
    HK_TRY(MsmRun<Fq>::alloc(L, pz, &rbL));
    HK_TRY(MsmRun<Fq>::alloc(L, ph, &rbh));
    HK_TRY(MsmRun<Fq2>::alloc(L, pb, &rb2));
    XYZZ<Fq>* res1 = L->alloc_n<XYZZ<Fq>>(4);
    XYZZ<Fq2>* res2 = L->alloc_n<XYZZ<Fq2>>(1);
    Affine<Fq>* oa = L->alloc_n<Affine<Fq>>(2);
    Affine<Fq2>* ob = L->alloc_n<Affine<Fq2>>(1);
    Fr* abc = L->alloc_n<Fr>(3 * m);
    if (!res1 || !res2 || !oa || !ob || !abc) return HK_ERR_NOMEM;
    // Fork: the five queries are independent once their scalars exist.  Side streams let the
    // latency-bound tails (segmented levels, bucket reduction) of one query hide under the
    // throughput-bound accumulation of another.
    //   main  : sort(z) -> A
    //   aux0  : B1      aux1 : B2 (G2)      aux2 : L      aux3 : witness map -> sort(h) -> H
    // HK_SERIAL_STREAMS=1 keeps everything on the lane's own stream: clean per-kernel times for profiling, and - with
    // 18 lanes and GPU_MAX_HW_QUEUES=18, one hardware queue per lane - the faster form for small circuits (DESIGN.md
    // section 5).  Not the default: 18 concurrent G2 tail kernels (2-3 KB of scratch per lane each) once exhausted the
    // runtime's scratch pool on BLS12-381 and the process aborted (HSA_STATUS_ERROR_OUT_OF_RESOURCES).
    hipStream_t axs[4] = {L->aux[0], L->aux[1], L->aux[2], L->aux[3]};
    static const bool serial = getenv("HK_SERIAL_STREAMS") != nullptr;
    if (serial) for (auto& a : axs) a = s;
    hipStream_t* ax = axs;
    // Every exit between the fork and the join - an HK_TRY / HK_HIP return included - must leave no side stream
    // running on this lane's arena: the next call on the lane resets the arena and would reuse live memory.
    struct JoinGuard {
        hipStream_t main; hipStream_t* aux; bool joined = false;
        ~JoinGuard() {
            if (joined) return;
            for (int i = 0; i < 4; i++) (void)hipStreamSynchronize(aux[i]);
            (void)hipStreamSynchronize(main);
        }
    } join_guard{s, axs};
    hipEvent_t ev_z = ev[16], ev_sorted = ev[17];
    HK_HIP(hipEventRecord(ev_z, s));                                           // z (and ext scalars) on device
    HK_HIP(hipStreamWaitEvent(ax[3], ev_z, 0));
    if (prof) HK_HIP(hipEventRecord(ev[5], ax[3]));
    HK_TRY(Q::run(ax[3], T, pk->csr[0], pk->csr[1], pk->csr[2], pk->n_inst, pk->n_c, zd, abc, pk->log_m));
    if (prof) HK_HIP(hipEventRecord(ev[6], ax[3]));                            // witness map done
    HK_TRY(MsmSort<Fr>::run(ax[3], ph, (const u32*)abc, 1, sbh, true));
    hipEvent_t kh0 = prof ? ev[12] : nullptr, kh1 = prof ? ev[13] : nullptr;
    HK_TRY(MsmRun<Fq>::run(ax[3], ph, pk->h_tab, (u32)m, 0, sbh, rbh, res1 + 3, kh0, kh1));
    HK_HIP(hipEventRecord(ev[7], ax[3]));                                      // H done
    HK_TRY(MsmSort<Fr>::run(s, pz, (const u32*)zext, 1, sb, true));
    HK_HIP(hipEventRecord(ev_sorted, s));
    HK_TRY(mark());                                                            // ev1: digits done
    HK_HIP(hipStreamWaitEvent(ax[2], ev_sorted, 0));
    const SortBufs* sbB = &sb;
    if (pk->b_compact) {
        // B1 / B2 over the non-infinity bases only: gather their scalars, sort those digits on aux0
        HK_HIP(hipStreamWaitEvent(ax[0], ev_z, 0));
        hipLaunchKernelGGL((k_gather<Fr>), dim3((pk->b_n + 255) / 256), dim3(256), 0, ax[0], zb, (const Fr*)zext,
                           (const u32*)pk->b_idx, pk->b_n, pk->n_ext);
        HK_TRY(MsmSort<Fr>::run(ax[0], pb, (const u32*)zb, 1, sbb, true));
        HK_HIP(hipEventRecord(ev[28], ax[0]));
        HK_HIP(hipStreamWaitEvent(ax[1], ev[28], 0));
        sbB = &sbb;
    } else {
        HK_HIP(hipStreamWaitEvent(ax[0], ev_sorted, 0));
        HK_HIP(hipStreamWaitEvent(ax[1], ev_sorted, 0));
    }
    HK_TRY(MsmRun<Fq2>::run(ax[1], pb, pk->b2_tab, pk->b_n, 0, *sbB, rb2, res2, nullptr, nullptr));
    HK_HIP(hipEventRecord(ev[4], ax[1]));                                      // B2 done
    HK_TRY(MsmRun<Fq>::run(ax[0], pb, pk->b1_tab, pk->b_n, 0, *sbB, rbB1, res1 + 1, prof ? ev[22] : nullptr,
                           prof ? ev[23] : nullptr));
    HK_HIP(hipEventRecord(ev[3], ax[0]));                                      // B1 done
    HK_TRY(MsmRun<Fq>::run(ax[2], pz, pk->l_tab, pk->l_n, pk->l_off, sb, rbL, res1 + 2, prof ? ev[24] : nullptr,
                           prof ? ev[25] : nullptr));
    HK_HIP(hipEventRecord(ev[18], ax[2]));                                     // L done
    HK_TRY(MsmRun<Fq>::run(s, pz, pk->a_tab, pk->n_ext, 0, sb, rbA, res1 + 0, prof ? ev[26] : nullptr,
                           prof ? ev[27] : nullptr));
    HK_TRY(mark());                                                            // ev2: A done
    // Join
    HK_HIP(hipStreamWaitEvent(s, ev[3], 0));
    HK_HIP(hipStreamWaitEvent(s, ev[4], 0));
    HK_HIP(hipStreamWaitEvent(s, ev[18], 0));
    HK_HIP(hipStreamWaitEvent(s, ev[7], 0));
    join_guard.joined = true;                                                  // main now depends on every side stream
    if (prof) HK_HIP(hipEventRecord(ev[19], s));                               // all queries done
    static const EndoSplit<2> endo_g1 = EndoOf<Fq>::split();
    hipLaunchKernelGGL((k_finish<Fr, Fq, Fq2>), dim3(3), dim3(64), 0, s, res1, res2, pk->consts_g1,
                       pk->consts_g2, small, oa, ob, oa + 1, endo_g1);
    HK_HIP(hipGetLastError());
    HK_HIP(hipMemcpyAsync(out_a, oa, sizeof(Affine<Fq>), hipMemcpyDeviceToHost, s));
    HK_HIP(hipMemcpyAsync(out_b, ob, sizeof(Affine<Fq2>), hipMemcpyDeviceToHost, s));
    HK_HIP(hipMemcpyAsync(out_c, oa + 1, sizeof(Affine<Fq>), hipMemcpyDeviceToHost, s));
    if (prof) HK_HIP(hipEventRecord(ev[8], s));
    HK_HIP(hipStreamSynchronize(s));
    if (prof) {
        hk_timings& t = L->timings;
        memset(&t, 0, sizeof(t));
        // the five queries run concurrently on side streams: each figure is the elapsed time on the
        // query's own stream since its fork point (they overlap, they do not add up to total_ms)
        t.total_ms = ev_ms(ev[0], ev[8]);
        t.digits_ms = ev_ms(ev[0], ev[1]);
        t.msm_a_ms = ev_ms(ev[1], ev[2]);
        t.msm_b_g1_ms = ev_ms(ev[1], ev[3]);
        t.msm_b_g2_ms = ev_ms(ev[1], ev[4]);
        t.msm_l_ms = ev_ms(ev[1], ev[18]);
        t.witness_map_ms = ev_ms(ev[5], ev[6]);
        t.msm_h_ms = ev_ms(ev[6], ev[7]);
        t.finish_ms = ev_ms(ev[19], ev[8]);
        // k_msm_accum0<Fq> launches of this proof: H (dense) + A, B1, L (sparse)
        t.accum_h_ms = ev_ms(ev[12], ev[13]);
        t.accum_kernel_ms = t.accum_h_ms + ev_ms(ev[22], ev[23]) + ev_ms(ev[24], ev[25]) + ev_ms(ev[26], ev[27]);
        t.accum_kernel_launches = 4;
    }
    return HK_OK;
}

}  // namespace hk
