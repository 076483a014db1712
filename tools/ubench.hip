// ubench.hip — gfx950 instruction / field-op throughput probes that size the MSM kernels
// (SURVEY.md §7 "microbenchmark v_mad_u64_u32 first").  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../hekaton_system_amd/csrc/ec.cuh"
using namespace hk;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorName(e), __LINE__); return 1; } } while (0)

__global__ void k_mad64(u64* out, int iters, u32 a, u32 b) {
    u64 x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    u32 m = a + threadIdx.x, n = b;
    for (int i = 0; i < iters; i++) {
        x0 = (u64)m * n + x0; x1 = (u64)m * n + x1; x2 = (u64)m * n + x2; x3 = (u64)m * n + x3;
        x4 = (u64)m * n + x4; x5 = (u64)m * n + x5; x6 = (u64)m * n + x6; x7 = (u64)m * n + x7;
        m = (u32)x0; n = (u32)(x4 >> 32);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_mullo(u32* out, int iters, u32 a) {
    u32 x0 = threadIdx.x + 1, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
        x0 *= a | x7; x1 *= a | x0; x2 *= a | x1; x3 *= a | x2; x4 *= a | x3; x5 *= a | x4; x6 *= a | x5; x7 *= a | x6;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_add32(u32* out, int iters, u32 a) {
    u32 x0 = threadIdx.x + 1, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
        x0 += a ^ x7; x1 += a ^ x0; x2 += a ^ x1; x3 += a ^ x2; x4 += a ^ x3; x5 += a ^ x4; x6 += a ^ x5; x7 += a ^ x6;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}
__global__ void k_fma64(double* out, int iters, double a) {
    double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
        x0 = fma(x0, a, x1); x1 = fma(x1, a, x2); x2 = fma(x2, a, x3); x3 = fma(x3, a, x4);
        x4 = fma(x4, a, x5); x5 = fma(x5, a, x6); x6 = fma(x6, a, x7); x7 = fma(x7, a, x0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <class F>
__global__ void k_fmul(const F* in, F* out, int iters) {
    size_t t = blockIdx.x * blockDim.x + threadIdx.x;
    F x = in[t], y = in[t + 1];
    for (int i = 0; i < iters; i++) { x = F::mul(x, y); y = F::mul(y, x); }
    out[t] = F::add(x, y);
}
template <class F>
__global__ void k_madd(const Affine<F>* in, XYZZ<F>* out, int iters, int npts) {
    size_t t = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ<F> acc = XYZZ<F>::from_affine(in[t % npts]);
    for (int i = 1; i <= iters; i++) acc = ec_madd(acc, in[(t + i * 7919u) % npts]);
    out[t] = acc;
}

// the real access pattern of k_msm_accum0: entry id from a list, then a dependent 64-B gather
template <class F, int WAVES>
__global__ void __launch_bounds__(64, WAVES) k_madd_idx(const Affine<F>* tab, const u32* idx, XYZZ<F>* out, int iters, int prefetch) {
    size_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const u32* my = idx + t * iters;
    XYZZ<F> acc = XYZZ<F>::inf();
    if (!prefetch) {
        for (int i = 0; i < iters; i++) acc = ec_madd(acc, tab[my[i]]);
    } else {
        Affine<F> nxt = tab[my[0]];
        for (int i = 0; i < iters; i++) {
            Affine<F> cur = nxt;
            if (i + 1 < iters) nxt = tab[my[i + 1]];
            acc = ec_madd(acc, cur);
        }
    }
    out[t] = acc;
}

template <class F, int WAVES>
__global__ void __launch_bounds__(64, WAVES) k_madd_occ(const Affine<F>* in, XYZZ<F>* out, int iters, int npts) {
    size_t t = blockIdx.x * blockDim.x + threadIdx.x;
    XYZZ<F> acc = XYZZ<F>::from_affine(in[t % npts]);
    for (int i = 1; i <= iters; i++) acc = ec_madd(acc, in[(t + i * 7919u) % npts]);
    out[t] = acc;
}

template <class K, class... A>
static float timeit(K k, dim3 g, dim3 b, A... a) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, g, b, 0, 0, a...);      // warm
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, g, b, 0, 0, a...);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <class F>
__global__ void k_mul_pairs(const F* x, const F* y, F* z, int n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) z[t] = F::canon(F::mul(x[t], y[t]));   // registers may be lazy, memory is canonical
}
template <class F>
static int check_mul(const char* name) {
    const int n = 1 << 14;
    std::vector<F> x(n), y(n), z(n);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (u32)(st >> 16); };
    auto gen = [&](F& f) {
        for (;;) {
            for (int i = 0; i < F::N; i++) f.v[i] = rnd();
            f.v[F::N - 1] &= 0x3fffffffu >> (F::N == 12 ? 1 : 0);
            bool lt = false;
            for (int i = F::N - 1; i >= 0; i--) { if (f.v[i] < F::Params::MOD[i]) { lt = true; break; } if (f.v[i] > F::Params::MOD[i]) break; }
            if (lt) return;
        }
    };
    for (int i = 0; i < n; i++) { gen(x[i]); gen(y[i]); }
    // edge values
    for (int i = 0; i < F::N; i++) { x[0].v[i] = 0; x[1].v[i] = F::Params::MOD[i]; y[1].v[i] = F::Params::MOD[i]; }
    x[1].v[0] -= 1; y[1].v[0] -= 1;
    F *dx, *dy, *dz;
    hipMalloc(&dx, n * sizeof(F)); hipMalloc(&dy, n * sizeof(F)); hipMalloc(&dz, n * sizeof(F));
    hipMemcpy(dx, x.data(), n * sizeof(F), hipMemcpyHostToDevice);
    hipMemcpy(dy, y.data(), n * sizeof(F), hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k_mul_pairs<F>), dim3(n / 64), dim3(64), 0, 0, dx, dy, dz, n);
    hipMemcpy(z.data(), dz, n * sizeof(F), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) { F w = F::mul(x[i], y[i]); if (!(w == z[i])) bad++; }
    printf("check %s device mul vs host C++ mul: %d mismatches of %d\n", name, bad, n);
    hipFree(dx); hipFree(dy); hipFree(dz);
    return bad;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
#if defined(HK_NO_ASM_MUL)
    printf("build: C++ CIOS multiplication\n");
#else
    printf("build: inline-asm product-scanning multiplication\n");
#endif
    if (check_mul<Fp<Bn254FqP>>("bn254 Fq") | check_mul<Fp<Bn254FrP>>("bn254 Fr") | check_mul<Fp<Bls381FrP>>("bls Fr") |
        check_mul<Fp<Bls381FqP>>("bls Fq")) {
#if !defined(HK_EXPERIMENT_NOREDUCE)
        return 2;
#endif
    }
    printf("device %s CUs=%d clock=%d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    const int CUs = prop.multiProcessorCount;
    void* buf; CK(hipMalloc(&buf, 512 << 20)); CK(hipMemset(buf, 1, 512 << 20));
    for (int wpc : {16}) {           // waves per CU
        dim3 g(CUs * wpc / 4), b(256);
        int iters = 20000;
        double lanes = (double)g.x * b.x;
        float t;
        t = timeit(k_mad64, g, b, (u64*)buf, iters, 12345u, 6789u);
        printf("waves/CU=%2d  v_mad_u64_u32: %7.2f Gop/s/CU  (%.2f Top/s chip)\n", wpc, lanes * iters * 8 / t / 1e6 / CUs, lanes * iters * 8 / t / 1e9);
        t = timeit(k_mullo, g, b, (u32*)buf, iters, 12345u);
        printf("waves/CU=%2d  v_mul_lo_u32 : %7.2f Gop/s/CU  (%.2f Top/s chip)\n", wpc, lanes * iters * 8 / t / 1e6 / CUs, lanes * iters * 8 / t / 1e9);
        t = timeit(k_add32, g, b, (u32*)buf, iters, 12345u);
        printf("waves/CU=%2d  v_add+xor    : %7.2f Gop/s/CU  (%.2f Top/s chip)\n", wpc, lanes * iters * 16 / t / 1e6 / CUs, lanes * iters * 16 / t / 1e9);
        t = timeit(k_fma64, g, b, (double*)buf, iters, 1.0000001);
        printf("waves/CU=%2d  v_fma_f64    : %7.2f Gop/s/CU  (%.2f Top/s chip)\n", wpc, lanes * iters * 8 / t / 1e6 / CUs, lanes * iters * 8 / t / 1e9);
    }
    typedef Fp<Bn254FqP> Fq;
    typedef Fp<Bls381FqP> Fq381;
    for (int wpc : {4, 8, 16}) {
        dim3 g(CUs * wpc), b(64);
        int iters = 2000;
        double lanes = (double)g.x * b.x;
        float t = timeit(k_fmul<Fq>, g, b, (const Fq*)buf, (Fq*)buf + (64 << 20) / 32, iters);
        printf("waves/CU=%2d  Fq254 mont mul : %8.2f Gmul/s chip   (%.1f ns per mul per lane)\n", wpc, lanes * iters * 2 / t / 1e6, t * 1e6 / (iters * 2));
        t = timeit(k_fmul<Fq381>, g, b, (const Fq381*)buf, (Fq381*)buf + (64 << 20) / 48, iters);
        printf("waves/CU=%2d  Fq381 mont mul : %8.2f Gmul/s chip   (%.1f ns per mul per lane)\n", wpc, lanes * iters * 2 / t / 1e6, t * 1e6 / (iters * 2));
    }
    // one lane alone: latency of a dependent mul chain
    {
        float t = timeit(k_fmul<Fq>, dim3(1), dim3(64), (const Fq*)buf, (Fq*)buf + (64 << 20) / 32, 20000);
        printf("single wave: Fq254 mul latency %.1f ns\n", t * 1e6 / 40000);
    }
    // fill bases with valid-looking data is not needed for timing (formulas are data independent)
    for (int wpc : {4, 8, 12, 16}) {
        dim3 g(CUs * wpc), b(64);
        int iters = 200;
        double lanes = (double)g.x * b.x;
        float t = timeit(k_madd<Fq>, g, b, (const Affine<Fq>*)buf, (XYZZ<Fq>*)((char*)buf + (256 << 20)), iters, 1 << 20);
        printf("waves/CU=%2d  G1-254 madd (random 64B gathers): %8.2f Gadd/s chip\n", wpc, lanes * iters / t / 1e6);
    }
    {
        // table-size sweep with the accumulate kernel's access pattern (index list -> gather)
        typedef Fp<Bn254FqP> Fq;
        size_t maxpts = (size_t)1 << 25;                 // 2 GiB of affine G1 points
        Affine<Fq>* tab; u32* idx; XYZZ<Fq>* outp;
        int iters = 128;
        size_t lanes = (size_t)CUs * 12 * 64;
        CK(hipMalloc(&tab, maxpts * sizeof(Affine<Fq>))); CK(hipMemset(tab, 1, maxpts * sizeof(Affine<Fq>)));
        CK(hipMalloc(&idx, lanes * iters * 4)); CK(hipMalloc(&outp, lanes * sizeof(XYZZ<Fq>)));
        std::vector<u32> h(lanes * iters);
        for (int lg : {20, 23, 25}) {
            unsigned long long st = 88172645463325252ull;
            for (auto& v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (u32)(st >> 20) & ((1u << lg) - 1); }
            CK(hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice));
            for (int pf : {0, 1}) {
                float t2 = timeit(k_madd_idx<Fq, 2>, dim3(lanes / 64), dim3(64), tab, idx, outp, iters, pf);
                float t3 = timeit(k_madd_idx<Fq, 3>, dim3(lanes / 64), dim3(64), tab, idx, outp, iters, pf);
                printf("G1 madd via index list, table 2^%d pts (%4zu MiB), prefetch=%d: occ2 %6.2f Gadd/s  occ3 %6.2f Gadd/s\n",
                       lg, ((size_t)64 << lg) >> 20, pf, lanes * (double)iters / t2 / 1e6, lanes * (double)iters / t3 / 1e6);
            }
        }
        hipFree(tab); hipFree(idx); hipFree(outp);
    }
    {
        typedef Fp2<Bn254FqP> F2;
        int iters = 100;
        for (int wpc : {4, 8, 12}) {
            dim3 g(CUs * wpc), b(64);
            double lanes = (double)g.x * b.x;
            float t1 = timeit(k_madd_occ<F2, 1>, g, b, (const Affine<F2>*)buf, (XYZZ<F2>*)((char*)buf + (256 << 20)), iters, 1 << 19);
            float t2 = timeit(k_madd_occ<F2, 2>, g, b, (const Affine<F2>*)buf, (XYZZ<F2>*)((char*)buf + (256 << 20)), iters, 1 << 19);
            printf("waves/CU=%2d  G2-254 madd  launch_bounds occ1: %6.2f Gadd/s   occ2: %6.2f Gadd/s\n", wpc, lanes * iters / t1 / 1e6, lanes * iters / t2 / 1e6);
        }
    }
    for (int wpc : {4, 8}) {
        dim3 g(CUs * wpc), b(64);
        int iters = 100;
        double lanes = (double)g.x * b.x;
        float t = timeit(k_madd<Fp2<Bn254FqP>>, g, b, (const Affine<Fp2<Bn254FqP>>*)buf, (XYZZ<Fp2<Bn254FqP>>*)((char*)buf + (256 << 20)), iters, 1 << 19);
        printf("waves/CU=%2d  G2-254 madd: %8.2f Gadd/s chip\n", wpc, lanes * iters / t / 1e6);
    }
    return 0;
}
