// ntt_bench.hip — stand-alone timing of the NTT pass kernels on the witness-map shape (3 vectors of 2^21 Fr),
// used to pick the pass schedule / kernel variant.  Build:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ntt_bench.hip -o tools/ntt_bench
// Prints per-variant time of one DIF chain + one DIT chain and checks every candidate bit-for-bit against the
// previous stage-by-stage kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../hekaton_system_amd/csrc/ntt.cuh"
using namespace hk;
typedef Fp<Bn254FrP> Fr;

// ---- the stage-by-stage (radix-2) pass kernel the product shipped before k_ntt_pass4; kept here as the A/B baseline
constexpr int OLD_THREADS = 256;
namespace hk {
// One pass of `nst` butterfly stages [lo, lo+nst) on a transform of size 2^logn, in place.
//   DIF (dit == 0): stages run from high to low, butterfly (u, v) -> (u + v, (u - v) * w)
//   DIT (dit == 1): stages run from low to high, butterfly (u, v) -> (u + v*w, u - v*w)
// tw: table of w_M^i, i < M/2, M = 2^log_table.  Batched over blockIdx.y (vectors `stride_vec` apart).
// Optional fused epilogue (post != 0): every element is multiplied by `scale` and, when post == 2, also by
// g^bitrev(index) from the 3x1024 power tables `pw` before it is stored — the "/m and coset shift"
// step that follows a DIF chain in the witness map, so it costs no extra HBM round trip.
template <class Fr>
__global__ void __launch_bounds__(OLD_THREADS)
k_ntt_pass_r2(Fr* __restrict__ data, size_t stride_vec, const Fr* __restrict__ tw, u32 logn, u32 log_table,
           u32 lo, u32 nst, int dit, int post, Fr scale, const Fr* __restrict__ pw) {
    extern __shared__ uint4 lds_raw[];
    Fr* lds = reinterpret_cast<Fr*>(lds_raw);
    Fr* vec = data + (size_t)blockIdx.y * stride_vec;
    u32 cols_bits = lo < (u32)3 ? lo : (u32)3;
    u32 rows = 1u << nst, cols = 1u << cols_bits;
    u32 tile_elems = rows << cols_bits;
    u32 mid_bits = lo - cols_bits;
    u32 t = blockIdx.x;
    u32 mid = t & ((1u << mid_bits) - 1u);
    u32 high = t >> mid_bits;
    size_t base = ((size_t)high << (lo + nst)) | ((size_t)mid << cols_bits);
    // load tile: element (r, c) lives at base | r << lo | c ; LDS index r * cols + c
    for (u32 e = threadIdx.x; e < tile_elems; e += OLD_THREADS) {
        u32 r = e >> cols_bits, c = e & (cols - 1);
        lds[e] = fr_load(&vec[base | ((size_t)r << lo) | c]);
    }
    __syncthreads();
    u32 half_count = tile_elems >> 1;
    for (u32 st = 0; st < nst; st++) {
        u32 ls = dit ? st : (nst - 1 - st);          // local stage (bit of r)
        u32 s = lo + ls;                              // global stage
        for (u32 bidx = threadIdx.x; bidx < half_count; bidx += OLD_THREADS) {
            // butterfly index -> (r without bit ls, c)
            u32 c = bidx & (cols - 1);
            u32 rr = bidx >> cols_bits;
            u32 r0 = ((rr >> ls) << (ls + 1)) | (rr & ((1u << ls) - 1u));
            u32 r1 = r0 | (1u << ls);
            u32 i0 = (r0 << cols_bits) | c, i1 = (r1 << cols_bits) | c;
            size_t g0 = base | ((size_t)r0 << lo) | c;           // global index of the upper element
            u32 j = (u32)(g0 & (((size_t)1 << s) - 1));
            Fr w = fr_load(&tw[(size_t)j << (log_table - s - 1)]);   // issued first: longest latency
            Fr u = lds[i0], v = lds[i1];
            if (dit) {
                v = Fr::mul(v, w);
                lds[i0] = Fr::add(u, v);
                lds[i1] = Fr::sub(u, v);
            } else {
                lds[i0] = Fr::add(u, v);
                lds[i1] = Fr::mul(Fr::sub(u, v), w);
            }
        }
        __syncthreads();
    }
    for (u32 e = threadIdx.x; e < tile_elems; e += OLD_THREADS) {
        u32 r = e >> cols_bits, c = e & (cols - 1);
        size_t gi = base | ((size_t)r << lo) | c;
        Fr x = lds[e];
        if (post) {
            x = Fr::mul(x, scale);
            if (post == 2) {
                u32 j = logn ? (__brev((u32)gi) >> (32 - logn)) : 0u;
                x = Fr::mul(x, pow_from_tables(pw, j, logn));
            }
        }
        fr_store(&vec[gi], x);
    }
}

}  // namespace hk

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorName(e), __LINE__); exit(1); } } while (0)

__global__ void k_fill(Fr* d, size_t n, u32 seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr x;
    u32 s = seed + (u32)i * 2654435761u;
    for (int k = 0; k < 8; k++) { s = s * 1664525u + 1013904223u; x.v[k] = s; }
    x.v[7] &= 0x0fffffffu;
    fr_store(&d[i], x);
}

struct Sched { u32 lo, nst, cols_bits; };

static void run_old(Fr* data, size_t stride, u32 batch, u32 logn, const Fr* tw, u32 log_table, int dit) {
    u32 bottom = logn < 11 ? logn : 11, rest = logn - bottom;
    u32 npass = (rest + 7) / 8;
    struct P { u32 lo, nst; } ps[8];
    int np = 0;
    ps[np++] = {0, bottom};
    u32 lo = bottom;
    for (u32 i = 0; i < npass; i++) { u32 nst = (rest - (lo - bottom) + (npass - i) - 1) / (npass - i); ps[np++] = {lo, nst}; lo += nst; }
    Fr one = Fr::one();
    for (int k = 0; k < np; k++) {
        const P& p = dit ? ps[k] : ps[np - 1 - k];
        u32 cb = p.lo < 3 ? p.lo : 3;
        u32 tl = p.nst + cb;
        hipLaunchKernelGGL((k_ntt_pass_r2<Fr>), dim3(1u << (logn - tl), batch), dim3(OLD_THREADS), sizeof(Fr) << tl, 0, data, stride, tw,
                           logn, log_table, p.lo, p.nst, dit, 0, one, (const Fr*)nullptr);
    }
}

static void run_new(Fr* data, size_t stride, u32 batch, u32 logn, const Fr* tws, int dit, const std::vector<Sched>& sc, int threads) {
    Fr one = Fr::one();
    for (size_t k = 0; k < sc.size(); k++) {
        const Sched& p = dit ? sc[k] : sc[sc.size() - 1 - k];
        u32 tl = p.nst + p.cols_bits;
        if (dit)
            hipLaunchKernelGGL((k_ntt_pass4<Fr, 1>), dim3(1u << (logn - tl), batch), dim3(threads), sizeof(Fr) << tl, 0, data, stride, tws, logn,
                               p.lo, p.nst, p.cols_bits, 0, 0u, one, (const Fr*)nullptr, (const Fr*)nullptr, one);
        else
            hipLaunchKernelGGL((k_ntt_pass4<Fr, 0>), dim3(1u << (logn - tl), batch), dim3(threads), sizeof(Fr) << tl, 0, data, stride, tws, logn,
                               p.lo, p.nst, p.cols_bits, 0, 0u, one, (const Fr*)nullptr, (const Fr*)nullptr, one);
    }
}

int main(int argc, char** argv) {
    u32 logn = argc > 1 ? atoi(argv[1]) : 21;
    u32 batch = 3;
    size_t n = (size_t)1 << logn;
    u32 log_table = logn;
    Fr *a, *b, *tw, *tws, *sq;
    CK(hipMalloc(&a, sizeof(Fr) * n * batch));
    CK(hipMalloc(&b, sizeof(Fr) * n * batch));
    CK(hipMalloc(&tw, sizeof(Fr) * (n / 2)));
    CK(hipMalloc(&tws, sizeof(Fr) * n));
    CK(hipMalloc(&sq, sizeof(Fr) * 32));
    // a genuine 2^logn-th root of unity so that old and new kernels can be compared on real twiddles
    std::vector<Fr> hsq(32);
    {
        Fr w; const u32 root[8] = HK_BN254_FR_ROOT;
        for (int i = 0; i < 8; i++) w.v[i] = root[i];
        for (u32 k = 0; k < 28 - logn; k++) w = Fr::sqr(w);
        for (u32 k = 0; k < logn; k++) { hsq[k] = w; w = Fr::sqr(w); }
    }
    CK(hipMemcpy(sq, hsq.data(), sizeof(Fr) * 32, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((k_pow_table<Fr>), dim3((u32)((n / 2 + 255) / 256)), dim3(256), 0, 0, tw, sq, (u32)(n / 2), logn - 1);
    hipLaunchKernelGGL((k_stage_tables<Fr>), dim3((u32)((n + 255) / 256)), dim3(256), 0, 0, tws, tw, log_table);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto fill = [&](Fr* d) { hipLaunchKernelGGL(k_fill, dim3((u32)((n * batch + 255) / 256)), dim3(256), 0, 0, d, n * batch, 12345u); };
    auto timeit = [&](const char* name, auto fn) {
        fn(); CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0, 0)); fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("%-44s %8.3f ms  (%.1f G butterflies/s)\n", name, best, 2.0 * batch * logn * (double)(n / 2) / best / 1e6);
        return best;
    };
    // reference result: old DIF then old DIT
    fill(a);
    timeit("radix-2 stage-by-stage kernel: DIF + DIT", [&] { run_old(a, n, batch, logn, tw, log_table, 0); run_old(a, n, batch, logn, tw, log_table, 1); });
    fill(a);
    run_old(a, n, batch, logn, tw, log_table, 0); run_old(a, n, batch, logn, tw, log_table, 1);
    CK(hipDeviceSynchronize());
    std::vector<u32> ref(n * batch * 8), got(n * batch * 8);
    CK(hipMemcpy(ref.data(), a, sizeof(Fr) * n * batch, hipMemcpyDeviceToHost));

    struct Cand { const char* name; std::vector<Sched> sc; int threads; };
    std::vector<Cand> cands;
    if (logn == 21) {
        cands.push_back({"r4 11+5+5 T=11 512thr", {{0, 11, 0}, {11, 5, 6}, {16, 5, 6}}, 512});
        cands.push_back({"r4 11+10 T=11 (cols 2) 512thr", {{0, 11, 0}, {11, 10, 1}}, 512});
        cands.push_back({"r4 9+6+6 T=10 256thr", {{0, 9, 0}, {9, 6, 4}, {15, 6, 4}}, 256});
        cands.push_back({"r4 10+6+5 T=10 256thr", {{0, 10, 0}, {10, 6, 4}, {16, 5, 5}}, 256});
        cands.push_back({"r4 10+6+5 T=10 128thr", {{0, 10, 0}, {10, 6, 4}, {16, 5, 5}}, 128});
        cands.push_back({"r4 11+6+4 T=11/10 256thr", {{0, 11, 0}, {11, 6, 4}, {17, 4, 6}}, 256});
        cands.push_back({"r4 11+10 T=11 (cols 2) 256thr", {{0, 11, 0}, {11, 10, 1}}, 256});
        cands.push_back({"r4 12+9 T=12 (cols 8) 512thr", {{0, 12, 0}, {12, 9, 3}}, 512});
    } else {
        u32 b0 = logn < 10 ? logn : 10;
        std::vector<Sched> sc = {{0, b0, 0}};
        u32 lo = b0;
        while (lo < logn) { u32 nst = logn - lo < 7 ? logn - lo : 7; sc.push_back({lo, nst, 10 - nst > lo ? lo : 10 - nst}); lo += nst; }
        cands.push_back({"r4 generic T=10 256thr", sc, 256});
    }
    for (auto& c : cands) {
        fill(a);
        timeit(c.name, [&] { run_new(a, n, batch, logn, tws, 0, c.sc, c.threads); run_new(a, n, batch, logn, tws, 1, c.sc, c.threads); });
        fill(a);
        run_new(a, n, batch, logn, tws, 0, c.sc, c.threads); run_new(a, n, batch, logn, tws, 1, c.sc, c.threads);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), a, sizeof(Fr) * n * batch, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < got.size(); i++) bad += got[i] != ref[i];
        printf("    parity vs radix-2 kernel: %s (%zu limbs differ)\n", bad ? "MISMATCH" : "bit-exact", bad);
    }
    return 0;
}
