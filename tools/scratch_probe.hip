// scratch_probe.hip — what the runtime reserves for private ("scratch") memory, measured, for DESIGN.md section 3c.
//
//   hipcc -O2 --offload-arch=gfx950 tools/scratch_probe.hip -o tools/scratch_probe -lhsa-runtime64
//   GPU_MAX_HW_QUEUES=20 tools/scratch_probe
//
// Prints (1) the agent's own limits, HSA_AMD_AGENT_INFO_SCRATCH_LIMIT_MAX ("shared across all queues created on this
// agent") and _CURRENT (a dispatch above it is served by a use-once allocation, one below it keeps its ring assigned to
// the queue), hsa_ext_amd.h; (2) for kernels that need B bytes of private memory per lane, B = 0 ... 3 KB as the MSM
// tails of libhekaton do (lib/kernel_meta.txt): the device memory that disappears (hipMemGetInfo) when the kernel runs
// on 1, 2, 4, 8 streams, whether it comes back after the streams are idle, and the launch-to-completion time of a
// 64-wave and of a chip-filling dispatch.  At most 8 streams carry scratch at once - what the default bench does anyway.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorName(e_), __LINE__); return 1; } } while (0)

template <int WORDS>
__global__ void __launch_bounds__(64) k_scratch(unsigned* out, unsigned seed, int rounds) {
    unsigned buf[WORDS > 0 ? WORDS : 1];
    unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned x = seed + t;
    if (WORDS > 0) {
        for (int i = 0; i < WORDS; i++) buf[i] = x + i;
        for (int r = 0; r < rounds; r++) {
            x = x * 1664525u + 1013904223u;
            unsigned j = x % (unsigned)(WORDS > 0 ? WORDS : 1);       // runtime index: the array stays in private memory
            buf[j] += x;
            x ^= buf[(j * 7u + 3u) % (unsigned)(WORDS > 0 ? WORDS : 1)];
        }
    } else {
        for (int r = 0; r < rounds; r++) x = x * 1664525u + 1013904223u;
    }
    out[t] = x;
}

static hsa_status_t agent_cb(hsa_agent_t a, void* data) {
    hsa_device_type_t type;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS || type != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    char name[64] = {0};
    hsa_agent_get_info(a, HSA_AGENT_INFO_NAME, name);
    uint64_t mx = 0, cur = 0;
    uint32_t cus = 0, waves_cu = 0, qmax = 0;
    hsa_status_t s1 = hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_SCRATCH_LIMIT_MAX, &mx);
    hsa_status_t s2 = hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_SCRATCH_LIMIT_CURRENT, &cur);
    hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_COMPUTE_UNIT_COUNT, &cus);
    hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_MAX_WAVES_PER_CU, &waves_cu);
    hsa_agent_get_info(a, HSA_AGENT_INFO_QUEUES_MAX, &qmax);
    printf("agent %s: SCRATCH_LIMIT_MAX %llu B (%.2f GiB, status %d), SCRATCH_LIMIT_CURRENT %llu B (%.1f MiB, status %d), %u CUs x %u waves, queues_max %u\n",
           name, (unsigned long long)mx, mx / 1073741824.0, (int)s1, (unsigned long long)cur, cur / 1048576.0, (int)s2, cus, waves_cu, qmax);
    (*(int*)data)++;
    return HSA_STATUS_SUCCESS;
}

template <int WORDS>
static int probe(const char* label, int max_streams) {
    hipFuncAttributes fa;
    CK(hipFuncGetAttributes(&fa, (const void*)k_scratch<WORDS>));
    size_t free0 = 0, total = 0;
    CK(hipDeviceSynchronize());
    CK(hipMemGetInfo(&free0, &total));
    std::vector<hipStream_t> st(max_streams);
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned* out = nullptr;
    const unsigned big_blocks = 256 * 32;                       // one wave per slot of the chip
    CK(hipMalloc(&out, (size_t)big_blocks * 64 * 4 * max_streams));
    size_t free_base = 0;
    CK(hipMemGetInfo(&free_base, &total));
    printf("%-10s private %5zu B/lane (compiler), ", label, (size_t)fa.localSizeBytes);
    for (int k = 1; k <= max_streams; k *= 2) {
        for (int i = 0; i < k; i++) hipLaunchKernelGGL(k_scratch<WORDS>, dim3(64), dim3(64), 0, st[i], out + (size_t)i * big_blocks * 64, 1u + i, 64);
        CK(hipDeviceSynchronize());
        size_t f = 0;
        CK(hipMemGetInfo(&f, &total));
        printf("%d streams: -%.1f MiB  ", k, (double)(free_base - f) / 1048576.0);
    }
    // chip-filling dispatch on every stream at once
    for (int i = 0; i < max_streams; i++) hipLaunchKernelGGL(k_scratch<WORDS>, dim3(big_blocks), dim3(64), 0, st[i], out + (size_t)i * big_blocks * 64, 7u + i, 64);
    CK(hipDeviceSynchronize());
    size_t f2 = 0;
    CK(hipMemGetInfo(&f2, &total));
    printf("| %d chip-filling dispatches: -%.1f MiB", max_streams, (double)(free_base - f2) / 1048576.0);
    // launch-to-completion time of small dispatches, one stream
    auto t0 = std::chrono::steady_clock::now();
    const int reps = 200;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_scratch<WORDS>, dim3(64), dim3(64), 0, st[0], out, 3u + r, 64);
    CK(hipStreamSynchronize(st[0]));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    printf(" | %.1f us per 64-wave dispatch back to back\n", us);
    for (auto& s : st) CK(hipStreamDestroy(s));
    CK(hipFree(out));
    CK(hipDeviceSynchronize());
    size_t f3 = 0;
    CK(hipMemGetInfo(&f3, &total));
    printf("           after destroying the streams: %.1f MiB still gone\n", (double)((long long)free0 - (long long)f3) / 1048576.0);
    return 0;
}

int main() {
    int n = 0;
    if (hsa_init() == HSA_STATUS_SUCCESS) {
        hsa_iterate_agents(agent_cb, &n);
    } else printf("hsa_init failed\n");
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    printf("GPU_MAX_HW_QUEUES=%s\n", q ? q : "(unset)");
    CK(hipSetDevice(0));
    size_t fr = 0, tot = 0;
    CK(hipMemGetInfo(&fr, &tot));
    printf("device memory: %.1f GiB total, %.1f GiB free\n", tot / 1073741824.0, fr / 1073741824.0);
    if (probe<0>("none", 8)) return 1;
    if (probe<64>("256 B", 8)) return 1;
    if (probe<176>("704 B", 8)) return 1;
    if (probe<313>("1252 B", 8)) return 1;
    if (probe<572>("2288 B", 8)) return 1;
    if (probe<764>("3056 B", 8)) return 1;
    return 0;
}
