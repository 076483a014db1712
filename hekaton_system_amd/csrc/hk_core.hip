#include <utility>
// hk_core.hip — context / lane management and the C ABI entry points (dispatch to per-curve code).
#include "hk_internal.h"
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <atomic>

namespace hk {

// ---- scratch budget ------------------------------------------------------------------------
// Measured on MI355X / ROCm 7.2 (tools/scratch_probe.hip, profiles/r03_scratch_probe.txt): a hardware queue that has
// run a kernel with B bytes of private memory per lane keeps a ring of B x 64 lanes x (CUs x 32 wave slots) bytes for
// as long as it lives - 1.49 GiB for the 3056-byte frame of k_msm_reduce_fused<Fp2<Bls381FqP>> - however few waves the
// kernel launches; the rings of all queues share HSA_AMD_AGENT_INFO_SCRATCH_LIMIT_MAX (32 GiB).  Streams map onto at
// most GPU_MAX_HW_QUEUES queues and any of them may end up running the curve's deepest kernel, so the bound that no
// setting of HK_MAX_LANES / HK_SERIAL_STREAMS can break is   queues x ring(deepest frame) + reserve <= limit.
static hsa_status_t scratch_limit_cb(hsa_agent_t a, void* data) {
    hsa_device_type_t type;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS || type != HSA_DEVICE_TYPE_GPU)
        return HSA_STATUS_SUCCESS;
    uint64_t mx = 0;
    if (hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_SCRATCH_LIMIT_MAX, &mx) == HSA_STATUS_SUCCESS && mx) {
        uint64_t* best = (uint64_t*)data;
        if (*best == 0 || mx < *best) *best = mx;
    }
    return HSA_STATUS_SUCCESS;
}
static uint64_t scratch_limit_bytes() {
    static const uint64_t v = [] {
        uint64_t best = 0;
        if (hsa_init() == HSA_STATUS_SUCCESS) {            // reference-counted; HIP holds its own reference
            (void)hsa_iterate_agents(scratch_limit_cb, &best);
            (void)hsa_shut_down();
        }
        return best ? best : (uint64_t)32 << 30;           // what this pool's MI355X report
    }();
    return v;
}
static std::atomic<size_t> g_deepest_frame{0};             // over every curve a context was created for in this process

// Ring pre-sizing.  A queue's ring is re-allocated every time a kernel with a deeper frame than it has seen arrives; rings
// that grow one after another through several sizes fragment the agent's scratch range, and a request can then fail with
// most of the budget free.  Every stream a lane creates therefore first runs one wave of a kernel whose frame is the
// deepest of the process (rounded up to 256 B), in stream-creation = queue order: each ring is allocated once, at its
// final size.  k_scratch_presize<W> owns W * 4 bytes of private memory (runtime-indexed, so it cannot live in registers).
template <int WORDS>
__global__ void __launch_bounds__(64) k_scratch_presize(unsigned* sink, unsigned seed) {
    unsigned buf[WORDS];
    unsigned x = seed + threadIdx.x;
    for (int i = 0; i < WORDS; i += 16) buf[i] = x + i;
    unsigned j = (x * 2654435761u) % (unsigned)WORDS;
    buf[j & ~15u] += x;
    if (sink && buf[(j * 7u) % (unsigned)WORDS & ~15u] == 0x9e3779b9u) *sink = x;     // never true in practice; keeps buf alive
}
typedef void (*presize_fn)(unsigned*, unsigned);
template <int... I> struct PresizeTable { static const presize_fn fn[sizeof...(I)]; };
template <int... I> const presize_fn PresizeTable<I...>::fn[sizeof...(I)] = {k_scratch_presize<64 * (I + 1)>...};
typedef PresizeTable<0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
                     30, 31> Presize;                        // frames of 256 B ... 8 KiB
static presize_fn presize_kernel_for(size_t frame, size_t* actual) {
    size_t idx = frame ? (frame + 255) / 256 - 1 : 0;
    if (idx > 31) idx = 31;
    for (; idx < 32; idx++) {                               // the compiler may add a few bytes: take the first that is deep enough
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, (const void*)Presize::fn[idx]) != hipSuccess) { (void)hipGetLastError(); break; }
        if ((size_t)fa.localSizeBytes >= frame || idx == 31) { *actual = (size_t)fa.localSizeBytes; return Presize::fn[idx]; }
    }
    *actual = frame;
    return nullptr;
}

hk_status scratch_budget_check(const CurveOps* ops, const hipDeviceProp_t& prop, void** presize_out) {
    // the deepest frame of EVERY curve the library carries, not only this context's: a process that opens a context of the
    // other curve later (the bench's BLS12-381 leg after its BN254 leg) then finds every queue's ring already at its final
    // size - no ring ever grows, and the moment the round-3 experiment aborted in (twenty rings wanted at once while the
    // previous context's were still held) does not arise
    size_t frame = ops->max_private_bytes ? ops->max_private_bytes() : 0;
    for (const CurveOps* o : {curve_ops_bn254(), curve_ops_bls381()})
        if (o && o->max_private_bytes && o->max_private_bytes() > frame) frame = o->max_private_bytes();
    size_t prev = g_deepest_frame.load();
    while (frame > prev && !g_deepest_frame.compare_exchange_weak(prev, frame)) {}
    frame = g_deepest_frame.load();
    size_t presized = frame;
    *presize_out = (void*)presize_kernel_for(frame, &presized);
    if (presized > frame) frame = presized;                 // the ring every queue will actually hold
    const char* q = getenv("GPU_MAX_HW_QUEUES");
    uint64_t queues = q && atoi(q) > 0 ? (uint64_t)atoi(q) : 4;          // the HIP runtime's default
    uint64_t slots = (uint64_t)prop.multiProcessorCount * (uint64_t)(prop.maxThreadsPerMultiProcessor / 64);
    uint64_t ring = (uint64_t)frame * 64 * slots;
    uint64_t limit = scratch_limit_bytes();
    const uint64_t reserve = (uint64_t)1 << 30;                          // other users of the agent (RCCL, torch, rocprofv3)
    if (queues * ring + reserve > limit) {
        uint64_t fit = ring ? (limit - reserve) / ring : queues;
        fprintf(stderr, "[hekaton] scratch budget: %llu hardware queues x %.2f GiB (deepest kernel frame %zu B per lane x 64 x "
                        "%llu wave slots) + 1 GiB reserve exceeds the agent's scratch limit of %.1f GiB; export "
                        "GPU_MAX_HW_QUEUES=%llu or less before the first HIP call%s\n",
                (unsigned long long)queues, ring / 1073741824.0, frame, (unsigned long long)slots, limit / 1073741824.0,
                (unsigned long long)fit, getenv("HK_PAIR_SERIAL") ? " (or unset HK_PAIR_SERIAL: its kernels carry the deepest frames)" : "");
        return HK_ERR_DEVICE;
    }
    return HK_OK;
}

static thread_local hk_timings tl_last_timings = {};

hk_status Lane::reserve(size_t bytes) {
    arena_off = 0;
    if (bytes <= arena_cap) return HK_OK;
    HK_HIP(hipStreamSynchronize(stream));
    if (owner) {
        // an idle lane of the context may hold an arena that is large enough (an earlier call of this size ran there):
        // trade arenas with it - the smallest that fits - instead of allocating (a hipMalloc of tens of MB takes
        // milliseconds, and which lane a call lands on is arbitrary).  An idle lane's stream has nothing in flight.
        std::lock_guard<std::mutex> lk(owner->mu);
        Lane* best = nullptr;
        for (Lane* l : owner->lanes)
            if (l != this && !l->busy && l->arena_cap >= bytes && (!best || l->arena_cap < best->arena_cap)) best = l;
        if (best) {
            std::swap(arena, best->arena);
            std::swap(arena_cap, best->arena_cap);
            return HK_OK;
        }
    }
    if (arena) retired.push_back(arena);          // not hipFree here: it would wait for every other lane's kernels
    arena = nullptr;
    arena_cap = 0;
    size_t want = bytes + bytes / 8 + (1u << 20);
    hipError_t e = hipMalloc((void**)&arena, want);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        fprintf(stderr, "[hekaton] scratch allocation of %zu bytes failed: %s\n", want, hipGetErrorName(e));
        return HK_ERR_NOMEM;
    }
    arena_cap = want;
    return HK_OK;
}

LaneGuard::LaneGuard(hk_ctx* c) : ctx(c), lane(nullptr) {
    std::unique_lock<std::mutex> lk(ctx->mu);
    for (;;) {
        // among the free lanes the one with the largest scratch arena: a lane that has to grow its arena frees the old one,
        // and hipFree waits for the whole device - with calls of mixed sizes side by side (the aggregator: 6 ms pairing
        // calls beside 1 ms folds) a big call on a small lane stalled everything for milliseconds
        for (Lane* l : ctx->lanes)
            if (!l->busy && (!lane || l->arena_cap > lane->arena_cap)) lane = l;
        if (lane) break;
        if (ctx->lanes.size() < ctx->max_lanes) {
            Lane* l = new Lane();
            (void)hipSetDevice(ctx->device);
            for (auto& e : l->ev) e = nullptr;
            bool ok = hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) == hipSuccess;
            for (auto& e : l->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
            for (auto& a : l->aux) ok = ok && hipStreamCreateWithFlags(&a, hipStreamNonBlocking) == hipSuccess;
            if (!ok) {                       // a half-built lane would run on the legacy default stream / null events
                (void)hipGetLastError();
                fprintf(stderr, "[hekaton] could not create the streams / events of a lane\n");
                for (auto& e : l->ev) if (e) (void)hipEventDestroy(e);
                for (auto& a : l->aux) if (a) (void)hipStreamDestroy(a);
                if (l->stream) (void)hipStreamDestroy(l->stream);
                delete l;
                break;
            }
            // The runtime hands hardware queues to streams round robin in creation order; with 20 queues and five streams
            // per lane the FIRST stream of lane k and of lane k + 4 share a queue, and two concurrent single-kernel calls
            // (the aggregator's sweeps and pairings, the witness programs) on those lanes run one after the other.  Lanes
            // 4 .. 7 therefore work on their second stream, lanes 8 .. 11 on their third, ...: queues 0, 5, 10, 15, then
            // 1, 6, 11, 16, then 2, 7, ...  (the five streams of a lane are interchangeable for hk_prove's fork / join).
            // Streams stay lazily created, lane by lane: creating all of them at once made a second context of the
            // process abort with HSA_STATUS_ERROR_OUT_OF_RESOURCES in its first scratch pre-sizing (twenty new queues
            // wanting their rings while the previous context's were still held).
            {
                // all five roles rotate (HK_LANE_ROT positions per group of four lanes, default 1): not only the working
                // stream of single-kernel calls moves to another queue, hk_prove's heavy side streams of lanes k and
                // k + 4 stop sharing queues role by role
                static const size_t step = [] { const char* e = getenv("HK_LANE_ROT"); return e ? (size_t)atoi(e) % 5 : (size_t)1; }();
                size_t rot = ((ctx->lanes.size() / 4) * step) % 5;
                if (rot) {
                    hipStream_t all[5] = {l->stream, l->aux[0], l->aux[1], l->aux[2], l->aux[3]};
                    l->stream = all[rot % 5];
                    for (size_t j = 0; j < 4; j++) l->aux[j] = all[(rot + 1 + j) % 5];
                }
            }
            if (ctx->presize_kernel && !getenv("HK_NO_SCRATCH_PRESIZE")) {
                // one wave of the deepest frame on each of the lane's streams, in creation order (see k_scratch_presize)
                presize_fn k = (presize_fn)ctx->presize_kernel;
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, l->stream, (unsigned*)nullptr, 1u);
                for (auto& a : l->aux) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, a, (unsigned*)nullptr, 1u);
                bool okp = hipStreamSynchronize(l->stream) == hipSuccess;
                for (auto& a : l->aux) okp = okp && hipStreamSynchronize(a) == hipSuccess;
                if (!okp) { (void)hipGetLastError(); fprintf(stderr, "[hekaton] scratch pre-sizing of a lane failed\n"); }
            }
            memset(&l->timings, 0, sizeof(l->timings));
            ctx->lanes.push_back(l);
            lane = l;
            break;
        }
        ctx->cv.wait(lk);
    }
    if (lane) { lane->busy = true; lane->owner = ctx; }
    lk.unlock();
    (void)hipSetDevice(ctx->device);
}

LaneGuard::~LaneGuard() {
    if (!lane) return;
    std::unique_lock<std::mutex> lk(ctx->mu);
    lane->busy = false;
    ctx->last = lane->timings;
    tl_last_timings = lane->timings;
    bool idle = true;
    for (Lane* l : ctx->lanes) idle = idle && !l->busy;
    if (idle)                                      // nothing of this context is on the GPU: outgrown arenas go now
        for (Lane* l : ctx->lanes) {
            if (l->retired.empty()) continue;
            (void)hipSetDevice(ctx->device);
            for (void* p : l->retired) (void)hipFree(p);
            l->retired.clear();
        }
    lk.unlock();
    ctx->cv.notify_one();
}

bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

hk_status to_device(Lane* L, const void* p, size_t bytes, const void** out) {
    if (bytes == 0) { *out = p; return HK_OK; }
    if (is_device_ptr(p)) { *out = p; return HK_OK; }
    void* d = L->alloc(bytes);
    if (!d) return HK_ERR_NOMEM;
    HK_HIP(hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, L->stream));
    *out = d;
    return HK_OK;
}

}  // namespace hk

using namespace hk;

extern "C" {

const char* hk_status_str(hk_status s) {
    switch (s) {
        case HK_OK: return "HK_OK";
        case HK_ERR_LEN: return "HK_ERR_LEN";
        case HK_ERR_DOMAIN_TOO_LARGE: return "HK_ERR_DOMAIN_TOO_LARGE";
        case HK_ERR_DEVICE: return "HK_ERR_DEVICE";
        case HK_ERR_ARG: return "HK_ERR_ARG";
        case HK_ERR_NOMEM: return "HK_ERR_NOMEM";
    }
    return "HK_ERR_UNKNOWN";
}

const char* hk_version(void) { return "hekaton-mi355x 0.1 (gfx950)"; }

hk_status hk_ctx_create(hk_curve curve, int device_id, hk_ctx** out) {
    if (!out) return HK_ERR_ARG;
    *out = nullptr;
    if (curve != HK_BN254 && curve != HK_BLS12_381) return HK_ERR_ARG;
    // Concurrent proofs want more hardware queues than the HIP runtime's default of 4 (GPU_MAX_HW_QUEUES=20: DESIGN.md
    // section 5).  The runtime reads that variable when it initialises and the library does NOT touch the process
    // environment (setenv is not safe against another thread's getenv in a multi-threaded host): the host exports it
    // before its first HIP call - the bindings and drivers of this repository do (capi.py, apps/hk_all_in_one.cpp,
    // INTEGRATION.md section 3).
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        fprintf(stderr, "[hekaton] no HIP device: this library has no CPU path\n");
        return HK_ERR_DEVICE;
    }
    if (device_id < 0 || device_id >= n) return HK_ERR_DEVICE;
    HK_HIP(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HK_HIP(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "[hekaton] device %d is %s; kernels are built for gfx950 only\n", device_id,
                prop.gcnArchName);
        return HK_ERR_DEVICE;
    }
    const hk::CurveOps* ops = curve == HK_BN254 ? curve_ops_bn254() : curve_ops_bls381();
    // no environment setting may be able to exhaust the runtime's scratch pool (the abort of round 2): refuse here
    void* presize = nullptr;
    HK_TRY(hk::scratch_budget_check(ops, prop, &presize));
    hk_ctx* c = new hk_ctx();
    c->presize_kernel = presize;
    c->curve = curve;
    c->device = device_id;
    c->ops = ops;
    memset(&c->last, 0, sizeof(c->last));
    const char* ml = getenv("HK_MAX_LANES");
    if (ml && atoi(ml) > 0) c->max_lanes = (size_t)atoi(ml);
    *out = c;
    return HK_OK;
}

void hk_ctx_destroy(hk_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->ops && ctx->ops->ctx_release) ctx->ops->ctx_release(ctx);
    for (Lane* l : ctx->lanes) {
        if (l->arena) (void)hipFree(l->arena);
        for (void* p : l->retired) (void)hipFree(p);
        if (l->pinned) (void)hipHostFree(l->pinned);
        for (auto& e : l->ev) (void)hipEventDestroy(e);
        for (auto& a : l->aux) if (a) (void)hipStreamDestroy(a);
        (void)hipStreamDestroy(l->stream);
        delete l;
    }
    delete ctx;
}

hk_status hk_ctx_sync(hk_ctx* ctx) {
    if (!ctx) return HK_ERR_ARG;
    HK_HIP(hipSetDevice(ctx->device));
    std::unique_lock<std::mutex> lk(ctx->mu);
    for (Lane* l : ctx->lanes) {
        HK_HIP(hipStreamSynchronize(l->stream));
        for (auto& a : l->aux) if (a) HK_HIP(hipStreamSynchronize(a));
    }
    return HK_OK;
}

hk_status hk_ctx_set_profiling(hk_ctx* ctx, int enable) {
    if (!ctx) return HK_ERR_ARG;
    ctx->profiling = enable;
    return HK_OK;
}

hk_status hk_ctx_last_timings(hk_ctx* ctx, hk_timings* out) {
    if (!ctx || !out) return HK_ERR_ARG;
    *out = hk::tl_last_timings;        // timings of the calling thread's most recent call
    return HK_OK;
}

hk_status hk_ctx_sizes(const hk_ctx* ctx, size_t* fr, size_t* fq, size_t* g1, size_t* g2) {
    if (!ctx) return HK_ERR_ARG;
    if (fr) *fr = ctx->ops->fr_bytes;
    if (fq) *fq = ctx->ops->fq_bytes;
    if (g1) *g1 = ctx->ops->g1_bytes;
    if (g2) *g2 = ctx->ops->g2_bytes;
    return HK_OK;
}

hk_status hk_dev_alloc(hk_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return HK_ERR_ARG;
    HK_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
    if (e != hipSuccess) { (void)hipGetLastError(); return HK_ERR_NOMEM; }
    return HK_OK;
}
hk_status hk_dev_free(hk_ctx* ctx, void* dptr) {
    if (!ctx) return HK_ERR_ARG;
    HK_HIP(hipSetDevice(ctx->device));
    HK_HIP(hipFree(dptr));
    return HK_OK;
}
hk_status hk_dev_upload(hk_ctx* ctx, void* dst_d, const void* src_h, size_t bytes) {
    if (!ctx) return HK_ERR_ARG;
    HK_HIP(hipSetDevice(ctx->device));
    HK_HIP(hipMemcpy(dst_d, src_h, bytes, hipMemcpyHostToDevice));
    return HK_OK;
}
hk_status hk_dev_download(hk_ctx* ctx, void* dst_h, const void* src_d, size_t bytes) {
    if (!ctx) return HK_ERR_ARG;
    HK_HIP(hipSetDevice(ctx->device));
    HK_HIP(hipMemcpy(dst_h, src_d, bytes, hipMemcpyDeviceToHost));
    return HK_OK;
}

hk_status hk_msm_g1(hk_ctx* ctx, const void* bases, size_t n_bases, const void* scalars, size_t n_scalars,
                    int mont, int checked, void* out) {
    if (!ctx || !out) return HK_ERR_ARG;
    return ctx->ops->msm(ctx, 1, bases, n_bases, scalars, n_scalars, mont, checked, out);
}
hk_status hk_msm_g2(hk_ctx* ctx, const void* bases, size_t n_bases, const void* scalars, size_t n_scalars,
                    int mont, int checked, void* out) {
    if (!ctx || !out) return HK_ERR_ARG;
    return ctx->ops->msm(ctx, 2, bases, n_bases, scalars, n_scalars, mont, checked, out);
}
hk_status hk_fixed_base_g1(hk_ctx* ctx, const void* base, const void* scalars, size_t n, int mont, void* out) {
    if (!ctx || !base || (n && (!scalars || !out))) return HK_ERR_ARG;
    return ctx->ops->fixed_base(ctx, 1, base, scalars, n, mont, out);
}
hk_status hk_fixed_base_g2(hk_ctx* ctx, const void* base, const void* scalars, size_t n, int mont, void* out) {
    if (!ctx || !base || (n && (!scalars || !out))) return HK_ERR_ARG;
    return ctx->ops->fixed_base(ctx, 2, base, scalars, n, mont, out);
}
hk_status hk_scalar_pairing_g1(hk_ctx* ctx, const void* points, const void* scalars, size_t n, void* out) {
    if (!ctx || (n && (!points || !scalars || !out))) return HK_ERR_ARG;
    return ctx->ops->scalar_pairing(ctx, 1, points, scalars, n, out);
}
hk_status hk_scalar_pairing_g2(hk_ctx* ctx, const void* points, const void* scalars, size_t n, void* out) {
    if (!ctx || (n && (!points || !scalars || !out))) return HK_ERR_ARG;
    return ctx->ops->scalar_pairing(ctx, 2, points, scalars, n, out);
}
hk_status hk_bases_upload(hk_ctx* ctx, int group, const void* bases, size_t n, hk_bases** out) {
    if (!ctx || !out || (group != 1 && group != 2) || (n && !bases)) return HK_ERR_ARG;
    return ctx->ops->bases_upload(ctx, group, bases, n, out);
}
void hk_bases_free(hk_bases* b) {
    if (b) b->ops->bases_free(b);
}
hk_status hk_msm_bases(hk_ctx* ctx, const hk_bases* b, const void* scalars, size_t n_scalars, int mont, int checked,
                       void* out) {
    if (!ctx || !b || !out || b->ctx != ctx) return HK_ERR_ARG;
    return ctx->ops->msm_bases(ctx, b, scalars, n_scalars, mont, checked, out);
}
hk_status hk_wprog_upload(hk_ctx* ctx, const uint32_t* ops, size_t n_ops, const uint32_t* refs, size_t n_refs,
                          const uint32_t* map, size_t n_v, size_t n_values, size_t n_inputs, hk_wprog** out) {
    if (!ctx || !ops || !map || !out || (n_refs && !refs)) return HK_ERR_ARG;
    return ctx->ops->wprog_upload(ctx, ops, n_ops, refs, n_refs, map, n_v, n_values, n_inputs, out);
}
void hk_wprog_free(hk_wprog* w) {
    if (w) w->ops->wprog_free(w);
}
hk_status hk_poseidon_path(hk_ctx* ctx, const void* consts_mont, size_t n_consts, const hk_poseidon_desc* leaf_hash,
                           const hk_poseidon_desc* node_hash, const void* leaf_mont, const void* siblings_mont,
                           const uint32_t* leaf_index, size_t depth, size_t batch, size_t n_v, size_t col0, void* z_out) {
    if (!ctx) return HK_ERR_ARG;
    return ctx->ops->poseidon_path(ctx, consts_mont, n_consts, leaf_hash, node_hash, leaf_mont, siblings_mont, leaf_index,
                                   depth, batch, n_v, col0, z_out);
}
hk_status hk_wprog_run(hk_ctx* ctx, const hk_wprog* w, const uint32_t* inputs, size_t batch, const uint32_t* full_cols,
                       const void* full_vals_mont, size_t n_full, void* z_out) {
    if (!ctx || !w || w->ctx != ctx || !z_out || (batch && !inputs) || (n_full && (!full_cols || !full_vals_mont))) return HK_ERR_ARG;
    return ctx->ops->wprog_run(ctx, w, inputs, batch, full_cols, full_vals_mont, n_full, z_out);
}
hk_status hk_assignment_scatter(hk_ctx* ctx, const uint32_t* full_cols, const void* full_vals_mont, size_t n_full, size_t batch,
                                size_t n_v, void* z_out) {
    if (!ctx || !z_out || (n_full && batch && (!full_cols || !full_vals_mont))) return HK_ERR_ARG;
    return ctx->ops->assignment_scatter(ctx, full_cols, full_vals_mont, n_full, batch, n_v, z_out);
}
hk_status hk_assignment_from_bits(hk_ctx* ctx, const void* bits, size_t n_v, const uint32_t* full_cols,
                                  const void* full_vals_mont, size_t n_full, void* z_out) {
    if (!ctx || !z_out || (n_v && !bits) || (n_full && (!full_cols || !full_vals_mont))) return HK_ERR_ARG;
    return ctx->ops->assignment_from_bits(ctx, bits, n_v, full_cols, full_vals_mont, n_full, z_out);
}
hk_status hk_gt_pow(hk_ctx* ctx, const void* gt_in, const void* scalars_mont, size_t n, void* gt_out) {
    if (!ctx || (n && (!gt_in || !scalars_mont || !gt_out))) return HK_ERR_ARG;
    return ctx->ops->gt_pow(ctx, gt_in, scalars_mont, n, gt_out, 1, 1);
}
hk_status hk_fq12_pow(hk_ctx* ctx, const void* gt_in, const void* scalars_mont, size_t n, void* gt_out) {
    if (!ctx || (n && (!gt_in || !scalars_mont || !gt_out))) return HK_ERR_ARG;
    return ctx->ops->gt_pow(ctx, gt_in, scalars_mont, n, gt_out, 0, 1);
}
hk_status hk_gt_pow_prod(hk_ctx* ctx, const void* gt_in, const void* scalars_mont, size_t n, size_t group_len, int in_gt,
                         void* gt_out) {
    if (!ctx || group_len == 0 || (n && (!gt_in || !scalars_mont || !gt_out))) return HK_ERR_ARG;
    return ctx->ops->gt_pow(ctx, gt_in, scalars_mont, n, gt_out, in_gt ? 1 : 0, group_len);
}
hk_status hk_points_lincomb_g1(hk_ctx* ctx, const void* const* vecs, const void* coeffs_mont, size_t k, size_t n, void* out) {
    if (!ctx || !vecs || !coeffs_mont || (n && !out)) return HK_ERR_ARG;
    return ctx->ops->points_lincomb(ctx, 1, vecs, coeffs_mont, k, n, out);
}
hk_status hk_points_lincomb_g2(hk_ctx* ctx, const void* const* vecs, const void* coeffs_mont, size_t k, size_t n, void* out) {
    if (!ctx || !vecs || !coeffs_mont || (n && !out)) return HK_ERR_ARG;
    return ctx->ops->points_lincomb(ctx, 2, vecs, coeffs_mont, k, n, out);
}
hk_status hk_points_fold_g1(hk_ctx* ctx, const void* lo, const void* hi, const void* coeffs2_mont, unsigned neg_mask, size_t n,
                            void* out) {
    if (!ctx || !ctx->ops) return HK_ERR_ARG;
    return ctx->ops->points_fold_g1(ctx, lo, hi, coeffs2_mont, neg_mask, n, out);
}
hk_status hk_points_fold_g2(hk_ctx* ctx, const void* lo, const void* hi, const void* coeffs4_mont, unsigned neg_mask, size_t n,
                            void* out) {
    if (!ctx || !ctx->ops) return HK_ERR_ARG;
    return ctx->ops->points_fold_g2(ctx, lo, hi, coeffs4_mont, neg_mask, n, out);
}
hk_status hk_points_fold_many_g1(hk_ctx* ctx, size_t k, const void* const* lo, const void* const* hi, const void* coeffs2_mont,
                                 unsigned neg_mask, size_t n, void* const* out) {
    if (!ctx || !ctx->ops) return HK_ERR_ARG;
    return ctx->ops->points_fold_many(ctx, 1, k, lo, hi, coeffs2_mont, neg_mask, n, out);
}
hk_status hk_points_fold_many_g2(hk_ctx* ctx, size_t k, const void* const* lo, const void* const* hi, const void* coeffs4_mont,
                                 unsigned neg_mask, size_t n, void* const* out) {
    if (!ctx || !ctx->ops) return HK_ERR_ARG;
    return ctx->ops->points_fold_many(ctx, 2, k, lo, hi, coeffs4_mont, neg_mask, n, out);
}
hk_status hk_pairing_pairs(hk_ctx* ctx, const void* const* lhs_g1, size_t n_lhs, const void* const* rhs_g2, size_t n_rhs,
                           const uint32_t* pair_lhs, const uint32_t* pair_rhs, size_t n_pairs, size_t n, void* gt_out) {
    if (!ctx || !lhs_g1 || !rhs_g2 || !pair_lhs || !pair_rhs || !gt_out) return HK_ERR_ARG;
    return ctx->ops->pairing_pairs(ctx, lhs_g1, n_lhs, rhs_g2, n_rhs, pair_lhs, pair_rhs, n_pairs, n, gt_out);
}
hk_status hk_pairing_products(hk_ctx* ctx, const void* const* lhs_g1, size_t n_lhs, const void* const* rhs_g2,
                              size_t n_rhs, size_t n, void* gt_out) {
    if (!ctx || !lhs_g1 || !rhs_g2 || !gt_out) return HK_ERR_ARG;
    return ctx->ops->pairing_products(ctx, lhs_g1, n_lhs, rhs_g2, n_rhs, n, gt_out);
}
hk_status hk_multi_pairing(hk_ctx* ctx, const void* g1, const void* g2, size_t n, void* gt_out) {
    if (!ctx || !gt_out || (n && (!g1 || !g2))) return HK_ERR_ARG;
    const void* l[1] = {n ? g1 : (const void*)gt_out};
    const void* r[1] = {n ? g2 : (const void*)gt_out};
    return ctx->ops->pairing_products(ctx, l, 1, r, 1, n, gt_out);
}
hk_status hk_ctx_gt_bytes(const hk_ctx* ctx, size_t* gt) {
    if (!ctx || !gt) return HK_ERR_ARG;
    *gt = ctx->ops->gt_bytes;
    return HK_OK;
}
hk_status hk_field_convert(hk_ctx* ctx, int which, const void* in, void* out, size_t n, int to_mont) {
    if (!ctx || (which != 0 && which != 1) || (n && (!in || !out))) return HK_ERR_ARG;
    return ctx->ops->field_convert(ctx, which, in, out, n, to_mont);
}
hk_status hk_ntt(hk_ctx* ctx, void* data, unsigned log_m, int inverse, int coset) {
    if (!ctx || !data) return HK_ERR_ARG;
    return ctx->ops->ntt(ctx, data, log_m, inverse, coset);
}
hk_status hk_witness_map(hk_ctx* ctx, const hk_csr* A, const hk_csr* B, const hk_csr* C, size_t n_inst,
                         size_t n_constraints, const void* z_mont, size_t n_v, void* h_out,
                         size_t h_capacity, size_t* m_out) {
    if (!ctx || !A || !B || !C || !z_mont || !h_out) return HK_ERR_ARG;
    return ctx->ops->witness_map(ctx, A, B, C, n_inst, n_constraints, z_mont, n_v, h_out, h_capacity, m_out);
}
hk_status hk_pk_upload(hk_ctx* ctx, const hk_pk_desc* desc, hk_pk** out) {
    if (!ctx || !desc || !out) return HK_ERR_ARG;
    return ctx->ops->pk_upload(ctx, desc, out);
}
void hk_pk_free(hk_pk* pk) {
    if (!pk) return;
    pk->ops->pk_free(pk);
}
hk_status hk_commit(hk_ctx* ctx, const hk_pk* pk, size_t stage, const void* w, size_t n, const void* kappa,
                    void* out) {
    if (!ctx || !pk || !kappa || !out) return HK_ERR_ARG;
    return ctx->ops->commit(ctx, pk, stage, w, n, kappa, out);
}
hk_status hk_commit_batch(hk_ctx* ctx, const hk_pk* pk, size_t stage, const void* w, size_t n, const void* kappas, size_t batch,
                          void* out) {
    if (!ctx || !pk || (batch && (!kappas || !out))) return HK_ERR_ARG;
    return ctx->ops->commit_batch(ctx, pk, stage, w, n, kappas, batch, out);
}
hk_status hk_prove(hk_ctx* ctx, const hk_pk* pk, const void* z, size_t n_v, const void* r, const void* s,
                   const void* kappas, size_t n_kappas, void* a, void* b, void* c) {
    if (!ctx || !pk || !z || !r || !s || !a || !b || !c) return HK_ERR_ARG;
    return ctx->ops->prove(ctx, pk, z, n_v, r, s, kappas, n_kappas, a, b, c);
}

}  // extern "C"

// ---- host utility: Keccak-f[1600], the permutation under the merlin transcripts of the aggregator (STROBE-128;
// distributed-prover/src/util.rs:22, aggregation.rs:219-222,276-278).  Plain C on the host: a pure-Python permutation costs
// 0.35 ms a call, 30 ms per aggregation.  state: 25 little-endian 64-bit lanes, lane (x, y) at x + 5 y.
extern "C" void hk_keccak_f1600(uint64_t* st) {
    static const uint64_t RC[24] = {
        0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
        0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
        0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
        0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};
    static const int ROT[5][5] = {{0, 36, 3, 41, 18}, {1, 44, 10, 45, 2}, {62, 6, 43, 15, 61}, {28, 55, 25, 21, 56}, {27, 20, 39, 8, 14}};
    auto rol = [](uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; };
    for (int rnd = 0; rnd < 24; rnd++) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) st[i] ^= d[i % 5];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(st[x + 5 * y], ROT[x][y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) st[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        st[0] ^= RC[rnd];
    }
}
