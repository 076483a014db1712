// hk_internal.h — context, lanes and scratch arenas shared by the translation units of libhekaton.
#pragma once
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hekaton.h"
#include "msm.cuh"

#define HK_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            fprintf(stderr, "[hekaton] HIP error %s at %s:%d: %s\n", hipGetErrorName(_e),    \
                    __FILE__, __LINE__, #expr);                                              \
            return HK_ERR_DEVICE;                                                            \
        }                                                                                    \
    } while (0)

#define HK_TRY(expr)                       \
    do {                                   \
        hk_status _s = (expr);             \
        if (_s != HK_OK) return _s;        \
    } while (0)

namespace hk {

// ---- curve traits -----------------------------------------------------------------------------
struct CurveBn254 {
    typedef Fp<Bn254FrP> Fr;
    typedef Fp<Bn254FqP> Fq;
    typedef Fp2<Bn254FqP> Fq2;
    static constexpr u32 FR_BITS = HK_BN254_FR_BITS;
    static constexpr u32 TWO_ADICITY = HK_BN254_TWO_ADICITY;
    static constexpr u32 ROOT[8] = HK_BN254_FR_ROOT;
    static constexpr u32 GEN[8] = HK_BN254_FR_GEN;
    static constexpr u32 GEN_INV[8] = HK_BN254_FR_GEN_INV;
};
struct CurveBls381 {
    typedef Fp<Bls381FrP> Fr;
    typedef Fp<Bls381FqP> Fq;
    typedef Fp2<Bls381FqP> Fq2;
    static constexpr u32 FR_BITS = HK_BLS12_381_FR_BITS;
    static constexpr u32 TWO_ADICITY = HK_BLS12_381_TWO_ADICITY;
    static constexpr u32 ROOT[8] = HK_BLS12_381_FR_ROOT;
    static constexpr u32 GEN[8] = HK_BLS12_381_FR_GEN;
    static constexpr u32 GEN_INV[8] = HK_BLS12_381_FR_GEN_INV;
};

// scalar field that goes with a coordinate field
template <class F> struct ScalarOf;
template <> struct ScalarOf<CurveBn254::Fq> { typedef CurveBn254::Fr type; };
template <> struct ScalarOf<CurveBn254::Fq2> { typedef CurveBn254::Fr type; };
template <> struct ScalarOf<CurveBls381::Fq> { typedef CurveBls381::Fr type; };
template <> struct ScalarOf<CurveBls381::Fq2> { typedef CurveBls381::Fr type; };

// ---- per-call lane: one stream + one grow-only scratch arena ---------------------------------------
struct Lane {
    hipStream_t stream = nullptr;
    char* arena = nullptr;
    size_t arena_cap = 0;
    size_t arena_off = 0;
    void* pinned = nullptr;       // small host staging buffer
    size_t pinned_cap = 0;
    hipEvent_t ev[32];
    hipStream_t aux[4] = {nullptr, nullptr, nullptr, nullptr};   // fork/join side streams of hk_prove
    bool busy = false;
    hk_timings timings;
    std::vector<void*> retired;   // outgrown arenas: freed when no call is in flight (hipFree waits for the whole device)
    hk_ctx* owner = nullptr;      // the context the lane belongs to (reserve() may trade arenas with an idle lane of it)

    hk_status reserve(size_t bytes);                 // ensure capacity (may sync + realloc), reset
    void* alloc(size_t bytes) {                      // bump allocation, 256-B aligned
        size_t off = (arena_off + 255) & ~(size_t)255;
        if (off + bytes > arena_cap) return nullptr;
        arena_off = off + bytes;
        return arena + off;
    }
    template <class T> T* alloc_n(size_t n) { return (T*)alloc(n * sizeof(T)); }
};

struct NttTables;   // ntt.hip

}  // namespace hk

namespace hk {
// per-curve entry points; each curve's translation unit fills one table
struct CurveOps {
    size_t fr_bytes, fq_bytes, g1_bytes, g2_bytes;
    hk_status (*msm)(hk_ctx*, int group, const void* bases, size_t n_bases, const void* scalars,
                     size_t n_scalars, int mont, int checked, void* out);
    hk_status (*ntt)(hk_ctx*, void* data, unsigned log_m, int inverse, int coset);
    hk_status (*witness_map)(hk_ctx*, const hk_csr* A, const hk_csr* B, const hk_csr* C, size_t n_inst,
                             size_t n_c, const void* z, size_t n_v, void* h_out, size_t h_cap, size_t* m_out);
    hk_status (*pk_upload)(hk_ctx*, const hk_pk_desc*, hk_pk**);
    void (*pk_free)(hk_pk*);
    hk_status (*commit)(hk_ctx*, const hk_pk*, size_t stage, const void* w, size_t n, const void* kappa,
                        void* out);
    hk_status (*prove)(hk_ctx*, const hk_pk*, const void* z, size_t n_v, const void* r, const void* s,
                       const void* kappas, size_t n_kappas, void* a, void* b, void* c);
    void (*ctx_release)(hk_ctx*);
    hk_status (*fixed_base)(hk_ctx*, int group, const void* base, const void* scalars, size_t n, int mont,
                            void* out);
    hk_status (*scalar_pairing)(hk_ctx*, int group, const void* points, const void* scalars, size_t n, void* out);
    hk_status (*field_convert)(hk_ctx*, int which, const void* in, void* out, size_t n, int to_mont);
    hk_status (*bases_upload)(hk_ctx*, int group, const void* bases, size_t n, hk_bases** out);
    void (*bases_free)(hk_bases*);
    hk_status (*msm_bases)(hk_ctx*, const hk_bases*, const void* scalars, size_t n_scalars, int mont, int checked,
                           void* out);
    hk_status (*pairing_products)(hk_ctx*, const void* const* lhs, size_t n_lhs, const void* const* rhs, size_t n_rhs,
                                  size_t n, void* out);
    size_t gt_bytes;
    hk_status (*points_lincomb)(hk_ctx*, int group, const void* const* vecs, const void* coeffs, size_t k, size_t n,
                                void* out);
    hk_status (*points_fold_g2)(hk_ctx*, const void* lo, const void* hi, const void* coeffs4, unsigned neg_mask, size_t n,
                                void* out);
    hk_status (*points_fold_g1)(hk_ctx*, const void* lo, const void* hi, const void* coeffs2, unsigned neg_mask, size_t n,
                                void* out);
    hk_status (*assignment_from_bits)(hk_ctx*, const void* bits, size_t n_v, const uint32_t* full_cols,
                                      const void* full_vals, size_t n_full, void* z_out);
    hk_status (*wprog_upload)(hk_ctx*, const uint32_t* ops, size_t n_ops, const uint32_t* refs, size_t n_refs,
                              const uint32_t* map, size_t n_v, size_t n_values, size_t n_inputs, hk_wprog** out);
    void (*wprog_free)(hk_wprog*);
    hk_status (*wprog_run)(hk_ctx*, const hk_wprog*, const uint32_t* inputs, size_t batch, const uint32_t* full_cols,
                           const void* full_vals, size_t n_full, void* z_out);
    hk_status (*gt_pow)(hk_ctx*, const void* gt_in, const void* scalars, size_t n, void* gt_out, int in_gt, size_t group_len);
    // largest private-memory frame (bytes per lane) among the curve's kernels: sizes the scratch ring the runtime
    // pins to every hardware queue that ever runs one of them
    size_t (*max_private_bytes)();
    hk_status (*poseidon_path)(hk_ctx*, const void* consts, size_t n_consts, const hk_poseidon_desc* leaf_hash,
                               const hk_poseidon_desc* node_hash, const void* leaf, const void* siblings,
                               const uint32_t* index, size_t depth, size_t batch, size_t n_v, size_t col0, void* z_out);
    hk_status (*points_fold_many)(hk_ctx*, int group, size_t k, const void* const* lo, const void* const* hi,
                                  const void* coeffs, unsigned neg_mask, size_t n, void* const* out);
    hk_status (*pairing_pairs)(hk_ctx*, const void* const* lhs, size_t n_lhs, const void* const* rhs, size_t n_rhs,
                               const uint32_t* pair_lhs, const uint32_t* pair_rhs, size_t n_pairs, size_t n, void* out);
    hk_status (*assignment_scatter)(hk_ctx*, const uint32_t* full_cols, const void* full_vals, size_t n_full, size_t batch,
                                    size_t n_v, void* z_out);
    hk_status (*commit_batch)(hk_ctx*, const hk_pk*, size_t stage, const void* w, size_t n, const void* kappas, size_t batch,
                              void* out);
};
const CurveOps* curve_ops_bn254();
const CurveOps* curve_ops_bls381();
}  // namespace hk

struct hk_ctx {
    hk_curve curve;
    const hk::CurveOps* ops = nullptr;
    int device;
    int profiling = 0;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<hk::Lane*> lanes;
    size_t max_lanes = 8;           // concurrent calls beyond this wait for a lane (HK_MAX_LANES)
    hk::NttTables* ntt = nullptr;
    hk_timings last;
    uint32_t max_lanes0 = 262144;   // level-0 accumulate lanes: 4 waves/SIMD x 1024 SIMDs x 64
    void* presize_kernel = nullptr; // k_scratch_presize<W> with the process's deepest frame (hk_core.hip)
    // window tables of hk_fixed_base, kept per base: a trusted setup and the aggregator's SRS multiply the two generators
    // again and again, and a table is a 248-step doubling chain (2 ms in G1, 5 - 6 ms in G2) in front of a 0.6 ms sweep.
    // At most FB_CACHE_MAX entries, never evicted; freed with the context.
    struct FbTable { int group; std::string base; void* table; bool ready; };
    enum { FB_CACHE_MAX = 8 };
    std::vector<FbTable> fb_cache;
};

struct hk_pk {
    const hk::CurveOps* ops;
    hk_ctx* ctx;
    void* impl;
};

struct hk_bases {
    const hk::CurveOps* ops;
    hk_ctx* ctx;
    void* impl;
};

namespace hk {
struct WprogImpl {                 // a class's word program on the device (witness.cuh)
    uint32_t *ops = nullptr, *refs = nullptr, *map = nullptr;
    uint32_t n_ops = 0, n_refs = 0, n_values = 0, n_inputs = 0;
    size_t n_v = 0;
};
}  // namespace hk
struct hk_wprog {
    const hk::CurveOps* ops;
    hk_ctx* ctx;
    hk::WprogImpl* impl;
};

namespace hk {

struct LaneGuard {
    hk_ctx* ctx;
    Lane* lane;
    LaneGuard(hk_ctx* c);
    ~LaneGuard();
};

bool is_device_ptr(const void* p);
// returns a device pointer for `p` (copies host data into lane scratch when needed)
hk_status to_device(Lane* L, const void* p, size_t bytes, const void** out);

}  // namespace hk
