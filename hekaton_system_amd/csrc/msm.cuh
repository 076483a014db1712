// msm.cuh — variable-base multi-scalar multiplication (Pippenger bucket method) for CDNA4.
//
// Replaces ark-ec `VariableBaseMSM::{msm_bigint,msm,msm_unchecked}` as called by the reference at
// cp-groth16/src/prover.rs:88,97,107,117,129 and committer.rs:89,113.  The result of an MSM is a
// unique group element, so any bucket schedule is bit-exact with arkworks once normalised to affine
// (SURVEY.md A.3); this schedule is designed for MI355X, not translated from ark-ec:
//
//   1. k_msm_hist / k_msm_scatter — every scalar is split into W signed c-bit digits with the
//      "add 2^(c-1) to every window" trick (digits independent, no carry chain); digits are
//      counting-sorted by bucket with a 128 KiB LDS histogram per workgroup (LDS atomics absorb the
//      heavy-hitter buckets that SHA-style 0/1 witnesses create) and one global atomic per
//      (workgroup, non-empty bucket).
//   2. k_msm_accum0 — the sorted entry list is cut into EQUAL slices, one per lane, regardless of
//      bucket boundaries (perfect load balance, 64-wide waves never idle on a short bucket); each lane
//      runs mixed XYZZ adds in registers, writes buckets it fully owns, and emits at most two boundary
//      partials.
//   3. k_msm_accum_lvl — the boundary partials (sorted by bucket by construction) are reduced by the
//      same equal-slice segmented scheme, level by level, until one lane remains.
//   4. k_msm_bucket_reduce / k_msm_window_sum / k_msm_final — weighted bucket sums, LDS tree, Horner.
//
// HBM-first trick: proving-key bases are static, HBM is 288 GB, so hk_pk_upload stores every base
// together with its 2^(c*WP*g) multiples ("shift tables", k_msm_build_tables).  All windows of a
// group then share one bucket set and the serial Horner tail shrinks from ~254 doublings to
// c*(WP-1) (zero when WP == 1).
#pragma once
#include <cstdlib>
#include "ec.cuh"

namespace hk {

constexpr int MSM_MAX_LEVELS = 16;
#ifndef HK_MSM_LVL_L
#define HK_MSM_LVL_L 16
#endif
constexpr u32 MSM_LVL_L = HK_MSM_LVL_L;          // entries per lane on levels >= 1: 6 levels for a full 2^18-lane level 0 (8: 10 levels)
constexpr int MSM_TAIL_THREADS = 256;  // levels whose lane count fits one workgroup run fused in k_msm_accum_tail
constexpr int MSM_WSUM_THREADS = 256;
#ifndef HK_MSM_SORT_THREADS
#define HK_MSM_SORT_THREADS 1024
#endif
constexpr int MSM_SORT_THREADS = HK_MSM_SORT_THREADS;   // build-time knob
// a sorted entry = sign << 31 | table group << gshift | scalar index, gshift = ceil(log2 n) <= 26 (so no division
// in the accumulate loop); 31 - gshift bits hold the group: >= 5 bits at the largest n, where c >= 13 gives F <= 20,
// and 64 groups (c = 4) only occur for n < 2^10
constexpr int MSM_ENTRY_GROUP_SHIFT = 26;
constexpr int MSM_LDS_COUNTERS = 32768; // 128 KiB of LDS per sort workgroup

struct MsmPlan {
    u32 n;            // scalars / entries per group
    u32 c;            // window bits
    u32 B;            // buckets per window = 2^(c-1)
    u32 W;            // total windows
    u32 WP;           // windows per group (share Horner); groups F = ceil(W/WP)
    u32 F;
    u32 NB;           // WP * B
    u32 n_levels;
    u32 T[MSM_MAX_LEVELS];     // lanes launched per level
    u32 Lmin0;
    u32 gshift;       // bit position of the group field in a sorted entry
    u32 chunk;        // scalars per sort workgroup
    u32 K;            // buckets per lane in bucket_reduce
    u32 kconst[10];   // sum_w 2^(c*w + c-1) as 32-bit limbs
};

// level bookkeeping shared by host and device: how many entries / lanes are live on level k
struct LevelInfo { u32 count, L, active; };
HK_HD LevelInfo msm_level_info(const MsmPlan& p, u32 E, int k) {
    LevelInfo li;
    li.count = E;
    u32 L = (E + p.T[0] - 1) / p.T[0];
    if (L < p.Lmin0) L = p.Lmin0;
    li.L = L;
    li.active = (E + L - 1) / L;
    for (int j = 1; j <= k; j++) {
        li.count = 2 * li.active;
        li.L = MSM_LVL_L;
        li.active = (li.count + li.L - 1) / li.L;
    }
    return li;
}

#if defined(__HIPCC__)

// ---- digit extraction -----------------------------------------------------------------------------
// s' = s + kconst; digit_w = ((s' >> c*w) & (2^c-1)) - 2^(c-1)  in [-2^(c-1), 2^(c-1)-1]
template <class Fr>
__device__ __forceinline__ void msm_load_scalar(const u32* scalars, size_t i, int is_mont,
                                                const MsmPlan& p, u32 (&sp)[10]) {
    Fr s;
    const uint4* src = reinterpret_cast<const uint4*>(scalars + i * Fr::N);
#pragma unroll
    for (int k = 0; k < Fr::N / 4; k++) {
        uint4 v = src[k];
        s.v[4 * k] = v.x; s.v[4 * k + 1] = v.y; s.v[4 * k + 2] = v.z; s.v[4 * k + 3] = v.w;
    }
    if (is_mont) s = Fr::from_mont(s);
    u64 carry = 0;
#pragma unroll
    for (int k = 0; k < Fr::N; k++) {
        carry += (u64)s.v[k] + p.kconst[k];
        sp[k] = (u32)carry;
        carry >>= 32;
    }
    carry += p.kconst[Fr::N];
    sp[Fr::N] = (u32)carry;
    sp[Fr::N + 1] = 0;
}

__device__ __forceinline__ int msm_digit(const u32 (&sp)[10], u32 w, u32 c) {
    u32 o = c * w;
    u32 limb = o >> 5, sh = o & 31;
    u64 two = ((u64)sp[limb + 1] << 32) | sp[limb];
    u32 e = (u32)(two >> sh) & ((1u << c) - 1u);
    return (int)e - (int)(1u << (c - 1));
}

// ---- counting sort, pass 1: histogram ------------------------------------------------------------
template <class Fr>
__global__ void __launch_bounds__(MSM_SORT_THREADS)
k_msm_hist(const u32* __restrict__ scalars, int is_mont, MsmPlan p, u32* __restrict__ count,
           short* __restrict__ digits) {
    __shared__ u32 h[MSM_LDS_COUNTERS];
    for (u32 b = threadIdx.x; b < p.NB; b += blockDim.x) h[b] = 0;
    __syncthreads();
    size_t base = (size_t)blockIdx.x * p.chunk;
    for (u32 k = threadIdx.x; k < p.chunk; k += blockDim.x) {
        size_t i = base + k;
        if (i >= p.n) break;
        u32 sp[10];
        msm_load_scalar<Fr>(scalars, i, is_mont, p, sp);
        for (u32 w = 0; w < p.W; w++) {
            int d = msm_digit(sp, w, p.c);
            // the signed digits are computed ONCE per scalar (Montgomery reduction + split) and kept, window-major,
            // for the two passes of k_msm_scatter: 2 bytes per digit, coalesced across the lanes of a wave
            digits[(size_t)w * p.n + i] = (short)d;
            if (d != 0) {
                u32 mag = d < 0 ? (u32)(-d) : (u32)d;
                atomicAdd(&h[(w % p.WP) * p.B + mag - 1], 1u);
            }
        }
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < p.NB; b += blockDim.x) {
        u32 v = h[b];
        if (v) atomicAdd(&count[b], v);
    }
}

// ---- exclusive scan of bucket counts (single workgroup) ---------------------------------------------
template <int UNUSED>
__global__ void __launch_bounds__(1024)
k_msm_scan(const u32* __restrict__ count, u32* __restrict__ start, u32* __restrict__ cursor, u32 NB) {
    __shared__ u32 part[1024];
    u32 per = (NB + 1023) / 1024;
    u32 lo = threadIdx.x * per, hi = min(lo + per, NB);
    u32 s = 0;
    for (u32 b = lo; b < hi; b++) s += count[b];
    part[threadIdx.x] = s;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (u32 off = 1; off < 1024; off <<= 1) {
        u32 v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    u32 run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (u32 b = lo; b < hi; b++) {
        start[b] = run;
        cursor[b] = run;
        run += count[b];
    }
    if (threadIdx.x == 1023) start[NB] = part[1023];
}

// ---- counting sort, pass 2: scatter entry ids ----------------------------------------------------------
// entry = sign << 31 | group << p.gshift | i.   Reads the digits k_msm_hist stored (c <= 16: a digit is an int16;
// -2^15 is the one value whose magnitude needs the unsigned form).
__device__ __forceinline__ u32 msm_mag(int d) { return d < 0 ? (u32)(-d) : (u32)d; }
template <class Fr>
__global__ void __launch_bounds__(MSM_SORT_THREADS)
k_msm_scatter(const short* __restrict__ digits, MsmPlan p, u32* __restrict__ cursor, u32* __restrict__ sorted) {
    __shared__ u32 h[MSM_LDS_COUNTERS];
    for (u32 b = threadIdx.x; b < p.NB; b += blockDim.x) h[b] = 0;
    __syncthreads();
    size_t base = (size_t)blockIdx.x * p.chunk;
    for (u32 k = threadIdx.x; k < p.chunk; k += blockDim.x) {
        size_t i = base + k;
        if (i >= p.n) break;
        for (u32 w = 0; w < p.W; w++) {
            int d = digits[(size_t)w * p.n + i];
            if (d != 0) atomicAdd(&h[(w % p.WP) * p.B + msm_mag(d) - 1], 1u);
        }
    }
    __syncthreads();
    // reserve this workgroup's range in every non-empty bucket; h[b] becomes the running position
    for (u32 b = threadIdx.x; b < p.NB; b += blockDim.x) {
        u32 v = h[b];
        if (v) h[b] = atomicAdd(&cursor[b], v);
    }
    __syncthreads();
    for (u32 k = threadIdx.x; k < p.chunk; k += blockDim.x) {
        size_t i = base + k;
        if (i >= p.n) break;
        for (u32 w = 0; w < p.W; w++) {
            int d = digits[(size_t)w * p.n + i];
            if (d != 0) {
                u32 pos = atomicAdd(&h[(w % p.WP) * p.B + msm_mag(d) - 1], 1u);
                sorted[pos] = ((w / p.WP) << p.gshift) | (u32)i | (d < 0 ? 0x80000000u : 0u);
            }
        }
    }
}

// ---- point I/O -----------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T ld_vec(const T* p) {
    static_assert(sizeof(T) % 16 == 0, "16-byte multiples");
    T r;
    const uint4* s = reinterpret_cast<const uint4*>(p);
    uint4* d = reinterpret_cast<uint4*>(&r);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 16); k++) d[k] = s[k];
    return r;
}
template <class T>
__device__ __forceinline__ void st_vec(T* p, const T& v) {
    uint4* d = reinterpret_cast<uint4*>(p);
    const uint4* s = reinterpret_cast<const uint4*>(&v);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 16); k++) d[k] = s[k];
}

// field-aware stores: registers may hold lazy [0,2p) values, memory is always canonical
template <class P>
__device__ __forceinline__ void st_vec(Fp<P>* p, const Fp<P>& v) {
    Fp<P> c = Fp<P>::canon(v);
    uint4* d = reinterpret_cast<uint4*>(p);
    const uint4* s = reinterpret_cast<const uint4*>(&c);
#pragma unroll
    for (int k = 0; k < (int)(sizeof(Fp<P>) / 16); k++) d[k] = s[k];
}
template <class P>
__device__ __forceinline__ void st_vec(Fp2<P>* p, const Fp2<P>& v) {
    st_vec(&p->c0, v.c0);
    st_vec(&p->c1, v.c1);
}
template <class F>
__device__ __forceinline__ void st_vec(Affine<F>* p, const Affine<F>& v) {
    st_vec(&p->x, v.x);
    st_vec(&p->y, v.y);
}
template <class F>
__device__ __forceinline__ void st_vec(XYZZ<F>* p, const XYZZ<F>& v) {
    st_vec(&p->x, v.x);
    st_vec(&p->y, v.y);
    st_vec(&p->zz, v.zz);
    st_vec(&p->zzz, v.zzz);
}

// largest b in [0, NB) with start[b] <= pos   (pos < start[NB])
__device__ __forceinline__ u32 msm_find_bucket(const u32* __restrict__ start, u32 NB, u32 pos) {
    u32 lo = 0, hi = NB;   // invariant: start[lo] <= pos < start[hi]
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (start[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

// occupancy the accumulate kernel is compiled for (waves per SIMD): 8-limb G1 needs 105 VGPRs (4 waves), 12-limb G1
// 145 (3 waves), 8-limb G2 207 (2 waves), 12-limb G2 spills at any occupancy (1 wave)
// the accumulate loop inlines the P == Q corner of the mixed add for every base field (G1); G2 follows the default
template <class F> struct AccumInlineCorner { static constexpr bool value = F::Params::N <= 8; };
template <class P> struct AccumInlineCorner<Fp<P>> { static constexpr bool value = true; };
#ifndef HK_ACCUM_PREFETCH
#define HK_ACCUM_PREFETCH 1            // 0 = off, 1 = 8-limb G1 only (shipped: H launch alone 3.86 -> 3.5-3.7 ms, bench
                                       // throughput unchanged), 2 = every G1, 3 = G1 and 8-limb G2 (build-time knob)
#endif
template <class F> struct AccumPrefetch { static constexpr bool value = HK_ACCUM_PREFETCH >= 3 && F::Params::N <= 8; };
template <class P> struct AccumPrefetch<Fp<P>> { static constexpr bool value = HK_ACCUM_PREFETCH >= 2 || (HK_ACCUM_PREFETCH == 1 && P::N <= 8); };
template <class F> struct AccumOcc { static constexpr int waves = 1; };
#ifndef HK_ACCUM_WAVES_G1_8LIMB
#define HK_ACCUM_WAVES_G1_8LIMB 4      // build-time experiment knob (make EXTRA=-DHK_ACCUM_WAVES_G1_8LIMB=5)
#endif
template <class P> struct AccumOcc<Fp<P>> { static constexpr int waves = P::N <= 8 ? HK_ACCUM_WAVES_G1_8LIMB : 3; };
template <class P> struct AccumOcc<Fp2<P>> { static constexpr int waves = P::N <= 8 ? 2 : 1; };   // 207 VGPRs with the asm add/sub

// ---- level 0: equal slices of the sorted entry list, mixed adds into registers ---------------------
// The sorted list was built for `n_entries` scalars per group; this base table has `n_bases` bases per
// group and its base j corresponds to scalar j + idx_off (the L-query is a suffix of the assignment:
// cp-groth16/src/prover.rs:111-117 vs :78-82).
template <class F>
__global__ void __launch_bounds__(64, AccumOcc<F>::waves)
k_msm_accum0(const Affine<F>* __restrict__ bases, u32 n_bases, u32 idx_off,
             const u32* __restrict__ sorted, const u32* __restrict__ start, MsmPlan p,
             XYZZ<F>* __restrict__ buckets, u32* __restrict__ pkeys, XYZZ<F>* __restrict__ ppts) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 E = start[p.NB];
    LevelInfo li = msm_level_info(p, E, 0);
    // No memset precedes this launch.  The reduction's ticket (one u32 behind the buckets) is cleared here; a bucket with
    // no entries is never read (the reductions see start[b] == start[b + 1]); every other bucket is written exactly once by
    // the lane in whose slice it STARTS - its sum when it also ends there, the neutral element when it continues into the
    // next slices, whose partial sums the level kernels then add to it.
    if (t == 0) *reinterpret_cast<u32*>(buckets + p.NB) = 0u;
    if (t >= li.active) return;
    u32 pos = t * li.L;
    u32 end = min(pos + li.L, E);
    u32 b = msm_find_bucket(start, p.NB, pos);
    bool head_partial = start[b] < pos;
    bool first = true;
    u32 first_key = b;
    u32 boundary = start[b + 1];
    XYZZ<F> acc = XYZZ<F>::inf();
    // boundary partials go straight to memory when they become known (keeping a second XYZZ value
    // live across the loop costs 32+ VGPRs, i.e. a wave of occupancy)
    st_vec(&ppts[2 * t], XYZZ<F>::inf());
    if constexpr (AccumPrefetch<F>::value) {
        // software pipeline, two deep: the entry of step pos + 2 and the 64-byte table row of step pos + 1 are
        // requested before the mixed add of step pos, so neither link of the dependent chain entry -> row -> add
        // stands between two adds of this wave
        u32 e_n = 0, e_nn = 0;
        bool v_n = false;
        Affine<F> P_n;
        auto request_row = [&](u32 e) {
            e_n = e;
            u32 g = (e & 0x7fffffffu) >> p.gshift;
            u32 i = e & ((1u << p.gshift) - 1u);
            v_n = i >= idx_off && i - idx_off < n_bases;
            if (v_n) P_n = ld_vec(&bases[(size_t)g * n_bases + (i - idx_off)]);     // never touches an empty table
        };
        if (pos < end) request_row(sorted[pos]);
        if (pos + 1 < end) e_nn = sorted[pos + 1];
        for (; pos < end; pos++) {
            Affine<F> P = P_n;
            u32 e = e_n;
            bool v = v_n;
            if (pos + 1 < end) request_row(e_nn);
            if (pos + 2 < end) e_nn = sorted[pos + 2];
            if (pos == boundary) {
                if (first && head_partial) st_vec(&ppts[2 * t], acc);
                else st_vec(&buckets[b], acc);
                first = false;
                acc = XYZZ<F>::inf();
                do { b++; boundary = start[b + 1]; } while (boundary <= pos);
            }
            if (v) {
                if (e >> 31) P.y = F::neg(P.y);
                acc = ec_madd<F, AccumInlineCorner<F>::value>(acc, P);
            }
        }
    } else
    for (; pos < end; pos++) {
        if (pos == boundary) {
            // bucket b ended exactly here
            if (first && head_partial) st_vec(&ppts[2 * t], acc);
            else st_vec(&buckets[b], acc);
            first = false;
            acc = XYZZ<F>::inf();
            do { b++; boundary = start[b + 1]; } while (boundary <= pos);
        }
        u32 e = sorted[pos];
        u32 g = (e & 0x7fffffffu) >> p.gshift;
        u32 i = e & ((1u << p.gshift) - 1u);
        if (i >= idx_off && i - idx_off < n_bases) {
            Affine<F> P = ld_vec(&bases[(size_t)g * n_bases + (i - idx_off)]);
            if (e >> 31) P.y = F::neg(P.y);
            acc = ec_madd<F, AccumInlineCorner<F>::value>(acc, P);
        }
    }
    bool tail_partial = end < boundary;     // bucket b continues in the next lane's slice
    bool is_tail = false;
    if (first && head_partial) st_vec(&ppts[2 * t], acc);          // single run that began before this slice
    else if (tail_partial) { is_tail = true; st_vec(&buckets[b], XYZZ<F>::inf()); }   // starts here, continues: neutral
    else st_vec(&buckets[b], acc);
    pkeys[2 * t] = first_key;
    pkeys[2 * t + 1] = b;
    st_vec(&ppts[2 * t + 1], is_tail ? acc : XYZZ<F>::inf());
}

// ---- levels >= 1: segmented reduction of boundary partials ---------------------------------------------
template <class F>
__device__ __forceinline__ void msm_accum_level(int level, u32 t, const LevelInfo& li, const u32* __restrict__ keys_in,
                                                const XYZZ<F>* __restrict__ pts_in, const MsmPlan& p,
                                                XYZZ<F>* __restrict__ buckets, u32* __restrict__ keys_out,
                                                XYZZ<F>* __restrict__ pts_out) {
    u32 pos = t * li.L;
    u32 end = min(pos + li.L, li.count);
    u32 key = keys_in[pos];
    bool head_partial = pos > 0 && keys_in[pos - 1] == key;
    bool first = true;
    u32 first_key = key;
    XYZZ<F> acc = XYZZ<F>::inf();
    XYZZ<F> head = XYZZ<F>::inf();
    for (; pos < end; pos++) {
        u32 kk = keys_in[pos];
        if (kk != key) {
            if (first && head_partial) head = acc;
            else if (!acc.is_inf()) st_vec(&buckets[key], ec_add_ni(ld_vec(&buckets[key]), acc));
            first = false;
            acc = XYZZ<F>::inf();
            key = kk;
        }
        XYZZ<F> q = ld_vec(&pts_in[pos]);
        if (!q.is_inf()) acc = ec_add(acc, q);
    }
    bool tail_partial = end < li.count && keys_in[end] == key;
    XYZZ<F> tail = XYZZ<F>::inf();
    if (first && head_partial) head = acc;
    else if (tail_partial) tail = acc;
    else if (!acc.is_inf()) st_vec(&buckets[key], ec_add_ni(ld_vec(&buckets[key]), acc));
    if (level + 1 < (int)p.n_levels) {
        keys_out[2 * t] = first_key;
        keys_out[2 * t + 1] = key;
        st_vec(&pts_out[2 * t], head);
        st_vec(&pts_out[2 * t + 1], tail);
    }
}

template <class F>
__global__ void __launch_bounds__(64)
k_msm_accum_lvl(int level, const u32* __restrict__ keys_in, const XYZZ<F>* __restrict__ pts_in,
                const u32* __restrict__ start, MsmPlan p, XYZZ<F>* __restrict__ buckets,
                u32* __restrict__ keys_out, XYZZ<F>* __restrict__ pts_out) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 E = start[p.NB];
    LevelInfo li = msm_level_info(p, E, level);
    if (t >= li.active) return;
    msm_accum_level<F>(level, t, li, keys_in, pts_in, p, buckets, keys_out, pts_out);
}

// every level from `level0` on (each at most MSM_TAIL_THREADS lanes wide) in ONE workgroup, ping-ponging the two
// partial buffers with a workgroup barrier between levels: replaces the last 3 launches of the level chain
template <class F>
__global__ void __launch_bounds__(MSM_TAIL_THREADS)
k_msm_accum_tail(int level0, u32* __restrict__ keys0, XYZZ<F>* __restrict__ pts0, u32* __restrict__ keys1,
                 XYZZ<F>* __restrict__ pts1, const u32* __restrict__ start, MsmPlan p, XYZZ<F>* __restrict__ buckets) {
    u32 t = threadIdx.x;
    u32 E = start[p.NB];
    for (int level = level0; level < (int)p.n_levels; level++) {
        LevelInfo li = msm_level_info(p, E, level);
        bool in0 = ((level - 1) & 1) == 0;           // level k reads buffer (k-1)&1 and writes buffer k&1
        if (t < li.active)
            msm_accum_level<F>(level, t, li, in0 ? keys0 : keys1, in0 ? pts0 : pts1, p, buckets, in0 ? keys1 : keys0,
                               in0 ? pts1 : pts0);
        __threadfence_block();
        __syncthreads();
    }
}

// ---- bucket reduction: lane j of window w' owns K consecutive buckets ---------------------------------
// sum_b (b+1) * bucket[b] over its range = local running sum + (j*K) * (plain sum)
template <class F>
__global__ void __launch_bounds__(64)
k_msm_bucket_reduce(const XYZZ<F>* __restrict__ buckets, const u32* __restrict__ start, MsmPlan p,
                    XYZZ<F>* __restrict__ out) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 J = p.B / p.K;
    if (t >= p.WP * J) return;
    u32 w = t / J, j = t - w * J;
    u32 b0 = w * p.B + j * p.K;
    const XYZZ<F>* bk = buckets + b0;
    XYZZ<F> run = XYZZ<F>::inf(), tot = XYZZ<F>::inf();
    for (int b = (int)p.K - 1; b >= 0; b--) {
        // a bucket without entries was never written (no memset): it is the neutral element
        XYZZ<F> q = start[b0 + b] == start[b0 + b + 1] ? XYZZ<F>::inf() : ld_vec(&bk[b]);
        run = ec_add_ni(run, q);
        tot = ec_add_ni(tot, run);
    }
    u32 wgt = j * p.K;
    if (wgt && !run.is_inf()) {
        XYZZ<F> acc = XYZZ<F>::inf();
        for (int bit = 31 - __clz(wgt); bit >= 0; bit--) {
            acc = ec_dbl_ni(acc);
            if ((wgt >> bit) & 1) acc = ec_add_ni(acc, run);
        }
        tot = ec_add_ni(tot, acc);
    }
    st_vec(&out[t], tot);
}

// WP == 1 (every table-backed MSM): bucket reduction, the sum over lanes and the final result in ONE launch.
// Every workgroup reduces its lanes' weighted bucket sums in LDS and publishes one partial; the workgroup that
// takes the last ticket adds the partials.  `ticket` sits right behind the buckets and is cleared by k_msm_accum0.
constexpr int MSM_REDUCE_THREADS = 128;
template <class F>
__global__ void __launch_bounds__(MSM_REDUCE_THREADS)
k_msm_reduce_fused(const XYZZ<F>* __restrict__ buckets, const u32* __restrict__ start, MsmPlan p,
                   XYZZ<F>* __restrict__ partial, u32* __restrict__ ticket, XYZZ<F>* __restrict__ res) {
    __shared__ XYZZ<F> sh[MSM_REDUCE_THREADS];
    __shared__ u32 last;
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    u32 J = p.B / p.K;
    XYZZ<F> tot = XYZZ<F>::inf();
    if (t < J) {
        const XYZZ<F>* bk = buckets + (size_t)t * p.K;
        const u32* st = start + (size_t)t * p.K;
        XYZZ<F> run = XYZZ<F>::inf();
        for (int b = (int)p.K - 1; b >= 0; b--) {
            XYZZ<F> q = st[b] == st[b + 1] ? XYZZ<F>::inf() : ld_vec(&bk[b]);      // empty bucket: never written
            run = ec_add_ni(run, q);
            tot = ec_add_ni(tot, run);
        }
        u32 wgt = t * p.K;
        if (wgt && !run.is_inf()) {
            XYZZ<F> acc = XYZZ<F>::inf();
            for (int bit = 31 - __clz(wgt); bit >= 0; bit--) {
                acc = ec_dbl_ni(acc);
                if ((wgt >> bit) & 1) acc = ec_add_ni(acc, run);
            }
            tot = ec_add_ni(tot, acc);
        }
    }
    sh[threadIdx.x] = tot;
    __syncthreads();
    for (u32 off = MSM_REDUCE_THREADS / 2; off >= 1; off >>= 1) {
        if (threadIdx.x < off) {
            // operands stay in LDS (the out-of-line add takes references): two private copies of 384 B each less per
            // lane for the 12-limb G2 flavour, i.e. a smaller scratch ring on every queue that runs this kernel
            XYZZ<F> t = ec_add_ni(sh[threadIdx.x], sh[threadIdx.x + off]);
            sh[threadIdx.x] = t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        st_vec(&partial[blockIdx.x], sh[0]);
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    XYZZ<F> acc = XYZZ<F>::inf();
    for (u32 j = threadIdx.x; j < gridDim.x; j += MSM_REDUCE_THREADS) acc = ec_add_ni(acc, ld_vec(&partial[j]));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (u32 off = MSM_REDUCE_THREADS / 2; off >= 1; off >>= 1) {
        if (threadIdx.x < off) {
            // operands stay in LDS (the out-of-line add takes references): two private copies of 384 B each less per
            // lane for the 12-limb G2 flavour, i.e. a smaller scratch ring on every queue that runs this kernel
            XYZZ<F> t = ec_add_ni(sh[threadIdx.x], sh[threadIdx.x + off]);
            sh[threadIdx.x] = t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) st_vec(res, sh[0]);
}

// one workgroup per window: LDS tree over the J lane results
template <class F>
__global__ void __launch_bounds__(MSM_WSUM_THREADS)
k_msm_window_sum(const XYZZ<F>* __restrict__ in, MsmPlan p, XYZZ<F>* __restrict__ wsum) {
    __shared__ XYZZ<F> sh[MSM_WSUM_THREADS];
    u32 J = p.B / p.K;
    u32 w = blockIdx.x;
    XYZZ<F> acc = XYZZ<F>::inf();
    for (u32 j = threadIdx.x; j < J; j += MSM_WSUM_THREADS) acc = ec_add_ni(acc, ld_vec(&in[(size_t)w * J + j]));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (u32 off = MSM_WSUM_THREADS / 2; off >= 1; off >>= 1) {
        if (threadIdx.x < off) {
            // operands stay in LDS (the out-of-line add takes references): two private copies of 384 B each less per
            // lane for the 12-limb G2 flavour, i.e. a smaller scratch ring on every queue that runs this kernel
            XYZZ<F> t = ec_add_ni(sh[threadIdx.x], sh[threadIdx.x + off]);
            sh[threadIdx.x] = t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) st_vec(&wsum[w], sh[0]);
}

// Horner over the WP window sums of one MSM: res = sum_w 2^(c w) S_w  (XYZZ out)
template <class F>
__global__ void k_msm_final(const XYZZ<F>* __restrict__ wsum, MsmPlan p, XYZZ<F>* __restrict__ res) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    XYZZ<F> acc = ld_vec(&wsum[p.WP - 1]);
    for (int w = (int)p.WP - 2; w >= 0; w--) {
        for (u32 k = 0; k < p.c; k++) acc = ec_dbl_ni(acc);
        acc = ec_add_ni(acc, ld_vec(&wsum[w]));
    }
    st_vec(res, acc);
}

template <class F>
__global__ void k_to_affine(const XYZZ<F>* __restrict__ in, Affine<F>* __restrict__ out, u32 n) {
    u32 t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    st_vec(&out[t], ec_to_affine(ld_vec(&in[t])));
}

// ---- shift tables: table[g][i] = 2^(c*WP*g) * base[i], affine, g = 0..F-1 ----------------------------
// One lane per base walks the groups; every step is c*WP doublings followed by one inversion.
template <class F>
__global__ void __launch_bounds__(64)
k_msm_build_tables(Affine<F>* __restrict__ table, u32 n, u32 F_groups, u32 shift_bits) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> cur = ld_vec(&table[i]);
    for (u32 g = 1; g < F_groups; g++) {
        XYZZ<F> x = XYZZ<F>::from_affine(cur);
        for (u32 k = 0; k < shift_bits; k++) x = ec_dbl_ni(x);
        cur = ec_to_affine(x);
        st_vec(&table[(size_t)g * n + i], cur);
    }
}

#endif  // __HIPCC__

// ---- host-side planning -----------------------------------------------------------------------------
// (re)derive the accumulate schedule of a plan for a kernel flavour that keeps `max_lanes0` lanes resident
// (lanes = waves/SIMD the kernel is compiled for x 1024 SIMDs x 64): one full round of equal slices
inline void msm_set_lanes(MsmPlan& p, u32 max_lanes0) {
    static const u32 lmin_env = [] { const char* e = getenv("HK_MSM_LMIN0"); return e && atoi(e) > 0 ? (u32)atoi(e) : 32u; }();
    p.Lmin0 = lmin_env;
    u64 emax = (u64)p.n * p.W;
    u64 t0 = (emax + p.Lmin0 - 1) / p.Lmin0;
    if (t0 > max_lanes0) t0 = max_lanes0;
    if (t0 == 0) t0 = 1;
    p.T[0] = (u32)t0;
    int k = 0;
    while (p.T[k] > 1 && k + 1 < MSM_MAX_LEVELS) {
        u64 cnt = 2ull * p.T[k];
        p.T[k + 1] = (u32)((cnt + MSM_LVL_L - 1) / MSM_LVL_L);
        k++;
    }
    p.n_levels = k + 1;
}

// Number of signed c-bit windows: the digits of s are read from s' = s + kconst_W (kconst_W = sum_{w<W} 2^(cw+c-1)), so W
// windows suffice iff (r - 1) + kconst_W < 2^(cW).  ceil((bits + 2) / c) always does; one window fewer does whenever the
// modulus leaves the room - BLS12-381's r = 0.906 * 2^255 at c = 16: 16 windows, not 17 (the 17th digit was always zero:
// no adds, but a 17th shift table per key and a 17th digit in every sort pass).
inline u32 msm_num_windows(u32 fr_bits, u32 c, const u32* mod, int n_limbs) {
    u32 W = (fr_bits + 2 + c - 1) / c;
    if (!mod || W < 2 || n_limbs > 8) return W;
    u32 Wt = W - 1;
    if (c * Wt < fr_bits || c * Wt > 288) return W;
    u32 sum[10] = {0}, kc[10] = {0};
    for (u32 w = 0; w < Wt; w++) {
        u32 bit = c * w + c - 1;
        kc[bit >> 5] |= 1u << (bit & 31);
    }
    u64 carry = 0;
    for (int i = 0; i < 10; i++) {                                  // sum = r + kconst_Wt  (= (r - 1) + kconst + 1)
        carry += (u64)(i < n_limbs ? mod[i] : 0) + kc[i];
        sum[i] = (u32)carry;
        carry >>= 32;
    }
    // sum <= 2^(c Wt)  <=>  every bit above c*Wt is clear, or sum == 2^(c Wt) exactly
    u32 top = c * Wt;
    bool above = false, below = false;
    for (u32 b = 0; b < 320; b++) {
        bool set = (sum[b >> 5] >> (b & 31)) & 1;
        if (b > top && set) above = true;
        if (b < top && set) below = true;
    }
    bool at = (sum[top >> 5] >> (top & 31)) & 1;
    bool ok = !above && (!at || !below);
    return ok ? Wt : W;
}

inline MsmPlan msm_make_plan(u32 n, u32 fr_bits, u32 c, u32 WP, u32 max_lanes0, const u32* mod = nullptr, int n_limbs = 0) {
    MsmPlan p;
    p.n = n;
    p.gshift = 1;
    while (((u64)1 << p.gshift) < n) p.gshift++;
    p.c = c;
    p.B = 1u << (c - 1);
    p.W = msm_num_windows(fr_bits, c, mod, n_limbs);
    if (WP > p.W) WP = p.W;
    p.WP = WP;
    p.F = (p.W + WP - 1) / WP;
    p.NB = WP * p.B;
    msm_set_lanes(p, max_lanes0);
    // sort workgroups: enough of them to fill 256 CUs, chunks of at least 1024 scalars
    u32 chunk = (n + 511) / 512;     // (n/128 measured slower: the scatter needs >= 2 workgroups per CU)
    if (chunk < 1024) chunk = 1024;
    p.chunk = chunk;
    p.K = p.B >= 8 ? 8 : p.B;
    for (int i = 0; i < 10; i++) p.kconst[i] = 0;
    for (u32 w = 0; w < p.W; w++) {
        u32 bit = c * w + c - 1;
        p.kconst[bit >> 5] |= 1u << (bit & 31);
    }
    return p;
}

}  // namespace hk
