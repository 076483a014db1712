/*
 * hekaton.h — C ABI of the MI355X-native Hekaton subcircuit prover (libhekaton.so).
 *
 * This is the drop-in boundary for ONE hot path of zhaowenlan1779/hekaton-system:
 * the per-subcircuit CP-Groth16 commit + prove step.  The reference has no FFI
 * (SURVEY.md F5); the entry points below are what a Rust shim binds to replace
 * the arkworks call sites of that path (the binding is shown in INTEGRATION.md).
 * Every entry point cites the reference interface it replaces (paths relative to
 * the reference repository root).
 *
 * Conventions
 *   - Field elements are little-endian limb arrays in MONTGOMERY form with
 *     R = 2^(64*N), exactly the in-memory form of ark-ff `Fp<MontBackend<_,N>>`
 *     (N = 4 for BN254 Fr/Fq and BLS12-381 Fr; N = 6 for BLS12-381 Fq), unless a
 *     parameter says "canonical" (ark `BigInt<N>`: plain integer, LE limbs).
 *   - G1 affine = x || y ; G2 affine = x.c0 || x.c1 || y.c0 || y.c1 ; the point at
 *     infinity is encoded as all-zero coordinates ((0,0) is on neither curve).
 *     Rust `Affine<P>{x,y,infinity}` is not repr(C): the shim repacks once at
 *     key-load time.
 *   - Pointers marked [h|d] may be host or device memory (detected with
 *     hipPointerGetAttributes); [h] must be host, [d] must be device.
 *   - All calls are thread-safe on a shared hk_ctx: each call runs on a private
 *     lane (HIP stream + scratch arena), mirroring `compute_responses`
 *     (mpi-snark/src/bin/node.rs:745-795) which proves from several OS threads.
 *   - There is NO CPU backend: hk_ctx_create fails with HK_ERR_DEVICE when no
 *     gfx950 device is usable.  Randomness (r, s, kappa) always comes from the
 *     caller (prover.rs:28-29, committer.rs:85); the library is deterministic.
 */
#ifndef HEKATON_H
#define HEKATON_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hk_ctx hk_ctx;   /* one per (process, device): lanes, twiddles, scratch */
typedef struct hk_pk hk_pk;     /* device-resident proving-key class (+ its R1CS matrices) */

typedef enum { HK_BN254 = 0, HK_BLS12_381 = 1 } hk_curve;

typedef enum {
    HK_OK = 0,
    HK_ERR_LEN = 1,               /* ark `msm` Err(min_len) on length mismatch   */
    HK_ERR_DOMAIN_TOO_LARGE = 2,  /* SynthesisError::PolynomialDegreeTooLarge                 */
    HK_ERR_DEVICE = 3,            /* no device / HIP runtime error                            */
    HK_ERR_ARG = 4,
    HK_ERR_NOMEM = 5
} hk_status;

/* One R1CS matrix in CSR form; mirrors one of ark_relations `ConstraintMatrices::{a,b,c}`
 * (Vec<Vec<(F, usize)>>): row i holds (val, col) pairs, col indexes instance||witness. */
typedef struct {
    const uint64_t* row_ptr;   /* [n_rows + 1]                 [h|d] */
    const uint32_t* col;       /* [nnz]                        [h|d] */
    const void*     val_mont;  /* [nnz] Fr, Montgomery         [h|d] */
    size_t n_rows;
    size_t nnz;
} hk_csr;

/* Everything `ProvingKey<E>` (cp-groth16/src/data_structures.rs:66-83) holds that the
 * prover touches, plus the circuit class's constraint matrices (identical for every
 * subcircuit of a class, so uploaded once with the key). All arrays [h|d], packed affine. */
typedef struct {
    const void* a_g;  size_t a_len;        /* pk.a_g   : n_v G1  (data_structures.rs:72) */
    const void* b_g;  size_t b_g_len;      /* pk.b_g   : n_v G1  (:74)                   */
    const void* b_h;  size_t b_h_len;      /* pk.b_h   : n_v G2  (:76)                   */
    const void* h_g;  size_t h_len;        /* pk.h_g   : m-1 G1  (:78)                   */
    const void* const* ck_stage;           /* pk.ck.deltas_abc_g[stage] (:113)           */
    const size_t* ck_len;
    size_t n_stages;
    const void* deltas_g;                  /* pk.deltas_g : n_stages G1 (:82)            */
    const void* last_delta_h;              /* pk.vk.deltas_h.last() : 1 G2 (:95-97)      */
    const void* alpha_g;                   /* pk.vk.alpha_g : 1 G1 (:36)                 */
    const void* beta_g;                    /* pk.beta_g : 1 G1 (:70)                     */
    const void* beta_h;                    /* pk.vk.beta_h : 1 G2 (:38)                  */
    const hk_csr* A; const hk_csr* B; const hk_csr* C;   /* may be NULL: then hk_prove is unavailable */
    size_t n_inst;                         /* cs.num_instance_variables()                */
    size_t n_constraints;                  /* cs.num_constraints()                       */
} hk_pk_desc;

/* Per-phase device timings of the last hk_prove / hk_commit on the calling thread's lane,
 * measured with HIP events on the lane's stream (milliseconds). Replaces the reference's
 * start_timer!/end_timer! brackets (cp-groth16/src/prover.rs:65-150). */
typedef struct {
    float total_ms;
    float digits_ms;       /* scalar from-Montgomery + signed-digit split + bucket sort   */
    float msm_a_ms;        /* "Compute A"        prover.rs:85-90   */
    float msm_b_g1_ms;     /* "Compute B in G1"  prover.rs:95-100  */
    float msm_b_g2_ms;     /* "Compute B in G2"  prover.rs:105-108 */
    float msm_l_ms;        /* "Compute L"        prover.rs:113-118 */
    float witness_map_ms;  /* "R1CS to QAP witness map" prover.rs:122-125 */
    float msm_h_ms;        /* "Compute H"        prover.rs:127-130 */
    float finish_ms;       /* "Finish C" + into_affine prover.rs:135-155, committer.rs:112-114 */
    float accum_kernel_ms; /* sum over this call's k_msm_accum0<Fq> launches (dominant kernel),  */
    uint32_t accum_kernel_launches;  /* timed on the kernel itself; and how many launches that was */
    float accum_h_ms;      /* the H-query launch alone (the dense one)                             */
} hk_timings;

const char* hk_status_str(hk_status s);
const char* hk_version(void);

/* ---- context ------------------------------------------------------------------------ */
hk_status hk_ctx_create(hk_curve curve, int device_id, hk_ctx** out);
void      hk_ctx_destroy(hk_ctx* ctx);
hk_status hk_ctx_sync(hk_ctx* ctx);                      /* drain every lane */
hk_status hk_ctx_set_profiling(hk_ctx* ctx, int enable); /* record hk_timings per call */
hk_status hk_ctx_last_timings(hk_ctx* ctx, hk_timings* out);
/* element sizes for this curve: Fr, Fq, G1 affine, G2 affine bytes */
hk_status hk_ctx_sizes(const hk_ctx* ctx, size_t* fr, size_t* fq, size_t* g1, size_t* g2);

/* device-memory plumbing so a host (Rust shim, ctypes) can keep inputs resident */
hk_status hk_dev_alloc(hk_ctx* ctx, size_t bytes, void** dptr);
hk_status hk_dev_free(hk_ctx* ctx, void* dptr);
hk_status hk_dev_upload(hk_ctx* ctx, void* dst_d, const void* src_h, size_t bytes);
hk_status hk_dev_download(hk_ctx* ctx, void* dst_h, const void* src_d, size_t bytes);

/* ---- primitives: one per arkworks call the hot path makes ----------------------------- */

/* VariableBaseMSM for G1 — replaces `G::Group::msm_bigint(&query[1..], assignment)`
 * (cp-groth16/src/prover.rs:167, scalars canonical BigInt) and `E::G1::msm(bases, scalars)`
 * (prover.rs:117,129; committer.rs:89, scalars Montgomery Fr) and `msm_unchecked`
 * (committer.rs:113).  checked != 0: HK_ERR_LEN when n_bases != n_scalars (ark `msm`);
 * checked == 0: zip to min(n_bases, n_scalars) (ark `msm_unchecked` / `msm_bigint`).
 * out_affine [h]: packed affine sum (infinity = zeros). */
hk_status hk_msm_g1(hk_ctx* ctx, const void* bases, size_t n_bases,
                    const void* scalars, size_t n_scalars,
                    int scalars_are_montgomery, int checked, void* out_affine);
/* Same for G2 — replaces prover.rs:107 (`calculate_coeff` over pk.b_h). */
hk_status hk_msm_g2(hk_ctx* ctx, const void* bases, size_t n_bases,
                    const void* scalars, size_t n_scalars,
                    int scalars_are_montgomery, int checked, void* out_affine);

/* ark-poly Radix2EvaluationDomain {fft,ifft}_in_place and the coset forms with shift
 * F::GENERATOR.  data [h|d]: 2^log_m Fr (Montgomery), natural order in
 * and out.  HK_ERR_DOMAIN_TOO_LARGE when log_m > TWO_ADICITY. */
hk_status hk_ntt(hk_ctx* ctx, void* data, unsigned log_m, int inverse, int coset);

/* R1CSToQAP::witness_map (LibsnarkReduction) — replaces `cs.map(QAP::witness_map::<_, D<_>>)`
 * at cp-groth16/src/prover.rs:123.  z_mont: full assignment instance||witness (n_v Fr).
 * h_out [h|d]: m Fr (Montgomery), natural order; *m_out = domain size. */
hk_status hk_witness_map(hk_ctx* ctx, const hk_csr* A, const hk_csr* B, const hk_csr* C,
                         size_t n_inst, size_t n_constraints,
                         const void* z_mont, size_t n_v,
                         void* h_out, size_t h_capacity, size_t* m_out);

/* Fixed-base batch scalar multiplication: out[i] = scalars[i] * base, normalised to affine.
 * Replaces `FixedBase::msm` + `normalize_batch` of the trusted setup (cp-groth16/src/generator.rs:
 * 134-224).  base [h|d]: one affine point; scalars [h|d]: n Fr;
 * out [h|d]: n packed affine points.  The base's window table (`FixedBase::get_window_table`: a 248-step doubling chain,
 * 2 ms in G1 and 5 - 6 ms in G2) is kept per context for the first 8 distinct HOST bases and reused by later calls
 * (HK_FB_NO_CACHE=1: rebuilt every call); results do not depend on it. */
hk_status hk_fixed_base_g1(hk_ctx* ctx, const void* base, const void* scalars, size_t n,
                           int scalars_are_montgomery, void* out);
hk_status hk_fixed_base_g2(hk_ctx* ctx, const void* base, const void* scalars, size_t n,
                           int scalars_are_montgomery, void* out);

/* N independent scalar multiplications out[i] = scalars[i] * points[i], batch-normalised to affine —
 * replaces `scalar_pairing` (distributed-prover/src/pairing_ops.rs:32-39: `*si * *ri` + `normalize_batch`),
 * called 8x per aggregation with N = #subcircuits (aggregation.rs:236-242,289-310).
 * points [h|d]: n packed affine; scalars [h|d]: n Fr (Montgomery); out [h|d]: n packed affine. */
hk_status hk_scalar_pairing_g1(hk_ctx* ctx, const void* points, const void* scalars, size_t n, void* out);
hk_status hk_scalar_pairing_g2(hk_ctx* ctx, const void* points, const void* scalars, size_t n, void* out);

/* ---- pairings of the aggregation path -------------------------------------------------------
 * GT elements cross the ABI as ark's `Fp12`: c0.c0.c0, c0.c0.c1, c0.c1.c0, ... c1.c2.c1 - 12 Fq, Montgomery
 * (hk_ctx_gt_bytes: 384 B on BN254, 576 B on BLS12-381).
 * hk_multi_pairing: prod_i e(g1[i], g2[i]) = `E::final_exponentiation(E::multi_miller_loop(left, right))` - replaces
 * `pairing(left, right)` (distributed-prover/src/pairing_ops.rs:9-29; pairs with an infinity member contribute 1,
 * n = 0 gives 1).  g1 [h|d]: n packed G1 affine; g2 [h|d]: n packed G2 affine; gt_out [h|d].
 * hk_pairing_products: every lhs vector against every rhs vector in ONE batched launch,
 * gt_out[a * n_rhs + b] = pairing(lhs_g1[a], rhs_g2[b]) - replaces the 4 x 4 `cross_terms` of
 * distributed-prover/src/aggregation.rs:255-263 and, with n_lhs = n_rhs = 1, the IPP commitments'
 * inner products (aggregation.rs:97-103,167-168).  All vectors have n elements. */
hk_status hk_multi_pairing(hk_ctx* ctx, const void* g1, const void* g2, size_t n, void* gt_out);
hk_status hk_pairing_products(hk_ctx* ctx, const void* const* lhs_g1, size_t n_lhs, const void* const* rhs_g2,
                              size_t n_rhs, size_t n, void* gt_out);
/* hk_pairing_pairs: the same batched launch for a LIST of (lhs vector, rhs vector) pairs instead of the full grid:
 * gt_out[p] = pairing(lhs_g1[pair_lhs[p]], rhs_g2[pair_rhs[p]]), p < n_pairs <= 64.  One GIPA round of the TIPA prover
 * (ark-ip-proofs `gipa` under distributed-prover/src/aggregation.rs:340) needs ten inner products between six G1 and six
 * G2 half-vectors - e(A_R, v1_L), e(w1_R, B_L), ... - each rhs vector's Miller lines are computed once and shared by the
 * pairs that use it.  pair_lhs, pair_rhs [h]: n_pairs indices into lhs_g1 / rhs_g2 (at most 64 pairs).  TWO rounds fit one
 * call: round k + 1's messages are inner products of the folded vectors, by bilinearity products of quarter-by-quarter
 * inner products of the current vectors raised to 1, c, 1 / c - sixty pairs out of twelve G1 and twelve G2 quarter
 * vectors, then hk_gt_pow_prod (hekaton_system_amd/tipa.py `_round_pair`). */
hk_status hk_pairing_pairs(hk_ctx* ctx, const void* const* lhs_g1, size_t n_lhs, const void* const* rhs_g2, size_t n_rhs,
                           const uint32_t* pair_lhs, const uint32_t* pair_rhs, size_t n_pairs, size_t n, void* gt_out);
hk_status hk_ctx_gt_bytes(const hk_ctx* ctx, size_t* gt);
/* gt_out[i] = gt_in[i]^scalars[i] in GT - `Commitment * scalar` (distributed-prover/src/aggregation.rs:171-174,328-332) and
 * the six GT powers per round of the TIPA verifier; one wavefront per element.  gt_in, gt_out [h|d]: n GT elements;
 * scalars_mont [h|d]: n Fr.  The inputs must lie IN GT (order r: pairing values and their products - what `PairingOutput`
 * holds): the exponent is split along the Frobenius, z^c = prod_j pi^j(z)^(k_j) with four parts of <= 67 bits, which is
 * z^c only there (hk_fq12_pow below takes any element; HK_GT_POW_PLAIN in the environment routes hk_gt_pow to it). */
hk_status hk_gt_pow(hk_ctx* ctx, const void* gt_in, const void* scalars_mont, size_t n, void* gt_out);
/* The same power for ANY Fq12 elements (the plain 254-step square-and-multiply chain): what a verifier uses on values it
 * has not produced itself - the GT members of a TIPA proof (ark's `PairingOutput` deserialises without a subgroup check). */
hk_status hk_fq12_pow(hk_ctx* ctx, const void* fq12_in, const void* scalars_mont, size_t n, void* fq12_out);
/* Grouped multi-exponentiation: gt_out[g] = prod_{j < group_len} gt_in[g * group_len + j]^scalars[g * group_len + j], n a
 * multiple of group_len, n / group_len <= 65535 groups.  The fold check of the TIPA verifier (ark-ip-proofs `gipa` verify
 * under distributed-prover/src/aggregation.rs:340): T' = T * prod_k TL_k^(c_k) TR_k^(1/c_k), likewise U and Z - three groups
 * of 2 log2(N) powers; the powers run one wavefront per element as in hk_gt_pow / hk_fq12_pow (in_gt != 0: the Frobenius
 * split, elements of GT only; 0: the plain chain, any Fq12 element), then one wavefront per group multiplies them up.
 * gt_in [h|d]: n elements; scalars_mont [h|d]: n Fr; gt_out [h|d]: n / group_len elements. */
hk_status hk_gt_pow_prod(hk_ctx* ctx, const void* gt_in, const void* scalars_mont, size_t n, size_t group_len, int in_gt,
                         void* gt_out);

/* Element-wise linear combination of k <= 8 point vectors: out[i] = sum_j coeffs[j] * vecs[j][i], batch-normalised to
 * affine.  Replaces the aggregator's `prepared_input = s0 + s1*x0 + s2*x1 + s3*x2` (distributed-prover/src/
 * aggregation.rs:192-203) and the `left` / `right` combinations of :293-326 (three constant-scalar `scalar_pairing`
 * sweeps + element-wise additions each) with ONE launch sharing one doubling chain.
 * vecs [h]: k pointers, each [h|d] to n packed affine points; coeffs_mont [h|d]: k Fr; out [h|d]: n packed affine. */
hk_status hk_points_lincomb_g1(hk_ctx* ctx, const void* const* vecs, const void* coeffs_mont, size_t k, size_t n, void* out);
hk_status hk_points_lincomb_g2(hk_ctx* ctx, const void* const* vecs, const void* coeffs_mont, size_t k, size_t n, void* out);
/* out[i] = lo[i] + sum_{j<4} s_j * coeffs4[j] * psi^j(hi[i]) in G2, s_j = -1 where bit j of neg_mask is set: the G2 fold
 * `lo + c * hi` of a TIPA / GIPA round (ark-ip-proofs `gipa`, called from distributed-prover/src/aggregation.rs:340) with the
 * challenge split by the caller into four ~64-bit parts along psi, the untwist-Frobenius-twist endomorphism of G2
 * (psi(Q) = [q mod r] Q: c = sum s_j coeffs4[j] lambda^j mod r with lambda = 6 x^2 on BN254, x on BLS12-381) - the shared
 * doubling chain is ~66 steps instead of 254.  lo, hi [h|d]: n G2 points; coeffs4_mont [h|d]: 4 Fr; out [h|d]: n G2. */
hk_status hk_points_fold_g2(hk_ctx* ctx, const void* lo, const void* hi, const void* coeffs4_mont, unsigned neg_mask, size_t n,
                            void* out);
/* Host utility (no device work): Keccak-f[1600] on 25 little-endian 64-bit lanes - the permutation under the merlin
 * transcripts (STROBE-128) the aggregator draws its challenges from (distributed-prover/src/util.rs:22,41-75). */
void hk_keccak_f1600(uint64_t* state25);

/* The same for G1 along the GLV endomorphism phi(x, y) = (beta x, y) (phi(P) = [lambda] P, lambda^2 + lambda + 1 = 0 mod r):
 * out[i] = lo[i] + s_0 coeffs2[0] hi[i] + s_1 coeffs2[1] phi(hi[i]), c = s_0 coeffs2[0] + s_1 coeffs2[1] lambda mod r with two
 * ~128-bit parts - the folds `A' = A_L + c A_R`, `w' = w_L + c w_R` of a round.  128 doubling steps instead of 254. */
hk_status hk_points_fold_g1(hk_ctx* ctx, const void* lo, const void* hi, const void* coeffs2_mont, unsigned neg_mask, size_t n,
                            void* out);
/* k <= 4 folds that share ONE scalar, as the folds of one GIPA round do (`A' = A_L + c A_R`, `w1' = ..`, `w2' = ..` with c;
 * `B'`, `v1'`, `v2'` with c^-1: ark-ip-proofs `gipa`, reached from distributed-prover/src/aggregation.rs:340 - there one rayon
 * sweep per vector): out[y][i] = lo[y][i] + c * hi[y][i], one launch and one normalisation for all k vectors.
 * lo, hi, out [h]: k pointers, each [h|d] to n packed affine points; coeffs / neg_mask as in hk_points_fold_g1 / _g2. */
hk_status hk_points_fold_many_g1(hk_ctx* ctx, size_t k, const void* const* lo, const void* const* hi, const void* coeffs2_mont,
                                 unsigned neg_mask, size_t n, void* const* out);
hk_status hk_points_fold_many_g2(hk_ctx* ctx, size_t k, const void* const* lo, const void* const* hi, const void* coeffs4_mont,
                                 unsigned neg_mask, size_t n, void* const* out);

/* ---- MSM over a RESIDENT base set ----------------------------------------------------------------------
 * Bases that are key material (the KZG / commitment-key powers of the aggregator's SRS, any static query) are
 * uploaded once; long sets together with their 2^(16 g) multiples, exactly like the proving-key queries, so that every later
 * MSM over them has no Horner tail (the 254 sequential doublings that bound a one-off MSM's latency).  Short sets (G1 up to
 * 8 192 bases, G2 up to 2 048) stay as they are - hk_msm_bases runs n element-wise endomorphism products and one sum over
 * them, 1.4 - 2.4 ms - because building the multiples costs 6 - 10 ms per set and the aggregator multiplies each of its sets
 * once per aggregation (HK_BASES_TABLES=1 in the environment builds them for G1 sets of any length).
 * replaces `G::Group::msm(&srs_powers_alpha, &witness_poly.coeffs)` / `..beta..` of the KZG openings
 * (distributed-prover/src/kzg.rs:151-152) and the static-key MSMs of TIPA (distributed-prover/src/aggregation.rs:337,
 * third-party ripp) once the SRS of `TIPA::setup` (aggregation.rs:60-135) is resident.
 * group: 1 = G1, 2 = G2.  hk_msm_bases: same scalar conventions and length semantics as hk_msm_g1/g2
 * (checked != 0: n_scalars must equal the number of bases, else HK_ERR_LEN; unchecked: zip to the shorter). */
typedef struct hk_bases hk_bases;
hk_status hk_bases_upload(hk_ctx* ctx, int group, const void* bases, size_t n, hk_bases** out);
void      hk_bases_free(hk_bases* b);
hk_status hk_msm_bases(hk_ctx* ctx, const hk_bases* b, const void* scalars, size_t n_scalars, int mont,
                       int checked, void* out);

/* Montgomery <-> canonical conversion of n field elements (which: 0 = Fr, 1 = Fq; to_mont != 0: out = in*R mod p,
 * else out = in/R mod p; canonical input must be < p).  in/out are host or device pointers and may alias.
 * replaces ark-ff `into_bigint()` / `from_bigint()` as ark-serialize calls them for every field element of a key
 * file or response (mpi-snark/src/bin/node.rs:231-237 `ProvingKeys::deserialize_uncompressed_unchecked`;
 * mpi-snark/src/lib.rs:68-71 `serialize_to_vec`): the wire format is canonical little-endian, the ABI Montgomery. */
hk_status hk_field_convert(hk_ctx* ctx, int which, const void* in, void* out, size_t n, int to_mont);

/* ---- witness materialisation ------------------------------------------------------------------
 * A gadget circuit's assignment is almost entirely bits (SHA-256: > 99.99 %).  The witness generator hands over ONE
 * BYTE per variable plus the few full-width values, and the assignment `cs.full_assignment()` would hold
 * (cp-groth16/src/constraint_synthesizer.rs:102-106: instance || witness, 32 B Montgomery each) is materialised in
 * HBM: z[i] = bits[i] ? 1 : 0, then z[full_cols[k]] = full_vals[k].  PCIe carries n_v bytes instead of 32 n_v.
 * bits [h|d]: n_v bytes (0 / 1; bits[0] = 1 for the constant); full_cols [h|d]: n_full column indices;
 * full_vals_mont [h|d]: n_full Fr; z_out [d]: n_v Fr, ready for hk_commit / hk_prove. */
hk_status hk_assignment_from_bits(hk_ctx* ctx, const void* bits, size_t n_v, const uint32_t* full_cols,
                                  const void* full_vals_mont, size_t n_full, void* z_out);

/* Witness generation ON the device for gadget circuits: a class's WORD PROGRAM (the dataflow of its bit gadgets at
 * 32-bit word granularity, recorded when its R1CS is built) + column map are uploaded once; hk_wprog_run then turns the
 * inputs of `batch` subcircuits (a leaf's 16 words, the 54 bytes of two child hashes) into their full Montgomery
 * assignments in HBM - replaces the witness side of `circuit.generate_constraints` (cp-groth16/src/prover.rs:70-75;
 * distributed-prover/src/tree_hash_circuit.rs:313-398) for a re-implemented gadget set (csrc/witness.cuh).
 *   ops [h]: n_ops x 8 u32 (opcode, a, b, c, imm, 0, 0, 0): 0 INPUT imm | 1 CONST imm | 2 XOR a b | 3 CH a b c | 4 AND a b |
 *            5 MAJ a b c | 6 ADD refs[a .. a+b) + imm -> TWO values (low word, carry) | 7 PACK4 refs[a .. a+4) | imm;
 *            8 SHA_ROUND refs[a .. a+9) = (a b c d e f g h w), imm = K_t -> ELEVEN values, those of the round's gadget
 *            entries in their order (rotr6^rotr11 of e, Sigma1, Ch, rotr2^rotr13 of a, Sigma0, a & b, Maj, low / carry of
 *            d + h + Sigma1 + Ch + w + K, low / carry of h + Sigma1 + Ch + w + Sigma0 + Maj + K) | 9 SHA_SCHED
 *            refs[a .. a+4) = (w[t-15] w[t-2] w[t-7] w[t-16]) -> SIX values (rotr7^rotr18, sigma0, rotr17^rotr19, sigma1,
 *            low / carry of sigma1 + w[t-7] + sigma0 + w[t-16]);
 *            an operand = value id | rotate-right << 20 | shift-right << 25; every entry defines the next value id(s)
 *   map [h]: n_v u32, value id << 5 | bit position, or 0xffffffff for instance / full-width columns
 *   hk_wprog_run: inputs [h|d] batch x n_inputs u32; full_cols [h|d] n_full columns; full_vals_mont [h|d] batch x n_full Fr;
 *            z_out [d]: batch x n_v Fr.  HK_ERR_ARG for a program that would index out of range. */
typedef struct hk_wprog hk_wprog;
hk_status hk_wprog_upload(hk_ctx* ctx, const uint32_t* ops, size_t n_ops, const uint32_t* refs, size_t n_refs,
                          const uint32_t* map, size_t n_v, size_t n_values, size_t n_inputs, hk_wprog** out);
void      hk_wprog_free(hk_wprog* w);
hk_status hk_wprog_run(hk_ctx* ctx, const hk_wprog* w, const uint32_t* inputs, size_t batch, const uint32_t* full_cols,
                       const void* full_vals_mont, size_t n_full, void* z_out);
/* The full-width values alone: z_out[b][full_cols[j]] = full_vals[b][j] for b < batch.  A subcircuit's bit columns do not
 * depend on the round's challenges, its running evaluations do (distributed-prover/src/subcircuit_circuit.rs:206-231): a
 * worker may run hk_wprog_run with n_full = 0 while the first round is still in flight and hand these in afterwards. */
hk_status hk_assignment_scatter(hk_ctx* ctx, const uint32_t* full_cols, const void* full_vals_mont, size_t n_full, size_t batch,
                                size_t n_v, void* z_out);

/* The Poseidon membership block of a subcircuit's assignment (the witness side of `verify_membership`,
 * distributed-prover/src/subcircuit_circuit.rs:233-252, with the hashes of poseidon_util.rs:26-107): for `batch`
 * subcircuits at once, the S-box chains and round states of the leaf hash (rate 3 over the 4 leaf fields) and of the
 * `depth` two-to-one hashes along the path, plus per level (bit, sibling, left input), written to columns
 * [col0, col0 + block) of each subcircuit's assignment in HBM - the order of hekaton_system_amd/sha_circuit.py
 * `poseidon_path_trace`.  hk_poseidon_desc: width t = rate + 1 (<= 4), S-box exponent (5 or 17), full / partial rounds,
 * offset (in Fr elements) of its constants inside `consts_mont` = ark[(rf + rp)][t] then mds[t][t].
 *   consts_mont [h|d]; leaf_mont [h|d] batch x 4 Fr; siblings_mont [h|d] batch x depth Fr (bottom-up: the leaf's sibling
 *   first); leaf_index [h|d] batch u32; z_out [d] batch x n_v Fr.  HK_ERR_ARG for a block that would not fit n_v. */
typedef struct { uint32_t t, alpha, full_rounds, partial_rounds, consts_offset; } hk_poseidon_desc;
hk_status hk_poseidon_path(hk_ctx* ctx, const void* consts_mont, size_t n_consts, const hk_poseidon_desc* leaf_hash,
                           const hk_poseidon_desc* node_hash, const void* leaf_mont, const void* siblings_mont,
                           const uint32_t* leaf_index, size_t depth, size_t batch, size_t n_v, size_t col0, void* z_out);

/* ---- proving-key residency -------------------------------------------------------------- */
hk_status hk_pk_upload(hk_ctx* ctx, const hk_pk_desc* desc, hk_pk** out);
void      hk_pk_free(hk_pk* pk);

/* ---- fused per-subcircuit calls (the unit of work the metric counts) --------------------- */

/* CommitmentBuilder::commit arithmetic (cp-groth16/src/committer.rs:87-91):
 * com = msm(ck[stage], w_stage) + kappa * last_delta_g.  w_stage_mont [h|d]: n Fr; HK_ERR_LEN
 * unless n == ck_len[stage] (committer.rs:83 assert). com_affine_out [h]. */
hk_status hk_commit(hk_ctx* ctx, const hk_pk* pk, size_t stage,
                    const void* w_stage_mont, size_t n, const void* kappa_mont,
                    void* com_affine_out);
/* The same for `batch` subcircuits of one proving-key class in ONE call - what a worker does for the stage-0 requests of the
 * subcircuits it holds (distributed-prover/src/worker.rs:91-146 under mpi-snark/src/bin/node.rs:500-506: one rayon task per
 * subcircuit there): com[b] = msm(ck[stage], w[b]) + kappa[b] * last_delta_g.  w_mont [h|d]: batch x n Fr, row after row;
 * kappas_mont [h]: batch Fr; coms_affine_out [h]: batch G1.  Short stages (the 16 stage-0 witnesses of a big-merkle subcircuit)
 * run as one set of launches; long ones as `batch` hk_commit calls. */
hk_status hk_commit_batch(hk_ctx* ctx, const hk_pk* pk, size_t stage, const void* w_mont, size_t n, const void* kappas_mont,
                          size_t batch, void* coms_affine_out);

/* CPGroth16::prove_last_stage (prover.rs:78-155) followed by CommitmentBuilder::prove's
 * kappa correction (committer.rs:112-114), everything after constraint synthesis:
 *   z_mont [h|d]  full assignment instance||witness, n_v Fr Montgomery, z[0] = 1
 *   r_mont,s_mont [h] the two blinders (prover.rs:28-29)
 *   kappas_mont [h]   n_kappas = n_stages-1 commitment randomizers (committer.rs:110-113)
 * Outputs [h]: proof.a (G1), proof.b (G2), proof.c (G1), packed affine. */
hk_status hk_prove(hk_ctx* ctx, const hk_pk* pk, const void* z_mont, size_t n_v,
                   const void* r_mont, const void* s_mont,
                   const void* kappas_mont, size_t n_kappas,
                   void* proof_a_g1, void* proof_b_g2, void* proof_c_g1);

#ifdef __cplusplus
}
#endif
#endif /* HEKATON_H */
