// cp_groth16.hpp — C++ host-side mirror of the reference's cp-groth16 / worker surface over the C ABI.
//
// The reference is Rust; no Rust toolchain exists in the build image, so the host side above the
// C ABI is written in C++ with the reference's names, argument meaning and error behaviour:
//   ProvingKey / CommitterKey / Proof            cp-groth16/src/data_structures.rs:7-16,66-114
//   MultiStageConstraintSystem                   cp-groth16/src/constraint_synthesizer.rs:14-117
//   MultiStageConstraintSynthesizer              cp-groth16/src/constraint_synthesizer.rs:119-134
//   CommitmentBuilder::{new,commit,prove}        cp-groth16/src/committer.rs:38-123
//   CPGroth16::prove_last_stage                  cp-groth16/src/prover.rs:53-156
//   Stage{0,1}Response, process_stage{0,1}_*     distributed-prover/src/worker.rs:20-195
// Header-only; all arithmetic is in libhekaton (HIP).  Field elements are opaque Montgomery byte
// strings on this side (the Rust side holds ark `Fp` values with the same bytes).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/hekaton.h"

namespace hekaton {

struct Error : std::runtime_error {
    hk_status status;
    Error(hk_status s, const std::string& what) : std::runtime_error(what + ": " + hk_status_str(s)), status(s) {}
};
inline void check(hk_status s, const char* what) { if (s != HK_OK) throw Error(s, what); }

typedef std::vector<uint8_t> Bytes;

struct Sizes { size_t fr, fq, g1, g2; };

class Context {
public:
    Context(hk_curve curve, int device) : curve_(curve) {
        check(hk_ctx_create(curve, device, &ctx_), "hk_ctx_create");
        check(hk_ctx_sizes(ctx_, &sz_.fr, &sz_.fq, &sz_.g1, &sz_.g2), "hk_ctx_sizes");
    }
    ~Context() { hk_ctx_destroy(ctx_); }
    Context(const Context&) = delete;
    hk_ctx* raw() const { return ctx_; }
    const Sizes& sizes() const { return sz_; }
private:
    hk_curve curve_;
    hk_ctx* ctx_ = nullptr;
    Sizes sz_;
};

struct Csr { std::vector<uint64_t> row_ptr; std::vector<uint32_t> col; Bytes val_mont; };

// data_structures.rs:66-83 (+ the class's constraint matrices, uploaded with the key)
struct ProvingKey {
    Bytes alpha_g, beta_g, beta_h, last_delta_h;      // vk.alpha_g, beta_g, vk.beta_h, vk.deltas_h.last()
    Bytes a_g, b_g, b_h, h_g, deltas_g;
    std::vector<Bytes> ck_deltas_abc_g;               // ck.deltas_abc_g[stage]
    Csr A, B, C;
    size_t n_inst = 0, n_constraints = 0;
    hk_pk* device = nullptr;

    void upload(const Context& ctx) {
        if (device) return;
        const Sizes& z = ctx.sizes();
        std::vector<const void*> ckp;
        std::vector<size_t> ckl;
        for (auto& c : ck_deltas_abc_g) { ckp.push_back(c.data()); ckl.push_back(c.size() / z.g1); }
        hk_csr a{A.row_ptr.data(), A.col.data(), A.val_mont.data(), A.row_ptr.size() - 1, A.col.size()};
        hk_csr b{B.row_ptr.data(), B.col.data(), B.val_mont.data(), B.row_ptr.size() - 1, B.col.size()};
        hk_csr c{C.row_ptr.data(), C.col.data(), C.val_mont.data(), C.row_ptr.size() - 1, C.col.size()};
        hk_pk_desc d{};
        d.a_g = a_g.data(); d.a_len = a_g.size() / z.g1;
        d.b_g = b_g.data(); d.b_g_len = b_g.size() / z.g1;
        d.b_h = b_h.data(); d.b_h_len = b_h.size() / z.g2;
        d.h_g = h_g.data(); d.h_len = h_g.size() / z.g1;
        d.ck_stage = ckp.data(); d.ck_len = ckl.data(); d.n_stages = ckp.size();
        d.deltas_g = deltas_g.data(); d.last_delta_h = last_delta_h.data();
        d.alpha_g = alpha_g.data(); d.beta_g = beta_g.data(); d.beta_h = beta_h.data();
        d.A = &a; d.B = &b; d.C = &c; d.n_inst = n_inst; d.n_constraints = n_constraints;
        check(hk_pk_upload(ctx.raw(), &d, &device), "hk_pk_upload");
    }
    ~ProvingKey() { if (device) hk_pk_free(device); }
};

struct Proof { Bytes a, b, c; std::vector<Bytes> ds; };          // data_structures.rs:7-16

// constraint_synthesizer.rs:14-117 — assignments only (matrices are static per class and live with the key)
struct MultiStageConstraintSystem {
    size_t fr_bytes;
    Bytes instance_assignment, witness_assignment;      // Montgomery Fr, concatenated
    std::vector<std::pair<size_t, size_t>> variable_range_for_stage;
    explicit MultiStageConstraintSystem(size_t fr, const Bytes& one_mont) : fr_bytes(fr), instance_assignment(one_mont) {}
    void initialize_stage() { size_t s = num_witness_variables(); variable_range_for_stage.push_back({s, s}); }
    void finalize_stage() { variable_range_for_stage.back().second = num_witness_variables(); }
    void new_input_variable(const uint8_t* v) { instance_assignment.insert(instance_assignment.end(), v, v + fr_bytes); }
    void new_witness_variable(const uint8_t* v) { witness_assignment.insert(witness_assignment.end(), v, v + fr_bytes); }
    size_t num_instance_variables() const { return instance_assignment.size() / fr_bytes; }
    size_t num_witness_variables() const { return witness_assignment.size() / fr_bytes; }
    Bytes current_stage_witness_assignment() const {                    // :96-99
        auto r = variable_range_for_stage.back();
        return Bytes(witness_assignment.begin() + r.first * fr_bytes, witness_assignment.begin() + r.second * fr_bytes);
    }
    Bytes full_assignment() const {                                     // :102-106
        Bytes z = instance_assignment;
        z.insert(z.end(), witness_assignment.begin(), witness_assignment.end());
        return z;
    }
};

struct MultiStageConstraintSynthesizer {                               // constraint_synthesizer.rs:119-134
    virtual ~MultiStageConstraintSynthesizer() {}
    virtual size_t total_num_stages() const = 0;
    size_t last_stage() const { return total_num_stages() - 1; }
    virtual void generate_constraints(size_t stage, MultiStageConstraintSystem& cs) = 0;
};

// draws one Montgomery Fr (the reference's `E::ScalarField::rand(rng)`)
typedef std::function<Bytes()> FrRng;

struct CPGroth16 {                                                     // prover.rs:53-156
    static Proof prove_last_stage(const Context& ctx, MultiStageConstraintSystem& cs,
                                  MultiStageConstraintSynthesizer& circuit, ProvingKey& pk, const Bytes& r,
                                  const Bytes& s, const std::vector<Bytes>& comm_rands) {
        circuit.generate_constraints(circuit.last_stage(), cs);        // prover.rs:70
        Bytes z = cs.full_assignment();
        Bytes kap;
        for (auto& k : comm_rands) kap.insert(kap.end(), k.begin(), k.end());
        const Sizes& sz = ctx.sizes();
        Proof p;
        p.a.resize(sz.g1); p.b.resize(sz.g2); p.c.resize(sz.g1);
        check(hk_prove(ctx.raw(), pk.device, z.data(), z.size() / sz.fr, r.data(), s.data(),
                       kap.empty() ? nullptr : kap.data(), comm_rands.size(), p.a.data(), p.b.data(), p.c.data()),
              "hk_prove");
        return p;
    }
};

class CommitmentBuilder {                                              // committer.rs:17-123
public:
    MultiStageConstraintSystem cs;
    MultiStageConstraintSynthesizer& circuit;
    CommitmentBuilder(const Context& ctx, MultiStageConstraintSynthesizer& c, ProvingKey& pk, const Bytes& one_mont)
        : cs(ctx.sizes().fr, one_mont), circuit(c), ctx_(ctx), pk_(pk) {
        if (!pk.device) throw std::logic_error("proving key not resident: call ProvingKey::upload (no CPU path)");
    }
    // committer.rs:55-98; returns (commitment, randomness) — randomness is the FIRST draw of rng
    std::pair<Bytes, Bytes> commit(const FrRng& rng) {
        circuit.generate_constraints(cur_stage_, cs);
        Bytes w = cs.current_stage_witness_assignment();
        if (cur_stage_ >= pk_.ck_deltas_abc_g.size()) throw std::out_of_range("no more values left in committing key");
        Bytes randomness = rng();
        Bytes com(ctx_.sizes().g1);
        check(hk_commit(ctx_.raw(), pk_.device, cur_stage_, w.empty() ? nullptr : w.data(), w.size() / ctx_.sizes().fr,
                        randomness.data(), com.data()), "hk_commit");   // HK_ERR_LEN <=> committer.rs:83 assert
        cur_stage_++;
        return {com, randomness};
    }
    // committer.rs:100-123
    Proof prove(const std::vector<Bytes>& comms, const std::vector<Bytes>& comm_rands, const FrRng& rng) {
        if (pk_.deltas_g.size() / ctx_.sizes().g1 != comm_rands.size() + 1)
            throw std::logic_error("assert_eq!(pk.deltas_g.len(), comm_rands.len() + 1)");   // committer.rs:112
        Bytes r = rng(), s = rng();                                      // prover.rs:28-29
        Proof p = CPGroth16::prove_last_stage(ctx_, cs, circuit, pk_, r, s, comm_rands);
        p.ds = comms;
        return p;
    }
private:
    const Context& ctx_;
    ProvingKey& pk_;
    size_t cur_stage_ = 0;
};

// ---- worker protocol (distributed-prover/src/worker.rs) -----------------------------------------------
struct Stage0Response { uint64_t subcircuit_idx; Bytes com; std::array<uint8_t, 32> com_seed; };   // :20-25
struct Stage1Response { uint64_t subcircuit_idx; Proof proof; };                                   // :49-52

}  // namespace hekaton
