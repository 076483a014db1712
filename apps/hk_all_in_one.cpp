// hk_all_in_one — native (C++) driver of one two-round proving job over the C ABI, the shape of the reference's
// single-process harness mpi-snark/src/bin/all_in_one.rs:109-196 (and of `node work`, node.rs:478-617, for one rank):
//
//   load the proving keys of the job's classes          (node.rs:231-237; here: raw ABI arrays exported by
//                                                        tools/export_job.py, one directory per class)
//   round 1: for every subcircuit, on T worker threads  process_stage0_request_get_cb (worker.rs:91-146):
//            kappa = Fr::rand(ChaCha12Rng(com_seed)), com = hk_commit(...)            -> Stage0Response
//            (short stages: one hk_commit_batch per key class; --no-batch-commit for one call per subcircuit)
//   (the coordinator's step between the rounds is not part of the worker path)
//   round 2: for every subcircuit                       process_stage1_request_with_cb (worker.rs:150-195):
//            hk_prove(...) with the SAME kappa re-derived from com_seed (worker.rs:236-241)  -> Stage1Response
//   write every response as the reference's ark-serialize bytes: <out>/stage0_resp_<i>.bin, stage1_resp_<i>.bin
//   (util.rs:79-91 cli_filenames), print one JSON line with the job's timing
//
// Host code above the C ABI is the C++ mirror of the cp-groth16 surface (hekaton_system_amd/csrc/host/*.hpp); all
// arithmetic is in libhekaton (HIP).  Assignments are uploaded once and stay resident (the metric's definition, §④);
// `--host-inputs` sends them over PCIe per proof instead.  BN254 (what the reference instantiates) by default,
// `--curve bls12_381` for the other build of the library (zcash point format on the wire).
//
// build:  make -C apps      (g++ -O2 -std=c++17 -pthread ... -lhekaton, rpath = hekaton_system_amd/lib)
// usage:  hk_all_in_one <job_dir> <out_dir> [--threads T] [--steps K] [--warmup W] [--device D] [--host-inputs] [--curve C]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include "../hekaton_system_amd/csrc/host/ark_serialize.hpp"
#include "../hekaton_system_amd/csrc/host/cp_groth16.hpp"

using namespace hekaton;

static Bytes rd(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot read " + path);
    return Bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
template <class T> static std::vector<T> rdv(const std::string& path) {
    Bytes b = rd(path);
    std::vector<T> v(b.size() / sizeof(T));
    memcpy(v.data(), b.data(), v.size() * sizeof(T));
    return v;
}
static void wr(const std::string& path, const Bytes& b) {
    std::ofstream f(path, std::ios::binary);
    f.write((const char*)b.data(), (std::streamsize)b.size());
    if (!f) throw std::runtime_error("cannot write " + path);
}

// one assignment of a class, resident in HBM (or on the host with --host-inputs)
struct Assignment {
    Bytes host;            // full assignment z (instance || witness), Montgomery
    void* dev = nullptr;
    size_t n_v = 0;
};

struct KeyClass {
    ProvingKey pk;
    size_t n0 = 0;         // stage-0 witnesses
    std::vector<Assignment> assigns;
};

struct Subcircuit {
    uint64_t cls, assign;
    std::array<uint8_t, 32> com_seed;
    Bytes r, s;
};

template <class F> static void parallel_for(size_t n, unsigned threads, F f) {
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    std::exception_ptr err;
    std::mutex m;
    for (unsigned t = 0; t < threads; t++)
        pool.emplace_back([&]() {
            try {
                for (size_t i = next++; i < n; i = next++) f(i);
            } catch (...) {
                std::lock_guard<std::mutex> g(m);
                if (!err) err = std::current_exception();
                next = n;
            }
        });
    for (auto& th : pool) th.join();
    if (err) std::rethrow_exception(err);
}

int main(int argc, char** argv) {
    // before any thread exists and before the first HIP call: hardware queues for the concurrent proofs (the library
    // itself never writes the environment); a value the user exported wins
    setenv("GPU_MAX_HW_QUEUES", "20", 0);
    if (argc < 3) {
        fprintf(stderr, "usage: %s <job_dir> <out_dir> [--threads T] [--steps K] [--warmup W] [--device D] [--host-inputs] [--curve bn254|bls12_381]\n", argv[0]);
        return 2;
    }
    std::string job = argv[1], out = argv[2];
    unsigned threads = 8, steps = 1, warmup = 0;
    int device = 0;
    bool host_inputs = false;
    bool bls = false;
    bool batch_commit = true;
    for (int i = 3; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--threads" && i + 1 < argc) threads = (unsigned)atoi(argv[++i]);
        else if (a == "--steps" && i + 1 < argc) steps = (unsigned)atoi(argv[++i]);
        else if (a == "--warmup" && i + 1 < argc) warmup = (unsigned)atoi(argv[++i]);
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--host-inputs") host_inputs = true;
        else if (a == "--no-batch-commit") batch_commit = false;
        else if (a == "--curve" && i + 1 < argc) { std::string c = argv[++i]; bls = c == "bls12_381"; if (!bls && c != "bn254") { fprintf(stderr, "unknown curve %s\n", c.c_str()); return 2; } }
        else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    try {
        Context ctx(bls ? HK_BLS12_381 : HK_BN254, device);
        const Sizes sz = ctx.sizes();
        std::unique_ptr<ArkCodecBn254> codec_bn(bls ? nullptr : new ArkCodecBn254(ctx));
        std::unique_ptr<ArkCodecBls381> codec_bls(bls ? new ArkCodecBls381(ctx) : nullptr);
        auto kappa_of = [&](const std::array<uint8_t, 32>& seed) {
            return bls ? commitment_randomness_bls12_381(seed) : commitment_randomness_bn254(seed);
        };
        auto dims = rdv<uint64_t>(job + "/job");                  // n_subcircuits, n_classes
        size_t n_sub = dims.at(0), n_cls = dims.at(1);
        auto t_load = std::chrono::steady_clock::now();
        std::vector<std::unique_ptr<KeyClass>> classes;
        for (size_t c = 0; c < n_cls; c++) {
            std::string d = job + "/class_" + std::to_string(c) + "/";
            auto kc = std::make_unique<KeyClass>();
            ProvingKey& pk = kc->pk;
            pk.a_g = rd(d + "a_g"); pk.b_g = rd(d + "b_g"); pk.b_h = rd(d + "b_h"); pk.h_g = rd(d + "h_g");
            pk.deltas_g = rd(d + "deltas_g"); pk.last_delta_h = rd(d + "last_delta_h"); pk.alpha_g = rd(d + "alpha_g");
            pk.beta_g = rd(d + "beta_g"); pk.beta_h = rd(d + "beta_h");
            pk.ck_deltas_abc_g = {rd(d + "ck0"), rd(d + "ck1")};
            for (auto m : {std::make_pair(&pk.A, "A"), std::make_pair(&pk.B, "B"), std::make_pair(&pk.C, "C")}) {
                m.first->row_ptr = rdv<uint64_t>(d + m.second + "_row_ptr");
                m.first->col = rdv<uint32_t>(d + m.second + "_col");
                m.first->val_mont = rd(d + m.second + "_val");
            }
            auto cd = rdv<uint64_t>(d + "dims");                  // n_inst, n_constraints, n0, n_assignments
            pk.n_inst = cd.at(0); pk.n_constraints = cd.at(1); kc->n0 = cd.at(2);
            pk.upload(ctx);
            // the big host copies are not needed once the key is resident
            for (Bytes* b : {&pk.a_g, &pk.b_g, &pk.b_h, &pk.h_g}) Bytes().swap(*b);
            for (size_t k = 0; k < cd.at(3); k++) {
                Assignment as;
                as.host = rd(d + "z_" + std::to_string(k));
                as.n_v = as.host.size() / sz.fr;
                if (!host_inputs) {
                    check(hk_dev_alloc(ctx.raw(), as.host.size(), &as.dev), "hk_dev_alloc");
                    check(hk_dev_upload(ctx.raw(), as.dev, as.host.data(), as.host.size()), "hk_dev_upload");
                }
                kc->assigns.push_back(std::move(as));
            }
            classes.push_back(std::move(kc));
        }
        auto subs_raw = rdv<uint64_t>(job + "/subs");             // (class, assignment) per subcircuit
        Bytes seeds = rd(job + "/com_seeds"), rs = rd(job + "/rs");
        if (subs_raw.size() != 2 * n_sub || seeds.size() != 32 * n_sub || rs.size() != 2 * sz.fr * n_sub)
            throw std::runtime_error("job files do not match n_subcircuits");
        std::vector<Subcircuit> subs(n_sub);
        for (size_t i = 0; i < n_sub; i++) {
            subs[i].cls = subs_raw[2 * i]; subs[i].assign = subs_raw[2 * i + 1];
            if (subs[i].cls >= n_cls || subs[i].assign >= classes[subs[i].cls]->assigns.size())
                throw std::runtime_error("subcircuit refers to a class / assignment that does not exist");
            memcpy(subs[i].com_seed.data(), &seeds[32 * i], 32);
            subs[i].r = Bytes(rs.begin() + 2 * sz.fr * i, rs.begin() + 2 * sz.fr * i + sz.fr);
            subs[i].s = Bytes(rs.begin() + 2 * sz.fr * i + sz.fr, rs.begin() + 2 * sz.fr * (i + 1));
        }
        double load_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_load).count();

        // round 1 as one hk_commit_batch per key class where the stage is short (big-merkle: 16 witnesses): the requests'
        // stage-0 witnesses row after row and their blinders, laid out once (input data, before the timed region)
        struct ClassRows { std::vector<size_t> members; Bytes rows, kappas; bool batched = false; };
        std::vector<ClassRows> crow(n_cls);
        for (size_t i = 0; i < n_sub; i++) crow[subs[i].cls].members.push_back(i);
        struct Task { size_t cls, sub; bool whole_class; };
        std::vector<Task> round1;
        for (size_t c = 0; c < n_cls; c++) {
            ClassRows& cr = crow[c];
            KeyClass& kc = *classes[c];
            cr.batched = batch_commit && !cr.members.empty() && (kc.n0 + 1) * cr.members.size() * 2 <= 65536;
            if (!cr.batched) { for (size_t i : cr.members) round1.push_back(Task{c, i, false}); continue; }
            for (size_t i : cr.members) {
                const Assignment& as = kc.assigns[subs[i].assign];
                cr.rows.insert(cr.rows.end(), as.host.begin() + kc.pk.n_inst * sz.fr, as.host.begin() + (kc.pk.n_inst + kc.n0) * sz.fr);
                Bytes kappa = kappa_of(subs[i].com_seed);                                // committer.rs:85
                cr.kappas.insert(cr.kappas.end(), kappa.begin(), kappa.end());
            }
            round1.push_back(Task{c, 0, true});
        }
        std::vector<Stage0Response> resp0(n_sub);
        std::vector<Stage1Response> resp1(n_sub);
        auto step = [&]() {
            // round 1 (worker.rs:91-146)
            parallel_for(round1.size(), threads, [&](size_t t) {
                const Task& tk = round1[t];
                KeyClass& kc = *classes[tk.cls];
                if (tk.whole_class) {
                    const ClassRows& cr = crow[tk.cls];
                    Bytes coms(sz.g1 * cr.members.size());
                    check(hk_commit_batch(ctx.raw(), kc.pk.device, 0, kc.n0 ? cr.rows.data() : nullptr, kc.n0, cr.kappas.data(),
                                          cr.members.size(), coms.data()), "hk_commit_batch");
                    for (size_t k = 0; k < cr.members.size(); k++) {
                        size_t i = cr.members[k];
                        resp0[i] = Stage0Response{(uint64_t)i, Bytes(coms.begin() + k * sz.g1, coms.begin() + (k + 1) * sz.g1), subs[i].com_seed};
                    }
                    return;
                }
                size_t i = tk.sub;
                const Subcircuit& sc = subs[i];
                const Assignment& as = kc.assigns[sc.assign];
                Bytes kappa = kappa_of(sc.com_seed);                                     // committer.rs:85
                Bytes com(sz.g1);
                const uint8_t* w0 = host_inputs ? as.host.data() + kc.pk.n_inst * sz.fr
                                                : (const uint8_t*)as.dev + kc.pk.n_inst * sz.fr;
                check(hk_commit(ctx.raw(), kc.pk.device, 0, kc.n0 ? w0 : nullptr, kc.n0, kappa.data(), com.data()), "hk_commit");
                resp0[i] = Stage0Response{(uint64_t)i, com, sc.com_seed};
            });
            // round 2 (worker.rs:150-195): kappa re-derived from the seed, the commitment travels inside the proof
            parallel_for(n_sub, threads, [&](size_t i) {
                const Subcircuit& sc = subs[i];
                KeyClass& kc = *classes[sc.cls];
                const Assignment& as = kc.assigns[sc.assign];
                Bytes kappa = kappa_of(sc.com_seed);
                Proof p;
                p.a.resize(sz.g1); p.b.resize(sz.g2); p.c.resize(sz.g1);
                const void* z = host_inputs ? (const void*)as.host.data() : as.dev;
                check(hk_prove(ctx.raw(), kc.pk.device, z, as.n_v, sc.r.data(), sc.s.data(), kappa.data(), 1,
                               p.a.data(), p.b.data(), p.c.data()), "hk_prove");
                p.ds = {resp0[i].com};
                resp1[i] = Stage1Response{(uint64_t)i, p};
            });
        };
        for (unsigned w = 0; w < warmup; w++) step();
        check(hk_ctx_sync(ctx.raw()), "hk_ctx_sync");
        auto t0 = std::chrono::steady_clock::now();
        for (unsigned k = 0; k < steps; k++) step();
        check(hk_ctx_sync(ctx.raw()), "hk_ctx_sync");
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (size_t i = 0; i < n_sub; i++) {
            wr(out + "/stage0_resp_" + std::to_string(i) + ".bin",
               bls ? codec_bls->stage0_response_to_wire(resp0[i]) : codec_bn->stage0_response_to_wire(resp0[i]));
            wr(out + "/stage1_resp_" + std::to_string(i) + ".bin",
               bls ? codec_bls->stage1_response_to_wire(resp1[i]) : codec_bn->stage1_response_to_wire(resp1[i]));
        }
        for (auto& kc : classes)
            for (auto& as : kc->assigns)
                if (as.dev) hk_dev_free(ctx.raw(), as.dev);
        printf("{\"driver\": \"hk_all_in_one (C++ over the C ABI)\", \"curve\": \"%s\", \"subcircuits\": %zu, \"classes\": %zu, \"threads\": %u, "
               "\"steps\": %u, \"warmup\": %u, \"ms_per_step\": %.3f, \"proofs_per_s\": %.3f, \"inputs\": \"%s\", "
               "\"key_load_s\": %.2f}\n",
               bls ? "bls12_381" : "bn254", n_sub, n_cls, threads, steps, warmup, dt / steps * 1e3, n_sub * steps / dt,
               host_inputs ? "host (PCIe per proof)" : "resident", load_s);
        return 0;
    } catch (const Error& e) {
        fprintf(stderr, "hekaton error: %s\n", e.what());
        return 3;
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 4;
    }
}
