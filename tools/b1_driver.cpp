// =============================================================================
// b1_driver.cpp -- the CALLER side of the drop-in boundary, as the reference runs it.
//
// ddo's ParallelSolver starts nb_threads worker threads; every one of them owns a DecisionDiagram (`let mut mdd =
// D::default();`, parallel.rs:580) and runs process_one_node in a loop (parallel.rs:391-437): restricted compile,
// maybe_update_best, and when the restricted DD was inexact a relaxed compile followed by drain_cutset.  This file is that
// loop in C++ over include/ddo_hip.h ONLY -- what `hip_mdd::HipMdd` (hip_mdd/src/lib.rs) does per thread once a Rust
// toolchain builds it -- so that the boundary's throughput can be measured and its results
// checked per compile with T = 64 ... 2048 concurrent callers.  It is measurement / test support, not part of the
// product library: it links against libddo_hip.so like any other client.
//
// Every thread draws work items from one counter; item k is sub-problem k % nsub.  A digest of each compile (what the
// DecisionDiagram trait lets a caller observe, the cut-set as an order-independent checksum) is compared with the digest
// the FIRST compile of that sub-problem produced; the caller (tests) compares those first digests with the oracle's.
// =============================================================================
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/ddo_hip.h"

extern "C" {

typedef struct b1_digest {
    int32_t status;            // DDO_OK / DDO_CUTOFF / error of ddo_mdd_compile; -1000: never compiled
    int32_t is_exact;
    int32_t has_best, has_best_exact;
    int64_t best_value, best_exact_value;
    uint64_t nodes_expanded, arcs, layers;
    uint64_t n_cutset;
    uint64_t cutset_hash;      // sum over the cut-set nodes of mix(state words, value, ub, depth, path length) -- order independent
} b1_digest;

typedef struct b1_config {
    int device;
    int cutset_type;           // DDO_LAST_EXACT_LAYER [| DDO_MDD_ENGINE_*]
    size_t width;              // FixedWidth
    int threads;               // worker threads == mdds
    uint64_t items;            // work items in all (item k = sub-problem k % nsub)
    int64_t best_lb;           // incumbent every restricted compile starts from; the relaxed one sees max(best_lb, best exact value of the restricted DD)
} b1_config;

typedef struct b1_totals {
    double seconds;            // wall time of the threads' loops (mdds created before, destroyed after)
    uint64_t compiles, nodes_expanded, arcs, layers, cutset_nodes, path_decisions;
    uint64_t mismatches;       // compiles whose digest differs from the first digest of their sub-problem
    uint64_t errors;           // compiles that returned an error
    uint64_t launches, requests;   // ddo_mdd_combine_stats of the engine over the run
    double kernel_ms;              // ... and the HIP-event time of those launches
    double compile_s, drain_s;     // thread-seconds inside ddo_mdd_compile / inside ddo_mdd_drain_cutset and the value queries, summed over the threads
} b1_totals;

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

struct DrainAcc {
    uint64_t n = 0, hash = 0, decisions = 0;
};
static void drain_cb(const ddo_subproblem* sp, void* user) {
    DrainAcc* a = (DrainAcc*)user;
    uint64_t h = 0x9e3779b97f4a7c15ULL;
    for (size_t k = 0; k < sp->state_words; ++k) h = mix64(h ^ sp->state[k]);
    h = mix64(h ^ (uint64_t)sp->value);
    h = mix64(h ^ (uint64_t)sp->ub);
    h = mix64(h ^ (uint64_t)sp->depth);
    a->hash += h;
    a->n += 1;
    a->decisions += sp->path_len;
}

static bool same(const b1_digest& a, const b1_digest& b) { return std::memcmp(&a, &b, sizeof(b1_digest)) == 0; }

static void digest_of(ddo_mdd* mdd, int status, const ddo_completion& c, bool drain, b1_digest& d, DrainAcc& acc) {
    std::memset(&d, 0, sizeof(d));
    d.status = status;
    if (status != DDO_OK) return;
    d.is_exact = c.is_exact;
    d.has_best = c.has_best_value;
    d.best_value = c.has_best_value ? c.best_value : 0;
    int64_t v = 0;
    d.has_best_exact = ddo_mdd_best_exact_value(mdd, &v);
    d.best_exact_value = d.has_best_exact ? v : 0;
    ddo_counters k{};
    ddo_mdd_last_counters(mdd, &k);
    d.nodes_expanded = k.nodes_expanded;
    d.arcs = k.arcs;
    d.layers = k.layers;
    if (drain) {
        DrainAcc a;
        ddo_mdd_drain_cutset(mdd, drain_cb, &a);
        d.n_cutset = a.n;
        d.cutset_hash = a.hash;
        acc.n += a.n;
        acc.decisions += a.decisions;
    }
}

/// states: nsub x state_words; restricted / relaxed: nsub digests each (first compile of every sub-problem; relaxed[i].status ==
/// -1000 when the restricted DD of sub-problem i was exact).  Returns 0, or a negative DDO_ERR_* when the mdds cannot be created.
int b1_run(const ddo_model* model, const b1_config* cfg, size_t nsub, const uint64_t* states, const int64_t* values, const int64_t* depths,
           b1_digest* restricted, b1_digest* relaxed, b1_totals* out) {
    if (!model || !cfg || !out || nsub == 0 || cfg->threads < 1) return DDO_ERR_INVALID;
    const int T = cfg->threads;
    const size_t ws = (size_t)ddo_model_state_words(model);
    std::vector<ddo_mdd*> mdds((size_t)T, nullptr);
    for (int t = 0; t < T; ++t) {
        mdds[(size_t)t] = ddo_mdd_create(model, cfg->device, cfg->cutset_type, cfg->width);
        if (!mdds[(size_t)t]) {
            for (ddo_mdd* m : mdds)
                if (m) ddo_mdd_destroy(m);
            return DDO_ERR_NO_DEVICE;
        }
    }
    std::vector<std::atomic<int>> first(nsub);   // 0: nobody compiled this sub-problem yet, 1: being written, 2: digests valid
    for (auto& f : first) f.store(0);
    for (size_t i = 0; i < nsub; ++i) restricted[i].status = relaxed[i].status = -1000;
    std::atomic<uint64_t> next{0}, compiles{0}, nodes{0}, arcs{0}, layers{0}, csn{0}, dec{0}, mism{0}, errs{0}, ns_compile{0}, ns_drain{0};
    auto now = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    uint64_t l0 = 0, r0 = 0, l1 = 0, r1 = 0;
    double k0 = 0, k1 = 0;
    ddo_mdd_combine_stats(mdds[0], &l0, &r0, &k0);
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    auto worker = [&](int t) {
        ddo_mdd* mdd = mdds[(size_t)t];
        ready.fetch_add(1);
        while (!go.load()) std::this_thread::yield();
        uint64_t my_c = 0, my_n = 0, my_a = 0, my_l = 0, my_tc = 0, my_td = 0;
        DrainAcc acc;
        for (;;) {
            const uint64_t k = next.fetch_add(1);
            if (k >= cfg->items) break;
            const size_t i = (size_t)(k % nsub);
            ddo_compile_input in{};
            in.comp_type = DDO_RESTRICTED;
            in.max_width = cfg->width;
            in.best_lb = cfg->best_lb;
            in.residual.state = states + i * ws;
            in.residual.state_words = ws;
            in.residual.value = values[i];
            in.residual.ub = INT64_MAX;
            in.residual.depth = (size_t)depths[i];
            in.residual.path = nullptr;
            in.residual.path_len = 0;
            ddo_completion c{};
            b1_digest d0, d1;
            uint64_t ta = now();
            int rc = ddo_mdd_compile(mdd, &in, &c);   // 1. RESTRICTION (parallel.rs:402-420)
            uint64_t tb = now();
            digest_of(mdd, rc, c, false, d0, acc);
            my_tc += tb - ta, my_td += now() - tb;
            ++my_c;
            my_n += d0.nodes_expanded, my_a += d0.arcs, my_l += d0.layers;
            if (rc < 0) errs.fetch_add(1);
            std::memset(&d1, 0, sizeof(d1));
            d1.status = -1000;
            if (rc == DDO_OK && !c.is_exact) {   // 2. RELAXATION (parallel.rs:425-434)
                in.comp_type = DDO_RELAXED;
                if (d0.has_best_exact && d0.best_exact_value > in.best_lb) in.best_lb = d0.best_exact_value;   // maybe_update_best
                ddo_completion c2{};
                ta = now();
                rc = ddo_mdd_compile(mdd, &in, &c2);
                tb = now();
                digest_of(mdd, rc, c2, rc == DDO_OK && !c2.is_exact, d1, acc);   // enqueue_cutset only when inexact (parallel.rs:431)
                my_tc += tb - ta, my_td += now() - tb;
                ++my_c;
                my_n += d1.nodes_expanded, my_a += d1.arcs, my_l += d1.layers;
                if (rc < 0) errs.fetch_add(1);
            }
            int expect = 0;
            if (first[i].compare_exchange_strong(expect, 1)) {
                restricted[i] = d0;
                relaxed[i] = d1;
                first[i].store(2);
            } else {
                while (first[i].load() != 2) std::this_thread::yield();
                if (!same(restricted[i], d0) || !same(relaxed[i], d1)) mism.fetch_add(1);
            }
        }
        compiles += my_c, nodes += my_n, arcs += my_a, layers += my_l, csn += acc.n, dec += acc.decisions;
        ns_compile += my_tc, ns_drain += my_td;
    };
    std::vector<std::thread> th;
    th.reserve((size_t)T);
    for (int t = 0; t < T; ++t) th.emplace_back(worker, t);
    while (ready.load() < T) std::this_thread::yield();
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true);
    for (auto& x : th) x.join();
    const auto t1 = std::chrono::steady_clock::now();
    ddo_mdd_combine_stats(mdds[0], &l1, &r1, &k1);
    for (ddo_mdd* m : mdds) ddo_mdd_destroy(m);
    out->seconds = std::chrono::duration<double>(t1 - t0).count();
    out->compiles = compiles, out->nodes_expanded = nodes, out->arcs = arcs, out->layers = layers;
    out->cutset_nodes = csn, out->path_decisions = dec, out->mismatches = mism, out->errors = errs;
    out->launches = l1 - l0, out->requests = r1 - r0, out->kernel_ms = k1 - k0;
    out->compile_s = (double)ns_compile.load() * 1e-9, out->drain_s = (double)ns_drain.load() * 1e-9;
    return DDO_OK;
}

}  // extern "C"
