// =============================================================================
// hip_mdd_shim.cpp -- TEST INFRASTRUCTURE: the drop-in plug `hip_mdd/src/lib.rs` describes, COMPILED (in C++: no Rust toolchain here).
//
// `HipMdd` below implements the DecisionDiagram interface (mdd.rs:75-114) the way the Rust shim does -- every method one call of
// include/ddo_hip.h -- and is handed to the ORACLE's restatement of the reference's solvers as their type parameter `D`:
// `SequentialSolver<BitSet, HipMdd>` / `ParallelSolver<BitSet, HipMdd>` (oracle/ddo_oracle.hpp, restating sequential.rs:202-527 and
// parallel.rs:287-641: the worker threads, the mutex-shared critical section, the NoDupFringe, MaxUB, the widths).  So what runs is
// the reference's HOST, statement for statement, over the device engine through nothing but the C ABI -- the configuration north_star
// names ("host and branch-and-bound fringe stay in Rust, calling HIP through a thin extern "C" FFI").  It proves the plug end to end:
//   * a sequential search through HipMdd explores exactly what the same search explores through the oracle's own Mdd (every
//     compile returns the same bits: tests/test_gpu_shim.py);
//   * with T worker threads the host's concurrent compile() calls meet in the engine's combining layer and share launches.
// Like the oracle it includes, this file is for tests and measurement only; the product library does not know it exists.
// =============================================================================
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/ddo_hip.h"
#include "../../oracle/ddo_oracle.hpp"
#include "../../oracle/models.hpp"

using namespace ddo;

namespace {

/// what hip_mdd::install keeps in its registry (hip_mdd/src/lib.rs: REGISTRY)
struct Registry {
    ddo_model* model = nullptr;
    int device = 0;
    int cutset_type = DDO_LAST_EXACT_LAYER;
    size_t max_width = 0;
    size_t words = 0;
    size_t nb_vars = 0;
    ddo_cache* cache = nullptr;   // the device-side SimpleCache `HipCache` wraps (hip_mdd::install(.., cache_entries)); nullptr == EmptyCache
};
Registry g_reg;

void to_words(const BitSet& s, size_t words, uint64_t* out) {
    for (size_t k = 0; k < words; ++k) out[k] = k < s.w.size() ? s.w[k] : 0;
}
BitSet from_words(const uint64_t* w, size_t words, size_t nb_vars) {
    BitSet b(nb_vars);
    for (size_t k = 0; k < words && k < b.w.size(); ++k) b.w[k] = w[k];
    return b;
}

/// `impl DecisionDiagram for HipMdd` (hip_mdd/src/lib.rs), in C++
class HipMdd {
    ddo_mdd* h = nullptr;
    /// The cut-set of the latest relaxed compile, ALREADY turned into SubProblems -- by compile(), in the worker's own thread.  The
    /// reference drains a cut-set inside the solver's critical section (parallel.rs:456-469): what drain_cutset does per node is
    /// serialised over all worker threads, so it only moves these into the closure.
    std::vector<SubProblem<BitSet>> ready;

  public:
    MddCounters counters, last_counters;
    HipMdd() {   // == D::default() (parallel.rs:580)
        h = ddo_mdd_create(g_reg.model, g_reg.device, g_reg.cutset_type, g_reg.max_width);
        if (!h) throw std::runtime_error(std::string("ddo_mdd_create: ") + ddo_last_error());
    }
    ddo_mdd* raw() const { return h; }
    HipMdd(const HipMdd&) = delete;
    HipMdd& operator=(const HipMdd&) = delete;
    ~HipMdd() {
        if (h) ddo_mdd_destroy(h);
    }

    std::optional<Completion> compile(const CompilationInput<BitSet>& in, Reason* why = nullptr) {
        std::vector<uint64_t> st(g_reg.words);
        to_words(*in.residual->state, g_reg.words, st.data());
        std::vector<ddo_decision> path(in.residual->path.size());
        for (size_t i = 0; i < path.size(); ++i) path[i] = ddo_decision{(int64_t)in.residual->path[i].variable, (int64_t)in.residual->path[i].value};
        volatile int stop = in.cutoff->must_stop() ? 1 : 0;   // (`&dyn Cutoff` hands out no address: must_stop() is folded into a flag, as in the Rust shim)
        ddo_compile_input ci{};
        ci.comp_type = in.comp_type == CompilationType::Exact ? DDO_EXACT : in.comp_type == CompilationType::Relaxed ? DDO_RELAXED : DDO_RESTRICTED;
        ci.max_width = std::min(in.max_width, g_reg.max_width);
        ci.best_lb = in.best_lb;
        ci.residual.state = st.data();
        ci.residual.state_words = g_reg.words;
        ci.residual.value = in.residual->value;
        ci.residual.ub = in.residual->ub;
        ci.residual.depth = in.residual->depth;
        ci.residual.path = path.empty() ? nullptr : path.data();
        ci.residual.path_len = path.size();
        ci.cutoff = &stop;
        ci.cache = g_reg.cache;   // null == EmptyCache; else the table HipCache wraps (same thresholds on both sides)
        ddo_completion out{};
        const int rc = ddo_mdd_compile(h, &ci, &out);
        if (rc == DDO_CUTOFF) {
            if (why) *why = Reason::CutoffOccurred;
            return std::nullopt;
        }
        if (rc != DDO_OK) throw std::runtime_error(std::string("ddo_mdd_compile: ") + ddo_last_error());
        ddo_counters k{};
        ddo_mdd_last_counters(h, &k);
        last_counters = MddCounters{k.nodes_expanded, k.arcs, k.layers, 1};
        counters.add(last_counters);
        Completion c;
        c.is_exact = out.is_exact != 0;
        if (out.has_best_value) c.best_value = (isize)out.best_value;
        ready.clear();
        if (in.comp_type == CompilationType::Relaxed && !c.is_exact) {
            // One FFI call for the whole cut-set (ddo_mdd_drain_cutset_rows) instead of a callback per node; nodes whose bound does not
            // exceed input.best_lb are left on the other side: both solvers' closures drop them (parallel.rs:461-463, sequential.rs:372-376).
            ddo_cutset_rows rows{};
            if (ddo_mdd_drain_cutset_rows(h, (int64_t)in.best_lb, &rows) != DDO_OK) throw std::runtime_error(std::string("ddo_mdd_drain_cutset_rows: ") + ddo_last_error());
            ready.resize(rows.count);
            const auto& head = in.residual->path;
            for (size_t i = 0; i < rows.count; ++i) {
                SubProblem<BitSet>& n = ready[i];
                n.state = std::make_shared<const BitSet>(from_words(rows.states + i * rows.state_words, rows.state_words, g_reg.nb_vars));
                n.value = (isize)rows.values[i];
                n.ub = (isize)rows.ubs[i];
                n.depth = rows.depths[i];
                n.path.reserve(head.size() + rows.path_lens[i]);
                n.path.assign(head.begin(), head.end());
                const ddo_decision* p = rows.paths + i * rows.path_stride;
                for (size_t k = 0; k < rows.path_lens[i]; ++k) n.path.push_back(Decision{(size_t)p[k].variable, (isize)p[k].value});
            }
        }
        return c;
    }
    bool is_exact() const { return ddo_mdd_is_exact(h) != 0; }
    std::optional<isize> best_value() const {
        int64_t v = 0;
        return ddo_mdd_best_value(h, &v) == 1 ? std::optional<isize>((isize)v) : std::nullopt;
    }
    std::optional<isize> best_exact_value() const {
        int64_t v = 0;
        return ddo_mdd_best_exact_value(h, &v) == 1 ? std::optional<isize>((isize)v) : std::nullopt;
    }
    std::optional<Solution> solution_of(int (*fn)(const ddo_mdd*, ddo_decision*, size_t*)) const {
        std::vector<ddo_decision> buf(2 * g_reg.nb_vars + 8);
        size_t len = buf.size();
        if (fn(h, buf.data(), &len) != 1) return std::nullopt;
        Solution s(len);
        for (size_t i = 0; i < len; ++i) s[i] = Decision{(size_t)buf[i].variable, (isize)buf[i].value};
        return s;
    }
    std::optional<Solution> best_solution() const { return solution_of(ddo_mdd_best_solution); }
    std::optional<Solution> best_exact_solution() const { return solution_of(ddo_mdd_best_exact_solution); }
    template <class F>
    void drain_cutset(F&& func) {   // mdd.rs:107-113 -- called under the solver's lock: nothing is built here
        for (SubProblem<BitSet>& n : ready) func(std::move(n));
        ready.clear();
    }
};

/// `impl Cache for HipCache` (hip_mdd/src/lib.rs), in C++: the solver's own calls (must_explore at the pop, update_threshold in the
/// parallel solver, clear) reach the device table through the ABI's host views; the compiles read and write it on the device.
class HipCache : public Cache<BitSet> {
  public:
    void initialize(const Problem<BitSet>&) override {}   // created by install()
    std::optional<Threshold> get_threshold(const BitSet& state, size_t depth) const override {
        if (!g_reg.cache) return std::nullopt;
        std::vector<uint64_t> st(g_reg.words);
        to_words(state, g_reg.words, st.data());
        int64_t v = 0;
        int e = 0;
        if (ddo_cache_get_threshold(g_reg.cache, st.data(), depth, &v, &e) != 1) return std::nullopt;
        return Threshold{(isize)v, e != 0};
    }
    void update_threshold(std::shared_ptr<const BitSet> state, size_t depth, isize value, bool explored) override {
        if (!g_reg.cache) return;
        std::vector<uint64_t> st(g_reg.words);
        to_words(*state, g_reg.words, st.data());
        ddo_cache_update_threshold(g_reg.cache, st.data(), depth, (int64_t)value, explored ? 1 : 0);
    }
    void clear_layer(size_t) override {}   // memory management only in the reference (parallel.rs:506-511)
    void clear() override {
        if (g_reg.cache) ddo_cache_clear(g_reg.cache);
    }
};

}  // namespace

extern "C" {

struct shim_out {
    int has_value, is_exact;
    int64_t best_value, best_lb, best_ub;
    uint64_t explored, nodes_expanded, arcs, layers, compiles;
    double wall_s;
    uint64_t launches, requests;   // ddo_mdd_combine_stats over the search
    int n_solution;
};

/// The reference's solver over the device: width 0 = NbUnassignedWidth, nthreads 0 = SequentialSolver, else ParallelSolver with that
/// many worker threads (each with its own HipMdd); pooled: the mdds are Pooled decision diagrams (hip_mdd::install_pooled).
/// solution: (variable, value) pairs, capacity 2 * n.  Returns 0, or -1 with the message on stderr.
int shim_misp_solve_ex(const char* path, uint64_t width, int nthreads, int device, double timeout_s, int pooled, uint64_t cache_entries,
                       shim_out* out, int64_t* solution);
int shim_misp_solve(const char* path, uint64_t width, int nthreads, int device, double timeout_s, int pooled, shim_out* out, int64_t* solution) {
    return shim_misp_solve_ex(path, width, nthreads, device, timeout_s, pooled, 0, out, solution);
}
/// cache_entries > 0: hip_mdd::install(.., cache_entries) -- the mdds take the device-side SimpleCache, the solver's `C` is HipCache
/// (the reference's Seq / ParCachingSolverLel, and with pooled = 1 Seq / ParCachingSolverPooled, solver/mod.rs:36, :38, :45, :47)
int shim_misp_solve_ex(const char* path, uint64_t width, int nthreads, int device, double timeout_s, int pooled, uint64_t cache_entries,
                       shim_out* out, int64_t* solution) {
    try {
        Misp pb = read_misp_instance(path);
        MispRelax relax(pb);
        MispRanking rank;
        // hip_mdd::install / install_pooled
        const size_t words = (pb.nb_vars + 63) / 64;
        std::vector<uint64_t> rows(pb.nb_vars * words, 0);
        std::vector<int64_t> w(pb.nb_vars);
        for (size_t i = 0; i < pb.nb_vars; ++i) {
            to_words(pb.neighbors[i], words, rows.data() + i * words);
            w[i] = pb.weight[i];
        }
        ddo_model* model = ddo_model_create_misp((int)pb.nb_vars, rows.data(), w.data());
        if (!model) throw std::runtime_error(std::string("ddo_model_create_misp: ") + ddo_last_error());
        ddo_cache* cache = nullptr;
        if (cache_entries) {
            cache = ddo_cache_create(model, device, (size_t)cache_entries);
            if (!cache) throw std::runtime_error(std::string("ddo_cache_create: ") + ddo_last_error());
        }
        g_reg = Registry{model, device, (pooled ? (DDO_FRONTIER | DDO_MDD_POOLED) : DDO_LAST_EXACT_LAYER) | (cache ? DDO_MDD_CACHING : 0),
                         width ? (size_t)width : pb.nb_vars, words, pb.nb_vars, cache};
        FixedWidth<BitSet> fixed(width);
        NbUnassignedWidth<BitSet> unassigned(pb.nb_vars);
        const WidthHeuristic<BitSet>& wh = width ? (const WidthHeuristic<BitSet>&)fixed : unassigned;
        EmptyDominanceChecker<BitSet> dom;
        NoCutoff nocut;
        TimeBudget budget(timeout_s > 0 ? timeout_s : 1e9);
        const Cutoff& cut = timeout_s > 0 ? (const Cutoff&)budget : nocut;
        MaxUB<BitSet> mx(rank);
        NoDupFringe<BitSet> fringe(mx);
        uint64_t l0 = 0, r0 = 0, l1 = 0, r1 = 0;
        {
            HipMdd probe;   // (keeps the engine alive over the whole search and reads its launch statistics)
            ddo_mdd* ph = probe.raw();
            ddo_mdd_combine_stats(ph, &l0, &r0, nullptr);
            const auto t0 = std::chrono::steady_clock::now();
            Completion c;
            std::optional<Solution> sol;
            MddCounters cnt;
            auto run = [&](auto& s) {
                c = s.maximize();
                out->best_lb = s.best_lower_bound();
                out->best_ub = s.best_upper_bound();
                out->explored = s.explored();
                sol = s.best_solution();
                cnt = s.counters();
            };
            if (nthreads <= 0 && cache) {
                SequentialSolver<BitSet, HipMdd, HipCache> s(pb, relax, rank, wh, dom, cut, fringe);
                run(s);
            } else if (nthreads <= 0) {
                SequentialSolver<BitSet, HipMdd> s(pb, relax, rank, wh, dom, cut, fringe);
                run(s);
            } else if (cache) {
                ParallelSolver<BitSet, HipMdd, HipCache> s(pb, relax, rank, wh, dom, cut, fringe, (size_t)nthreads);
                run(s);
            } else {
                ParallelSolver<BitSet, HipMdd> s(pb, relax, rank, wh, dom, cut, fringe, (size_t)nthreads);
                run(s);
            }
            out->wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            ddo_mdd_combine_stats(ph, &l1, &r1, nullptr);
            out->has_value = c.best_value.has_value();
            out->best_value = c.best_value.value_or(-1);
            out->is_exact = c.is_exact;
            out->nodes_expanded = cnt.nodes_expanded;
            out->arcs = cnt.arcs;
            out->layers = cnt.layers;
            out->compiles = cnt.compiles;
            out->launches = l1 - l0;
            out->requests = r1 - r0;
            out->n_solution = 0;
            if (sol && solution)
                for (const Decision& d : *sol) {
                    solution[2 * out->n_solution] = (int64_t)d.variable;
                    solution[2 * out->n_solution + 1] = d.value;
                    out->n_solution++;
                }
        }
        if (cache) ddo_cache_destroy(cache);
        ddo_model_destroy(model);
        g_reg = Registry{};
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "shim_misp_solve: %s\n", e.what());
        return -1;
    }
}

}  // extern "C"
