// =============================================================================
// oracle_capi.cpp -- C entry points over the CPU oracle, loaded with ctypes by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
//
// *** TEST INFRASTRUCTURE *** (see ddo_oracle.hpp).  The product library
// (ddo_amd/csrc) never links or calls this file.
// =============================================================================
#include <chrono>
#include <cstring>
#include <string>

#include "ddo_oracle.hpp"
#include "models.hpp"

using namespace ddo;

namespace {

struct MispHandle {
    Misp pb;
    size_t ws;  // 64-bit words per state
};

/// One recorded compile() of a sequential solve (input + everything observable).
struct TraceRec {
    int comp_type;  // 1 = Relaxed, 2 = Restricted (CompilationType order: Exact=0)
    uint64_t width;
    int64_t best_lb;
    std::vector<uint64_t> state;
    int64_t value;
    int64_t ub;
    uint64_t depth;
    // outputs
    int is_exact;
    int has_best;
    int64_t best_value;
    int has_best_exact;
    int64_t best_exact_value;
    uint64_t nodes_expanded, arcs, layers;
    // cut-set (relaxed only): flattened
    std::vector<uint64_t> cs_states;  // n_cs * ws
    std::vector<int64_t> cs_value, cs_ub;
    std::vector<uint64_t> cs_depth;
};

struct Trace {
    std::vector<TraceRec> recs;
    size_t ws = 0;
};

void state_to_words(const BitSet& s, size_t ws, uint64_t* out) {
    for (size_t k = 0; k < ws; ++k) out[k] = k < s.w.size() ? s.w[k] : 0;
}

void state_to_words(const KnapsackState& s, size_t, uint64_t* out) {   // device wire: word 0 capacity, word 1 depth
    out[0] = (uint64_t)s.capacity;
    out[1] = (uint64_t)s.depth;
}
void state_to_words(const Max2SatState& s, size_t, uint64_t* out) { pack_signed_vector(s.substates, s.depth, out); }
void state_to_words(const TsptwState& s, size_t ws, uint64_t* out) { pack_tsptw_state(s, (int)(ws - 2) / 3, out); }
void state_to_words(const McpState& s, size_t, uint64_t* out) { pack_signed_vector(s.benef, s.depth, out); }

template <class T, class D>
void record(Trace& tr, size_t ws, const SubProblem<T>& node, CompilationType t, size_t width, isize lb, D& mdd,
            bool with_cutset) {
    TraceRec r;
    r.comp_type = (int)t;
    r.width = width;
    r.best_lb = lb;
    r.state.resize(ws);
    state_to_words(*node.state, ws, r.state.data());
    r.value = node.value;
    r.ub = node.ub;
    r.depth = node.depth;
    r.is_exact = mdd.is_exact();
    auto bv = mdd.best_value();
    r.has_best = bv.has_value();
    r.best_value = bv.value_or(0);
    auto bev = mdd.best_exact_value();
    r.has_best_exact = bev.has_value();
    r.best_exact_value = bev.value_or(0);
    r.nodes_expanded = mdd.last_counters.nodes_expanded;
    r.arcs = mdd.last_counters.arcs;
    r.layers = mdd.last_counters.layers;
    if (with_cutset) {
        // drain_cutset consumes the cut-set; take a copy of the DD so the solver still sees it
        D copy = mdd;
        copy.drain_cutset([&](SubProblem<T> sp) {
            size_t off = r.cs_states.size();
            r.cs_states.resize(off + ws);
            state_to_words(*sp.state, ws, r.cs_states.data() + off);
            r.cs_value.push_back(sp.value);
            r.cs_ub.push_back(sp.ub);
            r.cs_depth.push_back(sp.depth);
        });
    }
    tr.recs.push_back(std::move(r));
}

}  // namespace

extern "C" {

struct oracle_solve_out {
    int has_value;
    int is_exact;
    int64_t best_value;
    int64_t best_lb;
    int64_t best_ub;
    uint64_t explored;
    uint64_t nodes_expanded;
    uint64_t arcs;
    uint64_t layers;
    uint64_t compiles;
    double wall_s;
    int n_solution;          // number of decisions written to `solution`
};

void* oracle_misp_load(const char* path) {
    try {
        auto* h = new MispHandle{read_misp_instance(path), 0};
        h->ws = (h->pb.nb_vars + 63) / 64;
        return h;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "oracle_misp_load: %s\n", e.what());
        return nullptr;
    }
}
void oracle_misp_free(void* h) { delete (MispHandle*)h; }
int oracle_misp_nb_vars(void* h) { return (int)((MispHandle*)h)->pb.nb_vars; }
int oracle_misp_state_words(void* h) { return (int)((MispHandle*)h)->ws; }
/// complement-adjacency rows (n * ws words) and weights (n)
void oracle_misp_export(void* hh, uint64_t* rows, int64_t* weights) {
    auto* h = (MispHandle*)hh;
    for (size_t i = 0; i < h->pb.nb_vars; ++i) {
        state_to_words(h->pb.neighbors[i], h->ws, rows + i * h->ws);
        weights[i] = h->pb.weight[i];
    }
}

/// Full branch-and-bound.  width == 0 -> NbUnassignedWidth (the policy of examples/misp/tests.rs);
/// nthreads == 0 -> SequentialSolver, else ParallelSolver with that many threads;
/// timeout_s <= 0 -> NoCutoff.  solution: (variable,value) pairs sorted by variable, capacity 2*n int64.
}  // extern "C"
template <class DD>
static int misp_solve_with(void* hh, uint64_t width, int nthreads, double timeout_s, oracle_solve_out* out, int64_t* solution) {
    auto* h = (MispHandle*)hh;
    Misp& pb = h->pb;
    MispRelax relax(pb);
    MispRanking rank;
    FixedWidth<BitSet> fixed(width);
    NbUnassignedWidth<BitSet> unassigned(pb.nb_vars);
    const WidthHeuristic<BitSet>& w = width ? (const WidthHeuristic<BitSet>&)fixed : unassigned;
    EmptyDominanceChecker<BitSet> dom;
    NoCutoff nocut;
    TimeBudget budget(timeout_s > 0 ? timeout_s : 1e9);
    const Cutoff& cut = timeout_s > 0 ? (const Cutoff&)budget : nocut;
    MaxUB<BitSet> mx(rank);
    NoDupFringe<BitSet> fringe(mx);

    auto t0 = std::chrono::steady_clock::now();
    Completion c;
    std::optional<Solution> sol;
    MddCounters cnt;
    if (nthreads <= 0) {
        SequentialSolver<BitSet, DD> s(pb, relax, rank, w, dom, cut, fringe);
        c = s.maximize();
        out->best_lb = s.best_lower_bound();
        out->best_ub = s.best_upper_bound();
        out->explored = s.explored();
        sol = s.best_solution();
        cnt = s.counters();
    } else {
        ParallelSolver<BitSet, DD> s(pb, relax, rank, w, dom, cut, fringe, (size_t)nthreads);
        c = s.maximize();
        out->best_lb = s.best_lower_bound();
        out->best_ub = s.best_upper_bound();
        out->explored = s.explored();
        sol = s.best_solution();
        cnt = s.counters();
    }
    auto t1 = std::chrono::steady_clock::now();
    out->wall_s = std::chrono::duration<double>(t1 - t0).count();
    out->has_value = c.best_value.has_value();
    out->best_value = c.best_value.value_or(-1);
    out->is_exact = c.is_exact;
    out->nodes_expanded = cnt.nodes_expanded;
    out->arcs = cnt.arcs;
    out->layers = cnt.layers;
    out->compiles = cnt.compiles;
    out->n_solution = 0;
    if (sol && solution) {
        for (const Decision& d : *sol) {
            solution[2 * out->n_solution] = (int64_t)d.variable;
            solution[2 * out->n_solution + 1] = d.value;
            out->n_solution++;
        }
    }
    return 0;
}
extern "C" {
int oracle_misp_solve(void* hh, uint64_t width, int nthreads, double timeout_s, oracle_solve_out* out, int64_t* solution) {
    return misp_solve_with<DefaultMDDLEL<BitSet>>(hh, width, nthreads, timeout_s, out, solution);
}
/// the same search over Pooled DDs (solver/mod.rs:34, :43: Par / SeqNoCachingSolverPooled; mdd/pooled.rs)
int oracle_misp_solve_pooled(void* hh, uint64_t width, int nthreads, double timeout_s, oracle_solve_out* out, int64_t* solution) {
    return misp_solve_with<Pooled<BitSet>>(hh, width, nthreads, timeout_s, out, solution);
}

}  // extern "C"
// ---- traced sequential solve: every compile() recorded for replay on the GPU -------------------
template <class DD>
static void* misp_trace_solve_with(void* hh, uint64_t width, uint64_t max_compiles, oracle_solve_out* out) {
    auto* h = (MispHandle*)hh;
    Misp& pb = h->pb;
    MispRelax relax(pb);
    MispRanking rank;
    FixedWidth<BitSet> fixed(width);
    NbUnassignedWidth<BitSet> unassigned(pb.nb_vars);
    const WidthHeuristic<BitSet>& w = width ? (const WidthHeuristic<BitSet>&)fixed : unassigned;
    EmptyDominanceChecker<BitSet> dom;
    struct CountCutoff : Cutoff {
        const Trace* tr;
        uint64_t max;
        bool must_stop() const override { return max && tr->recs.size() >= max; }
    } cut;
    MaxUB<BitSet> mx(rank);
    NoDupFringe<BitSet> fringe(mx);
    auto* tr = new Trace();
    tr->ws = h->ws;
    cut.tr = tr;
    cut.max = max_compiles;
    SequentialSolver<BitSet, DD> s(pb, relax, rank, w, dom, cut, fringe);
    s.on_compile = [&](const SubProblem<BitSet>& node, CompilationType t, size_t width_, isize lb, DD& mdd) {
        record(*tr, h->ws, node, t, width_, lb, mdd, t == CompilationType::Relaxed);
    };
    auto t0 = std::chrono::steady_clock::now();
    Completion c = s.maximize();
    auto t1 = std::chrono::steady_clock::now();
    if (out) {
        out->wall_s = std::chrono::duration<double>(t1 - t0).count();
        out->has_value = c.best_value.has_value();
        out->best_value = c.best_value.value_or(-1);
        out->is_exact = c.is_exact;
        out->best_lb = s.best_lower_bound();
        out->best_ub = s.best_upper_bound();
        out->explored = s.explored();
        out->nodes_expanded = s.counters().nodes_expanded;
        out->arcs = s.counters().arcs;
        out->layers = s.counters().layers;
        out->compiles = s.counters().compiles;
        out->n_solution = 0;
    }
    return tr;
}
extern "C" {
void* oracle_misp_trace_solve(void* hh, uint64_t width, uint64_t max_compiles, oracle_solve_out* out) {
    return misp_trace_solve_with<DefaultMDDLEL<BitSet>>(hh, width, max_compiles, out);
}
/// the traced search of SeqNoCachingSolverPooled (solver/mod.rs:43): every compile a Pooled DD (mdd/pooled.rs)
void* oracle_misp_trace_solve_pooled(void* hh, uint64_t width, uint64_t max_compiles, oracle_solve_out* out) {
    return misp_trace_solve_with<Pooled<BitSet>>(hh, width, max_compiles, out);
}
}  // extern "C"

namespace {
/// traced sequential solve of a signed-vector model (MAX2SAT, MCP): same records as the MISP trace, states packed as on
/// the device wire (two benefits per word + a depth word)
template <class T, class PB, class RELAX, class RANK>
Trace* traced_vector_solve(PB& pb, RELAX& relax, RANK& rank, size_t nvars, uint64_t width, uint64_t max_compiles, oracle_solve_out* out) {
    FixedWidth<T> fixed(width);
    NbUnassignedWidth<T> unassigned(nvars);
    const WidthHeuristic<T>& w = width ? (const WidthHeuristic<T>&)fixed : unassigned;
    EmptyDominanceChecker<T> dom;
    struct CountCutoff : Cutoff {
        const Trace* tr;
        uint64_t max;
        bool must_stop() const override { return max && tr->recs.size() >= max; }
    } cut;
    MaxUB<T> mx(rank);
    NoDupFringe<T> fringe(mx);
    auto* tr = new Trace();
    tr->ws = (nvars + 1) / 2 + 1;
    cut.tr = tr;
    cut.max = max_compiles;
    SequentialSolver<T> s(pb, relax, rank, w, dom, cut, fringe);
    s.on_compile = [&](const SubProblem<T>& node, CompilationType t, size_t width_, isize lb, DefaultMDDLEL<T>& mdd) {
        record<T>(*tr, tr->ws, node, t, width_, lb, mdd, t == CompilationType::Relaxed);
    };
    auto t0 = std::chrono::steady_clock::now();
    Completion c = s.maximize();
    if (out) {
        out->wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        out->has_value = c.best_value.has_value();
        out->best_value = c.best_value.value_or(-1);
        out->is_exact = c.is_exact;
        out->best_lb = s.best_lower_bound();
        out->best_ub = s.best_upper_bound();
        out->explored = s.explored();
        out->nodes_expanded = s.counters().nodes_expanded;
        out->arcs = s.counters().arcs;
        out->layers = s.counters().layers;
        out->compiles = s.counters().compiles;
        out->n_solution = 0;
    }
    return tr;
}
}  // namespace

extern "C" {
}  // extern "C"

namespace {
/// traced sequential solve with any decision-diagram type D (last exact layer / frontier cut-set) and cache type C
template <class T, class D, class C, class PB, class RELAX, class RANK>
Trace* traced_solve_dc(PB& pb, RELAX& relax, RANK& rank, size_t nvars, size_t ws, uint64_t width, uint64_t max_compiles,
                       oracle_solve_out* out, DominanceChecker<T>* domx = nullptr, const WidthHeuristic<T>* wx = nullptr) {
    FixedWidth<T> fixed(width);
    NbUnassignedWidth<T> unassigned(nvars);
    const WidthHeuristic<T>& w = wx ? *wx : (width ? (const WidthHeuristic<T>&)fixed : unassigned);
    EmptyDominanceChecker<T> dom0;
    DominanceChecker<T>& dom = domx ? *domx : (DominanceChecker<T>&)dom0;
    struct CountCutoff : Cutoff {
        const Trace* tr;
        uint64_t max;
        bool must_stop() const override { return max && tr->recs.size() >= max; }
    } cut;
    MaxUB<T> mx(rank);
    NoDupFringe<T> fringe(mx);
    auto* tr = new Trace();
    tr->ws = ws;
    cut.tr = tr;
    cut.max = max_compiles;
    SequentialSolver<T, D, C> s(pb, relax, rank, w, dom, cut, fringe);
    s.on_compile = [&](const SubProblem<T>& node, CompilationType t, size_t width_, isize lb, D& mdd) {
        record<T>(*tr, ws, node, t, width_, lb, mdd, t == CompilationType::Relaxed);
    };
    auto t0 = std::chrono::steady_clock::now();
    Completion c = s.maximize();
    if (out) {
        out->wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        out->has_value = c.best_value.has_value();
        out->best_value = c.best_value.value_or(-1);
        out->is_exact = c.is_exact;
        out->best_lb = s.best_lower_bound();
        out->best_ub = s.best_upper_bound();
        out->explored = s.explored();
        out->nodes_expanded = s.counters().nodes_expanded;
        out->arcs = s.counters().arcs;
        out->layers = s.counters().layers;
        out->compiles = s.counters().compiles;
        out->n_solution = 0;
    }
    return tr;
}
template <class T, class PB, class RELAX, class RANK>
Trace* traced_solve_any(PB& pb, RELAX& relax, RANK& rank, size_t nvars, size_t ws, uint64_t width, uint64_t max_compiles, int frontier,
                        int cache, oracle_solve_out* out, DominanceChecker<T>* dom = nullptr, const WidthHeuristic<T>* wx = nullptr) {
    if (frontier && cache) return traced_solve_dc<T, DefaultMDDFC<T>, SimpleCache<T>>(pb, relax, rank, nvars, ws, width, max_compiles, out, dom, wx);
    if (frontier) return traced_solve_dc<T, DefaultMDDFC<T>, EmptyCache<T>>(pb, relax, rank, nvars, ws, width, max_compiles, out, dom, wx);
    if (cache) return traced_solve_dc<T, DefaultMDDLEL<T>, SimpleCache<T>>(pb, relax, rank, nvars, ws, width, max_compiles, out, dom, wx);
    return traced_solve_dc<T, DefaultMDDLEL<T>, EmptyCache<T>>(pb, relax, rank, nvars, ws, width, max_compiles, out, dom, wx);
}
}  // namespace

extern "C" {
/// Traced SEQUENTIAL solve (sequential.rs) of an instance file with the decision-diagram / cache combination of the
/// reference's solver aliases (solver/mod.rs): frontier = 0 last-exact-layer cut-set, 1 frontier cut-set; cache = 0
/// EmptyCache, 1 SimpleCache.  kind: "misp" | "knapsack" | "max2sat" | "mcp".  Records are read like every other trace.
void* oracle_trace_solve_ex(const char* kind, const char* path, uint64_t width, uint64_t max_compiles, int frontier, int cache,
                            oracle_solve_out* out) {
    try {
        const std::string k(kind);
        if (k == "misp") {
            Misp pb = read_misp_instance(path);
            MispRelax relax(pb);
            MispRanking rank;
            return traced_solve_any<BitSet>(pb, relax, rank, pb.nb_vars, (pb.nb_vars + 63) / 64, width, max_compiles, frontier, cache, out);
        }
        if (k == "misp+pooled") {   // Pooled<T> (mdd/pooled.rs) with EmptyCache / SimpleCache: Seq{No,}CachingSolverPooled (solver/mod.rs:43, :47)
            Misp pb = read_misp_instance(path);
            MispRelax relax(pb);
            MispRanking rank;
            const size_t ws = (pb.nb_vars + 63) / 64;
            if (cache) return traced_solve_dc<BitSet, Pooled<BitSet>, SimpleCache<BitSet>>(pb, relax, rank, pb.nb_vars, ws, width, max_compiles, out);
            return traced_solve_dc<BitSet, Pooled<BitSet>, EmptyCache<BitSet>>(pb, relax, rank, pb.nb_vars, ws, width, max_compiles, out);
        }
        if (k == "knapsack" || k == "knapsack+dominance") {   // "+dominance": SimpleDominanceChecker(KPDominance), knapsack/main.rs:325
            Knapsack pb = read_knapsack_instance(path);
            KPRelax relax(pb);
            KPRanking rank;
            SimpleDominanceChecker<KnapsackState, KPDominance> dom(KPDominance{}, pb.nb_variables());
            return traced_solve_any<KnapsackState>(pb, relax, rank, pb.nb_variables(), 2, width, max_compiles, frontier, cache, out,
                                                   k == "knapsack" ? nullptr : &dom);
        }
        if (k == "max2sat") {
            Weighed2Sat inst = read_max2sat_instance(path);
            Max2Sat pb(inst);
            Max2SatRelax relax(pb);
            Max2SatRanking rank;
            return traced_solve_any<Max2SatState>(pb, relax, rank, pb.nb_variables(), (pb.nb_variables() + 1) / 2 + 1, width, max_compiles,
                                                  frontier, cache, out);
        }
        if (k == "mcp") {
            Mcp pb(read_mcp_instance(path));
            McpRelax relax(pb);
            McpRanking rank;
            return traced_solve_any<McpState>(pb, relax, rank, pb.nb_variables(), (pb.nb_variables() + 1) / 2 + 1, width, max_compiles, frontier,
                                              cache, out);
        }
        if (k == "tsptw" || k == "tsptw+dominance") {   // width 0: TsptwWidth(nb_vars, 1) as in examples/tsptw/tests.rs:42; else FixedWidth
            Tsptw pb(read_tsptw_instance(path));
            const size_t tws = 3 * (size_t)tsptw_set_words(pb.nb_variables()) + 2;   // words of a packed state (include/ddo_hip.h)
            TsptwRelax relax(pb);
            TsptwRanking rank;
            TsptwWidth tw(pb.nb_variables(), 1);
            SimpleDominanceChecker<TsptwState, TsptwDominance> dom(TsptwDominance(), pb.nb_variables());
            return traced_solve_any<TsptwState>(pb, relax, rank, pb.nb_variables(), tws, width, max_compiles, frontier, cache, out,
                                                k == "tsptw" ? nullptr : &dom, width ? nullptr : &tw);
        }
        std::fprintf(stderr, "oracle_trace_solve_ex: unknown kind %s\n", kind);
        return nullptr;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "oracle_trace_solve_ex: %s\n", e.what());
        return nullptr;
    }
}

/// traced solves of the signed-vector models: the trace is read with oracle_trace_len / _get / _get_cutset / _free;
/// every state has oracle_trace_state_words() words
void* oracle_max2sat_trace_solve(const char* path, uint64_t width, uint64_t max_compiles, oracle_solve_out* out) {
    try {
        Weighed2Sat inst = read_max2sat_instance(path);
        Max2Sat pb(inst);
        Max2SatRelax relax(pb);
        Max2SatRanking rank;
        return traced_vector_solve<Max2SatState>(pb, relax, rank, pb.nb_variables(), width, max_compiles, out);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "oracle_max2sat_trace_solve: %s\n", e.what());
        return nullptr;
    }
}
void* oracle_mcp_trace_solve(const char* path, uint64_t width, uint64_t max_compiles, oracle_solve_out* out) {
    try {
        Mcp pb(read_mcp_instance(path));
        McpRelax relax(pb);
        McpRanking rank;
        return traced_vector_solve<McpState>(pb, relax, rank, pb.nb_variables(), width, max_compiles, out);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "oracle_mcp_trace_solve: %s\n", e.what());
        return nullptr;
    }
}
uint64_t oracle_trace_state_words(void* t) { return ((Trace*)t)->ws; }
void oracle_trace_free(void* t) { delete (Trace*)t; }
uint64_t oracle_trace_len(void* t) { return ((Trace*)t)->recs.size(); }

struct oracle_trace_hdr {
    int comp_type;
    int is_exact;
    int has_best;
    int has_best_exact;
    uint64_t width;
    int64_t best_lb;
    int64_t value;
    int64_t ub;
    uint64_t depth;
    int64_t best_value;
    int64_t best_exact_value;
    uint64_t nodes_expanded, arcs, layers;
    uint64_t n_cutset;
};
void oracle_trace_get(void* t, uint64_t i, oracle_trace_hdr* hdr, uint64_t* state) {
    const TraceRec& r = ((Trace*)t)->recs[i];
    hdr->comp_type = r.comp_type;
    hdr->is_exact = r.is_exact;
    hdr->has_best = r.has_best;
    hdr->has_best_exact = r.has_best_exact;
    hdr->width = r.width;
    hdr->best_lb = r.best_lb;
    hdr->value = r.value;
    hdr->ub = r.ub;
    hdr->depth = r.depth;
    hdr->best_value = r.best_value;
    hdr->best_exact_value = r.best_exact_value;
    hdr->nodes_expanded = r.nodes_expanded;
    hdr->arcs = r.arcs;
    hdr->layers = r.layers;
    hdr->n_cutset = r.cs_value.size();
    std::memcpy(state, r.state.data(), r.state.size() * sizeof(uint64_t));
}
void oracle_trace_get_cutset(void* t, uint64_t i, uint64_t* states, int64_t* value, int64_t* ub, uint64_t* depth) {
    const TraceRec& r = ((Trace*)t)->recs[i];
    std::memcpy(states, r.cs_states.data(), r.cs_states.size() * sizeof(uint64_t));
    std::memcpy(value, r.cs_value.data(), r.cs_value.size() * sizeof(int64_t));
    std::memcpy(ub, r.cs_ub.data(), r.cs_ub.size() * sizeof(int64_t));
    std::memcpy(depth, r.cs_depth.data(), r.cs_depth.size() * sizeof(uint64_t));
}

// ---- one compile() on an arbitrary residual sub-problem -----------------------------------------
/// comp_type: 0 Exact, 1 Relaxed, 2 Restricted.  Cut-set buffers sized by the caller (cap entries).
/// Returns the number of cut-set entries (or -1 when cap is too small).
}  // extern "C"
template <class DD>
static int64_t misp_compile_with(void* hh, int comp_type, uint64_t width, int64_t best_lb, const uint64_t* state,
                                 int64_t value, uint64_t depth, oracle_trace_hdr* hdr, uint64_t cap, uint64_t* cs_states,
                                 int64_t* cs_value, int64_t* cs_ub, uint64_t* cs_depth, int64_t* best_path,
                                 int64_t* n_best_path) {
    auto* h = (MispHandle*)hh;
    Misp& pb = h->pb;
    MispRelax relax(pb);
    MispRanking rank;
    NoCutoff nocut;
    EmptyCache<BitSet> cache;
    EmptyDominanceChecker<BitSet> dom;
    SubProblem<BitSet> node;
    BitSet s(pb.nb_vars);
    for (size_t k = 0; k < h->ws; ++k) s.w[k] = state[k];
    node.state = std::make_shared<const BitSet>(s);
    node.value = value;
    node.depth = depth;
    node.ub = ISIZE_MAX;
    CompilationInput<BitSet> in{(CompilationType)comp_type, &pb, &relax, &rank, &nocut, width, &node, best_lb,
                                &cache, &dom};
    DD mdd;
    auto c = mdd.compile(in);
    if (!c) return -2;
    Trace tr;
    record(tr, h->ws, node, (CompilationType)comp_type, width, best_lb, mdd, true);
    const TraceRec& r = tr.recs[0];
    hdr->comp_type = comp_type;
    hdr->is_exact = r.is_exact;
    hdr->has_best = r.has_best;
    hdr->has_best_exact = r.has_best_exact;
    hdr->width = width;
    hdr->best_lb = best_lb;
    hdr->value = value;
    hdr->ub = ISIZE_MAX;
    hdr->depth = depth;
    hdr->best_value = r.best_value;
    hdr->best_exact_value = r.best_exact_value;
    hdr->nodes_expanded = r.nodes_expanded;
    hdr->arcs = r.arcs;
    hdr->layers = r.layers;
    hdr->n_cutset = r.cs_value.size();
    if (n_best_path) {
        *n_best_path = 0;
        auto sol = mdd.best_solution();
        if (sol && best_path) {
            for (const Decision& d : *sol) {
                best_path[2 * *n_best_path] = (int64_t)d.variable;
                best_path[2 * *n_best_path + 1] = d.value;
                (*n_best_path)++;
            }
        }
    }
    if (r.cs_value.size() > cap) return -1;
    if (!r.cs_value.empty()) {
        std::memcpy(cs_states, r.cs_states.data(), r.cs_states.size() * sizeof(uint64_t));
        std::memcpy(cs_value, r.cs_value.data(), r.cs_value.size() * sizeof(int64_t));
        std::memcpy(cs_ub, r.cs_ub.data(), r.cs_ub.size() * sizeof(int64_t));
        std::memcpy(cs_depth, r.cs_depth.data(), r.cs_depth.size() * sizeof(uint64_t));
    }
    return (int64_t)r.cs_value.size();
}
extern "C" {
int64_t oracle_misp_compile(void* hh, int comp_type, uint64_t width, int64_t best_lb, const uint64_t* state, int64_t value, uint64_t depth,
                            oracle_trace_hdr* hdr, uint64_t cap, uint64_t* cs_states, int64_t* cs_value, int64_t* cs_ub, uint64_t* cs_depth,
                            int64_t* best_path, int64_t* n_best_path) {
    return misp_compile_with<DefaultMDDLEL<BitSet>>(hh, comp_type, width, best_lb, state, value, depth, hdr, cap, cs_states, cs_value, cs_ub, cs_depth,
                                                    best_path, n_best_path);
}
/// the same compile as a Pooled DD (mdd/pooled.rs): its cut-set is the frontier, whatever the DD type of the caller
int64_t oracle_misp_compile_pooled(void* hh, int comp_type, uint64_t width, int64_t best_lb, const uint64_t* state, int64_t value, uint64_t depth,
                                   oracle_trace_hdr* hdr, uint64_t cap, uint64_t* cs_states, int64_t* cs_value, int64_t* cs_ub, uint64_t* cs_depth,
                                   int64_t* best_path, int64_t* n_best_path) {
    return misp_compile_with<Pooled<BitSet>>(hh, comp_type, width, best_lb, state, value, depth, hdr, cap, cs_states, cs_value, cs_ub, cs_depth,
                                             best_path, n_best_path);
}

// ---- knapsack (config C1: plumbing, CPU only) ----------------------------------------------------
/// profits/weights of n items, capacity; width 0 -> NbUnassignedWidth.  Returns optimum (or -1).
int64_t oracle_knapsack_solve(int n, const int64_t* profit, const uint64_t* weight, uint64_t capacity, uint64_t width,
                              int nthreads, int64_t* sol_values, oracle_solve_out* out) {
    std::vector<isize> p(profit, profit + n);
    std::vector<size_t> wv(weight, weight + n);
    Knapsack pb(capacity, p, wv);
    KPRelax relax(pb);
    KPRanking rank;
    FixedWidth<KnapsackState> fixed(width);
    NbUnassignedWidth<KnapsackState> unassigned(pb.nb_variables());
    const WidthHeuristic<KnapsackState>& w = width ? (const WidthHeuristic<KnapsackState>&)fixed : unassigned;
    EmptyDominanceChecker<KnapsackState> dom;
    NoCutoff cut;
    MaxUB<KnapsackState> mx(rank);
    NoDupFringe<KnapsackState> fringe(mx);
    Completion c;
    std::optional<Solution> sol;
    auto t0 = std::chrono::steady_clock::now();
    if (nthreads <= 0) {
        SequentialSolver<KnapsackState> s(pb, relax, rank, w, dom, cut, fringe);
        c = s.maximize();
        sol = s.best_solution();
        if (out) {
            out->explored = s.explored();
            out->best_lb = s.best_lower_bound();
            out->best_ub = s.best_upper_bound();
            out->nodes_expanded = s.counters().nodes_expanded;
            out->arcs = s.counters().arcs;
            out->layers = s.counters().layers;
            out->compiles = s.counters().compiles;
        }
    } else {
        ParallelSolver<KnapsackState> s(pb, relax, rank, w, dom, cut, fringe, (size_t)nthreads);
        c = s.maximize();
        sol = s.best_solution();
        if (out) {
            out->explored = s.explored();
            out->best_lb = s.best_lower_bound();
            out->best_ub = s.best_upper_bound();
            out->nodes_expanded = s.counters().nodes_expanded;
            out->arcs = s.counters().arcs;
            out->layers = s.counters().layers;
            out->compiles = s.counters().compiles;
        }
    }
    auto t1 = std::chrono::steady_clock::now();
    if (out) {
        out->wall_s = std::chrono::duration<double>(t1 - t0).count();
        out->has_value = c.best_value.has_value();
        out->best_value = c.best_value.value_or(-1);
        out->is_exact = c.is_exact;
        out->n_solution = sol ? (int)sol->size() : 0;
    }
    if (sol && sol_values)
        for (const Decision& d : *sol) sol_values[d.variable] = d.value;
    return c.best_value.value_or(-1);
}
int64_t oracle_knapsack_solve_file(const char* path, uint64_t width, int nthreads, oracle_solve_out* out) {
    try {
        Knapsack pb = read_knapsack_instance(path);
        std::vector<int64_t> p(pb.profit.begin(), pb.profit.end());
        std::vector<uint64_t> w(pb.weight.begin(), pb.weight.end());
        return oracle_knapsack_solve((int)p.size(), p.data(), w.data(), pb.capacity, width, nthreads, nullptr, out);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "oracle_knapsack_solve_file: %s\n", e.what());
        return -2;
    }
}

/// MAX2SAT (examples/max2sat): solves a .wcnf file the way examples/max2sat/tests.rs:42-63 does (NoDupFringe, MaxUB,
/// EmptyDominanceChecker, NoCutoff; width 0 = NbUnassignedWidth).  nthreads <= 0: SequentialSolver.
/// sol_values (n entries, may be null) receives +1 / -1 per variable.  Returns the optimum, -1 when there is none.
int64_t oracle_max2sat_solve_file(const char* path, uint64_t width, int nthreads, double time_budget_s, int64_t* sol_values,
                                  oracle_solve_out* out) {
    try {
        Weighed2Sat inst = read_max2sat_instance(path);
        Max2Sat pb(inst);
        Max2SatRelax relax(pb);
        Max2SatRanking rank;
        FixedWidth<Max2SatState> fixed(width);
        NbUnassignedWidth<Max2SatState> unassigned(pb.nb_variables());
        const WidthHeuristic<Max2SatState>& w = width ? (const WidthHeuristic<Max2SatState>&)fixed : unassigned;
        EmptyDominanceChecker<Max2SatState> dom;
        NoCutoff nocut;
        TimeBudget budget(time_budget_s > 0 ? time_budget_s : 1e9);
        const Cutoff& cut = time_budget_s > 0 ? (const Cutoff&)budget : (const Cutoff&)nocut;
        MaxUB<Max2SatState> mx(rank);
        NoDupFringe<Max2SatState> fringe(mx);
        Completion c;
        std::optional<Solution> sol;
        auto t0 = std::chrono::steady_clock::now();
        auto fill = [&](auto& s) {
            c = s.maximize();
            sol = s.best_solution();
            if (out) {
                out->explored = s.explored();
                out->best_lb = s.best_lower_bound();
                out->best_ub = s.best_upper_bound();
                out->nodes_expanded = s.counters().nodes_expanded;
                out->arcs = s.counters().arcs;
                out->layers = s.counters().layers;
                out->compiles = s.counters().compiles;
            }
        };
        if (nthreads <= 0) {
            SequentialSolver<Max2SatState> s(pb, relax, rank, w, dom, cut, fringe);
            fill(s);
        } else {
            ParallelSolver<Max2SatState> s(pb, relax, rank, w, dom, cut, fringe, (size_t)nthreads);
            fill(s);
        }
        if (out) {
            out->has_value = c.best_value.has_value() ? 1 : 0;
            out->is_exact = c.is_exact ? 1 : 0;
            out->best_value = c.best_value.value_or(-1);
            out->wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            out->n_solution = sol ? (int)sol->size() : 0;
        }
        if (sol && sol_values) {
            for (size_t i = 0; i < pb.nb_variables(); ++i) sol_values[i] = 0;
            for (const Decision& d : *sol) sol_values[d.variable] = d.value;
        }
        return c.best_value.value_or(-1);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "oracle_max2sat_solve_file: %s\n", e.what());
        return -2;
    }
}
/// nb_vars and number of distinct clauses of a .wcnf file (data.rs:119-126 checks 3 and 4 on debug2.wcnf)
int oracle_max2sat_instance_info(const char* path, uint64_t* nb_vars, uint64_t* nb_clauses) {
    try {
        Weighed2Sat inst = read_max2sat_instance(path);
        *nb_vars = inst.nb_vars;
        *nb_clauses = inst.weights.size();
        return 0;
    } catch (const std::exception&) {
        return -2;
    }
}
/// total weight of the clauses satisfied by an assignment (+1 / -1 per variable): independent check of a solution
int64_t oracle_max2sat_evaluate(const char* path, const int64_t* values) {
    Weighed2Sat inst = read_max2sat_instance(path);
    int64_t total = 0;
    for (const auto& e : inst.weights) {
        auto sat = [&](isize lit) { return (lit > 0) == (values[(lit < 0 ? -lit : lit) - 1] > 0); };
        if (sat(e.first.a) || sat(e.first.b)) total += e.second;
    }
    return total;
}

/// MCP (examples/mcp): solves a .mcp file the way examples/mcp/tests.rs:36-62 does; sol_values receives +1 (side S) /
/// -1 (side T) per vertex.  Returns the optimum (-1: none, -2: io error).
int64_t oracle_mcp_solve_file(const char* path, uint64_t width, int nthreads, int64_t* sol_values, oracle_solve_out* out) {
    try {
        Mcp pb(read_mcp_instance(path));
        McpRelax relax(pb);
        McpRanking rank;
        FixedWidth<McpState> fixed(width);
        NbUnassignedWidth<McpState> unassigned(pb.nb_variables());
        const WidthHeuristic<McpState>& w = width ? (const WidthHeuristic<McpState>&)fixed : unassigned;
        EmptyDominanceChecker<McpState> dom;
        NoCutoff cut;
        MaxUB<McpState> mx(rank);
        NoDupFringe<McpState> fringe(mx);
        Completion c;
        std::optional<Solution> sol;
        auto t0 = std::chrono::steady_clock::now();
        auto fill = [&](auto& s) {
            c = s.maximize();
            sol = s.best_solution();
            if (out) {
                out->explored = s.explored();
                out->best_lb = s.best_lower_bound();
                out->best_ub = s.best_upper_bound();
                out->nodes_expanded = s.counters().nodes_expanded;
                out->arcs = s.counters().arcs;
                out->layers = s.counters().layers;
                out->compiles = s.counters().compiles;
            }
        };
        if (nthreads <= 0) {
            SequentialSolver<McpState> s(pb, relax, rank, w, dom, cut, fringe);
            fill(s);
        } else {
            ParallelSolver<McpState> s(pb, relax, rank, w, dom, cut, fringe, (size_t)nthreads);
            fill(s);
        }
        if (out) {
            out->has_value = c.best_value.has_value() ? 1 : 0;
            out->is_exact = c.is_exact ? 1 : 0;
            out->best_value = c.best_value.value_or(-1);
            out->wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            out->n_solution = sol ? (int)sol->size() : 0;
        }
        if (sol && sol_values) {
            for (size_t i = 0; i < pb.nb_variables(); ++i) sol_values[i] = 0;
            for (const Decision& d : *sol) sol_values[d.variable] = d.value;
        }
        return c.best_value.value_or(-1);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "oracle_mcp_solve_file: %s\n", e.what());
        return -2;
    }
}
/// number of vertices of a .mcp file and the weight of the cut defined by `sides` (+1 / -1 per vertex): independent check
int64_t oracle_mcp_cut_weight(const char* path, const int64_t* sides, uint64_t* nb_vertices) {
    McpGraph g = read_mcp_instance(path);
    if (nb_vertices) *nb_vertices = g.nb_vertices;
    int64_t total = 0;
    if (sides)
        for (size_t a = 0; a < g.nb_vertices; ++a)
            for (size_t b = a + 1; b < g.nb_vertices; ++b)
                if (sides[a] * sides[b] < 0) total += g.at(a, b);
    return total;
}

/// TSPTW (examples/tsptw): solves a .dat file the way examples/tsptw/tests.rs:32-57 does -- DefaultCachingSolver
/// (frontier cut-set + SimpleCache), SimpleDominanceChecker(TsptwDominance), TsptwWidth(n, factor), NoDupFringe, MaxUB.
/// tour (n entries, may be null) receives the visiting order.  Returns the best value (= -10000 x tour length), or
/// INT64_MIN when no tour exists / INT64_MIN + 1 on io errors.
int64_t oracle_tsptw_solve_file(const char* path, uint64_t width_factor, int nthreads, int64_t* tour, oracle_solve_out* out) {
    try {
        Tsptw pb(read_tsptw_instance(path));
        TsptwRelax relax(pb);
        TsptwRanking rank;
        TsptwWidth width(pb.nb_variables(), width_factor ? (size_t)width_factor : 1);
        SimpleDominanceChecker<TsptwState, TsptwDominance> dom(TsptwDominance(), pb.nb_variables());
        NoCutoff cut;
        MaxUB<TsptwState> mx(rank);
        NoDupFringe<TsptwState> fringe(mx);
        auto t0 = std::chrono::steady_clock::now();
        ParallelSolver<TsptwState, DefaultMDDFC<TsptwState>, SimpleCache<TsptwState>> s(pb, relax, rank, width, dom, cut, fringe,
                                                                                          (size_t)(nthreads > 0 ? nthreads : 1));
        Completion c = s.maximize();
        std::optional<Solution> sol = s.best_solution();
        if (out) {
            out->explored = s.explored();
            out->best_lb = s.best_lower_bound();
            out->best_ub = s.best_upper_bound();
            out->nodes_expanded = s.counters().nodes_expanded;
            out->arcs = s.counters().arcs;
            out->layers = s.counters().layers;
            out->compiles = s.counters().compiles;
            out->has_value = c.best_value.has_value() ? 1 : 0;
            out->is_exact = c.is_exact ? 1 : 0;
            out->best_value = c.best_value.value_or(0);
            out->wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            out->n_solution = sol ? (int)sol->size() : 0;
        }
        if (sol && tour) {
            for (size_t i = 0; i < pb.nb_variables(); ++i) tour[i] = -1;
            for (const Decision& d : *sol) tour[d.variable] = d.value;   // variable k = k-th move of the tour
        }
        return c.best_value ? (int64_t)*c.best_value : INT64_MIN;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "oracle_tsptw_solve_file: %s\n", e.what());
        return INT64_MIN + 1;
    }
}
/// Independent check of a tour (the visiting order after the depot, ending with 0): total travel + waiting time in
/// 1/10000 units, or -1 when a time window is missed / a node is visited twice.
int64_t oracle_tsptw_tour_length(const char* path, const int64_t* tour, uint64_t* nb_nodes) {
    TsptwInstance inst = read_tsptw_instance(path);
    if (nb_nodes) *nb_nodes = inst.nb_nodes;
    if (!tour) return -1;
    std::vector<char> seen(inst.nb_nodes, 0);
    size_t here = 0, now = 0;
    for (size_t k = 0; k < inst.nb_nodes; ++k) {
        if (tour[k] < 0 || tour[k] >= inst.nb_nodes) return -1;
        const size_t j = (size_t)tour[k];
        if (seen[j] || (j == 0 && k + 1 != inst.nb_nodes)) return -1;
        seen[j] = 1;
        now += inst.distances[here][j];
        if (now < inst.timewindows[j].earliest) now = inst.timewindows[j].earliest;
        if (now > inst.timewindows[j].latest) return -1;
        here = j;
    }
    return here == 0 ? (int64_t)now : -1;
}

}  // extern "C"
