// =============================================================================
// kat_main.cpp -- pins the CPU oracle against the reference's own known-answer
// tests.  Every case cites the reference test it restates (inputs and expected
// values only; fixtures are re-expressed, not copied).
//
// *** TEST INFRASTRUCTURE ***  Built by oracle/Makefile into oracle/_build/kat.
// Prints "ok <name>" / "FAIL <name>: ..." per case; exit status = #failures.
// =============================================================================
#include <cstdio>
#include <map>
#include <string>

#include "ddo_oracle.hpp"
#include "models.hpp"

using namespace ddo;

static int g_fail = 0;
static std::string g_case;
#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) {                                                                \
            std::printf("FAIL %s: %s (line %d)\n", g_case.c_str(), #cond, __LINE__);  \
            g_fail++;                                                                 \
            return;                                                                   \
        }                                                                             \
    } while (0)
#define CASE(name) static void name()
#define RUN(name)                                  \
    do {                                           \
        g_case = #name;                            \
        int before = g_fail;                       \
        name();                                    \
        if (g_fail == before) std::printf("ok %s\n", #name); \
    } while (0)
// a DD case instantiated for the default DD (clean.rs tests) and for the pooled one (the same-named tests of pooled.rs)
#define RUN_DD(name)                                                                         \
    do {                                                                                     \
        g_case = #name " [Mdd, clean.rs]";                                                   \
        int before = g_fail;                                                                 \
        name<DefaultMDDLEL<DummyState>, DefaultMDDFC<char>>();                               \
        if (g_fail == before) std::printf("ok %s\n", g_case.c_str());                        \
        g_case = #name " [Pooled, pooled.rs]";                                               \
        before = g_fail;                                                                     \
        name<Pooled<DummyState>, Pooled<char>>();                                            \
        if (g_fail == before) std::printf("ok %s\n", g_case.c_str());                        \
    } while (0)

// ---------------------------------------------------------------------------
// Fixtures of clean.rs:2552-2667 (DummyProblem & co)
// ---------------------------------------------------------------------------
struct DummyState {
    isize value;
    size_t depth;
    bool operator==(const DummyState& o) const { return value == o.value && depth == o.depth; }
};
namespace ddo {
template <>
struct StateHash<DummyState> {
    size_t operator()(const DummyState& s) const { return (size_t)(s.value * 1000003 + (isize)s.depth); }
};
template <>
struct StateHash<char> {
    size_t operator()(const char& c) const { return (size_t)c; }
};
}  // namespace ddo

struct DummyProblem : Problem<DummyState> {
    size_t nb_variables() const override { return 3; }
    isize initial_value() const override { return 0; }
    DummyState initial_state() const override { return {0, 0}; }
    DummyState transition(const DummyState& s, Decision d) const override { return {s.value + d.value, s.depth + 1}; }
    isize transition_cost(const DummyState&, const DummyState&, Decision d) const override { return d.value; }
    std::optional<Variable> next_variable(size_t depth, StateIter<DummyState>&) const override {
        if (depth < 3) return Variable{depth};
        return std::nullopt;
    }
    void for_each_in_domain(Variable var, const DummyState&, DecisionCallback& f) const override {
        for (isize d = 0; d <= 2; ++d) f.apply(Decision{var.id, d});
    }
};
struct DummyInfeasibleProblem : DummyProblem {
    void for_each_in_domain(Variable, const DummyState&, DecisionCallback&) const override {}
};
struct DummyRelax : Relaxation<DummyState> {
    DummyState merge(StateIter<DummyState>& s) const override {
        const DummyState* f = s.next();
        return {100, f->depth};
    }
    isize relax(const DummyState&, const DummyState&, const DummyState&, Decision, isize) const override { return 20; }
    isize fast_upper_bound(const DummyState& s) const override { return (isize)(3 - s.depth) * 10; }
};
struct DummyRanking : StateRanking<DummyState> {
    int compare(const DummyState& a, const DummyState& b) const override {
        return a.value < b.value ? 1 : (a.value > b.value ? -1 : 0);  // cmp().reverse()
    }
};

// Fixtures of clean.rs:2080-2181 (LocBoundsAndThresholdsExample*)
struct LocbPb : Problem<char> {
    size_t nb_variables() const override { return 4; }
    char initial_state() const override { return 'r'; }
    isize initial_value() const override { return 0; }
    std::optional<Variable> next_variable(size_t, StateIter<char>& it) const override {
        const char* c = it.next();
        switch (c ? *c : 'z') {
            case 'r': return Variable{0};
            case 'a': case 'b': return Variable{1};
            case 'c': case 'd': case 'M': case 'e': case 'f': return Variable{2};
            case 'g': case 'h': case 'i': return Variable{0};
            default: return std::nullopt;
        }
    }
    void for_each_in_domain(Variable var, const char& s, DecisionCallback& f) const override {
        std::vector<isize> v;
        switch (s) {
            case 'r': v = {10, 7}; break;
            case 'a': v = {2}; break;
            case 'b': v = {3, 6, 5}; break;
            case 'M': v = {4}; break;
            case 'e': v = {0}; break;
            case 'f': v = {1, 2}; break;
            case 'g': case 'h': case 'i': v = {0}; break;
            default: break;
        }
        for (isize x : v) f.apply(Decision{var.id, x});
    }
    char transition(const char& s, Decision d) const override {
        if (s == 'r' && d.value == 10) return 'a';
        if (s == 'r' && d.value == 7) return 'b';
        if (s == 'a' && d.value == 2) return 'c';
        if (s == 'b' && d.value == 3) return 'd';
        if (s == 'b' && d.value == 6) return 'e';
        if (s == 'b' && d.value == 5) return 'f';
        if (s == 'M' && d.value == 4) return 'g';
        if (s == 'e' && d.value == 0) return 'h';
        if (s == 'f' && d.value == 1) return 'h';
        if (s == 'f' && d.value == 2) return 'i';
        return 't';
    }
    isize transition_cost(const char&, const char&, Decision d) const override { return d.value; }
};
struct LocbRelax : Relaxation<char> {
    char merge(StateIter<char>&) const override { return 'M'; }
    isize relax(const char&, const char&, const char&, Decision, isize cost) const override { return cost; }
    isize fast_upper_bound(const char& s) const override {
        switch (s) {
            case 'r': return 30;
            case 'a': case 'b': return 20;
            case 'M': case 'e': case 'f': return 10;
            default: return 0;
        }
    }
};
struct CmpChar : StateRanking<char> {
    int compare(const char& a, const char& b) const override { return a < b ? -1 : (a > b ? 1 : 0); }
};

template <class S>
static SubProblem<S> root_of(S s) {
    SubProblem<S> r;
    r.state = std::make_shared<const S>(s);
    r.value = 0;
    r.ub = ISIZE_MAX;
    r.depth = 0;
    return r;
}

struct DummyEnv {
    DummyProblem pb;
    DummyInfeasibleProblem infeasible;
    DummyRelax relax;
    DummyRanking rank;
    NoCutoff nocut;
    CutoffAlways always;
    EmptyCache<DummyState> ecache;
    SimpleCache<DummyState> scache;
    EmptyDominanceChecker<DummyState> dom;
    SubProblem<DummyState> root = root_of(DummyState{0, 0});
    CompilationInput<DummyState> input(CompilationType t, size_t w, isize lb, bool simple_cache = false,
                                       bool infeas = false, bool cut = false) {
        return CompilationInput<DummyState>{t, infeas ? (Problem<DummyState>*)&infeasible : &pb, &relax, &rank,
                                            cut ? (Cutoff*)&always : (Cutoff*)&nocut, w, &root, lb,
                                            simple_cache ? (Cache<DummyState>*)&scache : &ecache, &dom};
    }
};

static bool sol_eq(const Solution& s, std::initializer_list<Decision> e) { return s == Solution(e); }

// clean.rs:1155-1188
template <class DD, class DDC>
static void exact_completely_unrolls_the_mdd_no_matter_its_width() {
    DummyEnv e;
    DD mdd;
    CHECK(mdd.compile(e.input(CompilationType::Exact, 1, ISIZE_MIN)).has_value());
    CHECK(mdd.best_solution().has_value());
    CHECK(mdd.best_value() == std::optional<isize>(6));
    CHECK(sol_eq(*mdd.best_solution(), {{2, 2}, {1, 2}, {0, 2}}));
}
// clean.rs:1191-1224
template <class DD, class DDC>
static void restricted_drops_the_less_interesting_nodes() {
    DummyEnv e;
    DD mdd;
    CHECK(mdd.compile(e.input(CompilationType::Restricted, 1, ISIZE_MIN)).has_value());
    CHECK(mdd.best_value() == std::optional<isize>(6));
    CHECK(sol_eq(*mdd.best_solution(), {{2, 2}, {1, 2}, {0, 2}}));
}
// clean.rs:1227-1315
template <class DD, class DDC>
static void completion_must_be_coherent_with_outcome() {
    for (CompilationType t : {CompilationType::Exact, CompilationType::Restricted, CompilationType::Relaxed}) {
        DummyEnv e;
        DD mdd;
        auto c = mdd.compile(e.input(t, 1, ISIZE_MIN));
        CHECK(c.has_value());
        CHECK(c->is_exact == mdd.is_exact());
        CHECK(c->best_value == mdd.best_value());
    }
}
// clean.rs:1323-1403
template <class DD, class DDC>
static void fails_with_cutoff_when_cutoff_occurs() {
    for (CompilationType t : {CompilationType::Exact, CompilationType::Restricted, CompilationType::Relaxed}) {
        DummyEnv e;
        DD mdd;
        Reason why;
        auto c = mdd.compile(e.input(t, 1, ISIZE_MIN, false, false, true), &why);
        CHECK(!c.has_value());
        CHECK(why == Reason::CutoffOccurred);
    }
}
// clean.rs:1406-1440
template <class DD, class DDC>
static void relaxed_merges_the_less_interesting_nodes() {
    DummyEnv e;
    DD mdd;
    CHECK(mdd.compile(e.input(CompilationType::Relaxed, 1, ISIZE_MIN)).has_value());
    CHECK(mdd.best_value() == std::optional<isize>(24));
    CHECK(sol_eq(*mdd.best_solution(), {{2, 2}, {1, 0}, {0, 2}}));
}
// clean.rs:1443-1471
template <class DD, class DDC>
static void relaxed_populates_the_cutset_and_will_not_squash_first_layer() {
    DummyEnv e;
    DD mdd;
    CHECK(mdd.compile(e.input(CompilationType::Relaxed, 1, ISIZE_MIN)).has_value());
    size_t n = 0;
    mdd.drain_cutset([&](SubProblem<DummyState>) { n++; });
    CHECK(n == 3);
}
// clean.rs:1474-1614
template <class DD, class DDC>
static void exactness_flags() {
    {   // an_exact_mdd_must_be_exact
        DummyEnv e; DD mdd;
        CHECK(mdd.compile(e.input(CompilationType::Exact, 1, ISIZE_MIN)).has_value());
        CHECK(mdd.is_exact());
    }
    {   // a_relaxed_mdd_is_exact_as_long_as_no_merge_occurs (w = 10)
        DummyEnv e; DD mdd;
        CHECK(mdd.compile(e.input(CompilationType::Relaxed, 10, ISIZE_MIN)).has_value());
        CHECK(mdd.is_exact());
    }
    {   // a_relaxed_mdd_is_not_exact_when_a_merge_occurred (w = 1)
        DummyEnv e; DD mdd;
        CHECK(mdd.compile(e.input(CompilationType::Relaxed, 1, ISIZE_MIN)).has_value());
        CHECK(!mdd.is_exact());
    }
    {   // a_restricted_mdd_is_exact_as_long_as_no_restriction_occurs (w = 10)
        DummyEnv e; DD mdd;
        CHECK(mdd.compile(e.input(CompilationType::Restricted, 10, ISIZE_MIN)).has_value());
        CHECK(mdd.is_exact());
    }
    {   // a_restricted_mdd_is_not_exact_when_a_restriction_occurred (w = 1)
        DummyEnv e; DD mdd;
        CHECK(mdd.compile(e.input(CompilationType::Restricted, 1, ISIZE_MIN)).has_value());
        CHECK(!mdd.is_exact());
    }
}
// clean.rs:1616-1668
template <class DD, class DDC>
static void when_the_problem_is_infeasible_there_is_no_solution() {
    DummyEnv e;
    DD mdd;
    CHECK(mdd.compile(e.input(CompilationType::Exact, 10, ISIZE_MIN, false, true)).has_value());
    CHECK(!mdd.best_solution().has_value());
    CHECK(!mdd.best_value().has_value());
}
// clean.rs:1670-1749
template <class DD, class DDC>
static void skips_node_with_an_ub_less_than_best_known_lb() {
    for (CompilationType t : {CompilationType::Exact, CompilationType::Restricted, CompilationType::Relaxed}) {
        DummyEnv e;
        DD mdd;
        CHECK(mdd.compile(e.input(t, (size_t)-1, 1000)).has_value());
        CHECK(!mdd.best_solution().has_value());
    }
}
// clean.rs:1751-1843
template <class DD, class DDC>
static void skips_nodes_with_a_value_less_than_known_threshold() {
    for (CompilationType t : {CompilationType::Exact, CompilationType::Restricted, CompilationType::Relaxed}) {
        DummyEnv e;
        e.scache.initialize(e.pb);
        for (isize v = 0; v <= 2; ++v)
            e.scache.update_threshold(std::make_shared<const DummyState>(DummyState{v, 1}), 1, v, true);
        DD mdd;
        CHECK(mdd.compile(e.input(t, (size_t)-1, ISIZE_MIN, true)).has_value());
        CHECK(!mdd.best_solution().has_value());
    }
}
// clean.rs:1845-1948
template <class DD, class DDC>
static void computes_thresholds_when_exact() {
    for (CompilationType t : {CompilationType::Restricted, CompilationType::Relaxed}) {
        DummyEnv e;
        e.scache.initialize(e.pb);
        DD mdd;
        CHECK(mdd.compile(e.input(t, 10, ISIZE_MIN, true)).has_value());
        CHECK(mdd.is_exact());
        for (size_t depth = 0; depth <= 3; ++depth)
            for (isize v = 0; v <= (isize)(2 * depth); ++v) {
                auto th = e.scache.get_threshold(DummyState{v, depth}, depth);
                CHECK(th.has_value());
                CHECK(th->value == (isize)(2 * depth));
                CHECK(th->explored);
            }
    }
}
// clean.rs:1951-2055
template <class DD, class DDC>
static void computes_thresholds_when_all_pruned() {
    for (CompilationType t : {CompilationType::Restricted, CompilationType::Relaxed}) {
        DummyEnv e;
        e.scache.initialize(e.pb);
        DD mdd;
        CHECK(mdd.compile(e.input(t, 10, 15, true)).has_value());
        CHECK(mdd.is_exact());
        for (size_t depth = 0; depth <= 2; ++depth)
            for (isize v = 0; v <= (isize)(2 * depth); ++v) {
                auto th = e.scache.get_threshold(DummyState{v, depth}, depth);
                CHECK(th.has_value());
                CHECK(th->value == (isize)(1 + 2 * depth));
                CHECK(th->explored);
            }
        for (isize v = 0; v <= 6; ++v) CHECK(!e.scache.get_threshold(DummyState{v, 3}, 3).has_value());
    }
}

struct LocbEnv {
    LocbPb pb;
    LocbRelax relax;
    CmpChar rank;
    NoCutoff nocut;
    SimpleCache<char> cache;
    EmptyDominanceChecker<char> dom;
    SubProblem<char> root = root_of('r');
    LocbEnv() { cache.initialize(pb); }
    CompilationInput<char> input(isize lb) {
        return CompilationInput<char>{CompilationType::Relaxed, &pb, &relax, &rank, &nocut, 3, &root, lb, &cache, &dom};
    }
    bool th(char c, size_t d, isize v, bool explored) {
        auto t = cache.get_threshold(c, d);
        return t && t->value == v && t->explored == explored;
    }
    bool none(char c, size_t d) { return !cache.get_threshold(c, d).has_value(); }
};

// clean.rs:2184-2242 (LEL cut-set)
CASE(relaxed_computes_local_bounds_and_thresholds_1) {
    LocbEnv e;
    DefaultMDDLEL<char> mdd;
    CHECK(mdd.compile(e.input(0)).has_value());
    CHECK(!mdd.is_exact());
    CHECK(mdd.best_value() == std::optional<isize>(16));
    std::map<char, isize> v;
    mdd.drain_cutset([&](SubProblem<char> n) { v[*n.state] = n.ub; });
    CHECK(v.size() == 2);
    CHECK(v['a'] == 16);
    CHECK(v['b'] == 14);
    CHECK(e.th('r', 0, 0, true));
    CHECK(e.th('a', 1, 10, false));
    CHECK(e.th('b', 1, 7, false));
    CHECK(e.none('M', 2) && e.none('e', 2) && e.none('f', 2));
    CHECK(e.none('g', 3) && e.none('h', 3) && e.none('i', 3) && e.none('t', 4));
}
// clean.rs:2245-2321 (FRONTIER cut-set)
template <class DD, class DDC>
static void relaxed_computes_local_bounds_and_thresholds_2() {
    LocbEnv e;
    DDC mdd;
    CHECK(mdd.compile(e.input(0)).has_value());
    CHECK(!mdd.is_exact());
    CHECK(mdd.best_value() == std::optional<isize>(16));
    std::map<char, isize> v;
    mdd.drain_cutset([&](SubProblem<char> n) { v[*n.state] = n.ub; });
    CHECK(v.size() == 4);
    CHECK(v['a'] == 16);
    CHECK(v['b'] == 14);
    CHECK(v['h'] == 13);
    CHECK(v['i'] == 14);
    CHECK(e.th('r', 0, 0, true));
    CHECK(e.th('a', 1, 10, false));
    CHECK(e.th('b', 1, 7, false));
    CHECK(e.none('M', 2));
    CHECK(e.th('e', 2, 13, true));
    CHECK(e.th('f', 2, 12, true));
    CHECK(e.none('g', 3));
    CHECK(e.th('h', 3, 13, false));
    CHECK(e.th('i', 3, 14, false));
    CHECK(e.none('t', 4));
}
// clean.rs:2324-2398 (FRONTIER cut-set, best_lb = 15)
template <class DD, class DDC>
static void relaxed_computes_local_bounds_and_thresholds_with_pruning() {
    LocbEnv e;
    DDC mdd;
    CHECK(mdd.compile(e.input(15)).has_value());
    CHECK(!mdd.is_exact());
    CHECK(mdd.best_value() == std::optional<isize>(16));
    std::map<char, isize> v;
    mdd.drain_cutset([&](SubProblem<char> n) { v[*n.state] = n.ub; });
    CHECK(v.size() == 2);
    CHECK(v['a'] == 16);
    CHECK(v['b'] == 14);
    CHECK(e.th('r', 0, 0, true));
    CHECK(e.th('a', 1, 10, false));
    CHECK(e.th('b', 1, 8, false));
    CHECK(e.none('M', 2));
    CHECK(e.th('e', 2, 15, true));
    CHECK(e.th('f', 2, 13, true));
    CHECK(e.none('g', 3));
    CHECK(e.th('h', 3, 15, true));
    CHECK(e.th('i', 3, 15, true));
    CHECK(e.none('t', 4));
}

// ---------------------------------------------------------------------------
// Fringe KATs (usize states ranked by value: fringe/no_duplicate.rs:326-663)
// ---------------------------------------------------------------------------
struct UsizeRanking : StateRanking<size_t> {   // no_duplicate.rs:528-535
    int compare(const size_t& a, const size_t& b) const override { return a < b ? -1 : (a > b ? 1 : 0); }
};
static SubProblem<size_t> fnode(size_t state, isize value, isize ub, std::vector<Decision> path = {}, size_t depth = 0) {
    SubProblem<size_t> s;   // no_duplicate.rs:655-663
    s.state = std::make_shared<const size_t>(state);
    s.value = value;
    s.ub = ub;
    s.path = std::move(path);
    s.depth = depth;
    return s;
}
template <class F>
static std::vector<size_t> pop_all(F& f) {
    std::vector<size_t> out;
    while (auto n = f.pop()) out.push_back(*n->state);
    return out;
}
// no_duplicate.rs:405-410
CASE(nodup_pop_off_an_empty_fringe_is_none) {
    UsizeRanking rank;
    MaxUB<size_t> mx(rank);
    NoDupFringe<size_t> f(mx);
    CHECK(!f.pop().has_value());
    CHECK(f.is_empty());
    f.clear();
    CHECK(f.is_empty());
    f.push(fnode(42, 0, 0));
    CHECK(!f.is_empty());
    f.clear();
    CHECK(f.is_empty());
}
// no_duplicate.rs:413-470: largest ub then value; the lower-value duplicate of state 5 never shows up
CASE(nodup_pops_largest_ub_then_lp) {
    UsizeRanking rank;
    MaxUB<size_t> mx(rank);
    NoDupFringe<size_t> f(mx);
    f.push(fnode(1, 1, 1));
    f.push(fnode(5, 5, 5));   // f
    f.push(fnode(2, 2, 2));
    f.push(fnode(4, 4, 4));
    f.push(fnode(3, 3, 3));
    f.push(fnode(5, 4, 5));   // e: same state as f, lower value
    CHECK(f.len() == 5);
    auto n = f.pop();
    CHECK(*n->state == 5 && n->value == 5 && n->ub == 5);
    for (size_t k : {4, 3, 2, 1}) {
        n = f.pop();
        CHECK(*n->state == k && n->value == (isize)k && n->ub == (isize)k);
    }
    CHECK(!f.pop().has_value());
}
// no_duplicate.rs:474-506: of two copies the one with the longest path (value) is kept, with ITS path
CASE(nodup_keeps_the_copy_with_longest_path) {
    UsizeRanking rank;
    MaxUB<size_t> mx(rank);
    NoDupFringe<size_t> f(mx);
    f.push(fnode(5, 4, 5, {{0, 4}}, 1));
    f.push(fnode(5, 5, 5, {{1, 5}}, 1));
    auto n = f.pop();
    CHECK(n->value == 5 && n->ub == 5 && n->depth == 1);
    CHECK(n->path.size() == 1 && n->path[0].variable == 1 && n->path[0].value == 5);
}
// no_duplicate.rs:555-575
CASE(nodup_popped_in_order) {
    UsizeRanking rank;
    MaxUB<size_t> mx(rank);
    NoDupFringe<size_t> f(mx);
    for (size_t i = 1; i <= 5; ++i) f.push(fnode(i, 10, 99 + (isize)i));
    CHECK(f.len() == 5);
    CHECK((pop_all(f) == std::vector<size_t>{5, 4, 3, 2, 1}));
    CHECK(f.len() == 0 && f.is_empty());
}
// no_duplicate.rs:577-601
CASE(nodup_pushing_same_node_multiple_times_does_not_alter_pop_order) {
    UsizeRanking rank;
    MaxUB<size_t> mx(rank);
    NoDupFringe<size_t> f(mx);
    for (int rep = 0; rep < 5; ++rep)
        for (size_t i = 1; i <= 5; ++i) f.push(fnode(i, 10, 99 + (isize)i));
    CHECK(f.len() == 5);
    CHECK((pop_all(f) == std::vector<size_t>{5, 4, 3, 2, 1}));
}
// no_duplicate.rs:603-640
CASE(nodup_pushing_nodes_triggers_reordering_if_lplen_is_better_up) {
    UsizeRanking rank;
    MaxUB<size_t> mx(rank);
    NoDupFringe<size_t> f(mx);
    for (size_t i = 1; i <= 5; ++i) f.push(fnode(i, 10, 99 + (isize)i));     // ub 100..104
    for (size_t i = 1; i <= 5; ++i) f.push(fnode(i, 15, 101 - (isize)i));    // ub 100..96
    for (size_t i = 1; i <= 5; ++i) f.push(fnode(i, 20, 91 + (isize)i));     // ub 92..96
    CHECK(f.len() == 5);
    CHECK((pop_all(f) == std::vector<size_t>{5, 4, 3, 2, 1}));
}
// simple.rs:173-226 + subproblem_ranking.rs:44-75 (MaxUB doc example)
CASE(simple_fringe_pops_by_ub_then_value) {
    CmpChar rank;
    MaxUB<char> mx(rank);
    SimpleFringe<char> f(mx);
    auto sp = [](char c, isize value, isize ub) {
        SubProblem<char> s;
        s.state = std::make_shared<const char>(c);
        s.value = value;
        s.ub = ub;
        return s;
    };
    f.push(sp('a', 42, 300));
    f.push(sp('b', 2, 100));
    f.push(sp('c', 24, 150));
    f.push(sp('d', 13, 13));
    f.push(sp('e', 65, 700));
    f.push(sp('f', 19, 100));
    for (char c : {'e', 'a', 'c', 'f', 'b', 'd'}) CHECK(*f.pop()->state == c);
    CHECK(f.is_empty());
}

// ---------------------------------------------------------------------------
// Width KATs (width.rs:884-1075)
// ---------------------------------------------------------------------------
CASE(width_heuristics) {
    SubProblem<char> s;
    s.state = std::make_shared<const char>('a');
    s.path = {{0, 0}, {1, 0}};
    FixedWidth<char> fw(5);
    NbUnassignedWidth<char> nu(5);
    CHECK(fw.max_width(s) == 5);
    CHECK(nu.max_width(s) == 3);
    Times<char> t(2, nu);
    CHECK(t.max_width(s) == 6);
    Times<char> t0(0, nu);
    CHECK(t0.max_width(s) == 1);  // never 0
    DivBy<char> d(2, nu);
    CHECK(d.max_width(s) == 1);
    DivBy<char> d9(9, nu);
    CHECK(d9.max_width(s) == 1);  // never 0
}

// ---------------------------------------------------------------------------
// Solver KATs: knapsack README instance (parallel.rs:902-1151, sequential.rs:539-1092, README.md:246-292)
// ---------------------------------------------------------------------------
template <class SolverT>
static bool solve_kp(Knapsack& pb, size_t w, isize expect, std::vector<isize> expect_sol, bool parallel) {
    KPRelax relax(pb);
    KPRanking rank;
    FixedWidth<KnapsackState> fixed(w);
    NbUnassignedWidth<KnapsackState> unassigned(pb.nb_variables());
    const WidthHeuristic<KnapsackState>& width = w ? (const WidthHeuristic<KnapsackState>&)fixed : unassigned;
    EmptyDominanceChecker<KnapsackState> dom;
    NoCutoff cut;
    MaxUB<KnapsackState> mx(rank);
    SimpleFringe<KnapsackState> fringe(mx);
    SolverT solver(pb, relax, rank, width, dom, cut, fringe);
    (void)parallel;
    Completion c = solver.maximize();
    if (!c.is_exact || c.best_value != std::optional<isize>(expect)) return false;
    auto sol = solver.best_solution();
    if (!sol) return false;
    if (!expect_sol.empty()) {
        if (sol->size() != expect_sol.size()) return false;
        for (size_t i = 0; i < sol->size(); ++i)
            if ((*sol)[i].variable != i || (*sol)[i].value != expect_sol[i]) return false;
    }
    return solver.best_lower_bound() == expect && solver.best_upper_bound() == expect && solver.gap() == 0.0;
}
CASE(knapsack_readme_instance_220) {
    Knapsack pb(50, {60, 100, 120}, {10, 20, 30});
    using Seq = SequentialSolver<KnapsackState>;
    using SeqFc = SequentialSolver<KnapsackState, DefaultMDDFC<KnapsackState>, SimpleCache<KnapsackState>>;
    using SeqLelCache = SequentialSolver<KnapsackState, DefaultMDDLEL<KnapsackState>, SimpleCache<KnapsackState>>;
    using SeqFcNoCache = SequentialSolver<KnapsackState, DefaultMDDFC<KnapsackState>, EmptyCache<KnapsackState>>;
    using Par = ParallelSolver<KnapsackState>;
    using ParFc = ParallelSolver<KnapsackState, DefaultMDDFC<KnapsackState>, SimpleCache<KnapsackState>>;
    // w = 0 selects NbUnassignedWidth, the policy of the reference tests
    CHECK(solve_kp<Seq>(pb, 0, 220, {0, 1, 1}, false));
    CHECK(solve_kp<SeqFc>(pb, 0, 220, {0, 1, 1}, false));
    CHECK(solve_kp<SeqLelCache>(pb, 0, 220, {0, 1, 1}, false));
    CHECK(solve_kp<SeqFcNoCache>(pb, 0, 220, {0, 1, 1}, false));
    CHECK(solve_kp<Par>(pb, 0, 220, {0, 1, 1}, true));
    CHECK(solve_kp<ParFc>(pb, 0, 220, {0, 1, 1}, true));
    // solver/mod.rs:34-47: Seq/Par(No)CachingSolverPooled (the knapsack model impacts every state: Pooled == the default DD here)
    CHECK((solve_kp<SequentialSolver<KnapsackState, Pooled<KnapsackState>, EmptyCache<KnapsackState>>>(pb, 0, 220, {0, 1, 1}, false)));
    CHECK((solve_kp<SequentialSolver<KnapsackState, Pooled<KnapsackState>, SimpleCache<KnapsackState>>>(pb, 0, 220, {0, 1, 1}, false)));
    CHECK((solve_kp<ParallelSolver<KnapsackState, Pooled<KnapsackState>, EmptyCache<KnapsackState>>>(pb, 0, 220, {0, 1, 1}, true)));
    CHECK((solve_kp<ParallelSolver<KnapsackState, Pooled<KnapsackState>, SimpleCache<KnapsackState>>>(pb, 0, 220, {0, 1, 1}, true)));
    CHECK(solve_kp<Seq>(pb, 1, 220, {0, 1, 1}, false));
    CHECK(solve_kp<Seq>(pb, 100, 220, {0, 1, 1}, false));   // README.md:246-292 uses FixedWidth(100)
}
// parallel.rs 7-item instance ("maximizes_yields_the_optimum_2")
CASE(knapsack_seven_items) {
    Knapsack pb(50, {60, 210, 12, 5, 100, 120, 110}, {10, 45, 20, 4, 20, 30, 50});
    using Seq = SequentialSolver<KnapsackState>;
    using ParFc = ParallelSolver<KnapsackState, DefaultMDDFC<KnapsackState>, SimpleCache<KnapsackState>>;
    using Par = ParallelSolver<KnapsackState>;
    CHECK(solve_kp<Seq>(pb, 0, 220, {0, 0, 0, 0, 1, 1, 0}, false));
    CHECK(solve_kp<Par>(pb, 0, 220, {0, 0, 0, 0, 1, 1, 0}, true));
    CHECK(solve_kp<ParFc>(pb, 0, 220, {0, 0, 0, 0, 1, 1, 0}, true));
    CHECK(solve_kp<Seq>(pb, 2, 220, {0, 0, 0, 0, 1, 1, 0}, false));
}

// ---------------------------------------------------------------------------
// BitSet ordering (bit-set 0.5.3 Ord = lexicographic over ascending members)
// ---------------------------------------------------------------------------
static int naive_cmp(const BitSet& a, const BitSet& b) {
    std::vector<size_t> x, y;
    a.for_each([&](size_t i) { x.push_back(i); });
    b.for_each([&](size_t i) { y.push_back(i); });
    if (x < y) return -1;
    if (y < x) return 1;
    return 0;
}
CASE(bitset_ord_is_lexicographic_on_members) {
    uint64_t seed = 0x9E3779B97F4A7C15ULL;
    auto rnd = [&]() {
        seed ^= seed << 13;
        seed ^= seed >> 7;
        seed ^= seed << 17;
        return seed;
    };
    for (int it = 0; it < 20000; ++it) {
        BitSet a(130), b(130);
        for (int k = 0; k < 3; ++k) {
            uint64_t m = rnd() & rnd();
            if (it % 3 == 0) m &= rnd();
            a.w[k] = m;
            b.w[k] = (it % 5 == 0) ? m : (rnd() & rnd());
        }
        if (it % 7 == 0) b.w[2] = a.w[2], b.w[1] = a.w[1];
        a.w[2] &= 3;
        b.w[2] &= 3;
        CHECK(a.cmp(b) == naive_cmp(a, b));
        CHECK(b.cmp(a) == -naive_cmp(a, b));
    }
}


// ---------------------------------------------------------------------------
// Pooled on a model that implements is_impacted_by (MISP, examples/misp/main.rs:145-147): the reference has no unit test with long
// arcs (its Dummy fixtures impact every state), so these are PROPERTIES, checked against brute force on seeded random graphs:
// exact Pooled == the optimum == exact default DD; restricted <= optimum <= relaxed; the B&B over Pooled DDs (frontier cut-set,
// with and without cache) proves the optimum; nodes that the variable does not impact are not copied (fewer expansions).
// ---------------------------------------------------------------------------
static Misp random_misp(size_t n, unsigned seed, unsigned pct_edge) {
    Misp pb;
    pb.nb_vars = n;
    pb.weight.resize(n);
    uint64_t x = seed * 2654435761u + 12345u;
    auto rnd = [&]() { x = x * 6364136223846793005ULL + 1442695040888963407ULL; return (unsigned)(x >> 33); };
    for (size_t i = 0; i < n; ++i) pb.weight[i] = 1 + (isize)(rnd() % 5);
    pb.neighbors.assign(n, BitSet::full(n));          // complement rows: start from "compatible with everything"
    for (size_t a = 0; a < n; ++a)
        for (size_t b = a + 1; b < n; ++b)
            if (rnd() % 100 < pct_edge) {              // an edge a-b: not compatible
                pb.neighbors[a].remove(b);
                pb.neighbors[b].remove(a);
            }
    return pb;
}
static isize brute_force_misp(const Misp& pb) {
    const size_t n = pb.nb_vars;
    isize best = 0;
    for (uint32_t m = 0; m < (1u << n); ++m) {
        isize v = 0;
        bool ok = true;
        for (size_t a = 0; a < n && ok; ++a) {
            if (!((m >> a) & 1)) continue;
            v += pb.weight[a];
            for (size_t b = a + 1; b < n; ++b)
                if (((m >> b) & 1) && !pb.neighbors[a].contains(b)) { ok = false; break; }
        }
        if (ok && v > best) best = v;
    }
    return best;
}
template <class SolverT>
static std::optional<isize> solve_misp_with(const Misp& pb, size_t w) {
    MispRelax relax(pb);
    MispRanking rank;
    FixedWidth<BitSet> width(w);
    EmptyDominanceChecker<BitSet> dom;
    NoCutoff cut;
    MaxUB<BitSet> mx(rank);
    SimpleFringe<BitSet> fringe(mx);
    SolverT solver(pb, relax, rank, width, dom, cut, fringe);
    Completion c = solver.maximize();
    if (!c.is_exact) return std::nullopt;
    return c.best_value;
}
CASE(pooled_long_arcs_on_misp) {
    for (unsigned seed = 1; seed <= 6; ++seed) {
        Misp pb = random_misp(12 + seed % 3, seed, 20 + 10 * (seed % 4));
        const isize opt = brute_force_misp(pb);
        MispRelax relax(pb);
        MispRanking rank;
        NoCutoff nocut;
        EmptyCache<BitSet> cache;
        EmptyDominanceChecker<BitSet> dom;
        SubProblem<BitSet> root = root_of(pb.initial_state());
        auto input = [&](CompilationType t, size_t w) {
            return CompilationInput<BitSet>{t, &pb, &relax, &rank, &nocut, w, &root, ISIZE_MIN, &cache, &dom};
        };
        Pooled<BitSet> pooled;
        DefaultMDDLEL<BitSet> plain;
        CHECK(pooled.compile(input(CompilationType::Exact, 1)).has_value());
        CHECK(plain.compile(input(CompilationType::Exact, 1)).has_value());
        CHECK(pooled.is_exact() && pooled.best_value() == std::optional<isize>(opt));
        CHECK(plain.best_value() == std::optional<isize>(opt));
        CHECK(pooled.last_counters.nodes_expanded < plain.last_counters.nodes_expanded);   // long arcs: no copies of unaffected nodes
        // the best path is a solution of that value
        isize v = 0;
        const Solution best = *pooled.best_solution();
        for (const Decision& d : best) v += d.value == MISP_YES ? pb.weight[d.variable] : 0;
        CHECK(v == opt);
        for (size_t w : {2, 3, 5}) {
            Pooled<BitSet> r, x;
            CHECK(r.compile(input(CompilationType::Restricted, w)).has_value());
            CHECK(x.compile(input(CompilationType::Relaxed, w)).has_value());
            CHECK(!r.best_value() || *r.best_value() <= opt);
            CHECK(x.best_value() && *x.best_value() >= opt);
            if (x.is_exact()) CHECK(*x.best_value() == opt);
            // every cut-set node is an exact node with a bound that does not exceed the relaxed DD's
            x.drain_cutset([&](SubProblem<BitSet> n) {
                if (n.ub > *x.best_value() || n.value > opt) g_fail++;
            });
            using SeqP = SequentialSolver<BitSet, Pooled<BitSet>, EmptyCache<BitSet>>;
            using SeqPC = SequentialSolver<BitSet, Pooled<BitSet>, SimpleCache<BitSet>>;
            using ParP = ParallelSolver<BitSet, Pooled<BitSet>, EmptyCache<BitSet>>;
            CHECK(solve_misp_with<SeqP>(pb, w) == std::optional<isize>(opt));    // SeqNoCachingSolverPooled
            CHECK(solve_misp_with<SeqPC>(pb, w) == std::optional<isize>(opt));   // SeqCachingSolverPooled
            CHECK(solve_misp_with<ParP>(pb, w) == std::optional<isize>(opt));    // ParNoCachingSolverPooled
        }
    }
}

int main(int argc, char** argv) {
    (void)argc;
    (void)argv;
    RUN_DD(exact_completely_unrolls_the_mdd_no_matter_its_width);
    RUN_DD(restricted_drops_the_less_interesting_nodes);
    RUN_DD(completion_must_be_coherent_with_outcome);
    RUN_DD(fails_with_cutoff_when_cutoff_occurs);
    RUN_DD(relaxed_merges_the_less_interesting_nodes);
    RUN_DD(relaxed_populates_the_cutset_and_will_not_squash_first_layer);
    RUN_DD(exactness_flags);
    RUN_DD(when_the_problem_is_infeasible_there_is_no_solution);
    RUN_DD(skips_node_with_an_ub_less_than_best_known_lb);
    RUN_DD(skips_nodes_with_a_value_less_than_known_threshold);
    RUN_DD(computes_thresholds_when_exact);
    RUN_DD(computes_thresholds_when_all_pruned);
    RUN(relaxed_computes_local_bounds_and_thresholds_1);
    RUN_DD(relaxed_computes_local_bounds_and_thresholds_2);
    RUN_DD(relaxed_computes_local_bounds_and_thresholds_with_pruning);
    RUN(nodup_pop_off_an_empty_fringe_is_none);
    RUN(nodup_pops_largest_ub_then_lp);
    RUN(nodup_keeps_the_copy_with_longest_path);
    RUN(nodup_popped_in_order);
    RUN(nodup_pushing_same_node_multiple_times_does_not_alter_pop_order);
    RUN(nodup_pushing_nodes_triggers_reordering_if_lplen_is_better_up);
    RUN(simple_fringe_pops_by_ub_then_value);
    RUN(width_heuristics);
    RUN(knapsack_readme_instance_220);
    RUN(knapsack_seven_items);
    RUN(bitset_ord_is_lexicographic_on_members);
    RUN(pooled_long_arcs_on_misp);
    std::printf("%s (%d failure%s)\n", g_fail ? "FAILED" : "ALL OK", g_fail, g_fail == 1 ? "" : "s");
    return g_fail;
}
