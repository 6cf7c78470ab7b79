// =============================================================================
// models.hpp -- CPU restatement of the ddo example models that sit on the hot
// path: MISP (examples/misp/main.rs), 0/1 knapsack (examples/knapsack/main.rs) and
// MAX2SAT (examples/max2sat/{data,model,relax,heuristics}.rs), MCP (examples/mcp/{graph,model,relax}.rs) and TSPTW
// (examples/tsptw/{instance,state,model,relax,heuristics,dominance}.rs).
//
// *** TEST INFRASTRUCTURE (parity oracle + CPU baseline), see ddo_oracle.hpp ***
//
// MISP states are `bit_set::BitSet` in the reference (crate bit-set 0.5.3 on
// bit-vec 0.6.3, pinned in /root/reference/Cargo.lock:96-106; the crate source
// is NOT vendored under /root/reference).  The semantics relied on -- set
// algebra on blocks, len = popcount, Eq/Hash/Ord over the ascending sequence
// of members -- are restated here from the crate's documented behaviour and
// anchored on the reference's call sites (examples/misp/main.rs:70-208) and
// its known-answer tests (examples/misp/tests.rs).
// =============================================================================
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <regex>
#include <sstream>
#include <stdexcept>
#include <string>

#include "ddo_oracle.hpp"

namespace ddo {

// ---------------------------------------------------------------------------
// BitSet: heap-allocated block vector like bit_vec::BitVec (one allocation per
// state, as in the reference).
// ---------------------------------------------------------------------------
struct BitSet {
    std::vector<uint64_t> w;

    BitSet() = default;
    explicit BitSet(size_t nbits) : w((nbits + 63) / 64, 0) {}
    static BitSet full(size_t n) {
        BitSet b(n);
        for (size_t i = 0; i < n; ++i) b.insert(i);
        return b;
    }
    void insert(size_t i) {
        if (i / 64 >= w.size()) w.resize(i / 64 + 1, 0);
        w[i / 64] |= (uint64_t)1 << (i % 64);
    }
    void remove(size_t i) {
        if (i / 64 < w.size()) w[i / 64] &= ~((uint64_t)1 << (i % 64));
    }
    bool contains(size_t i) const { return i / 64 < w.size() && ((w[i / 64] >> (i % 64)) & 1); }
    void intersect_with(const BitSet& o) {
        for (size_t k = 0; k < w.size(); ++k) w[k] &= (k < o.w.size() ? o.w[k] : 0);
    }
    void union_with(const BitSet& o) {
        if (o.w.size() > w.size()) w.resize(o.w.size(), 0);
        for (size_t k = 0; k < o.w.size(); ++k) w[k] |= o.w[k];
    }
    size_t len() const {
        size_t c = 0;
        for (uint64_t x : w) c += (size_t)__builtin_popcountll(x);
        return c;
    }
    /// BitSet::iter(): ascending members
    template <class F>
    void for_each(F&& f) const {
        for (size_t k = 0; k < w.size(); ++k) {
            uint64_t x = w[k];
            while (x) {
                f(k * 64 + (size_t)__builtin_ctzll(x));
                x &= x - 1;
            }
        }
    }
    /// Eq over members, independent of allocated length
    bool operator==(const BitSet& o) const {
        size_t n = std::max(w.size(), o.w.size());
        for (size_t k = 0; k < n; ++k) {
            uint64_t a = k < w.size() ? w[k] : 0, b = k < o.w.size() ? o.w[k] : 0;
            if (a != b) return false;
        }
        return true;
    }
    /// Ord: lexicographic order of the ascending member lists:
    /// at the lowest differing bit p, the set containing p has the smaller
    /// next member, hence is the smaller sequence -- unless the other set has
    /// no member >= p at all (it is then a strict prefix, hence smaller).
    int cmp(const BitSet& o) const {
        size_t n = std::max(w.size(), o.w.size());
        for (size_t k = 0; k < n; ++k) {
            uint64_t a = k < w.size() ? w[k] : 0, b = k < o.w.size() ? o.w[k] : 0;
            uint64_t x = a ^ b;
            if (x) {
                uint64_t t = x & (~x + 1);
                bool a_has = (a & t) != 0;
                // does the set NOT containing p have any member above p ?
                const BitSet& other = a_has ? o : *this;
                bool other_has_more = false;
                uint64_t above = ~(t | (t - 1));
                uint64_t ow = k < other.w.size() ? other.w[k] : 0;
                if (ow & above) other_has_more = true;
                for (size_t j = k + 1; !other_has_more && j < other.w.size(); ++j)
                    if (other.w[j]) other_has_more = true;
                if (a_has) return other_has_more ? -1 : 1;   // a has p; b continues later => a < b; else b is prefix => b < a
                else return other_has_more ? 1 : -1;
            }
        }
        return 0;
    }
};

/// fxhash 0.2.1 (Cargo.lock:402-403), 64-bit: h = (rotl(h,5) ^ x) * K per
/// usize written.  bit-set's Hash feeds every member as a usize -- O(popcount)
/// per hash, which is part of the reference's CPU cost.
template <>
struct StateHash<BitSet> {
    size_t operator()(const BitSet& s) const {
        uint64_t h = 0;
        s.for_each([&](size_t i) {
            h = ((h << 5) | (h >> 59)) ^ (uint64_t)i;
            h *= 0x517cc1b727220a95ULL;
        });
        return (size_t)h;
    }
};

// ---------------------------------------------------------------------------
// MISP -- examples/misp/main.rs
// ---------------------------------------------------------------------------
constexpr isize MISP_YES = 1, MISP_NO = 0;

/// main.rs:37-51
struct Misp : Problem<BitSet> {
    bool canonical_ties() const override { return true; }   // equal-valued best arcs: an exact best path wins (Problem::canonical_ties)
    size_t nb_vars = 0;
    std::vector<BitSet> neighbors;  // COMPLEMENT adjacency rows (bit i itself stays set)
    std::vector<isize> weight;

    size_t nb_variables() const override { return nb_vars; }
    /// main.rs:69-71
    BitSet initial_state() const override { return BitSet::full(nb_vars); }
    isize initial_value() const override { return 0; }
    /// main.rs:77-85
    BitSet transition(const BitSet& state, Decision d) const override {
        BitSet res = state;
        res.remove(d.variable);
        if (d.value == MISP_YES) res.intersect_with(neighbors[d.variable]);
        return res;
    }
    /// main.rs:87-93
    isize transition_cost(const BitSet&, const BitSet&, Decision d) const override {
        return d.value == MISP_NO ? 0 : weight[d.variable];
    }
    /// main.rs:95-102
    void for_each_in_domain(Variable var, const BitSet& state, DecisionCallback& f) const override {
        if (state.contains(var.id)) {
            f.apply(Decision{var.id, MISP_YES});
            f.apply(Decision{var.id, MISP_NO});
        } else {
            f.apply(Decision{var.id, MISP_NO});
        }
    }
    /// main.rs:109-143: the variable occurring in the fewest states of the next
    /// layer (counts > 0), first index on ties (Iterator::min_by_key keeps the first).
    std::optional<Variable> next_variable(size_t, StateIter<BitSet>& next_layer) const override {
        static thread_local std::vector<size_t> heu;
        heu.assign(nb_vars, 0);
        while (const BitSet* s = next_layer.next()) s->for_each([&](size_t i) { heu[i] += 1; });
        std::optional<Variable> best;
        size_t best_cnt = 0;
        for (size_t i = 0; i < nb_vars; ++i) {
            if (heu[i] > 0 && (!best || heu[i] < best_cnt)) {
                best = Variable{i};
                best_cnt = heu[i];
            }
        }
        return best;
    }
    /// main.rs:145-147
    bool is_impacted_by(Variable var, const BitSet& state) const override { return state.contains(var.id); }
};

/// main.rs:168-194
struct MispRelax : Relaxation<BitSet> {
    const Misp& pb;
    explicit MispRelax(const Misp& p) : pb(p) {}
    BitSet merge(StateIter<BitSet>& states) const override {
        BitSet s(pb.nb_vars);
        while (const BitSet* x = states.next()) s.union_with(*x);
        return s;
    }
    isize relax(const BitSet&, const BitSet&, const BitSet&, Decision, isize cost) const override { return cost; }
    isize fast_upper_bound(const BitSet& state) const override {
        isize sum = 0;
        state.for_each([&](size_t i) { sum += pb.weight[i]; });
        return sum;
    }
};

/// main.rs:201-209
struct MispRanking : StateRanking<BitSet> {
    int compare(const BitSet& a, const BitSet& b) const override {
        size_t la = a.len(), lb = b.len();
        if (la != lb) return la < lb ? -1 : 1;
        return a.cmp(b);
    }
};

/// main.rs:258-317.  Regexes: comment `^c\s.*$`, `^p\s+edge\s+(\d+)\s+(\d+)$`,
/// `^n\s+(\d+)\s+(-?\d+)`, `^e\s+(\d+)\s+(\d+)`; anything else is a format error.
inline Misp read_misp_instance(const std::string& fname) {
    std::ifstream f(fname);
    if (!f) throw std::runtime_error("io error: cannot open " + fname);
    static const std::regex comment(R"(^c\s.*$)");
    static const std::regex pb_decl(R"(^p\s+edge\s+(\d+)\s+(\d+)$)");
    static const std::regex node_decl(R"(^n\s+(\d+)\s+(-?\d+))");
    static const std::regex edge_decl(R"(^e\s+(\d+)\s+(\d+))");
    Misp g;
    std::string raw;
    while (std::getline(f, raw)) {
        size_t b = raw.find_first_not_of(" \t\r\n\f\v");
        if (b == std::string::npos) continue;
        size_t e = raw.find_last_not_of(" \t\r\n\f\v");
        std::string line = raw.substr(b, e - b + 1);
        std::smatch m;
        if (std::regex_match(line, comment)) continue;
        if (std::regex_match(line, m, pb_decl)) {
            size_t n = std::stoull(m[1]);
            g.nb_vars = n;
            g.neighbors.assign(n, BitSet::full(n));
            g.weight.assign(n, 1);
            continue;
        }
        if (std::regex_search(line, m, node_decl)) {
            size_t n = std::stoull(m[1]);
            isize w = std::stoll(m[2]);
            g.weight.at(n - 1) = w;
            continue;
        }
        if (std::regex_search(line, m, edge_decl)) {
            size_t src = std::stoull(m[1]) - 1, dst = std::stoull(m[2]) - 1;
            g.neighbors.at(src).remove(dst);
            g.neighbors.at(dst).remove(src);
            continue;
        }
        throw std::runtime_error("ill formed instance");
    }
    return g;
}

// ---------------------------------------------------------------------------
// Knapsack -- examples/knapsack/main.rs
// ---------------------------------------------------------------------------
/// main.rs:37-44
struct KnapsackState {
    size_t depth;
    size_t capacity;
    bool operator==(const KnapsackState& o) const { return depth == o.depth && capacity == o.capacity; }
};
template <>
struct StateHash<KnapsackState> {
    size_t operator()(const KnapsackState& s) const {
        uint64_t h = 0;
        h = (((h << 5) | (h >> 59)) ^ (uint64_t)s.depth) * 0x517cc1b727220a95ULL;
        h = (((h << 5) | (h >> 59)) ^ (uint64_t)s.capacity) * 0x517cc1b727220a95ULL;
        return (size_t)h;
    }
};

constexpr isize TAKE_IT = 1, LEAVE_IT_OUT = 0;

/// main.rs:53-72
struct Knapsack : Problem<KnapsackState> {
    bool canonical_ties() const override { return true; }   // equal-valued best arcs: an exact best path wins (Problem::canonical_ties)
    size_t capacity;
    std::vector<isize> profit;
    std::vector<size_t> weight;
    std::vector<size_t> order;

    Knapsack(size_t capacity, std::vector<isize> profit, std::vector<size_t> weight)
        : capacity(capacity), profit(std::move(profit)), weight(std::move(weight)) {
        order.resize(this->profit.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        // sort_unstable_by_key(OrderedFloat(-profit/weight)); ties: index order (documented deviation:
        // Rust's unstable sort leaves tie order unspecified)
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
            double ka = -(double)this->profit[a] / (double)this->weight[a];
            double kb = -(double)this->profit[b] / (double)this->weight[b];
            return ka < kb;
        });
    }
    size_t nb_variables() const override { return profit.size(); }
    /// main.rs:93-99
    void for_each_in_domain(Variable var, const KnapsackState& s, DecisionCallback& f) const override {
        if (s.capacity >= weight[var.id]) f.apply(Decision{var.id, TAKE_IT});
        f.apply(Decision{var.id, LEAVE_IT_OUT});
    }
    KnapsackState initial_state() const override { return KnapsackState{0, capacity}; }
    isize initial_value() const override { return 0; }
    /// main.rs:106-113
    KnapsackState transition(const KnapsackState& s, Decision d) const override {
        KnapsackState r = s;
        r.depth += 1;
        if (d.value == TAKE_IT) r.capacity -= weight[d.variable];
        return r;
    }
    isize transition_cost(const KnapsackState&, const KnapsackState&, Decision d) const override {
        return profit[d.variable] * d.value;
    }
    /// main.rs:118-125
    std::optional<Variable> next_variable(size_t depth, StateIter<KnapsackState>&) const override {
        if (depth < nb_variables()) return Variable{order[depth]};
        return std::nullopt;
    }
};

/// main.rs:146-184
struct KPRelax : Relaxation<KnapsackState> {
    const Knapsack& pb;
    explicit KPRelax(const Knapsack& p) : pb(p) {}
    /// max_by_key(capacity): the LAST maximum
    KnapsackState merge(StateIter<KnapsackState>& states) const override {
        const KnapsackState* best = nullptr;
        while (const KnapsackState* s = states.next())
            if (!best || s->capacity >= best->capacity) best = s;
        return *best;
    }
    isize relax(const KnapsackState&, const KnapsackState&, const KnapsackState&, Decision, isize cost) const override {
        return cost;
    }
    isize fast_upper_bound(const KnapsackState& state) const override {
        size_t depth = state.depth;
        isize max_profit = 0;
        size_t cap = state.capacity;
        while (cap > 0 && depth < pb.profit.size()) {
            size_t item = pb.order[depth];
            if (cap >= pb.weight[item]) {
                max_profit += pb.profit[item];
                cap -= pb.weight[item];
            } else {
                double ratio = (double)cap / (double)pb.weight[item];
                double p = ratio * (double)pb.profit[item];
                max_profit += (isize)std::floor(p);
                cap = 0;
            }
            depth += 1;
        }
        return max_profit;
    }
};

/// main.rs:187-194
struct KPRanking : StateRanking<KnapsackState> {
    int compare(const KnapsackState& a, const KnapsackState& b) const override {
        return a.capacity < b.capacity ? -1 : (a.capacity > b.capacity ? 1 : 0);
    }
};

/// main.rs:267-303: lines starting with 'c' skipped; first line "n capacity";
/// then n lines "profit weight".
/// examples/knapsack/main.rs:198-218: states of one depth are comparable; more capacity and more value dominate
struct KPDominance {
    using Key = size_t;
    struct KeyHash { size_t operator()(const Key& k) const { return std::hash<size_t>()(k); } };
    struct KeyEq { bool operator()(const Key& a, const Key& b) const { return a == b; } };
    std::optional<Key> get_key(std::shared_ptr<const KnapsackState> s) const { return s->depth; }
    size_t nb_dimensions(const KnapsackState&) const { return 1; }
    isize get_coordinate(const KnapsackState& s, size_t) const { return (isize)s.capacity; }
    bool use_value() const { return true; }
};

inline Knapsack read_knapsack_instance(const std::string& fname) {
    std::ifstream f(fname);
    if (!f) throw std::runtime_error("io error: cannot open " + fname);
    std::string line;
    bool is_first = true;
    size_t n = 0, count = 0, capa = 0;
    std::vector<isize> profit;
    std::vector<size_t> weight;
    while (std::getline(f, line)) {
        if (!line.empty() && line[0] == 'c') continue;
        if (is_first) {
            is_first = false;
            std::istringstream ss(line);
            ss >> n >> capa;
        } else {
            if (count >= n) break;
            std::istringstream ss(line);
            isize p;
            size_t w;
            if (!(ss >> p >> w)) continue;
            profit.push_back(p);
            weight.push_back(w);
            count += 1;
        }
    }
    return Knapsack(capa, std::move(profit), std::move(weight));
}

// ===========================================================================
// MAX2SAT  (examples/max2sat/{data.rs, model.rs, relax.rs, heuristics.rs})
// ===========================================================================
/// model.rs:55-60: the marginal benefit of setting each variable to true, plus the depth
struct Max2SatState {
    size_t depth = 0;
    std::vector<isize> substates;
    bool operator==(const Max2SatState& o) const { return depth == o.depth && substates == o.substates; }
    /// model.rs:76-79
    isize rank() const {
        isize r = 0;
        for (isize x : substates) r += x < 0 ? -x : x;
        return r;
    }
};
template <>
struct StateHash<Max2SatState> {
    size_t operator()(const Max2SatState& s) const {
        uint64_t h = 0;
        auto mix = [&](uint64_t w) { h = (((h << 5) | (h >> 59)) ^ w) * 0x517cc1b727220a95ULL; };
        mix((uint64_t)s.depth);
        mix((uint64_t)s.substates.size());
        for (isize x : s.substates) mix((uint64_t)x);
        return (size_t)h;
    }
};

/// data.rs:31-53: a clause over literals (+-(1 + variable)), smaller literal first
struct BinaryClause {
    isize a, b;
    BinaryClause(isize x, isize y) : a(std::min(x, y)), b(std::max(x, y)) {}
    bool is_tautology() const { return a == -b; }
    bool is_unit() const { return a == b; }
};
/// data.rs:58-62.  `insert` semantics: a clause listed twice keeps its LAST weight.
struct Weighed2Sat {
    size_t nb_vars = 0;
    std::vector<std::pair<BinaryClause, isize>> weights;   // unique clauses, in order of first appearance
    void insert(BinaryClause c, isize w) {
        for (auto& e : weights)
            if (e.first.a == c.a && e.first.b == c.b) {
                e.second = w;
                return;
            }
        weights.emplace_back(c, w);
    }
};

/// data.rs:67-116.  The four line patterns of the reference, tried in its order on the trimmed line:
///   comment  ^c\s.*$            problem  ^p\s+wcnf\s+(\d+)\s+(\d+)
///   binary   ^(-?\d+)\s+(-?\d+)\s+(-?\d+)\s+0      unit  ^(-?\d+)\s+(-?\d+)-?\s+0
inline Weighed2Sat read_max2sat_instance(const std::string& fname) {
    std::ifstream f(fname);
    if (!f) throw std::runtime_error("io error: cannot open " + fname);
    static const std::regex comment(R"(^c\s.*$)");
    static const std::regex pb_decl(R"(^p\s+wcnf\s+(\d+)\s+(\d+))");
    static const std::regex bin_decl(R"(^(-?\d+)\s+(-?\d+)\s+(-?\d+)\s+0)");
    static const std::regex unit_decl(R"(^(-?\d+)\s+(-?\d+)-?\s+0)");
    Weighed2Sat inst;
    std::string line;
    while (std::getline(f, line)) {
        size_t b = line.find_first_not_of(" \t\r\n"), e = line.find_last_not_of(" \t\r\n");
        if (b == std::string::npos) continue;
        line = line.substr(b, e - b + 1);
        std::smatch m;
        if (std::regex_search(line, m, comment)) continue;
        if (std::regex_search(line, m, pb_decl)) {
            inst.nb_vars = (size_t)std::stoull(m[1].str());
            continue;
        }
        if (std::regex_search(line, m, bin_decl)) {
            inst.insert(BinaryClause((isize)std::stoll(m[2].str()), (isize)std::stoll(m[3].str())), (isize)std::stoll(m[1].str()));
            continue;
        }
        if (std::regex_search(line, m, unit_decl)) {
            isize x = (isize)std::stoll(m[2].str());
            inst.insert(BinaryClause(x, x), (isize)std::stoll(m[1].str()));
            continue;
        }
    }
    return inst;
}

/// model.rs:91-346.  DP model of Bergman, Cire, van Hoeve (INFORMS J. Comp. 2016).
struct Max2Sat : Problem<Max2SatState> {
    bool canonical_ties() const override { return true; }   // ranking is not a total order: see Problem::canonical_ties
    static constexpr isize T = 1, F = -1;                       // model.rs:29-31
    size_t nb_vars = 0;
    isize initial = 0;
    std::vector<isize> weights;                                   // (2n)^2, indexed by offset(x, y)
    std::vector<isize> sum_of_clause_weights;
    std::vector<size_t> order;                                    // vars_by_sum_of_clause_weights (worst first)
    std::vector<isize> nk, estimates;

    static isize lit_t(size_t v) { return 1 + (isize)v; }         // model.rs:34-46
    static isize lit_f(size_t v) { return -(1 + (isize)v); }
    static isize pos(isize x) { return x > 0 ? x : 0; }           // model.rs:49-51
    static size_t var_of(isize lit) { return (size_t)((lit < 0 ? -lit : lit) - 1); }   // idx, model.rs:105-107
    static size_t mk_lit(isize x) {                               // model.rs:108-113
        size_t a = var_of(x);
        return a + a + (x > 0 ? 1 : 0);
    }
    size_t offset(isize x, isize y) const {                       // model.rs:154-159
        isize a = std::min(x, y), b = std::max(x, y);
        return mk_lit(a) * 2 * nb_vars + mk_lit(b);
    }
    isize weight(isize x, isize y) const { return weights[offset(x, y)]; }   // model.rs:150-152

    explicit Max2Sat(const Weighed2Sat& inst) {                   // model.rs:115-148
        nb_vars = inst.nb_vars;
        const size_t n = nb_vars;
        weights.assign((2 * n) * (2 * n), 0);
        sum_of_clause_weights.assign(n, 0);
        for (const auto& e : inst.weights) {
            const BinaryClause& c = e.first;
            weights[offset(c.a, c.b)] = e.second;
            sum_of_clause_weights[var_of(c.a)] += e.second;
            if (!c.is_unit()) sum_of_clause_weights[var_of(c.b)] += e.second;
            if (c.is_tautology()) initial += e.second;
        }
        // sort_unstable_by_key(sum_of_clause_weights): ties keep index order here (documented deviation: the
        // reference's unstable sort leaves them unspecified)
        order.resize(n);
        for (size_t i = 0; i < n; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(),
                         [&](size_t a, size_t b) { return sum_of_clause_weights[a] < sum_of_clause_weights[b]; });
        estimates.assign(n, 0);
        for (size_t k = 0; k < n; ++k) estimates[k] = precompute_estimate(k);
        nk.assign(n, 0);
        for (size_t k = 0; k < n; ++k) {                          // model.rs:178-185
            isize sum = 0;
            for (size_t i = 0; i < k; ++i) sum += weight(lit_t(order[i]), lit_f(order[i]));
            nk[k] = sum;
        }
    }
    /// model.rs:193-229
    isize precompute_estimate(size_t k) const {
        const size_t n = nb_vars;
        isize sum = 0;
        for (size_t i = k; i < n; ++i) {
            const size_t vi = order[i];
            for (size_t j = i + 1; j < n; ++j) {
                const size_t vj = order[j];
                const isize tt = weight(lit_t(vi), lit_t(vj)), tf = weight(lit_t(vi), lit_f(vj));
                const isize ft = weight(lit_f(vi), lit_t(vj)), ff = weight(lit_f(vi), lit_f(vj));
                const isize wtt = tt + tf + ft, wtf = tt + tf + ff, wft = tt + ft + ff, wff = tf + ft + ff;
                sum += std::max(std::max(wtt, wtf), std::max(wft, wff));
            }
            sum += weight(lit_t(vi), lit_f(vi)) + std::max(weight(lit_t(vi), lit_t(vi)), weight(lit_f(vi), lit_f(vi)));
        }
        return sum;
    }
    /// model.rs:231-240
    isize fast_upper_bound(const Max2SatState& s) const {
        return s.rank() + estimates[s.depth] - initial + nk[s.depth];
    }

    size_t nb_variables() const override { return nb_vars; }
    Max2SatState initial_state() const override {
        Max2SatState s;
        s.substates.assign(nb_vars, 0);
        return s;
    }
    isize initial_value() const override { return initial; }      // sum of all tautologies
    void for_each_in_domain(Variable var, const Max2SatState&, DecisionCallback& f) const override {
        f.apply(Decision{var.id, T});
        f.apply(Decision{var.id, F});
    }
    /// model.rs:275-293.  varset(state) = the first n - (depth + 1) variables of the order (model.rs:161-170)
    Max2SatState transition(const Max2SatState& s, Decision d) const override {
        const size_t k = d.variable;
        Max2SatState r = s;
        r.depth += 1;
        r.substates[k] = 0;
        const size_t nfree = nb_vars - (s.depth + 1);
        for (size_t i = 0; i < nfree; ++i) {
            const size_t l = order[i];
            if (d.value == F) r.substates[l] += weight(lit_t(k), lit_t(l)) - weight(lit_t(k), lit_f(l));
            else r.substates[l] += weight(lit_f(k), lit_t(l)) - weight(lit_f(k), lit_f(l));
        }
        return r;
    }
    /// model.rs:294-329
    isize transition_cost(const Max2SatState& s, const Max2SatState&, Decision d) const override {
        const size_t k = d.variable;
        const size_t nfree = nb_vars - (s.depth + 1);
        if (d.value == F) {
            isize sum = weight(lit_f(k), lit_f(k));
            for (size_t i = 0; i < nfree; ++i) {
                const size_t l = order[i];
                const isize wff = weight(lit_f(k), lit_f(l)), wft = weight(lit_f(k), lit_t(l));
                const isize wtt = weight(lit_t(k), lit_t(l)), wtf = weight(lit_t(k), lit_f(l));
                sum += (wff + wft) + std::min(pos(s.substates[l]) + wtt, pos(-s.substates[l]) + wtf);
            }
            return pos(-s.substates[k]) + sum;
        }
        isize sum = weight(lit_t(k), lit_t(k));
        for (size_t i = 0; i < nfree; ++i) {
            const size_t l = order[i];
            const isize wtt = weight(lit_t(k), lit_t(l)), wtf = weight(lit_t(k), lit_f(l));
            const isize wff = weight(lit_f(k), lit_f(l)), wft = weight(lit_f(k), lit_t(l));
            sum += (wtf + wtt) + std::min(pos(s.substates[l]) + wft, pos(-s.substates[l]) + wff);
        }
        return pos(s.substates[k]) + sum;
    }
    /// model.rs:330-346: the depth of the first state of the next layer decides; none when the layer is empty
    std::optional<Variable> next_variable(size_t, StateIter<Max2SatState>& next_layer) const override {
        const Max2SatState* s = next_layer.next();
        if (!s) return std::nullopt;
        if (s->depth < nb_vars) return Variable{order[nb_vars - s->depth - 1]};
        return std::nullopt;
    }
};

/// relax.rs:42-90
struct Max2SatRelax : Relaxation<Max2SatState> {
    const Max2Sat& pb;
    explicit Max2SatRelax(const Max2Sat& p) : pb(p) {}
    /// relax.rs:46-77: per variable, the smallest |benefit| with the common sign; 0 when the signs disagree
    Max2SatState merge(StateIter<Max2SatState>& it) const override {
        std::vector<const Max2SatState*> states;
        while (const Max2SatState* s = it.next()) states.push_back(s);
        Max2SatState out;
        out.depth = states[0]->depth;
        out.substates.assign(pb.nb_vars, 0);
        for (size_t v = 0; v < pb.nb_vars; ++v) {
            isize sign = 0, min_benef = ISIZE_MAX;
            bool same = true;
            for (const Max2SatState* s : states) {
                const isize sub = s->substates[v];
                min_benef = std::min(min_benef, sub < 0 ? -sub : sub);
                if (sign == 0 && sub != 0) sign = sub < 0 ? -1 : 1;
                else if (sign * sub < 0) {
                    same = false;
                    break;
                }
            }
            if (same) out.substates[v] = sign * min_benef;
        }
        return out;
    }
    /// relax.rs:78-84
    isize relax(const Max2SatState&, const Max2SatState& dst, const Max2SatState& merged, Decision, isize cost) const override {
        isize c = cost;
        for (size_t v = 0; v < pb.nb_vars; ++v) {
            const isize a = dst.substates[v], b = merged.substates[v];
            c += (a < 0 ? -a : a) - (b < 0 ? -b : b);
        }
        return c;
    }
    isize fast_upper_bound(const Max2SatState& s) const override { return pb.fast_upper_bound(s); }
};

/// Signed-vector states on the wire (include/ddo_hip.h, ddo_model_create_max2sat / _mcp): benefit v is the 32-bit half
/// (v & 1) of word v / 2, the word after the last pair holds the depth.
inline void pack_signed_vector(const std::vector<isize>& b, size_t depth, uint64_t* out) {
    const size_t np = (b.size() + 1) / 2;
    for (size_t k = 0; k < np; ++k) {
        const uint64_t lo = (uint32_t)(int32_t)b[2 * k];
        const uint64_t hi = 2 * k + 1 < b.size() ? (uint32_t)(int32_t)b[2 * k + 1] : 0u;
        out[k] = lo | (hi << 32);
    }
    out[np] = (uint64_t)depth;
}
/// DETERMINISTIC TIE-BREAK shared with the device engine ("Exactness of selection"): the
/// reference ranks these states by sum |benefit| alone and leaves ties to the iteration order of its FxHashMap
/// (unpinned); oracle and device both break them by the packed state words, compared as
/// unsigned 64-bit numbers from word 0 (the larger word ranks higher).  Parity with the reference itself stays on
/// optimum + proof for these models; oracle <-> device parity becomes bit-exact per compile.
inline int compare_signed_vectors(const std::vector<isize>& a, size_t da, const std::vector<isize>& b, size_t db) {
    const size_t np = (std::max(a.size(), b.size()) + 1) / 2;
    auto word = [](const std::vector<isize>& v, size_t d, size_t k, size_t npairs) -> uint64_t {
        if (k == npairs) return (uint64_t)d;
        const uint64_t lo = 2 * k < v.size() ? (uint32_t)(int32_t)v[2 * k] : 0u;
        const uint64_t hi = 2 * k + 1 < v.size() ? (uint32_t)(int32_t)v[2 * k + 1] : 0u;
        return lo | (hi << 32);
    };
    for (size_t k = 0; k <= np; ++k) {
        const uint64_t x = word(a, da, k, np), y = word(b, db, k, np);
        if (x != y) return x < y ? -1 : 1;
    }
    return 0;
}

/// heuristics.rs:30-37, then the shared tie-break
struct Max2SatRanking : StateRanking<Max2SatState> {
    int compare(const Max2SatState& a, const Max2SatState& b) const override {
        const isize x = a.rank(), y = b.rank();
        if (x != y) return x < y ? -1 : 1;
        return compare_signed_vectors(a.substates, a.depth, b.substates, b.depth);
    }
};

// ===========================================================================
// MCP -- maximum cut  (examples/mcp/{graph.rs, model.rs, relax.rs})
// ===========================================================================
/// graph.rs:30-85: adjacency matrix of a weighted undirected graph
struct McpGraph {
    size_t nb_vertices = 0;
    std::vector<isize> adj;
    explicit McpGraph(size_t n = 0) : nb_vertices(n), adj(n * n, 0) {}
    isize at(size_t x, size_t y) const { return adj[x * nb_vertices + y]; }
    void add_bidir_edge(size_t x, size_t y, isize w) {
        adj[x * nb_vertices + y] = w;
        adj[y * nb_vertices + x] = w;
    }
    /// graph.rs:37-42: halved because every edge sits twice in the matrix
    isize sum_of_negative_edges() const {
        isize s = 0;
        for (isize w : adj)
            if (w < 0) s += w;
        return s / 2;
    }
};
/// graph.rs:48-79: "c " comment lines, "<vars> <edges>" (re)creates the graph, "<src> <dst> <w>" adds an edge (1-based)
inline McpGraph read_mcp_instance(const std::string& fname) {
    std::ifstream f(fname);
    if (!f) throw std::runtime_error("io error: cannot open " + fname);
    static const std::regex graph_decl(R"(^(\d+)\s+(\d+)$)");
    static const std::regex edge_decl(R"(^(\d+)\s+(\d+)\s+(-?\d+)$)");
    McpGraph g(0);
    std::string line;
    while (std::getline(f, line)) {
        size_t b = line.find_first_not_of(" \t\r\n"), e = line.find_last_not_of(" \t\r\n");
        if (b == std::string::npos) continue;
        line = line.substr(b, e - b + 1);
        if (line.rfind("c ", 0) == 0) continue;
        std::smatch m;
        if (std::regex_match(line, m, graph_decl)) {
            g = McpGraph((size_t)std::stoull(m[1].str()));
            continue;
        }
        if (std::regex_match(line, m, edge_decl))
            g.add_bidir_edge((size_t)std::stoull(m[1].str()) - 1, (size_t)std::stoull(m[2].str()) - 1, (isize)std::stoll(m[3].str()));
    }
    return g;
}

/// model.rs:27-31
struct McpState {
    std::vector<isize> benef;
    uint16_t depth = 0;
    bool operator==(const McpState& o) const { return depth == o.depth && benef == o.benef; }
};
template <>
struct StateHash<McpState> {
    size_t operator()(const McpState& s) const {
        uint64_t h = 0;
        auto mix = [&](uint64_t w) { h = (((h << 5) | (h >> 59)) ^ w) * 0x517cc1b727220a95ULL; };
        mix((uint64_t)s.benef.size());
        for (isize x : s.benef) mix((uint64_t)x);
        mix((uint64_t)s.depth);
        return (size_t)h;
    }
};

/// model.rs:37-130: vertices are assigned to side S (+1) or T (-1) in natural order; the first one is fixed to S
struct Mcp : Problem<McpState> {
    bool canonical_ties() const override { return true; }   // ranking is not a total order: see Problem::canonical_ties
    static constexpr isize SIDE_S = 1, SIDE_T = -1;
    McpGraph graph;
    explicit Mcp(McpGraph g) : graph(std::move(g)) {}
    size_t nb_variables() const override { return graph.nb_vertices; }
    McpState initial_state() const override {
        McpState s;
        s.benef.assign(nb_variables(), 0);
        return s;
    }
    isize initial_value() const override { return graph.sum_of_negative_edges(); }
    void for_each_in_domain(Variable var, const McpState& s, DecisionCallback& f) const override {   // model.rs:60-67
        f.apply(Decision{var.id, SIDE_S});
        if (s.depth != 0) f.apply(Decision{var.id, SIDE_T});
    }
    /// model.rs:69-78: benefits of the variables before the branching one are dropped to 0
    McpState transition(const McpState& s, Decision d) const override {
        const size_t n = nb_variables(), x = d.variable;
        McpState r;
        r.benef.assign(n, 0);
        for (size_t v = x; v < n; ++v) r.benef[v] = s.benef[v] + d.value * graph.at(x, v);
        r.depth = (uint16_t)(1 + s.depth);
        return r;
    }
    /// model.rs:80-86, 99-130
    isize transition_cost(const McpState& s, const McpState&, Decision d) const override {
        if (s.depth == 0) return 0;
        const size_t n = nb_variables(), x = d.variable;
        auto iabs = [](isize a) { return a < 0 ? -a : a; };
        isize sum = 0;
        if (d.value == SIDE_S) {
            for (size_t v = x; v < n; ++v) {
                const isize skl = s.benef[v], wkl = graph.at(x, v);
                if (skl * wkl <= 0) sum += std::min(iabs(skl), iabs(wkl));
            }
            return std::max<isize>(0, -s.benef[x]) + sum;
        }
        for (size_t v = x; v < n; ++v) {
            const isize skl = s.benef[v], wkl = graph.at(x, v);
            if (skl * wkl >= 0) sum += std::min(iabs(skl), iabs(wkl));
        }
        return std::max<isize>(0, s.benef[x]) + sum;
    }
    std::optional<Variable> next_variable(size_t depth, StateIter<McpState>&) const override {   // model.rs:88-96
        if (depth < nb_variables()) return Variable{depth};
        return std::nullopt;
    }
};

/// relax.rs:27-184
struct McpRelax : Relaxation<McpState> {
    const Mcp& pb;
    isize vr;
    std::vector<isize> nk, estimates;
    explicit McpRelax(const Mcp& p) : pb(p), vr(p.initial_value()) {
        const size_t n = pb.nb_variables();
        estimates.assign(n + 1, 0);                    // relax.rs:58-80: positive edges among vertices >= depth
        for (size_t d = 0; d <= n; ++d) {
            isize v = 0;
            for (size_t a = d; a < n; ++a)
                for (size_t b = a + 1; b < n; ++b)
                    if (pb.graph.at(a, b) > 0) v += pb.graph.at(a, b);
            estimates[d] = v;
        }
        nk.assign(n + 1, 0);                           // relax.rs:83-106: negative edges among vertices < depth
        for (size_t d = 0; d <= n; ++d) {
            isize v = 0;
            for (size_t j = 0; j < d; ++j)
                for (size_t i = 0; i < j; ++i)
                    if (pb.graph.at(i, j) < 0) v += pb.graph.at(i, j);
            nk[d] = v;
        }
    }
    /// relax.rs:141-176: same sign everywhere -> the value closest to zero; mixed signs -> 0
    McpState merge(StateIter<McpState>& it) const override {
        std::vector<const McpState*> nodes;
        while (const McpState* s = it.next()) nodes.push_back(s);
        McpState out;
        out.depth = nodes[0]->depth;
        out.benef.assign(pb.nb_variables(), 0);
        for (size_t v = 0; v < pb.nb_variables(); ++v) {
            bool posi = false, nega = false;
            for (const McpState* s : nodes) {
                if (s->benef[v] < 0) nega = true;
                else if (s->benef[v] > 0) posi = true;
                if (posi && nega) break;
            }
            if (posi && !nega) {
                isize m = ISIZE_MAX;
                for (const McpState* s : nodes) m = std::min(m, s->benef[v]);
                out.benef[v] = m;
            } else if (nega && !posi) {
                isize m = ISIZE_MAX;
                for (const McpState* s : nodes) m = std::min(m, s->benef[v] < 0 ? -s->benef[v] : s->benef[v]);
                out.benef[v] = -m;
            }
        }
        return out;
    }
    /// relax.rs:115-121
    isize relax(const McpState&, const McpState& dst, const McpState& mrg, Decision, isize c) const override {
        for (size_t v = 0; v < pb.nb_variables(); ++v) {
            const isize a = dst.benef[v], b = mrg.benef[v];
            c += (a < 0 ? -a : a) - (b < 0 ? -b : b);
        }
        return c;
    }
    /// relax.rs:123-130
    isize fast_upper_bound(const McpState& s) const override {
        const size_t k = s.depth;
        isize marginal = 0;
        for (size_t v = k; v < s.benef.size(); ++v) marginal += s.benef[v] < 0 ? -s.benef[v] : s.benef[v];
        return marginal + estimates[k] - vr + nk[k];
    }
};

/// model.rs:154-163, then the tie-break shared with the device (see compare_signed_vectors)
struct McpRanking : StateRanking<McpState> {
    int compare(const McpState& a, const McpState& b) const override {
        isize xa = 0, xb = 0;
        for (isize v : a.benef) xa += v < 0 ? -v : v;
        for (isize v : b.benef) xb += v < 0 ? -v : v;
        if (xa != xb) return xa < xb ? -1 : 1;
        return compare_signed_vectors(a.benef, a.depth, b.benef, b.depth);
    }
};

// ===========================================================================
// TSPTW -- travelling salesman with time windows
// (examples/tsptw/{instance.rs, state.rs, model.rs, relax.rs, heuristics.rs, dominance.rs})
// ===========================================================================
/// `smallbitset::Set256` (crate smallbitset 0.7.1, Cargo.lock; source not under /root/reference): a 256-bit set with
/// ascending iteration; restated from its use in examples/tsptw.
struct Set256 {
    uint64_t w[4] = {0, 0, 0, 0};
    void add(size_t i) { w[i >> 6] |= 1ULL << (i & 63); }
    void remove(size_t i) { w[i >> 6] &= ~(1ULL << (i & 63)); }
    bool contains(size_t i) const { return (w[i >> 6] >> (i & 63)) & 1ULL; }
    void union_with(const Set256& o) { for (int k = 0; k < 4; ++k) w[k] |= o.w[k]; }
    void inter_with(const Set256& o) { for (int k = 0; k < 4; ++k) w[k] &= o.w[k]; }
    void diff_with(const Set256& o) { for (int k = 0; k < 4; ++k) w[k] &= ~o.w[k]; }
    Set256 flip() const { Set256 r; for (int k = 0; k < 4; ++k) r.w[k] = ~w[k]; return r; }
    size_t len() const { size_t c = 0; for (int k = 0; k < 4; ++k) c += (size_t)__builtin_popcountll(w[k]); return c; }
    bool operator==(const Set256& o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
    template <class F>
    void for_each(F f) const {   // ascending; f returns false to stop
        for (int k = 0; k < 4; ++k) {
            uint64_t x = w[k];
            while (x) {
                size_t i = (size_t)k * 64 + (size_t)__builtin_ctzll(x);
                x &= x - 1;
                if (!f(i)) return;
            }
        }
    }
};

/// instance.rs:27-109
struct TsptwTimeWindow {
    size_t earliest, latest;
};
struct TsptwInstance {
    uint16_t nb_nodes = 0;
    std::vector<std::vector<size_t>> distances;
    std::vector<TsptwTimeWindow> timewindows;
};
/// `(x * 10000.0) as usize` on an f32 (instance.rs:87, 97-98): single-precision product, truncated, saturating at 0
inline size_t tsptw_fixed_point(const std::string& tok) {
    const float v = std::strtof(tok.c_str(), nullptr);
    const float m = v * 10000.0f;
    if (!(m > 0.0f)) return 0;
    return (size_t)m;
}
inline TsptwInstance read_tsptw_instance(const std::string& fname) {
    std::ifstream f(fname);
    if (!f) throw std::runtime_error("io error: cannot open " + fname);
    TsptwInstance inst;
    size_t lc = 0;
    std::string line;
    while (std::getline(f, line)) {
        size_t b = line.find_first_not_of(" \t\r\n"), e = line.find_last_not_of(" \t\r\n");
        if (b == std::string::npos) continue;
        line = line.substr(b, e - b + 1);
        if (line[0] == '#') continue;
        std::istringstream ss(line);
        std::string tok;
        if (lc == 0) {
            ss >> tok;
            inst.nb_nodes = (uint16_t)std::stoul(tok);
            inst.distances.assign(inst.nb_nodes, std::vector<size_t>(inst.nb_nodes, 0));
        } else if (lc >= 1 && lc <= inst.nb_nodes) {
            size_t j = 0;
            while (ss >> tok) {
                if (j < inst.nb_nodes) inst.distances[lc - 1][j] = tsptw_fixed_point(tok);
                ++j;
            }
        } else {
            std::string a, c;
            ss >> a >> c;
            inst.timewindows.push_back(TsptwTimeWindow{tsptw_fixed_point(a), tsptw_fixed_point(c)});
        }
        lc += 1;
    }
    return inst;
}

/// state.rs:34-101.  Position: a node, or (relaxed) one node among a set; elapsed time: fixed or an interval
struct TsptwState {
    bool pos_virtual = false;
    uint16_t pos_node = 0;
    Set256 pos_set;
    bool fuzzy = false;
    size_t t_earliest = 0, t_latest = 0;      // fixed: both = duration
    Set256 must_visit;
    bool has_maybe = false;
    Set256 maybe_visit;
    uint16_t depth = 0;
    bool same_position(const TsptwState& o) const {
        return pos_virtual == o.pos_virtual && (pos_virtual ? pos_set == o.pos_set : pos_node == o.pos_node);
    }
    bool operator==(const TsptwState& o) const {
        return same_position(o) && fuzzy == o.fuzzy && t_earliest == o.t_earliest && t_latest == o.t_latest &&
               must_visit == o.must_visit && has_maybe == o.has_maybe && (!has_maybe || maybe_visit == o.maybe_visit) &&
               depth == o.depth;
    }
    size_t earliest() const { return t_earliest; }
};
template <>
struct StateHash<TsptwState> {
    size_t operator()(const TsptwState& s) const {
        uint64_t h = 0;
        auto mix = [&](uint64_t w) { h = (((h << 5) | (h >> 59)) ^ w) * 0x517cc1b727220a95ULL; };
        mix(s.pos_virtual);
        if (s.pos_virtual) for (int k = 0; k < 4; ++k) mix(s.pos_set.w[k]);
        else mix(s.pos_node);
        mix(s.fuzzy);
        mix(s.t_earliest);
        mix(s.t_latest);
        for (int k = 0; k < 4; ++k) mix(s.must_visit.w[k]);
        mix(s.has_maybe);
        if (s.has_maybe) for (int k = 0; k < 4; ++k) mix(s.maybe_visit.w[k]);
        mix(s.depth);
        return (size_t)h;
    }
};

/// model.rs:30-217
struct Tsptw : Problem<TsptwState> {
    bool canonical_ties() const override { return true; }   // equal-valued best arcs: an exact best path wins (Problem::canonical_ties)
    TsptwInstance instance;
    TsptwState initial;
    explicit Tsptw(TsptwInstance inst) : instance(std::move(inst)) {
        for (size_t i = 1; i < instance.nb_nodes; ++i) initial.must_visit.add(i);
    }
    size_t nb_variables() const override { return instance.nb_nodes; }
    TsptwState initial_state() const override { return initial; }
    isize initial_value() const override { return 0; }

    size_t min_distance_to(const TsptwState& s, size_t j) const {   // model.rs:194-204
        if (!s.pos_virtual) return instance.distances[s.pos_node][j];
        size_t m = std::numeric_limits<size_t>::max();
        s.pos_set.for_each([&](size_t i) { m = std::min(m, instance.distances[i][j]); return true; });
        return m;
    }
    size_t max_distance_to(const TsptwState& s, size_t j) const {   // model.rs:205-215
        if (!s.pos_virtual) return instance.distances[s.pos_node][j];
        size_t m = 0;
        s.pos_set.for_each([&](size_t i) { m = std::max(m, instance.distances[i][j]); return true; });
        return m;
    }
    bool can_move_to(const TsptwState& s, size_t j) const {          // model.rs:150-157
        return s.t_earliest + min_distance_to(s, j) <= instance.timewindows[j].latest;
    }
    /// model.rs:65-94
    void for_each_in_domain(Variable var, const TsptwState& s, DecisionCallback& f) const override {
        if ((size_t)s.depth == nb_variables() - 1) {
            if (can_move_to(s, 0)) f.apply(Decision{var.id, 0});
            return;
        }
        bool ok = true;
        s.must_visit.for_each([&](size_t i) { ok = can_move_to(s, i); return ok; });
        if (!ok) return;
        s.must_visit.for_each([&](size_t i) { f.apply(Decision{var.id, (isize)i}); return true; });
        if (s.has_maybe)
            s.maybe_visit.for_each([&](size_t i) {
                if (can_move_to(s, i)) f.apply(Decision{var.id, (isize)i});
                return true;
            });
    }
    /// model.rs:95-115 with arrival_time :158-193
    TsptwState transition(const TsptwState& s, Decision d) const override {
        const size_t j = (size_t)d.value;
        TsptwState r;
        r.must_visit = s.must_visit;
        r.must_visit.remove(j);
        r.has_maybe = s.has_maybe;
        r.maybe_visit = s.maybe_visit;
        if (r.has_maybe) r.maybe_visit.remove(j);
        r.pos_virtual = false;
        r.pos_node = (uint16_t)j;
        r.depth = (uint16_t)(s.depth + 1);
        // earliest / latest arrival if we never had to wait
        size_t mn = s.t_earliest + min_distance_to(s, j);
        size_t mx = (s.fuzzy ? s.t_latest : s.t_earliest) + max_distance_to(s, j);
        const TsptwTimeWindow tw = instance.timewindows[j];
        if (mn == mx) {
            r.fuzzy = false;
            r.t_earliest = r.t_latest = std::max(mn, tw.earliest);
        } else {
            size_t e = std::max(mn, tw.earliest), l = std::min(mx, tw.latest);
            if (e == l) {
                r.fuzzy = false;
                r.t_earliest = r.t_latest = e;
            } else {
                r.fuzzy = true;
                r.t_earliest = e;
                r.t_latest = l;
            }
        }
        return r;
    }
    /// model.rs:116-139: minimisation seen as maximisation of the negated travel + waiting time
    isize transition_cost(const TsptwState& s, const TsptwState&, Decision d) const override {
        const size_t j = (size_t)d.value;
        const size_t travel = min_distance_to(s, j);
        const size_t arrive = s.t_earliest + travel;
        const size_t waiting = arrive < instance.timewindows[j].earliest ? instance.timewindows[j].earliest - arrive : 0;
        return -(isize)(travel + waiting);
    }
    std::optional<Variable> next_variable(size_t depth, StateIter<TsptwState>&) const override {   // model.rs:140-147
        if (depth == nb_variables()) return std::nullopt;
        return Variable{depth};
    }
};

/// relax.rs:32-265
struct TsptwRelax : Relaxation<TsptwState> {
    const Tsptw& pb;
    std::vector<size_t> cheapest_edge;
    explicit TsptwRelax(const Tsptw& p) : pb(p) {     // relax.rs:50-63: cheapest edge ENTERING each node
        const size_t n = pb.nb_variables();
        for (size_t i = 0; i < n; ++i) {
            size_t m = std::numeric_limits<size_t>::max();
            for (size_t j = 0; j < n; ++j)
                if (i != j) m = std::min(m, pb.instance.distances[j][i]);
            cheapest_edge.push_back(m);
        }
    }
    /// relax.rs:169-191 with RelaxHelper :65-166
    TsptwState merge(StateIter<TsptwState>& it) const override {
        uint16_t depth = 0;
        Set256 position, all_must, all_maybe;
        Set256 all_agree = Set256().flip();
        size_t earliest = std::numeric_limits<size_t>::max(), latest = 0;
        while (const TsptwState* s = it.next()) {
            depth = std::max(depth, s->depth);
            if (s->pos_virtual) position.union_with(s->pos_set);
            else position.add(s->pos_node);
            earliest = std::min(earliest, s->t_earliest);
            latest = std::max(latest, s->fuzzy ? s->t_latest : s->t_earliest);
            all_agree.inter_with(s->must_visit);
            all_must.union_with(s->must_visit);
            if (s->has_maybe) all_maybe.union_with(s->maybe_visit);
        }
        TsptwState r;
        r.depth = depth;
        r.pos_virtual = true;
        r.pos_set = position;
        r.fuzzy = earliest != latest;
        r.t_earliest = earliest;
        r.t_latest = r.fuzzy ? latest : earliest;
        r.must_visit = all_agree;
        Set256 maybe = all_maybe;
        maybe.union_with(all_must);
        maybe.diff_with(all_agree);
        r.has_maybe = maybe.len() > 0;
        if (r.has_maybe) r.maybe_visit = maybe;
        return r;
    }
    isize relax(const TsptwState&, const TsptwState&, const TsptwState&, Decision, isize cost) const override { return cost; }
    /// relax.rs:196-264
    isize fast_upper_bound(const TsptwState& s) const override {
        size_t complete_tour = pb.nb_variables() - (size_t)s.depth;
        std::vector<size_t> tmp;
        size_t mandatory = 0;
        size_t back_to_depot = std::numeric_limits<size_t>::max();
        bool infeasible = false;
        s.must_visit.for_each([&](size_t i) {
            complete_tour -= 1;
            mandatory += cheapest_edge[i];
            back_to_depot = std::min(back_to_depot, pb.instance.distances[i][0]);
            if (s.t_earliest + cheapest_edge[i] > pb.instance.timewindows[i].latest) {
                infeasible = true;
                return false;
            }
            return true;
        });
        if (infeasible) return ISIZE_MIN;
        if (s.has_maybe) {
            size_t violations = 0;
            s.maybe_visit.for_each([&](size_t i) {
                tmp.push_back(cheapest_edge[i]);
                back_to_depot = std::min(back_to_depot, pb.instance.distances[i][0]);
                if (s.t_earliest + cheapest_edge[i] > pb.instance.timewindows[i].latest) violations += 1;
                return true;
            });
            if (tmp.size() - violations < complete_tour) return ISIZE_MIN;
            std::sort(tmp.begin(), tmp.end());
            for (size_t k = 0; k < complete_tour && k < tmp.size(); ++k) mandatory += tmp[k];
        }
        if (mandatory == 0) {
            size_t here;
            if (!s.pos_virtual) here = pb.instance.distances[s.pos_node][0];
            else {
                here = std::numeric_limits<size_t>::max();
                s.pos_set.for_each([&](size_t x) { here = std::min(here, pb.instance.distances[x][0]); return true; });
            }
            back_to_depot = std::min(back_to_depot, here);
        }
        const size_t total = mandatory + back_to_depot;
        if (s.t_earliest + total > pb.instance.timewindows[0].latest) return ISIZE_MIN;
        return -(isize)total;
    }
};

/// TsptwState on the device wire (include/ddo_hip.h, ddo_model_create_tsptw): 5 words; instances of at most 64 nodes
/// words of one node set on the wire for nb_nodes nodes (include/ddo_hip.h): 1 up to 64 nodes, 2 up to 128, 4 up to 256
inline int tsptw_set_words(size_t nb_nodes) { return nb_nodes <= 64 ? 1 : (nb_nodes <= 128 ? 2 : 4); }
/// the state in 3K + 2 words: position set | must_visit | maybe_visit (K words each) | elapsed | node, flags, depth
inline void pack_tsptw_state(const TsptwState& s, int K, uint64_t* out) {
    for (int q = 0; q < K; ++q) {
        out[q] = s.pos_virtual ? s.pos_set.w[q] : 0;
        out[K + q] = s.must_visit.w[q];
        out[2 * K + q] = s.has_maybe ? s.maybe_visit.w[q] : 0;
    }
    out[3 * K] = (uint64_t)(uint32_t)s.t_earliest | ((uint64_t)(uint32_t)(s.fuzzy ? s.t_latest : s.t_earliest) << 32);
    out[3 * K + 1] = (s.pos_virtual ? 0ULL : (uint64_t)s.pos_node) | (s.pos_virtual ? 1ULL << 16 : 0) | (s.fuzzy ? 1ULL << 17 : 0) |
                     (s.has_maybe ? 1ULL << 18 : 0) | ((uint64_t)s.depth << 32);
}
/// heuristics.rs:26-51: the depth; ties (every pair of one layer) fall to the packed state words like for the other models
/// whose ranking is not a total order (compare_signed_vectors): the reference leaves them to its hash map's order
struct TsptwRanking : StateRanking<TsptwState> {
    int compare(const TsptwState& a, const TsptwState& b) const override {
        if (a.depth != b.depth) return a.depth < b.depth ? -1 : 1;
        // (packed with four words per set: the words beyond the instance's K are zero on both sides, so the order is the one
        // of the K-word packing the device compares)
        uint64_t x[14], y[14];
        pack_tsptw_state(a, 4, x);
        pack_tsptw_state(b, 4, y);
        for (int k = 0; k < 14; ++k)
            if (x[k] != y[k]) return x[k] < y[k] ? -1 : 1;
        return 0;
    }
};
struct TsptwWidth : WidthHeuristic<TsptwState> {
    size_t nb_vars, factor;
    TsptwWidth(size_t n, size_t f) : nb_vars(n), factor(f) {}
    size_t max_width(const SubProblem<TsptwState>& sp) const override { return nb_vars * (sp.depth + 1) * factor; }
};

/// dominance.rs:26-60: states with the same (position, must_visit) are compared on their value alone
struct TsptwDominance {
    using Key = std::shared_ptr<const TsptwState>;
    struct KeyHash {
        size_t operator()(const Key& k) const {
            uint64_t h = 0;
            auto mix = [&](uint64_t w) { h = (((h << 5) | (h >> 59)) ^ w) * 0x517cc1b727220a95ULL; };
            mix(k->pos_virtual);
            if (k->pos_virtual) for (int i = 0; i < 4; ++i) mix(k->pos_set.w[i]);
            else mix(k->pos_node);
            for (int i = 0; i < 4; ++i) mix(k->must_visit.w[i]);
            return (size_t)h;
        }
    };
    struct KeyEq {
        bool operator()(const Key& a, const Key& b) const { return a->same_position(*b) && a->must_visit == b->must_visit; }
    };
    std::optional<Key> get_key(std::shared_ptr<const TsptwState> s) const { return s; }
    size_t nb_dimensions(const TsptwState&) const { return 0; }
    isize get_coordinate(const TsptwState&, size_t) const { return 0; }
    bool use_value() const { return true; }
};

}  // namespace ddo
