// =============================================================================
// host_solver.cpp -- branch-and-bound host over the device engine: the Solver
// half of include/ddo_hip.h.
//
// Mirrors ParallelSolver (/root/reference/ddo/src/implementation/solver/parallel.rs:287-641)
// with `nb_concurrent` sub-problems compiled concurrently on the device instead of
// `nb_threads` OS threads, a NoDupFringe (fringe/no_duplicate.rs:52-324) ordered by
// MaxUB (heuristics/subproblem_ranking.rs:76-91) over MispRanking, FixedWidth /
// NbUnassignedWidth (heuristics/width.rs:166-171, 397-402) and NoCutoff / TimeBudget
// (heuristics/cutoff.rs:160-163, 302-323).  The reference host is Rust; there is no
// Rust toolchain in this image, so the host above the C ABI is C++ (INTEGRATION.md
// shows the Rust binding a maintainer would add).
//
// Storage: a fringe entry does not own its state and path.  Every relaxed DD
// contributes one ref-counted CutsetBlock (states, values, per-node decision rows and
// a link to the parent sub-problem's path); entries are (block, row) handles, so a
// push is a hash probe plus a heap sift and no copy.
// =============================================================================
#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <thread>
#include <unordered_map>

#include "../../include/ddo_hip.h"
#include "engine.hpp"

using namespace ddo_hip;

namespace {

constexpr int64_t I64_MIN = std::numeric_limits<int64_t>::min();
constexpr int64_t I64_MAX = std::numeric_limits<int64_t>::max();

/// The cut-set of one relaxed DD (or the problem root), shared by its fringe entries.
struct CutsetBlock {
    int refs = 0;
    CutsetBlock* parent = nullptr;   // block holding the sub-problem this DD was compiled from
    int parent_row = 0;
    int depth = 0;                   // depth of every node of this block
    int path_len = 0;                // decisions per row (relative to the parent sub-problem)
    int ws = 0;
    std::vector<uint64_t> states;    // rows x ws
    std::vector<int32_t> values;
    std::vector<uint32_t> paths;     // rows x path_len, node first (clean.rs:329-343 order)
    std::vector<int32_t> row_len;    // frontier cut-set: decisions (== layers below the parent) per row; empty: path_len for all
    std::vector<int32_t> row_plen;   // Pooled (mdd/pooled.rs): decisions per row -- one per EXPANDED ancestor, fewer than the layers (row_len) below the parent
    const uint64_t* state(int row) const { return states.data() + (size_t)row * ws; }
};

void block_ref(CutsetBlock* b) {
    if (b) b->refs++;
}
void block_unref(CutsetBlock* b) {
    while (b && --b->refs == 0) {
        CutsetBlock* p = b->parent;
        delete b;
        b = p;
    }
}

/// path of (block,row) from the problem root: parent path first, then this row's decisions
void materialize_path(const Model* model, const CutsetBlock* b, int row, std::vector<ddo_decision>& out) {
    std::vector<std::pair<const CutsetBlock*, int>> chain;
    while (b) {
        chain.push_back({b, row});
        row = b->parent_row;
        b = b->parent;
    }
    for (auto it = chain.rbegin(); it != chain.rend(); ++it) {
        const CutsetBlock* blk = it->first;
        const uint32_t* p = blk->paths.data() + (size_t)it->second * blk->path_len;
        const int plen = !blk->row_plen.empty() ? blk->row_plen[(size_t)it->second] : blk->row_len.empty() ? blk->path_len : blk->row_len[(size_t)it->second];
        // rows are stored node first (towards the parent): a frontier row uses the first plen entries of its stride
        for (int k = 0; k < plen; ++k) out.push_back(model->path_decision(p[k]));
    }
}

struct Entry {   // SubProblem (common.rs:75-87) as a handle
    CutsetBlock* block;
    int32_t row;
    int32_t depth;
    int64_t value;
    int64_t ub;
    uint64_t hash;
    int32_t plen = -1;   // decisions on the path from the problem root when that is not `depth` (Pooled decision diagrams), else -1
};

uint64_t hash_words(const uint64_t* s, int ws) {
    uint64_t h = 0x9E3779B97F4A7C15ULL;
    for (int k = 0; k < ws; ++k) {
        h ^= s[k];
        h *= 0xFF51AFD7ED558CCDULL;
        h ^= h >> 32;
    }
    return h;
}

/// NoDupFringe<MaxUB<MispRanking>> -- no_duplicate.rs:52-324
class NoDupFringe {
  public:
    explicit NoDupFringe(const Model* m) : model_(m), ws_(m->ws) { rehash(1024); }
    ~NoDupFringe() { clear(); }

    size_t len() const { return heap_.size(); }
    bool empty() const { return heap_.empty(); }

    /// no_duplicate.rs:88-140.  Takes one reference on e.block when the entry is stored.
    void push(const Entry& e) {
        const uint64_t* st = e.block->state(e.row);
        size_t slot = find_slot(st, e.hash);
        if (table_[slot] != EMPTY) {
            uint32_t id = table_[slot];
            Entry& old = nodes_[id];
            const int64_t old_lp = old.value, old_ub = old.ub;
            Entry cand = e;
            cand.ub = std::max(e.ub, old_ub);                       // :102
            const bool up = compare(cand, old) > 0;                 // :104
            if (e.value > old_lp) {                                 // :110-112 keep the longer path
                block_ref(cand.block);
                block_unref(old.block);
                old = cand;
            }
            if (e.ub > old_ub) nodes_[id].ub = e.ub;                // :113-115
            if (up) bubble_up(id);
            return;
        }
        uint32_t id;
        if (!recycle_.empty()) {
            id = recycle_.back();
            recycle_.pop_back();
            nodes_[id] = e;
        } else {
            id = (uint32_t)nodes_.size();
            nodes_.push_back(e);
            pos_.push_back(0);
        }
        block_ref(e.block);
        table_[slot] = id;
        if (++used_ * 10 > table_.size() * 6) rehash(table_.size() * 2);
        heap_.push_back(id);
        pos_[id] = (uint32_t)heap_.size() - 1;
        bubble_up(id);
    }

    /// no_duplicate.rs:144-164.  The caller inherits the entry's block reference.
    bool pop(Entry& out) {
        if (heap_.empty()) return false;
        uint32_t id = heap_[0];
        heap_[0] = heap_.back();
        heap_.pop_back();
        if (!heap_.empty()) {
            pos_[heap_[0]] = 0;
            bubble_down(heap_[0]);
        }
        out = nodes_[id];
        erase_from_table(id);
        recycle_.push_back(id);
        return true;
    }
    const Entry* peek() const { return heap_.empty() ? nullptr : &nodes_[heap_[0]]; }

    void clear() {
        for (uint32_t id : heap_) block_unref(nodes_[id].block);
        heap_.clear();
        nodes_.clear();
        pos_.clear();
        recycle_.clear();
        std::fill(table_.begin(), table_.end(), EMPTY);
        used_ = 0;
    }

  private:
    static constexpr uint32_t EMPTY = 0xFFFFFFFFu;
    /// MaxUB::compare (subproblem_ranking.rs:86-90): ub, then value, then the state ranking
    int compare(const Entry& l, const Entry& r) const {
        if (l.ub != r.ub) return l.ub < r.ub ? -1 : 1;
        if (l.value != r.value) return l.value < r.value ? -1 : 1;
        return model_->compare_states(l.block->state(l.row), r.block->state(r.row));
    }
    size_t find_slot(const uint64_t* st, uint64_t h) const {
        size_t mask = table_.size() - 1, slot = (size_t)h & mask;
        for (;;) {
            uint32_t id = table_[slot];
            if (id == EMPTY) return slot;
            const Entry& n = nodes_[id];
            if (n.hash == h && std::memcmp(n.block->state(n.row), st, (size_t)ws_ * 8) == 0) return slot;
            slot = (slot + 1) & mask;
        }
    }
    void erase_from_table(uint32_t id) {   // linear probing, backward-shift deletion
        size_t mask = table_.size() - 1, slot = (size_t)nodes_[id].hash & mask;
        while (table_[slot] != id) slot = (slot + 1) & mask;
        size_t hole = slot;
        for (;;) {
            slot = (slot + 1) & mask;
            uint32_t other = table_[slot];
            if (other == EMPTY) break;
            size_t home = (size_t)nodes_[other].hash & mask;
            // can `other` move into the hole ?  (its home must not lie cyclically in (hole, slot])
            bool between = hole <= slot ? (home > hole && home <= slot) : (home > hole || home <= slot);
            if (!between) {
                table_[hole] = other;
                hole = slot;
            }
        }
        table_[hole] = EMPTY;
        used_--;
    }
    void rehash(size_t cap) {
        std::vector<uint32_t> old;
        old.swap(table_);
        table_.assign(cap, EMPTY);
        size_t mask = cap - 1;
        for (uint32_t id : old) {
            if (id == EMPTY) continue;
            size_t slot = (size_t)nodes_[id].hash & mask;
            while (table_[slot] != EMPTY) slot = (slot + 1) & mask;
            table_[slot] = id;
        }
    }
    void bubble_up(uint32_t id) {          // :227-242
        size_t me = pos_[id];
        while (me > 0) {
            size_t par = (me - 1) / 2;
            if (compare(nodes_[heap_[me]], nodes_[heap_[par]]) <= 0) break;
            std::swap(heap_[me], heap_[par]);
            pos_[heap_[me]] = (uint32_t)me;
            pos_[heap_[par]] = (uint32_t)par;
            me = par;
        }
    }
    void bubble_down(uint32_t id) {        // :244-259
        size_t me = pos_[id], size = heap_.size();
        for (;;) {
            size_t l = 2 * me + 1, r = l + 1, kid;
            if (l >= size) break;
            if (r >= size) kid = l;
            else kid = compare(nodes_[heap_[l]], nodes_[heap_[r]]) > 0 ? l : r;
            if (compare(nodes_[heap_[me]], nodes_[heap_[kid]]) >= 0) break;
            std::swap(heap_[me], heap_[kid]);
            pos_[heap_[me]] = (uint32_t)me;
            pos_[heap_[kid]] = (uint32_t)kid;
            me = kid;
        }
    }

    const Model* model_;
    int ws_;
    std::vector<Entry> nodes_;
    std::vector<uint32_t> pos_, heap_, recycle_, table_;
    size_t used_ = 0;
};


// ---------------------------------------------------------------------------------------------
// Lazy block fringe: SimpleFringe semantics (fringe/simple.rs:35-62: a max-heap on MaxUB, no
// de-duplication) without ever materialising the nodes on the host.  Every relaxed DD leaves its
// cut-set as ONE block in the device node pool; the host receives only (value, ub) per node,
// counting-sorts them once, and keeps a heap of blocks keyed by each block's best remaining node.
// A push costs O(rows) integer work per block instead of a hash probe + heap sift per node, and the
// next batch reads its residual states straight from HBM.
// ---------------------------------------------------------------------------------------------
struct DevBlock {
    int refs = 0;
    DevBlock* parent = nullptr;
    int parent_row = 0;
    uint64_t off = NO_POOL_SRC;      // pool offset (NO_POOL_SRC: the problem root, state inline)
    int rows = 0, depth = 0, lel = 0;
    int64_t cap_ub = I64_MAX;        // ub of the sub-problem the DD was compiled from (parallel.rs:460)
    std::vector<int32_t> value, ub;
    std::vector<uint32_t> order;     // rows sorted by (ub, value) descending
    std::vector<uint8_t> mine;       // sharded search, root cut-set only: 1 = this rank owns the row (hash of the state)
    std::vector<std::vector<ddo_decision>> host_paths;   // imported sub-problems: decisions from the problem root, per row
    size_t cursor = 0;
    uint64_t id = 0;
    int64_t head_ub() const { return std::min<int64_t>(cap_ub, ub[order[cursor]]); }
    int64_t head_value() const { return value[order[cursor]]; }
};
void dev_ref(DevBlock* b) {
    if (b) b->refs++;
}
void dev_unref(DevBlock* b) {
    while (b && --b->refs == 0) {
        DevBlock* p = b->parent;
        delete b;
        b = p;
    }
}
struct LazyItem {
    DevBlock* block;
    int row;
    int depth;
    int64_t value, ub;
};
class LazyFringe {
  public:
    ~LazyFringe() { clear(); }
    size_t len() const { return open_; }
    bool empty() const { return heap_.empty(); }
    /// rows with min(cap_ub, ub) > best_lb are ordered by (ub, value) descending with two counting sorts
    void push_block(DevBlock* b, int64_t best_lb) {
        if (prepare_block(b, best_lb)) commit_block(b);
    }
    /// Thread-safe half of push_block: filters and orders the rows of `b` (b->order).  False: nothing to enqueue.
    /// Rows of the root cut-set that belong to another rank (b->mine) are left out: the owner of a row is a function
    /// of its STATE, never of its position -- row positions come from device atomics and differ between ranks.
    static bool prepare_block(DevBlock* b, int64_t best_lb) {
        const int n = b->rows;
        std::vector<uint32_t> tmp;
        tmp.reserve(n);
        int32_t vmin = INT32_MAX, vmax = INT32_MIN, umin = INT32_MAX, umax = INT32_MIN;
        for (int j = 0; j < n; ++j) {
            if (std::min<int64_t>(b->cap_ub, b->ub[j]) <= best_lb) continue;
            if (!b->mine.empty() && !b->mine[j]) continue;
            tmp.push_back((uint32_t)j);
            vmin = std::min(vmin, b->value[j]);
            vmax = std::max(vmax, b->value[j]);
            umin = std::min(umin, b->ub[j]);
            umax = std::max(umax, b->ub[j]);
        }
        if (tmp.empty()) return false;
        auto counting = [&](const std::vector<int32_t>& key, int32_t lo, int32_t hi) {
            const int64_t range = (int64_t)hi - lo + 1;
            if (range > (1 << 22)) {   // huge key range: comparison sort
                std::stable_sort(tmp.begin(), tmp.end(), [&](uint32_t x, uint32_t y) { return key[x] > key[y]; });
                return;
            }
            std::vector<uint32_t> cnt((size_t)range + 1, 0), out(tmp.size());
            for (uint32_t j : tmp) cnt[(size_t)(hi - key[j]) + 1]++;      // descending
            for (size_t i = 1; i < cnt.size(); ++i) cnt[i] += cnt[i - 1];
            for (uint32_t j : tmp) out[cnt[(size_t)(hi - key[j])]++] = j;
            tmp.swap(out);
        };
        counting(b->value, vmin, vmax);   // LSD: secondary key first
        counting(b->ub, umin, umax);
        b->order.swap(tmp);
        b->cursor = 0;
        return true;
    }
    /// Sequential half: the prepared block enters the heap.
    void commit_block(DevBlock* b) {
        b->id = next_id_++;
        dev_ref(b);
        open_ += b->order.size();
        heap_.push_back(b);
        sift_up(heap_.size() - 1);
    }
    /// best remaining node by (ub, value); false when nothing with ub > best_lb is left
    bool pop(LazyItem& out, int64_t best_lb) {
        while (!heap_.empty()) {
            DevBlock* b = heap_[0];
            if (b->head_ub() <= best_lb) {   // sorted by ub: the rest of the block is irrelevant too
                if (b == heap_[0] && heap_.size() >= 1) {
                    // the heap top carries the largest ub of the whole fringe: nothing relevant is left (parallel.rs:531-535)
                    clear();
                    return false;
                }
            }
            const int j = (int)b->order[b->cursor];
            out.block = b;
            out.row = j;
            out.depth = b->depth;
            out.value = b->value[j];
            out.ub = std::min<int64_t>(b->cap_ub, b->ub[j]);
            dev_ref(b);   // the caller's reference
            b->cursor++;
            open_--;
            if (b->cursor >= b->order.size()) {
                heap_[0] = heap_.back();
                heap_.pop_back();
                if (!heap_.empty()) sift_down(0);
                dev_unref(b);
            } else {
                sift_down(0);
            }
            return true;
        }
        return false;
    }
    int64_t best_ub() const { return heap_.empty() ? I64_MIN : heap_[0]->head_ub(); }
    void clear() {
        for (DevBlock* b : heap_) dev_unref(b);
        heap_.clear();
        open_ = 0;
    }

  private:
    static bool less(const DevBlock* a, const DevBlock* b) {   // MaxUB (subproblem_ranking.rs:86-90) on the block heads
        const int64_t ua = a->head_ub(), ub_ = b->head_ub();
        if (ua != ub_) return ua < ub_;
        const int64_t va = a->head_value(), vb = b->head_value();
        if (va != vb) return va < vb;
        return a->id > b->id;
    }
    void sift_up(size_t i) {
        while (i > 0) {
            size_t p = (i - 1) / 2;
            if (!less(heap_[p], heap_[i])) break;
            std::swap(heap_[p], heap_[i]);
            i = p;
        }
    }
    void sift_down(size_t i) {
        const size_t n = heap_.size();
        for (;;) {
            size_t l = 2 * i + 1, r = l + 1, m = i;
            if (l < n && less(heap_[m], heap_[l])) m = l;
            if (r < n && less(heap_[m], heap_[r])) m = r;
            if (m == i) break;
            std::swap(heap_[m], heap_[i]);
            i = m;
        }
    }
    std::vector<DevBlock*> heap_;
    size_t open_ = 0;
    uint64_t next_id_ = 0;
};

}  // namespace

struct ddo_solver {
    Model* model = nullptr;
    ddo_solver_config cfg{};
    std::shared_ptr<Engine> engine;
    CacheTable* cache = nullptr;         // SimpleCache in device memory (cfg.cache_entries > 0)
    DominanceTable* dominance = nullptr; // SimpleDominanceChecker in device memory (cfg.dominance_entries > 0)
    NoDupFringe* fringe = nullptr;
    LazyFringe* lazy = nullptr;          // DDO_FRINGE_LAZY
    std::vector<LazyItem> litems, flight;   // flight: the batch currently on the device
    // Critical (parallel.rs:32-81)
    uint64_t explored = 0;
    int64_t best_lb = I64_MIN;
    int64_t best_ub = I64_MAX;
    bool has_sol = false;
    std::vector<ddo_decision> best_sol;
    bool aborted = false;       // abort_proof
    bool initialized = false;
    bool finished = false;
    ddo_counters counters{};
    std::chrono::steady_clock::time_point t_start;
    // optional statistics (DDO_HIP_STATS=1): per-DD layers / nodes / widest layer
    std::vector<uint32_t> st_layers, st_maxw;
    std::vector<uint64_t> st_nodes;
    uint64_t st_clk[32] = {0};
    double st_t_fill = 0, st_t_launch = 0, st_t_wait = 0, st_t_wait_last = 0, st_t_wait_flush = 0;
    uint64_t st_bk_cnt[20] = {0}, st_bk_nodes[20] = {0}, st_bk_clk[20] = {0};   // by log2 of the DD's widest layer
    uint64_t st_push = 0, st_push_dup = 0;
    uint64_t st_recycled = 0;
    double st_host_pop = 0, st_host_run = 0, st_host_post = 0, st_host_fetch = 0;
    bool want_stats = false;
    // bench support (ddo_solver_bench_freeze / _bench_step): frozen batches of sub-problems compiled again and again
    std::vector<std::vector<LazyItem>> frozen;
    int64_t frozen_lb = I64_MIN;
    uint64_t frozen_mark = 0;
    size_t frozen_next = 0;
    bool bench_mode = false;
    // scratch
    std::vector<DDInput> inputs;
    std::vector<Entry> items;
    std::vector<HostResult> results;

    ~ddo_solver() {
        if (std::getenv("DDO_HIP_TIMES")) {   // host-side clocks only (DDO_HIP_STATS also makes every DD account its phases: slower)
            std::fprintf(stderr, "[ddo times] host s: pop %.3f run %.3f post %.3f (of which fetch %.3f) | dispatch: lists + inputs %.3f, launch() %.3f, "
                         "wait for a lower tier %.3f, wait for the last tier of the previous step %.3f (+ %.3f in flush)\n", st_host_pop, st_host_run, st_host_post, st_host_fetch,
                         st_t_fill, st_t_launch, st_t_wait, st_t_wait_last, st_t_wait_flush);
            for (size_t t = 0; t < tiers.size(); ++t)
                std::fprintf(stderr, "[ddo times] tier %zu: %llu launches, %llu sub-problems, %llu retried, kernels %.1f ms\n", t, (unsigned long long)st_tier_launch[t],
                             (unsigned long long)st_tier_items[t], (unsigned long long)st_tier_retry[t], tiers[t]->kernel_ms());
        }
        if (want_stats && !st_layers.empty()) {
            auto pct = [](std::vector<uint64_t> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
            std::vector<uint64_t> a(st_layers.begin(), st_layers.end()), b(st_maxw.begin(), st_maxw.end());
            uint64_t tl = 0, tn = 0;
            for (auto x : a) tl += x;
            for (auto x : st_nodes) tn += x;
            uint64_t tc = 0;
            for (int q = 0; q < 8; ++q) tc += st_clk[q];
            std::fprintf(stderr, "[ddo stats] host s: pop %.3f run %.3f post %.3f (of which fetch %.3f) | pushes %llu | node pool backed %.1f GB\n", st_host_pop,
                         st_host_run, st_host_post, st_host_fetch, (unsigned long long)st_push, engine ? engine->pool_capacity() / 1073741824.0 : 0.0);
            std::fprintf(stderr, "[ddo stats] dispatch s: lists + inputs %.3f, launch() %.3f, wait for a lower tier %.3f, wait for the last tier of the previous step %.3f\n",
                         st_t_fill, st_t_launch, st_t_wait, st_t_wait_last);
            if (engine && engine->engine_kind() == 1) {   // the layer-rebuilding engine's marks (DD1_TICK in misp_dd_core.hpp)
                const double d = 1e3 * (double)std::max<uint64_t>(1, tl);
                std::fprintf(stderr, "[ddo stats] layer-rebuilding engine, kcycles per layer: variable + cache filter %.1f, dominance filters %.1f, select %.1f, "
                             "classify + positions + merge %.1f, layer bookkeeping %.1f, expand + dedup %.1f, after the last layer %.1f, terminal layer %.1f\n",
                             st_clk[7] / d, st_clk[0] / d, st_clk[1] / d, st_clk[3] / d, st_clk[5] / d, st_clk[2] / d, st_clk[4] / d, st_clk[6] / d);
            } else
            std::fprintf(stderr, "[ddo stats] device kcycles per layer: misc(var,sweep,freelist,final,backward) %.1f select %.1f classify+tie-break %.1f victims+merge %.1f expand+dedup %.1f (unused %.1f %.1f) hand-over %.1f | total %.1f kcycles/layer, %.1f Mcycles/DD\n",
                         st_clk[0] / 1e3 / std::max<uint64_t>(1, tl), st_clk[1] / 1e3 / std::max<uint64_t>(1, tl), st_clk[2] / 1e3 / std::max<uint64_t>(1, tl),
                         st_clk[3] / 1e3 / std::max<uint64_t>(1, tl), st_clk[4] / 1e3 / std::max<uint64_t>(1, tl), st_clk[5] / 1e3 / std::max<uint64_t>(1, tl),
                         st_clk[6] / 1e3 / std::max<uint64_t>(1, tl), st_clk[7] / 1e3 / std::max<uint64_t>(1, tl),
                         tc / 1e3 / std::max<uint64_t>(1, tl), tc / 1e6 / a.size());
            for (size_t t = 0; t < tiers.size(); ++t)
                std::fprintf(stderr, "[ddo stats] tier %zu: layer capacity %d, %d slots x %d threads, LDS %zu B | %llu launches, %llu sub-problems, %llu retried (%.1f %%), kernels %.1f ms\n",
                             t, tiers[t]->is_tier() ? tiers[t]->cap_width() : (int)tiers[t]->max_width(), tiers[t]->nslots(), tiers[t]->threads(), tiers[t]->lds_bytes(),
                             (unsigned long long)st_tier_launch[t], (unsigned long long)st_tier_items[t], (unsigned long long)st_tier_retry[t],
                             100.0 * st_tier_retry[t] / std::max<uint64_t>(1, st_tier_items[t]), tiers[t]->kernel_ms());
            if (engine && engine->kernel_ms() > 0)   // busy share of the slots: sum of per-DD shader clocks vs slots x kernel time
                std::fprintf(stderr, "[ddo stats] slot utilisation: %.1f %% (sum of DD cycles %.3g over %d slots x %.1f ms of kernels at 2.4 GHz)\n",
                             100.0 * (double)tc / (2.4e6 * engine->kernel_ms() * engine->nslots()), (double)tc, engine->nslots(), engine->kernel_ms());
            std::fprintf(stderr, "[ddo stats] per layer: lex-ties %.1f lex-selects %.3f digit-rounds %.2f lex-words %.2f (tied at word start %.1f) | worklist %.1f (contain var %.1f) records %.1f victims %.1f squashes %.3f |",
                         (double)st_clk[8] / tl, (double)st_clk[9] / tl, (double)st_clk[10] / tl, (double)st_clk[11] / tl, (double)st_clk[16] / tl,
                         (double)st_clk[12] / tl, (double)st_clk[15] / tl, (double)st_clk[13] / tl, (double)st_clk[14] / tl, (double)st_clk[17] / tl);
            for (int q = 18; q < 24; ++q) std::fprintf(stderr, " aux%d %.2f", q - 8, st_clk[q] / 1e3 / std::max<uint64_t>(1, tl));
            if (!st_clk[31] && (st_clk[24] | st_clk[25]))   // (without PROBES the slots carry finer ticks of the tie-break)
                std::fprintf(stderr, " | classify sweep %.2f tie-break: gather %.2f pivot %.2f partition %.2f kcycles/layer", st_clk[24] / 1e3 / std::max<uint64_t>(1, tl),
                             st_clk[25] / 1e3 / std::max<uint64_t>(1, tl), st_clk[26] / 1e3 / std::max<uint64_t>(1, tl), st_clk[27] / 1e3 / std::max<uint64_t>(1, tl));
            if (st_clk[31]) {   // thread 0's own node in expand (make PROBES=1): the dependent chain, cycles per probed node
                const double np = (double)st_clk[31];
                std::fprintf(stderr, " | expand chain per node (thread 0, %.0f probes): loads %.0f, children + stores + fence %.0f, table insert NO %.0f + its bookkeeping %.0f, table insert YES %.0f + its bookkeeping %.0f, record %.0f",
                             np, st_clk[24] / np, st_clk[25] / np, st_clk[26] / np, st_clk[29] / np, st_clk[27] / np, st_clk[30] / np, st_clk[28] / np);
            }
            std::fprintf(stderr, " | recycled merges per layer %.4f", (double)st_recycled / tl);
            std::fprintf(stderr, "\n");
            for (int bk = 0; bk < 20; ++bk)
                if (st_bk_cnt[bk])
                    std::fprintf(stderr, "[ddo stats] widest layer <= %6u: %10llu DDs %14llu nodes (%5.1f %%) %8.3f Gcycles (%5.1f %%)  %.0f cycles/node\n", 1u << bk,
                                 (unsigned long long)st_bk_cnt[bk], (unsigned long long)st_bk_nodes[bk], 100.0 * st_bk_nodes[bk] / std::max<uint64_t>(1, tn),
                                 st_bk_clk[bk] / 1e9, 100.0 * st_bk_clk[bk] / std::max<uint64_t>(1, tc), (double)st_bk_clk[bk] / std::max<uint64_t>(1, st_bk_nodes[bk]));
            std::fprintf(stderr, "[ddo stats] DDs %zu  layers: mean %.1f p50 %llu p90 %llu max %llu | widest layer: p10 %llu p50 %llu p75 %llu p90 %llu p99 %llu max %llu | nodes/DD: mean %.0f p50 %llu p90 %llu max %llu | nodes/layer mean %.1f\n",
                         a.size(), (double)tl / a.size(), (unsigned long long)pct(a, .5), (unsigned long long)pct(a, .9), (unsigned long long)pct(a, 1.0),
                         (unsigned long long)pct(b, .1), (unsigned long long)pct(b, .5), (unsigned long long)pct(b, .75), (unsigned long long)pct(b, .9), (unsigned long long)pct(b, .99), (unsigned long long)pct(b, 1.0),
                         (double)tn / a.size(), (unsigned long long)pct(st_nodes, .5), (unsigned long long)pct(st_nodes, .9), (unsigned long long)pct(st_nodes, 1.0), (double)tn / std::max<uint64_t>(1, tl));
        }
        for (auto& b : frozen)
            for (LazyItem& e : b) dev_unref(e.block);
        for (LazyItem& e : flight) dev_unref(e.block);
        for (auto& pr : todo)
            for (LazyItem& e : pr.first) dev_unref(e.block);
        delete fringe;
        delete lazy;
        delete cache;
        delete dominance;
    }

    long engine_width() const {   // the largest width any sub-problem can be given: sizes the device workspace
        size_t w = 1;
        for (int d = 0; d <= model->n; ++d) w = std::max(w, ddo_width_heuristic(&cfg, (size_t)model->n, (size_t)d));
        return (long)w;
    }
    /// WidthHeuristic::max_width (width.rs:168-170 / :399-401; path.len() == depth for MISP)
    /// (a Pooled sub-problem's path is shorter than its depth: NbUnassignedWidth counts the path, width.rs:399-401)
    int width_of(const Entry& e) const { return (int)ddo_width_heuristic(&cfg, (size_t)model->n, (size_t)std::max(e.plen >= 0 ? e.plen : e.depth, 0)); }
    bool budget_exhausted() const {
        if (cfg.time_budget_s <= 0) return false;
        double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        return el >= cfg.time_budget_s;
    }

    /// parallel.rs:368-385
    void initialize() {
        if (initialized) return;
        initialized = true;
        t_start = std::chrono::steady_clock::now();
        CutsetBlock* root = new CutsetBlock();
        root->ws = model->ws;
        root->states.assign(model->ws, 0);
        model->initial_state(root->states.data());   // Problem::initial_state
        root->values.push_back((int32_t)model->initial_value);
        Entry e{root, 0, 0, model->initial_value, I64_MAX, hash_words(root->states.data(), model->ws)};
        block_ref(root);   // keep alive while pushing
        fringe->push(e);
        block_unref(root);
    }

    /// maybe_update_best (parallel.rs:446-453)
    void maybe_update_best(const Entry& it, const HostResult& r) {
        if (!r.hdr.has_best_exact) return;
        int64_t v = r.hdr.best_exact_value;
        if (v > best_lb) {
            best_lb = v;
            best_sol.clear();
            materialize_path(model, it.block, it.row, best_sol);
            const std::vector<uint32_t>& p = r.hdr.exact_same_as_best ? r.best_path : r.exact_path;
            for (uint32_t x : p) best_sol.push_back(model->path_decision(x));
            has_sol = true;
        }
    }

    /// enqueue_cutset (parallel.rs:456-469); at the root of a sharded run the cut-set is dealt
    /// round-robin over the ranks in fringe order.
    void enqueue_cutset(const Entry& it, HostResult& r) {
        if (r.n_cutset == 0) return;
        CutsetBlock* b = new CutsetBlock();
        b->parent = it.block;
        b->parent_row = it.row;
        block_ref(it.block);
        b->depth = it.depth + r.cs_path_len;
        b->path_len = r.cs_path_len;
        b->ws = model->ws;
        b->states = std::move(r.cs_state);
        b->values = std::move(r.cs_value);
        b->paths = std::move(r.cs_path);
        b->row_len = std::move(r.cs_depth);
        b->row_plen = std::move(r.cs_plen);
        block_ref(b);
        std::vector<int> order(r.n_cutset);
        for (int j = 0; j < r.n_cutset; ++j) order[j] = j;
        const bool deal = cfg.world_size > 1 && it.depth == 0 && it.block->parent == nullptr;
        if (!b->row_len.empty() && !deal)
            // frontier cut-set: the reference collects it bottom-up (clean.rs:586-606), so the deepest nodes reach the fringe
            // first -- and the NoDupFringe keeps the FIRST of two copies of a state with equal values (no_duplicate.rs:110).
            // MISP nodes keep their state from layer to layer, so one cut-set can hold the same state at two depths.
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return b->row_len[(size_t)x] > b->row_len[(size_t)y]; });
        if (deal) {
            std::sort(order.begin(), order.end(), [&](int x, int y) {
                int64_t ux = std::min<int64_t>(it.ub, r.cs_ub[x]), uy = std::min<int64_t>(it.ub, r.cs_ub[y]);
                if (ux != uy) return ux > uy;
                if (b->values[x] != b->values[y]) return b->values[x] > b->values[y];
                const int cs = model->compare_states(b->state(x), b->state(y));
                if (cs != 0) return cs > 0;
                // a frontier cut-set can hold one state at two depths with equal bound and value: the depth makes the order
                // strict and total, so that every rank sorts the rows alike and a row is owned by exactly one of them
                const int dx = b->row_len.empty() ? 0 : b->row_len[(size_t)x], dy = b->row_len.empty() ? 0 : b->row_len[(size_t)y];
                return dx > dy;
            });
        }
        for (int k = 0; k < r.n_cutset; ++k) {
            const int j = order[k];
            if (deal && (k % cfg.world_size) != cfg.rank) continue;
            int64_t ub = std::min<int64_t>(it.ub, r.cs_ub[j]);   // :460
            if (ub > best_lb) {                                   // :461
                const int dj = b->row_len.empty() ? b->depth : it.depth + b->row_len[(size_t)j];
                Entry e{b, j, dj, b->values[j], ub, hash_words(b->state(j), model->ws)};
                if (!b->row_plen.empty()) e.plen = (it.plen >= 0 ? it.plen : it.depth) + b->row_plen[(size_t)j];
                fringe->push(e);
                st_push += 1;
            }
        }
        block_unref(b);
    }

    /// problem-root path of a pool node: decisions stored as bit strings + per-block variable lists in HBM
    /// per export call: what materialize_pool_path has read from the device so far -- a block's header and branching vertices,
    /// the decisions of a (block, row) segment.  The nodes a rank hands over are the best of its fringe: siblings and cousins
    /// that share nearly all of their ancestor chains (round 3: 1.5 M nodes handed over in a two-rank search cost minutes of
    /// 8-byte device reads, one per header, vertex list and path word of every ancestor of every node).
    struct BlockHdr { PoolBlockHeader h; std::vector<uint32_t> lvar; };
    std::unordered_map<const DevBlock*, BlockHdr> x_hdr;
    std::map<std::pair<const DevBlock*, int>, std::vector<ddo_decision>> x_seg;
    bool x_cache = false;

    int materialize_pool_path(const DevBlock* b, int row, std::vector<ddo_decision>& out) {
        std::vector<std::pair<const DevBlock*, int>> chain;
        while (b) {
            chain.push_back({b, row});
            row = b->parent_row;
            b = b->parent;
        }
        for (auto it = chain.rbegin(); it != chain.rend(); ++it) {
            const DevBlock* blk = it->first;
            if (!blk->host_paths.empty()) {   // handed over by another rank: the path travelled with the node
                const auto& hp = blk->host_paths[(size_t)it->second];
                out.insert(out.end(), hp.begin(), hp.end());
                continue;
            }
            if (blk->off == NO_POOL_SRC || blk->lel == 0) continue;
            if (x_cache) {
                auto f = x_seg.find({blk, it->second});
                if (f != x_seg.end()) {
                    out.insert(out.end(), f->second.begin(), f->second.end());
                    continue;
                }
            }
            BlockHdr local;
            BlockHdr* bh = &local;
            bool have = false;
            if (x_cache) {
                auto f = x_hdr.find(blk);
                if (f != x_hdr.end()) {
                    bh = &f->second;
                    have = true;
                }
            }
            int rc;
            if (!have) {
                if ((rc = engine->read_pool(blk->off, &local.h, sizeof(local.h))) != DDO_OK) return rc;
                local.lvar.resize(local.h.lel);
                if ((rc = engine->read_pool(blk->off + local.h.off_lvar, local.lvar.data(), (size_t)local.h.lel * 4)) != DDO_OK) return rc;
                if (x_cache) bh = &(x_hdr[blk] = std::move(local));
            }
            const PoolBlockHeader& h = bh->h;
            std::vector<uint64_t> bits(h.pw);
            if ((rc = engine->read_pool_strided(blk->off + h.off_paths + (uint64_t)it->second * 8, bits.data(), h.pw, (size_t)h.rows * 8)) != DDO_OK) return rc;
            const size_t before = out.size();
            for (uint32_t tr = 0; tr < h.lel; ++tr)   // root side first
                out.push_back(ddo_decision{(int64_t)bh->lvar[tr], (int64_t)((bits[tr >> 6] >> (tr & 63)) & 1ULL)});
            if (x_cache) x_seg[{blk, it->second}].assign(out.begin() + (long)before, out.end());
        }
        return DDO_OK;
    }

    // ---- work hand-over between ranks ("optional send/recv of sub-problem batches") ----------
    /// Pops up to `max_count` open sub-problems in fringe order and writes them out as self-contained records: state
    /// words, value, ub, depth and the decisions from the problem root (path_off[i] .. path_off[i+1] in `paths`).
    /// Stops early when `path_cap` decisions do not suffice.  The nodes leave this solver for good.
    int export_nodes(size_t max_count, uint64_t* states, int64_t* value, int64_t* ub, int64_t* depth, uint64_t* path_off,
                     ddo_decision* paths, size_t path_cap, size_t* n_out) {
        *n_out = 0;
        int rc = lazy ? flush_lazy() : DDO_OK;
        if (rc != DDO_OK) return rc;
        const int ws = model->ws;
        size_t np = 0;
        path_off[0] = 0;
        std::vector<ddo_decision> path;
        struct CacheScope {   // the caches live for this call only (a DevBlock may be freed and its address reused afterwards)
            ddo_solver* s;
            explicit CacheScope(ddo_solver* s_) : s(s_) { s->x_cache = true; }
            ~CacheScope() {
                s->x_cache = false;
                s->x_hdr.clear();
                s->x_seg.clear();
            }
        } scope(this);
        while (*n_out < max_count) {
            path.clear();
            const size_t i = *n_out;
            if (lazy) {
                LazyItem it;
                // a node whose path may not fit is not popped: worst case is one decision per variable
                if (np + (size_t)model->n > path_cap || !lazy->pop(it, best_lb)) break;
                std::vector<uint64_t> row((size_t)std::max(ws, engine->words_per_state_device()), 0);
                if (it.block->off == NO_POOL_SRC) model->initial_state(row.data());
                else {
                    // (the block's header comes out of the call's cache once its path segment below has been built; here: the one
                    // strided copy of the state's words)
                    PoolBlockHeader h;
                    auto fh = x_hdr.find(it.block);
                    if (fh != x_hdr.end()) h = fh->second.h;
                    else if ((rc = engine->read_pool(it.block->off, &h, sizeof(h))) != DDO_OK) return rc;
                    const size_t nwords = std::min<size_t>(h.ws, row.size());
                    if ((rc = engine->read_pool_strided(it.block->off + h.off_states + (uint64_t)it.row * 8, row.data(), nwords, (size_t)h.rows * 8)) != DDO_OK) return rc;
                }
                std::memcpy(states + i * (size_t)ws, row.data(), (size_t)ws * 8);
                value[i] = it.value;
                ub[i] = it.ub;
                depth[i] = it.depth;
                rc = materialize_pool_path(it.block, it.row, path);
                dev_unref(it.block);
                if (rc != DDO_OK) return rc;
            } else {
                Entry e;
                if (np + (size_t)model->n > path_cap || !fringe->pop(e)) break;
                if (e.ub <= best_lb) {   // nothing relevant is left (parallel.rs:531-535)
                    block_unref(e.block);
                    fringe->clear();
                    break;
                }
                std::memcpy(states + i * (size_t)ws, e.block->state(e.row), (size_t)ws * 8);
                value[i] = e.value;
                ub[i] = e.ub;
                depth[i] = e.depth;
                materialize_path(model, e.block, e.row, path);
                block_unref(e.block);
            }
            std::memcpy(paths + np, path.data(), path.size() * sizeof(ddo_decision));
            np += path.size();
            path_off[i + 1] = np;
            *n_out += 1;
        }
        return DDO_OK;
    }

    /// Takes over sub-problems exported by another rank's solver (same model).  With the lazy fringe the states are
    /// written into this device's node pool (one block per depth), the paths stay on the host.
    int import_nodes(size_t count, const uint64_t* states, const int64_t* value, const int64_t* ub, const int64_t* depth,
                     const uint64_t* path_off, const ddo_decision* paths) {
        if (count == 0) return DDO_OK;
        if (!initialized) {   // every rank compiles the root itself (its share of the root cut-set is its initial fringe)
            set_error("ddo_solver_import_subproblems: call ddo_solver_step at least once before importing");
            return DDO_ERR_INVALID;
        }
        const int ws = model->ws;
        if (lazy) {
            int rc = flush_lazy();
            if (rc != DDO_OK) return rc;
            std::map<int64_t, std::vector<size_t>> by_depth;
            for (size_t i = 0; i < count; ++i)
                if (ub[i] > best_lb) by_depth[depth[i]].push_back(i);
            const uint32_t wsT = (uint32_t)engine->words_per_state_device();
            for (auto& kv : by_depth) {
                const std::vector<size_t>& idx = kv.second;
                const uint32_t rows = (uint32_t)idx.size();
                std::vector<uint8_t> blk((size_t)pool_block_bytes(rows, wsT, 0), 0);
                PoolBlockHeader* h = (PoolBlockHeader*)blk.data();
                h->rows = rows;
                h->ws = wsT;
                h->lel = 0;
                h->depth = (uint32_t)kv.first;
                h->parent_off = NO_POOL_SRC;
                h->parent_row = 0;
                h->pw = 0;
                h->off_lvar = 64;
                h->off_states = 64;
                h->off_paths = h->off_states + (uint64_t)wsT * rows * 8;
                h->off_values = h->off_paths;
                h->off_ubs = h->off_values + (((uint64_t)rows * 4 + 7) & ~7ULL);
                uint64_t* st = (uint64_t*)(blk.data() + h->off_states);
                int32_t* pv = (int32_t*)(blk.data() + h->off_values);
                int32_t* pu = (int32_t*)(blk.data() + h->off_ubs);
                DevBlock* b = new DevBlock();
                b->rows = (int)rows;
                b->depth = (int)kv.first;
                b->lel = 0;
                b->value.resize(rows);
                b->ub.resize(rows);
                b->host_paths.resize(rows);
                for (uint32_t j = 0; j < rows; ++j) {
                    const size_t i = idx[j];
                    for (int k = 0; k < ws; ++k) st[(size_t)k * rows + j] = states[i * (size_t)ws + (size_t)k];
                    pv[j] = b->value[j] = (int32_t)value[i];
                    pu[j] = b->ub[j] = (int32_t)std::min<int64_t>(ub[i], INT32_MAX);
                    b->host_paths[j].assign(paths + path_off[i], paths + path_off[i + 1]);
                }
                if ((rc = engine->pool_append(blk.data(), blk.size(), &b->off)) != DDO_OK) {
                    delete b;
                    return rc;
                }
                dev_ref(b);
                lazy->push_block(b, best_lb);
                dev_unref(b);
            }
        } else {
            for (size_t i = 0; i < count; ++i) {
                if (ub[i] <= best_lb) continue;
                CutsetBlock* b = new CutsetBlock();   // one block per node: paths differ in length and variables
                b->depth = (int)depth[i];
                b->path_len = (int)(path_off[i + 1] - path_off[i]);
                b->ws = ws;
                b->states.assign(states + i * (size_t)ws, states + (i + 1) * (size_t)ws);
                b->values.push_back((int32_t)value[i]);
                b->paths.resize((size_t)b->path_len);
                for (int k = 0; k < b->path_len; ++k) {
                    const ddo_decision& d = paths[path_off[i] + (size_t)k];
                    b->paths[(size_t)k] = model->path_word(d);
                }
                block_ref(b);
                Entry imported{b, 0, b->depth, value[i], ub[i], hash_words(b->states.data(), ws)};
                if (cfg.pooled) imported.plen = b->path_len;
                fringe->push(imported);
                block_unref(b);
            }
        }
        finished = false;
        return DDO_OK;
    }
    /// Sharded search: the root cut-set is dealt over the ranks.  Every rank compiles the root
    /// itself and obtains the same SET of rows, but in an order that depends on the scheduling of device atomics, so
    /// the owner of a row is hash(state) % world_size -- the states are read back from the node pool once.
    int deal_root_rows(DevBlock* b) {
        PoolBlockHeader h;
        int rc = engine->read_pool(b->off, &h, sizeof(h));
        if (rc != DDO_OK) return rc;
        std::vector<uint64_t> st((size_t)h.ws * h.rows);
        if ((rc = engine->read_pool(b->off + h.off_states, st.data(), st.size() * 8)) != DDO_OK) return rc;
        b->mine.assign(h.rows, 0);
        std::vector<uint64_t> row(std::max<uint32_t>(h.ws, (uint32_t)model->ws));
        for (uint32_t j = 0; j < h.rows; ++j) {
            for (uint32_t k = 0; k < h.ws; ++k) row[k] = st[(size_t)k * h.rows + j];   // word-major in the pool
            b->mine[j] = (int)(hash_words(row.data(), model->ws) % (uint64_t)cfg.world_size) == cfg.rank ? 1 : 0;
        }
        return DDO_OK;
    }

    /// results of a finished lazy launch -> incumbent, counters, new cut-set blocks (parallel.rs:420-434)
    /// a finished launch whose results are not folded into the fringe yet: its items and the raw result records (2 per item;
    /// skip[i]: the item was handed up to the next tier, its records mean nothing)
    struct Done {
        std::vector<LazyItem> first;
        Engine::RawBatch raw;
        std::vector<uint8_t> skip;
        int tier = 0;
    };
    std::vector<Done> todo;

    /// Folds the results of one finished launch into counters, incumbent and fringe, reading the result records and the
    /// output arena where the device left them (pinned host memory, Engine::fetch_raw).
    int absorb_lazy(Done& d) {
        std::vector<LazyItem>& its = d.first;
        const Engine::RawBatch& raw = d.raw;
        int err = DDO_OK;
        std::vector<DevBlock*> fresh_blocks;
        if ((size_t)raw.count != its.size()) err = DDO_ERR_INTERNAL;
        for (size_t i = 0; i < its.size() && err == DDO_OK; ++i) {
            if (!d.skip.empty() && d.skip[i]) continue;
            for (int k = 0; k < 2; ++k) {
                const DDResult& h = raw.hdr[2 * i + k];
                if (h.status == ST_NOT_RUN) continue;
                const bool overflow = h.status == ST_OK && h.arena_off + h.arena_bytes > raw.arena_used;
                if (h.status != ST_OK || overflow) {
                    const bool capacity = overflow || h.status == ST_ERR_CAPACITY || h.status <= -100;
                    set_error(capacity ? "device capacity exhausted (site " + std::to_string(h.status) +
                                             "): node pool / output arena / workspace; raise DDO_HIP_POOL_GB or use DDO_FRINGE_NODUP"
                                       : "device compile failed with status " + std::to_string(h.status));
                    err = h.status == ST_CUTOFF ? DDO_CUTOFF : (capacity ? DDO_ERR_CAPACITY : DDO_ERR_INTERNAL);
                    break;
                }
                counters.nodes_expanded += h.nodes_expanded;
                counters.arcs += h.arcs;
                counters.layers += h.layers;
                counters.compiles += 1;
                st_tier_nodes[d.tier] += h.nodes_expanded;
                if (want_stats) {
                    st_layers.push_back((uint32_t)h.layers);
                    st_maxw.push_back(h.max_width_seen);
                    st_nodes.push_back(h.nodes_expanded);
                    for (int q = 0; q < 32; ++q) st_clk[q] += h.phase_clk[q];
                    st_recycled += h.recycled_merges;
                    int bk = 0;
                    while (bk < 19 && (1u << bk) < h.max_width_seen) ++bk;
                    st_bk_cnt[bk] += 1;
                    st_bk_nodes[bk] += h.nodes_expanded;
                    for (int q = 0; q < 8; ++q) st_bk_clk[bk] += h.phase_clk[q];
                }
                const uint8_t* base = raw.arena + h.arena_off;
                if (!bench_mode && h.has_best_exact && (int64_t)h.best_exact_value > best_lb) {   // maybe_update_best
                    best_lb = h.best_exact_value;
                    best_sol.clear();
                    if ((err = materialize_pool_path(its[i].block, its[i].row, best_sol)) != DDO_OK) break;
                    // (the best exact path: stored once when it is also the best path -- the kernel then points exact_off at it)
                    const uint32_t* p = (const uint32_t*)(base + h.exact_off);
                    for (int q = 0; q < h.exact_len; ++q) best_sol.push_back(model->path_decision(p[q]));
                    has_sol = true;
                }
                const bool exact = h.is_exact || h.has_exact_best_path;
                if (k == 1 && !exact && h.n_cutset > 0 && h.pool_off != NO_POOL_SRC) {   // enqueue_cutset
                    DevBlock* b = new DevBlock();
                    b->parent = its[i].block;
                    b->parent_row = its[i].row;
                    dev_ref(its[i].block);
                    b->off = h.pool_off;
                    b->rows = h.n_cutset;
                    b->lel = h.cs_path_stride;
                    b->depth = its[i].depth + h.cs_path_stride;
                    b->cap_ub = its[i].ub;
                    const int32_t* v = (const int32_t*)(base + h.cs_value_off);
                    const int32_t* u = (const int32_t*)(base + h.cs_ub_off);
                    b->value.assign(v, v + h.n_cutset);
                    b->ub.assign(u, u + h.n_cutset);
                    dev_ref(b);
                    st_push += (uint64_t)b->rows;
                    if (cfg.world_size > 1 && its[i].block->parent == nullptr && its[i].block->off == NO_POOL_SRC &&
                        (err = deal_root_rows(b)) != DDO_OK) {
                        dev_unref(b);
                        break;
                    }
                    fresh_blocks.push_back(b);
                }
            }
        }
        // The cut-set blocks are filtered against the final incumbent and ordered (two counting sorts each) by a few
        // host threads, then enter the heap in submission order: this is the only host work of a step that is
        // proportional to the number of cut-set nodes, and it runs behind the next launch.
        if (!fresh_blocks.empty()) {
            std::vector<char> keep(fresh_blocks.size(), 0);
            const int nthreads = (int)std::min<size_t>(std::max(1u, std::min(16u, std::thread::hardware_concurrency() / 4)), fresh_blocks.size());
            auto work = [&](int t) {
                for (size_t q = (size_t)t; q < fresh_blocks.size(); q += (size_t)nthreads)
                    keep[q] = LazyFringe::prepare_block(fresh_blocks[q], best_lb) ? 1 : 0;
            };
            if (nthreads > 1) {
                std::vector<std::thread> pool;
                for (int t = 1; t < nthreads; ++t) pool.emplace_back(work, t);
                work(0);
                for (std::thread& th : pool) th.join();
            } else {
                work(0);
            }
            for (size_t q = 0; q < fresh_blocks.size(); ++q) {
                if (keep[q] && !bench_mode) lazy->commit_block(fresh_blocks[q]);   // bench: the ordered block is dropped
                dev_unref(fresh_blocks[q]);
            }
        }
        for (size_t i = 0; i < its.size(); ++i)
            if (d.skip.empty() || !d.skip[i]) dev_unref(its[i].block);   // (handed-up items carried their reference to the next tier)
        its.clear();
        return err;
    }

    // ---- capacity tiers -----------------------------------------------------------------------------------------
    // A best-first search is made of SMALL decision diagrams (brock400_1, W = 10 000: half of the 72 M DDs never hold more
    // than 57 nodes in a layer; DDs that stay within 256 nodes take 37 % of the device time for 4 % of the nodes), and
    // a narrow DD is pure latency: a layer costs a fixed chain of dependent memory round trips.  The full-width engine
    // runs one DD per CU (its LDS); a tier engine has slots for narrow DDs only and runs 12 of them per CU.  Every
    // sub-problem starts in the lowest tier its depth has not been seen to outgrow; a DD that outgrows a tier comes
    // back ST_RETRY (cheap: it failed while it was still small) and moves up.  Results are identical by construction:
    // a tier never squashes, so whatever it completes is what the full-width engine would have produced.
    std::vector<std::shared_ptr<Engine>> tiers;   // ascending capacity; tiers.back() == engine
    int flight_tier = -1;                          // engine of the launch in flight (flight = its items)
    static constexpr int HINT_DEPTHS = 1024;
    struct TierHint { uint32_t tried[4] = {0, 0, 0, 0}, retried[4] = {0, 0, 0, 0}; };
    std::vector<TierHint> hints;
    uint64_t probe_ctr = 0;
    uint64_t st_tier_items[5] = {0, 0, 0, 0, 0}, st_tier_retry[5] = {0, 0, 0, 0, 0}, st_tier_launch[5] = {0, 0, 0, 0, 0}, st_tier_nodes[5] = {0, 0, 0, 0, 0};

    bool pending() const { return !flight.empty() || !todo.empty(); }

    // how wide a DD gets depends on its depth and on how far its bound lies above the incumbent (everything whose rough upper
    // bound does not beat the incumbent is pruned): the retry statistics are kept per (depth, ub - incumbent) cell
    static constexpr int HINT_SLACKS = 16;
    int64_t hint_lb = 0;
    int tier_skip_pct = 90;   // a tier is skipped by the cells in which more than this share of its DDs outgrew it
    bool hint_by_slack = true;
    size_t hint_cell(const LazyItem& e) const {
        const size_t d = (size_t)std::min(std::max(e.depth, 0), HINT_DEPTHS - 1);
        if (!hint_by_slack) return d * HINT_SLACKS;
        const int64_t sl = e.ub - hint_lb;
        return d * HINT_SLACKS + (size_t)std::min<int64_t>(std::max<int64_t>(sl, 0), HINT_SLACKS - 1);
    }
    int start_tier(const LazyItem& e) {
        const int T = (int)tiers.size();
        if (T <= 1) return 0;
        if (hints.empty()) hints.resize((size_t)HINT_DEPTHS * HINT_SLACKS);
        const TierHint& h = hints[hint_cell(e)];
        const bool probe = !bench_mode && (++probe_ctr & 63) == 0;   // keep sampling the lower tiers: the search moves on (a frozen bench batch does not)
        for (int t = 0; t + 1 < T; ++t)
            if (probe || h.tried[t] < 32 || (uint64_t)h.retried[t] * 100 < (uint64_t)h.tried[t] * (uint64_t)tier_skip_pct) return t;
        return T - 1;
    }
    void note_tier(const LazyItem& e, int t, bool retried) {
        if (t + 1 >= (int)tiers.size()) return;
        TierHint& h = hints[hint_cell(e)];
        h.tried[t] += 1;
        h.retried[t] += retried ? 1 : 0;
        if (h.tried[t] >= 8192) {
            h.tried[t] >>= 1;
            h.retried[t] >>= 1;
        }
    }
    static void drop_items(std::vector<LazyItem>& v) {
        for (LazyItem& e : v) dev_unref(e.block);
        v.clear();
    }
    int absorb_todo() {
        int err = DDO_OK;
        for (Done& d : todo) {
            if (err == DDO_OK) err = absorb_lazy(d);
            else {
                for (size_t i = 0; i < d.first.size(); ++i)
                    if (d.skip.empty() || !d.skip[i]) dev_unref(d.first[i].block);
                d.first.clear();
            }
        }
        todo.clear();
        return err;
    }

    /// one synchronous launch of `cur` on tier t: the finished items go to `todo` (not folded in yet), the handed-up ones to `up`
    int run_tier_sync(int t, std::vector<LazyItem>& cur, int64_t lb, std::vector<LazyItem>& up) {
        DDInput* staged = tiers[(size_t)t]->stage_inputs((int)cur.size());
        if (!staged) {
            drop_items(cur);
            return DDO_ERR_INTERNAL;
        }
        fill_lazy_inputs(cur, lb, staged);
        int rc = tiers[(size_t)t]->launch(nullptr, (int)cur.size());
        if (rc != DDO_OK) {
            drop_items(cur);
            return rc;
        }
        st_tier_launch[t] += 1;
        st_tier_items[t] += cur.size();
        return settle(t, cur, up);
    }
    /// waits for tier t's launch of `cur`, splits it by status: handed-up items (ST_RETRY) move to `up`, the records of the
    /// others are queued in `todo`.  Takes over `cur`.
    int settle(int t, std::vector<LazyItem>& cur, std::vector<LazyItem>& up) {
        Done dn;
        auto t0 = std::chrono::steady_clock::now();
        int rc = tiers[(size_t)t]->wait();
        st_t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (rc == DDO_OK) rc = tiers[(size_t)t]->peek_retry(dn.skip);
        if (rc == DDO_OK && dn.skip.size() != cur.size()) rc = DDO_ERR_INTERNAL;
        if (rc == DDO_OK) rc = tiers[(size_t)t]->fetch_raw(dn.raw);
        if (rc != DDO_OK) {
            drop_items(cur);
            return rc;
        }
        for (size_t i = 0; i < cur.size(); ++i) {
            note_tier(cur[i], t, dn.skip[i] != 0);
            if (dn.skip[i]) {
                st_tier_retry[t] += 1;
                up.push_back(cur[i]);
            }
        }
        dn.first = std::move(cur);
        dn.tier = t;
        cur.clear();
        todo.push_back(std::move(dn));
        return DDO_OK;
    }
    /// The launch a step left in flight has to leave the device: its finished items are queued in `todo`; items it handed up
    /// (only when the launch left in flight was not the last tier's) are appended to lists[tier + 1].
    int settle_flight(std::vector<std::vector<LazyItem>>& lists) {
        if (flight_tier < 0) return DDO_OK;
        const int t = flight_tier;
        flight_tier = -1;
        std::vector<LazyItem> up;
        int rc = settle(t, flight, up);
        flight.clear();
        if (rc != DDO_OK) return rc;
        if (!up.empty()) {
            if (t + 1 >= (int)tiers.size()) {
                drop_items(up);
                return DDO_ERR_INTERNAL;   // the last tier has nowhere to hand a DD
            }
            for (LazyItem& e : up) lists[(size_t)t + 1].push_back(e);
        }
        return DDO_OK;
    }

    /// Compiles `batch` (restricted + relaxed DD per sub-problem, parallel.rs:391-437) through the tiers.  A lower tier's
    /// launch is waited for (who is handed up?) before the next tier's goes out; the LAST launch of a call -- the first tier
    /// above which nothing is queued: nearly always the dense tier, the long one -- stays in flight when the call returns, and
    /// the host's share of the step (folding the finished records of this and the previous step into counters, incumbent and
    /// fringe) runs while it is on the device.  Whatever that launch hands up joins the next call's batch.
    /// Takes over the references held by `batch`.
    int dispatch(std::vector<LazyItem>& batch, int64_t lb, bool rewind) {
        const int T = (int)tiers.size();
        std::vector<std::vector<LazyItem>> lists((size_t)T);
        hint_lb = lb > -((int64_t)1 << 40) ? lb : 0;
        for (LazyItem& e : batch) lists[(size_t)start_tier(e)].push_back(e);
        batch.clear();
        auto fail = [&](int rc) {
            for (auto& l : lists) drop_items(l);
            return rc;
        };
        auto t_run0 = std::chrono::steady_clock::now();
        // tiers run one after the other (a full-width workgroup owns a whole CU): the launch in flight must have left
        int rc = settle_flight(lists);
        st_t_wait_last += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_run0).count();
        if (rc != DDO_OK) return fail(rc);
        auto host_share = [&]() -> int {
            auto t1 = std::chrono::steady_clock::now();
            int r = absorb_todo();
            st_host_post += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
            return r;
        };
        bool shared = false;
        for (int t = 0; t < T; ++t) {
            if (lists[(size_t)t].empty()) continue;
            std::vector<LazyItem>& cur = lists[(size_t)t];
            bool last = true;   // nothing queued above: this launch ends the call (what it hands up joins the next one)
            for (int u = t + 1; u < T; ++u) last = last && lists[(size_t)u].empty();
            t_run0 = std::chrono::steady_clock::now();
            DDInput* staged = tiers[(size_t)t]->stage_inputs((int)cur.size());
            if (!staged) return fail(DDO_ERR_INTERNAL);
            fill_lazy_inputs(cur, lb, staged);
            if (rewind) {
                tiers[(size_t)t]->set_pool_rewind(frozen_mark);
                rewind = false;
            }
            auto tl0 = std::chrono::steady_clock::now();
            st_t_fill += std::chrono::duration<double>(tl0 - t_run0).count();
            if ((rc = tiers[(size_t)t]->launch(nullptr, (int)cur.size())) != DDO_OK) return fail(rc);
            st_t_launch += std::chrono::duration<double>(std::chrono::steady_clock::now() - tl0).count();
            st_tier_launch[t] += 1;
            st_tier_items[t] += cur.size();
            if (want_stats) std::fprintf(stderr, "[ddo stats] tier %d: launch of %zu sub-problems (fringe %zu open, best_lb %lld)\n", t, cur.size(),
                                         lazy->len(), (long long)lb);
            if (last) {
                flight.swap(cur);
                flight_tier = t;
                shared = true;
                rc = host_share();   // ... while the launch runs
                if (rc != DDO_OK) {
                    std::vector<std::vector<LazyItem>> none((size_t)T);
                    (void)settle_flight(none);
                    for (auto& l : none) drop_items(l);
                    (void)absorb_todo();
                    return fail(rc);
                }
                break;
            }
            if ((rc = settle(t, cur, lists[(size_t)t + 1])) != DDO_OK) return fail(rc);
        }
        if (!shared) rc = host_share();   // (an empty batch: nothing was launched)
        st_host_run += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_run0).count();
        return rc;
    }

    /// waits for the launch in flight (if any), runs what it handed up through the tiers above, and folds every finished
    /// result into the fringe
    int flush_lazy() {
        if (!lazy) return DDO_OK;
        auto t0 = std::chrono::steady_clock::now();
        const int T = (int)tiers.size();
        std::vector<std::vector<LazyItem>> lists((size_t)T);
        int rc = settle_flight(lists);
        st_t_wait_flush += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (int t = 0; t < T && rc == DDO_OK; ++t) {
            if (lists[(size_t)t].empty()) continue;
            std::vector<LazyItem> up;
            rc = run_tier_sync(t, lists[(size_t)t], frozen.empty() ? best_lb : (bench_mode ? frozen_lb : best_lb), up);
            if (rc == DDO_OK && !up.empty()) {
                if (t + 1 >= T) {
                    drop_items(up);
                    rc = DDO_ERR_INTERNAL;
                } else {
                    for (LazyItem& e : up) lists[(size_t)t + 1].push_back(e);
                }
            }
        }
        for (auto& l : lists) drop_items(l);
        st_host_run += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        auto t1 = std::chrono::steady_clock::now();
        int rc2 = absorb_todo();
        st_host_post += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        return rc != DDO_OK ? rc : rc2;
    }

    /// input records of a launch, written straight into the engine's pinned staging buffer.  A sub-problem whose state is a
    /// row of a pool block names (block, row); only the problem root carries its state inline (the 128 state bytes of the
    /// other records are never read by the device and stay untouched).
    void fill_lazy_inputs(const std::vector<LazyItem>& its, int64_t lb, DDInput* out) {
        const int64_t lim = (int64_t)1 << 40;
        const int64_t blb = std::max(-lim, std::min(lim, lb));
        const bool fixed = cfg.width_policy == DDO_WIDTH_FIXED && !cfg.width_times && !cfg.width_div_by;
        for (size_t i = 0; i < its.size(); ++i) {
            DDInput& in = out[i];
            in.comp_type = CT_RESTRICTED;
            in.flags = IN_FUSED | IN_FILTER_CUTSET | IN_POOL_OUT;
            in.width = fixed ? (int32_t)cfg.width : (int32_t)ddo_width_heuristic(&cfg, (size_t)model->n, (size_t)std::max(its[i].depth, 0));
            in.value = (int32_t)its[i].value;
            in.depth = its[i].depth;
            in.pad = 0;
            in.best_lb = blb;
            in.src_off = its[i].block->off;
            in.src_row = (uint32_t)its[i].row;
            in.pad2 = 0;
            if (in.src_off == NO_POOL_SRC) {
                std::memset(in.state, 0, sizeof(in.state));
                model->initial_state(in.state);
            }
        }
    }

    /// Bench support: takes the next `nbatches` x nb_concurrent sub-problems off the fringe (MaxUB order, as step()
    /// would) and FREEZES them together with the incumbent and the pool head.  bench_step() then compiles batch
    /// (k mod nbatches) exactly as step() does -- same launch, same pipelining, same host work on the results -- but
    /// folds nothing into the fringe, so every step of a timed region is the same piece of work whatever --steps and
    /// --warmup are.  Returns the number of frozen batches.
    int bench_freeze(int nbatches, int stride) {
        if (!lazy) {
            set_error("ddo_solver_bench_freeze needs DDO_FRINGE_LAZY");
            return DDO_ERR_UNSUPPORTED;
        }
        int rc = flush_lazy();
        if (rc != DDO_OK) return rc;
        const int B = std::max(1, cfg.nb_concurrent);
        LazyItem it;
        uint64_t popped = 0;
        stride = std::max(1, stride);
        for (int b = 0; b < nbatches; ++b) {
            std::vector<LazyItem> batch;
            while ((int)batch.size() < B && lazy->pop(it, best_lb)) {   // pop() hands over one reference
                if (popped++ % (uint64_t)stride == 0) batch.push_back(it);   // every stride-th node in fringe order
                else dev_unref(it.block);
            }
            if (batch.empty()) break;
            if (batch.size() > 2)
                std::stable_sort(batch.begin(), batch.end(), [](const LazyItem& a, const LazyItem& b2) {
                    if (a.depth != b2.depth) return a.depth < b2.depth;
                    return (a.ub - a.value) > (b2.ub - b2.value);
                });
            frozen.push_back(std::move(batch));
        }
        frozen_lb = best_lb;
        if ((rc = engine->read_pool_head(&frozen_mark)) != DDO_OK) return rc;
        bench_mode = true;
        frozen_next = 0;
        return (int)frozen.size();
    }
    int bench_step() {
        if (!bench_mode || frozen.empty()) {
            set_error("ddo_solver_bench_step: call ddo_solver_bench_freeze first");
            return DDO_ERR_INVALID;
        }
        std::vector<LazyItem> batch = frozen[frozen_next++ % frozen.size()];
        for (LazyItem& e : batch) dev_ref(e.block);   // dispatch / absorb_lazy release one reference per item
        explored += batch.size();
        int rc = dispatch(batch, frozen_lb, true);
        return rc == DDO_OK ? 1 : rc;
    }

    /// step() with the lazy block fringe: payload in the device node pool, (value, ub) keys on the host.
    /// Software pipeline: the batch popped in this call is launched BEFORE the results of the previous one are
    /// turned into fringe blocks, so the host work overlaps the device (the sub-problems of a batch therefore do not
    /// see the cut-sets of the batch right before them -- the same staleness as nb_concurrent racing threads).
    int step_lazy() {
        if (!initialized) {
            initialized = true;
            t_start = std::chrono::steady_clock::now();
            engine->pool_reset();
            DevBlock* root = new DevBlock();
            root->rows = 1;
            root->value.push_back(0);
            root->ub.push_back(INT32_MAX);
            lazy->push_block(root, I64_MIN);
        }
        if (finished) return 0;
        if (aborted) return DDO_CUTOFF;
        if (lazy->empty() || (int)lazy->len() < std::max(1, cfg.nb_concurrent) / 2) {
            // nothing (or too little) to overlap with: take the results of the launch in flight first
            int rc = flush_lazy();
            if (rc != DDO_OK) return rc;
        }
        if (lazy->empty() && !pending()) {
            if (cfg.world_size <= 1) best_ub = best_lb;
            finished = true;
            return 0;
        }
        if (budget_exhausted()) {
            flush_lazy();
            aborted = true;
            best_ub = std::max(best_lb, lazy->best_ub());   // abort_search (parallel.rs:479-489): the best open bound
            lazy->clear();
            return DDO_CUTOFF;
        }
        auto t_pop0 = std::chrono::steady_clock::now();
        litems.clear();
        const int B = std::max(1, cfg.nb_concurrent);
        LazyItem it;
        while ((int)litems.size() < B && lazy->pop(it, best_lb)) {
            litems.push_back(it);
            explored += 1;
        }
        if (litems.empty()) {
            int rc = flush_lazy();
            if (rc != DDO_OK) return rc;
            if (lazy->empty()) {
                if (cfg.world_size <= 1) best_ub = best_lb;
                finished = true;
                return 0;
            }
            return 1;
        }
        if (cfg.world_size <= 1) {
            // the batch still in flight was popped earlier, with bounds at least as large, and is not absorbed yet: the
            // open bound is the largest of both (only then does gap() never dip below the true bound mid-search)
            int64_t top = litems[0].ub;
            for (const LazyItem& e : flight) top = std::max(top, e.ub);
            for (const auto& pr : todo)
                for (const LazyItem& e : pr.first) top = std::max(top, e.ub);
            best_ub = top == INT32_MAX ? I64_MAX : top;
        }
        // longest-processing-time-first: shallow sub-problems with a lot of slack are the big DDs
        if (litems.size() > 2)
            std::stable_sort(litems.begin(), litems.end(), [](const LazyItem& a, const LazyItem& b) {
                if (a.depth != b.depth) return a.depth < b.depth;
                return (a.ub - a.value) > (b.ub - b.value);
            });
        st_host_pop += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pop0).count();
        int rc = dispatch(litems, best_lb, false);
        if (rc != DDO_OK) return rc;
        return 1;
    }

    /// One round of get_workload + process_one_node for up to nb_concurrent sub-problems.
    int step() {
        if (lazy) return step_lazy();
        initialize();
        if (finished) return 0;
        if (aborted) return DDO_CUTOFF;
        // ---- get_workload (parallel.rs:500-559)
        if (fringe->empty()) {
            if (cfg.world_size <= 1) best_ub = best_lb;   // :512-515
            finished = true;
            return 0;
        }
        if (budget_exhausted()) {                            // TimeBudget -> abort_search (:479-489)
            aborted = true;
            const Entry* top = fringe->peek();
            best_ub = top ? top->ub : best_lb;
            fringe->clear();
            return DDO_CUTOFF;
        }
        items.clear();
        const int B = std::max(1, cfg.nb_concurrent);
        Entry nn;
        while ((int)items.size() < B && fringe->pop(nn)) {
            if (nn.ub <= best_lb) {                          // :531-535 nothing relevant is left
                // SequentialSolver pops (and counts) every remaining node, each one skipped by
                // process_one_node (sequential.rs:452-456, 398-400): same outcome, different `explored`
                if (cfg.sequential) explored += 1 + fringe->len();
                block_unref(nn.block);
                fringe->clear();
                break;
            }
            items.push_back(nn);
            explored += 1;                                   // :553
        }
        if (items.empty()) {
            if (cfg.world_size <= 1) best_ub = best_lb;
            finished = true;
            return 0;
        }
        if (cfg.world_size <= 1) best_ub = items[0].ub;     // best-first: the first popped bounds the rest
        // longest-processing-time-first: the device work queue hands items out in order, so put the
        // sub-problems with the most remaining vertices (the biggest DDs) first to shorten the tail
        if (items.size() > 2)
            std::stable_sort(items.begin(), items.end(), [&](const Entry& a, const Entry& b) {
                return model->popcount(a.block->state(a.row)) > model->popcount(b.block->state(b.row));
            });
        // ---- process_one_node on the device (parallel.rs:391-437), restricted + relaxed fused
        inputs.resize(items.size());
        const int64_t lim = (int64_t)1 << 40;
        for (size_t i = 0; i < items.size(); ++i) {
            DDInput& in = inputs[i];
            std::memset(&in, 0, sizeof(in));
            in.comp_type = CT_RESTRICTED;
            in.flags = IN_FUSED | IN_FILTER_CUTSET | ((cfg.cutset_type == DDO_FRONTIER && !cfg.pooled) ? IN_FRONTIER : 0u);
            // DefaultCachingSolver: must_explore at the pop (sequential.rs:341 / parallel.rs:537-549; the parallel solver
            // also marks the node explored), thresholds and cache filter inside the compiles
            if (cache) in.flags |= IN_CACHE | IN_MUST_EXPLORE | (cfg.sequential ? 0u : IN_MARK_EXPLORED);
            if (dominance) in.flags |= IN_DOMINANCE;
            in.width = width_of(items[i]);
            in.value = (int32_t)items[i].value;
            in.depth = items[i].depth;
            in.best_lb = std::max(-lim, std::min(lim, best_lb));
            in.src_off = NO_POOL_SRC;
            std::memcpy(in.state, items[i].block->state(items[i].row), (size_t)model->ws * 8);
        }
        auto t_run0 = std::chrono::steady_clock::now();
        int rc = engine->run_batch(inputs.data(), (int)inputs.size(), results, cache, dominance);
        auto t_run1 = std::chrono::steady_clock::now();
        st_host_run += std::chrono::duration<double>(t_run1 - t_run0).count();
        if (rc != DDO_OK) {
            for (Entry& e : items) block_unref(e.block);
            return rc;
        }
        int err = DDO_OK;
        for (size_t i = 0; i < items.size() && err == DDO_OK; ++i) {
            if (results[2 * i].hdr.status == ST_ERR_CAPACITY || results[2 * i + 1].hdr.status == ST_ERR_CAPACITY ||
                results[2 * i].hdr.status <= -100 || results[2 * i + 1].hdr.status <= -100) {
                // the shared output arena overflowed: redo this sub-problem on its own (and if even one sub-problem's cut-set
                // -- a frontier cut-set holds nodes of every layer -- does not fit, the arena grows).  The failed run left no
                // thresholds in the cache (misp_dd_core.hpp), but it did answer Cache::must_explore and mark the node explored
                // at the pop: the second run must not ask again (it would be told "already explored" and dropped).
                std::vector<HostResult> solo;
                DDInput again = inputs[i];
                again.flags &= ~(IN_MUST_EXPLORE | IN_MARK_EXPLORED);
                int rc2 = engine->run_solo_growing(again, solo, cache, dominance);
                if (rc2 != DDO_OK || solo.size() < 2 || solo[0].hdr.status == ST_ERR_CAPACITY || solo[1].hdr.status == ST_ERR_CAPACITY ||
                    solo[0].hdr.status <= -100 || solo[1].hdr.status <= -100) {
                    const int st0 = solo.size() >= 2 ? std::min(solo[0].hdr.status, solo[1].hdr.status) : 0;
                    if (st0 == ST_ERR_LPOOL || st0 == ST_ERR_APOOL)
                        set_error("device compile failed: a decision diagram outgrew the per-slot pools of kept layers / arcs "
                                  "(DDO_HIP_LPOOL_M, DDO_HIP_APOOL_M: pool sizes in millions of records)");
                    else if (st0 == ST_ERR_ARENA || rc2 != DDO_OK)
                        set_error("device compile failed: output arena too small for one sub-problem");
                    else
                        set_error("device compile failed: a decision diagram outgrew its workspace (node slots / tables; site " + std::to_string(st0) + ")");
                    err = DDO_ERR_CAPACITY;
                    break;
                }
                results[2 * i] = std::move(solo[0]);
                results[2 * i + 1] = std::move(solo[1]);
            }
            if (results[2 * i].hdr.status == ST_SKIPPED) {   // Cache::must_explore said no: the node is dropped
                if (!cfg.sequential) explored -= 1;           // parallel.rs:553 counts what passes; sequential.rs:456 every pop
                continue;
            }
            for (int k = 0; k < 2; ++k) {
                HostResult* r = &results[2 * i + k];
                if (r->hdr.status == ST_NOT_RUN) continue;
                if (r->hdr.status != ST_OK) {
                    set_error("device compile failed with status " + std::to_string(r->hdr.status));
                    err = r->hdr.status == ST_CUTOFF ? DDO_CUTOFF : DDO_ERR_INTERNAL;
                    break;
                }
                counters.nodes_expanded += r->hdr.nodes_expanded;
                counters.arcs += r->hdr.arcs;
                counters.layers += r->hdr.layers;
                counters.compiles += 1;
                if (want_stats) {
                    st_layers.push_back((uint32_t)r->hdr.layers);
                    st_maxw.push_back(r->hdr.max_width_seen);
                    st_nodes.push_back(r->hdr.nodes_expanded);
                    for (int q = 0; q < 32; ++q) st_clk[q] += r->hdr.phase_clk[q];
                    st_recycled += r->hdr.recycled_merges;
                }
                maybe_update_best(items[i], *r);
                const bool exact = r->hdr.is_exact || r->hdr.has_exact_best_path;
                if (k == 1 && !exact) enqueue_cutset(items[i], *r);
            }
        }
        for (Entry& e : items) block_unref(e.block);
        st_host_post += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_run1).count();
        if (err != DDO_OK) return err;
        return 1;
    }
};

extern "C" {

size_t ddo_width_heuristic(const ddo_solver_config* cfg, size_t nb_vars, size_t depth) {
    if (!cfg) return 0;
    size_t w;
    // the inner policy's RAW value (it may be 0: NbUnassignedWidth at depth == nb_vars, TsptwWidth with factor 0); the decorators
    // apply their own max(1, .) like the reference's, and the engine never gets less than 1
    if (cfg->width_policy == DDO_WIDTH_FIXED) w = cfg->width;                                             // width.rs:168-170
    else if (cfg->width_policy == DDO_WIDTH_TSPTW) w = nb_vars * (depth + 1) * (size_t)cfg->width;        // tsptw/heuristics.rs:48-52
    else w = nb_vars > depth ? nb_vars - depth : 0;                                                        // width.rs:399-401
    if (cfg->width_times > 0) w = std::max<size_t>(1, cfg->width_times * w);                               // Times, width.rs:638-641: 1.max(k * inner)
    if (cfg->width_div_by > 0) w = std::max<size_t>(1, w / cfg->width_div_by);                             // DivBy, width.rs:877-880: 1.max(inner / k)
    return std::max<size_t>(1, w);
}

ddo_solver* ddo_solver_create(const ddo_model* model, const ddo_solver_config* cfg) {
    if (!model || !cfg) {
        set_error("ddo_solver_create: null argument");
        return nullptr;
    }
    if (cfg->width_policy == DDO_WIDTH_FIXED && cfg->width < 1) {
        set_error("ddo_solver_create: FixedWidth must be >= 1");
        return nullptr;
    }
    ddo_solver* s = new ddo_solver();
    s->model = const_cast<Model*>(&model->m);
    s->cfg = *cfg;
    if (s->cfg.world_size < 1) s->cfg.world_size = 1;
    if (s->cfg.rank < 0 || s->cfg.rank >= s->cfg.world_size) s->cfg.rank = 0;
    if (s->cfg.cutset_type == 0) s->cfg.cutset_type = DDO_LAST_EXACT_LAYER;
    if (s->cfg.pooled) {
        if (s->model->kind != MODEL_MISP || cfg->fringe != DDO_FRINGE_NODUP || s->cfg.dominance_entries > 0) {
            set_error("ddo_solver_create: pooled = 1 (the four *SolverPooled aliases, solver/mod.rs:34, :38, :43, :47) needs a MISP model, DDO_FRINGE_NODUP "
                      "and no dominance checker");
            delete s;
            return nullptr;
        }
        s->cfg.cutset_type = DDO_FRONTIER;   // the only cut-set a Pooled DD has (pooled.rs:543-566); computed by the pooled kernel itself
    }
    const bool keep_layers = !s->cfg.pooled && (s->cfg.cutset_type == DDO_FRONTIER || s->cfg.cache_entries > 0 || s->cfg.dominance_entries > 0);
    if ((s->cfg.cutset_type != DDO_LAST_EXACT_LAYER && s->cfg.cutset_type != DDO_FRONTIER) || (keep_layers && cfg->fringe == DDO_FRINGE_LAZY)) {
        set_error("ddo_solver_create: cutset_type must be LAST_EXACT_LAYER or FRONTIER; a frontier cut-set or a cache need DDO_FRINGE_NODUP");
        delete s;
        return nullptr;
    }
    // the lazy fringe keeps its nodes in the engine's device pool and leaves launches in flight: it owns its engine
    s->engine = cfg->fringe == DDO_FRINGE_LAZY ? Engine::create_private(s->model, cfg->device, s->engine_width())
                                               : Engine::get(s->model, cfg->device, s->engine_width(),
                                                             s->cfg.pooled ? Engine::ENGINE_POOLED : keep_layers ? Engine::ENGINE_KEEP_LAYERS : 0);
    if (!s->engine) {
        delete s;
        return nullptr;
    }
    if (s->cfg.cache_entries > 0 && !(s->cache = CacheTable::create(s->model, cfg->device, s->cfg.cache_entries))) {
        delete s;
        return nullptr;
    }
    if (s->cfg.dominance_entries > 0 && !(s->dominance = DominanceTable::create(s->model, cfg->device, s->cfg.dominance_entries))) {
        delete s;
        return nullptr;
    }
    s->fringe = new NoDupFringe(s->model);
    if (s->cfg.fringe == DDO_FRINGE_LAZY) {
        if (s->engine->engine_kind() != 2 || !s->engine->has_pool()) {
            set_error("DDO_FRINGE_LAZY needs the in-place device engine and its node pool");
            delete s;
            return nullptr;
        }
        s->lazy = new LazyFringe();
        // capacity tiers below the full-width engine (see ddo_solver::dispatch): only where the width leaves room for them
        std::vector<std::pair<int, int>> spec;   // (layer capacity, threads per workgroup)
        if (s->cfg.width_policy == DDO_WIDTH_FIXED && s->cfg.width >= 4096) {
            spec.push_back({256, 64});
            if (s->cfg.width >= 4096) spec.push_back({1024, 128});
            // (a wide capacity tier, "4096:256" in DDO_HIP_TIERS -- four 256-thread workgroups per CU, Engine::init: mid_ -- takes
            // 6 M of the brock400_1 search's DDs off the dense tier and needs as long for them: 12.4 + 36.6 s against 49.4 s.  What
            // limits those DDs is the memory system all CUs share, not the latency one CU can hide: not in the default list.)
        }
        if (const char* env = std::getenv("DDO_HIP_TIERS")) {   // "0" = none, "256:64,2048:256" = explicit list
            spec.clear();
            std::string v(env);
            size_t pos = 0;
            while (pos < v.size() && v != "0") {
                size_t end = v.find(',', pos);
                if (end == std::string::npos) end = v.size();
                const std::string tok = v.substr(pos, end - pos);
                const size_t colon = tok.find(':');
                const int w = std::atoi(tok.c_str());
                const int th = colon == std::string::npos ? (w <= 512 ? 64 : (w <= 1024 ? 128 : 256)) : std::atoi(tok.c_str() + colon + 1);
                if (w >= 8 && s->cfg.width_policy == DDO_WIDTH_FIXED && 2 * (size_t)w <= s->cfg.width && spec.size() < 3) spec.push_back({w, th});
                pos = end + 1;
            }
        }
        for (auto& sp : spec) {
            auto t = Engine::create_tier(s->model, cfg->device, s->engine.get(), sp.first, sp.second);
            if (!t) {
                delete s;
                return nullptr;
            }
            s->tiers.push_back(t);
        }
        // the dense tier: full layer capacity, two decision diagrams per CU (DDO_HIP_DENSE=0 switches it off); it exists only
        // where its LDS footprint fits twice -- not a failure otherwise
        const char* denv = std::getenv("DDO_HIP_DENSE");
        if (s->cfg.width_policy == DDO_WIDTH_FIXED && s->engine->engine_kind() == 2 && (denv ? std::atoi(denv) != 0 : s->cfg.width >= 2048)) {
            auto t = Engine::create_tier(s->model, cfg->device, s->engine.get(), (int)s->cfg.width, 512);
            if (t) s->tiers.push_back(t);
        }
    }
    s->tiers.push_back(s->engine);
    // every engine of this solver has its workspace now: the node pool starts to map what the first large launches will reserve,
    // in the background, while the search compiles its first (small) steps
    if (s->lazy) s->engine->pool_expect(std::max(1, s->cfg.nb_concurrent));
    s->want_stats = std::getenv("DDO_HIP_STATS") != nullptr;
    if (const char* env = std::getenv("DDO_HIP_TIER_SKIP")) s->tier_skip_pct = std::max(1, std::min(100, std::atoi(env)));
    if (const char* env = std::getenv("DDO_HIP_HINT_SLACK")) s->hint_by_slack = std::atoi(env) != 0;   // 0: per-depth cells only (A/B)
    return s;
}
void ddo_solver_destroy(ddo_solver* s) { delete s; }

int ddo_solver_step(ddo_solver* s) {
    if (!s) return DDO_ERR_INVALID;
    return s->step();
}
int ddo_solver_bench_freeze(ddo_solver* s, int nbatches, int stride) {
    if (!s || nbatches < 1 || stride < 1) return DDO_ERR_INVALID;
    return s->bench_freeze(nbatches, stride);
}
uint64_t ddo_solver_bench_frozen(const ddo_solver* s) {
    uint64_t n = 0;
    if (s)
        for (const auto& b : s->frozen) n += b.size();
    return n;
}
int ddo_solver_bench_step(ddo_solver* s) {
    if (!s) return DDO_ERR_INVALID;
    return s->bench_step();
}
int ddo_solver_export_subproblems(ddo_solver* s, size_t max_count, uint64_t* states, int64_t* value, int64_t* ub, int64_t* depth,
                                   uint64_t* path_off, ddo_decision* paths, size_t path_cap, size_t* count) {
    if (!s || !states || !value || !ub || !depth || !path_off || !paths || !count) return DDO_ERR_INVALID;
    return s->export_nodes(max_count, states, value, ub, depth, path_off, paths, path_cap, count);
}
int ddo_solver_import_subproblems(ddo_solver* s, size_t count, const uint64_t* states, const int64_t* value, const int64_t* ub,
                                   const int64_t* depth, const uint64_t* path_off, const ddo_decision* paths) {
    if (!s || (count && (!states || !value || !ub || !depth || !path_off || !paths))) return DDO_ERR_INVALID;
    return s->import_nodes(count, states, value, ub, depth, path_off, paths);
}
int ddo_solver_flush(ddo_solver* s) {
    if (!s) return DDO_ERR_INVALID;
    return s->flush_lazy();
}

int ddo_solver_maximize(ddo_solver* s, ddo_completion* out) {
    if (!s) return DDO_ERR_INVALID;
    int rc;
    while ((rc = s->step()) == 1) {
    }
    if (rc < 0) return rc;
    s->flush_lazy();
    if (out) {   // parallel.rs:604-606
        out->is_exact = s->aborted ? 0 : 1;
        out->has_best_value = s->has_sol ? 1 : 0;
        out->best_value = s->has_sol ? s->best_lb : 0;
    }
    return rc == DDO_CUTOFF ? DDO_CUTOFF : DDO_OK;
}
int ddo_solver_best_value(const ddo_solver* s, int64_t* value) {
    if (!s || !s->has_sol) return 0;
    if (value) *value = s->best_lb;
    return 1;
}
int ddo_solver_best_solution(const ddo_solver* s, ddo_decision* buf, size_t* len) {
    if (!s || !s->has_sol) return 0;
    if (!len) return DDO_ERR_INVALID;
    if (!buf || *len < s->best_sol.size()) {
        *len = s->best_sol.size();
        return DDO_ERR_CAPACITY;
    }
    std::vector<ddo_decision> sol = s->best_sol;   // parallel.rs:605: sorted by variable
    std::sort(sol.begin(), sol.end(), [](const ddo_decision& a, const ddo_decision& b) { return a.variable < b.variable; });
    std::memcpy(buf, sol.data(), sol.size() * sizeof(ddo_decision));
    *len = sol.size();
    return 1;
}
int64_t ddo_solver_best_lower_bound(const ddo_solver* s) { return s->best_lb; }
int64_t ddo_solver_best_upper_bound(const ddo_solver* s) { return s->best_ub; }
int ddo_solver_set_primal(ddo_solver* s, int64_t value, const ddo_decision* solution, size_t len) {
    if (!s) return DDO_ERR_INVALID;
    if (value > s->best_lb) {   // parallel.rs:630-636
        s->best_lb = value;
        s->best_sol.assign(solution, solution + len);
        s->has_sol = true;
    }
    return DDO_OK;
}
double ddo_solver_gap(const ddo_solver* s) {   // solver.rs:80-93
    int64_t ub = s->best_ub, lb = s->best_lb;
    if (ub == I64_MAX || lb == I64_MIN) return 1.0;
    double aub = std::abs((double)ub), alb = std::abs((double)lb);
    double u = std::max(aub, alb), l = std::min(aub, alb);
    if (u == 0.0) return 0.0;
    return (u - l) / u;
}
uint64_t ddo_solver_explored(const ddo_solver* s) { return s->explored; }
int ddo_solver_counters(const ddo_solver* s, ddo_counters* out) {
    if (!s || !out) return DDO_ERR_INVALID;
    *out = s->counters;
    return DDO_OK;
}
int ddo_solver_import_lower_bound(ddo_solver* s, int64_t best_lb) {
    if (!s) return DDO_ERR_INVALID;
    // the value comes from another rank's incumbent: the bound is valid here, its solution stays there
    if (best_lb > s->best_lb) {
        s->best_lb = best_lb;
        s->has_sol = false;
        s->best_sol.clear();
    }
    return DDO_OK;
}
uint64_t ddo_solver_fringe_len(const ddo_solver* s) {
    if (!s) return 0;
    if (!s->initialized) return 1;
    if (s->lazy) return s->lazy->len();
    return s->fringe->len();
}
int64_t ddo_solver_fringe_best_ub(const ddo_solver* s) {
    if (s->lazy) return s->lazy->best_ub();
    const Entry* top = s->fringe->peek();
    return top ? top->ub : I64_MIN;
}
/// One EPOCH of a sharded search in one call (ddo_amd/distributed.py: DistributedSearch.maximize): takes the reduced vector of the
/// previous epoch's all-reduce (`in`, may be NULL: [0] = the global incumbent), runs search steps until `min_ms` milliseconds have
/// passed (at least one step, at most `max_steps`; a rank whose fringe is empty does none), and fills `out` with this rank's
/// contribution to the next MAX all-reduce: [incumbent, 1 if work remains, 1 if cut off, open nodes, -open nodes, best open bound,
/// -best open bound] (the last two INT64_MIN / 2 without open nodes).  Returns what the last step returned.
int ddo_solver_epoch(ddo_solver* s, const int64_t* in, int64_t* out, int max_steps, double min_ms) {
    if (!s || !out) return DDO_ERR_INVALID;
    const int64_t LOW = -(1LL << 62);
    if (in && in[0] > LOW) ddo_solver_import_lower_bound(s, in[0]);
    const auto t0 = std::chrono::steady_clock::now();
    int rc = 0;
    for (int k = 0; k < std::max(1, max_steps); ++k) {
        if (k > 0 && std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() >= min_ms) break;
        rc = s->step();
        if (rc != 1) break;
    }
    if (rc < 0) return rc;
    const int64_t open = (int64_t)ddo_solver_fringe_len(s);
    const int64_t top = open > 0 ? ddo_solver_fringe_best_ub(s) : LOW;
    out[0] = std::max<int64_t>(s->best_lb, LOW);
    out[1] = rc == 1 ? 1 : 0;
    out[2] = (rc == DDO_CUTOFF || s->aborted) ? 1 : 0;
    out[3] = open;
    out[4] = -open;
    out[5] = std::max<int64_t>(top, LOW);
    out[6] = open > 0 ? -top : LOW;
    return rc;
}
int ddo_solver_tier_count(const ddo_solver* s) { return s ? (int)s->tiers.size() : 0; }
int ddo_solver_tier_stats(const ddo_solver* s, int t, ddo_tier_stats* out) {
    if (!s || !out || t < 0 || t >= (int)s->tiers.size()) return DDO_ERR_INVALID;
    const auto& e = s->tiers[(size_t)t];
    std::memset(out, 0, sizeof(*out));
    out->kernel_ms = e->kernel_ms();
    out->launches = e->launches();
    out->subproblems = s->st_tier_items[t];
    out->retried = s->st_tier_retry[t];
    if (t + 1 < (int)s->tiers.size()) out->nodes_expanded = s->st_tier_nodes[t];
    else {   // the last tier's results are folded in by the pipeline: everything the lower tiers did not finish
        uint64_t lower = 0;
        for (int k = 0; k + 1 < (int)s->tiers.size(); ++k) lower += s->st_tier_nodes[k];
        out->nodes_expanded = s->counters.nodes_expanded > lower ? s->counters.nodes_expanded - lower : 0;
    }
    out->layer_capacity = e->is_tier() ? e->cap_width() : (int32_t)e->max_width();
    out->threads = e->threads();
    out->slots = e->nslots();
    out->dense = e->is_dense() ? 1 : 0;
    out->lds_bytes = e->lds_bytes();
    return DDO_OK;
}

int ddo_solver_device_time(const ddo_solver* s, double* kernel_ms, uint64_t* launches) {
    if (!s) return DDO_ERR_INVALID;
    double ms = 0;
    uint64_t n = 0;
    for (const auto& t : s->tiers) {
        ms += t->kernel_ms();
        n += t->launches();
    }
    if (kernel_ms) *kernel_ms = ms;
    if (launches) *launches = n;
    return DDO_OK;
}

}  // extern "C"
