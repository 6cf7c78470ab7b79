// =============================================================================
// dd_types.h -- POD records shared by the device kernel and the host side of
// the engine (device<->host wire format of one compile).
// =============================================================================
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) && !defined(DDO_HOST_EMULATION)
#define DD_HD __host__ __device__
#else
#define DD_HD
#endif

namespace ddo_hip {

constexpr int MAX_WS = 72;               // 64-bit words per state at the wire (MISP uses up to 16: n <= 1024; signed-vector models n <= 142: 71 pair words + depth)
constexpr int MAX_VEC_VARS = 2 * (MAX_WS - 1);   // variables of a signed-vector state (two 32-bit benefits per word)
constexpr uint32_t NONE32 = 0xFFFFFFFFu;

// comp_type values follow include/ddo_hip.h (== mdd.rs:41-48 order)
constexpr int CT_EXACT = 0, CT_RELAXED = 1, CT_RESTRICTED = 2;

// EngineParams.model_kind
constexpr int MODEL_MISP = 0, MODEL_KNAPSACK = 1, MODEL_MCP = 2, MODEL_MAX2SAT = 3, MODEL_TSPTW = 4;

// DDInput.flags
constexpr uint32_t IN_FUSED = 1u;          // restricted, then (if inexact) relaxed: parallel.rs:391-437 on device
constexpr uint32_t IN_FILTER_CUTSET = 2u;  // emit only cut-set nodes with ub > best_lb (parallel.rs:461)
constexpr uint32_t IN_WANT_PATHS = 4u;     // always emit best paths (else only when value > best_lb)
constexpr uint32_t IN_POOL_OUT = 8u;       // keep the cut-set in the device node pool; ship only (ub, value) per node
constexpr uint32_t IN_FRONTIER = 16u;      // CUTSET_TYPE == FRONTIER (clean.rs:586-606) instead of the last exact layer
constexpr uint32_t IN_CACHE = 32u;         // SimpleCache behind the compile: _filter_with_cache, thresholds, cache updates
constexpr uint32_t IN_MUST_EXPLORE = 64u;  // solver pop: Cache::must_explore first (sequential.rs:341, parallel.rs:537); ST_SKIPPED when it says no
constexpr uint32_t IN_DOMINANCE = 256u;     // SimpleDominanceChecker behind the compile: _filter_with_dominance (clean.rs:689-708)
constexpr uint32_t IN_PATH_BITS = 512u;     // in-place engine, cut-set into the arena: paths as BIT rows (one decision bit per layer, ceil(lel / 64) words per node)
                                             // plus the branching variables once per DD (DDResult::cs_lvar_off) instead of one u32 per decision and node
constexpr uint32_t IN_MARK_EXPLORED = 128u; // ... and update_threshold(state, depth, value, true) when it says yes (parallel.rs:538 only)
constexpr uint64_t NO_POOL_SRC = ~0ULL;    // DDInput.src_off: the residual state is inline (not a pool row)

/// A cut-set block in the device node pool (all offsets in bytes from the block start, 8-byte aligned):
///   header 64 B | lvar u32[lel] | states u64[ws][rows] | path bits u64[pw][rows] | value i32[rows] | ub i32[rows]
struct PoolBlockHeader {
    uint32_t rows, ws, lel, depth;     // depth of every row (common.rs:86)
    uint64_t parent_off;               // block holding the sub-problem this DD was compiled from (NO_POOL_SRC: root)
    uint32_t parent_row, pw;           // pw = words of path bits per row = ceil(lel / 64)
    uint64_t off_lvar, off_states, off_paths, off_values, off_ubs;
};
DD_HD inline uint64_t pool_block_bytes(uint32_t rows, uint32_t ws, uint32_t lel) {
    const uint64_t pw = (lel + 63) / 64;
    uint64_t b = 64;
    b += ((uint64_t)lel * 4 + 7) & ~7ULL;
    b += (uint64_t)ws * rows * 8;
    b += pw * rows * 8;
    b += ((uint64_t)rows * 4 + 7) & ~7ULL;
    b += ((uint64_t)rows * 4 + 7) & ~7ULL;
    return b;
}

/// One sub-problem to compile (mdd.rs:51-71 CompilationInput, minus the trait objects).
struct DDInput {
    int32_t comp_type;
    uint32_t flags;
    int32_t width;
    int32_t value;     // residual.value
    int32_t depth;     // residual.depth
    int32_t pad;
    int64_t best_lb;
    uint64_t src_off;  // NO_POOL_SRC, or the pool block whose row `src_row` is the residual state
    uint32_t src_row;
    uint32_t pad2;
    uint64_t state[MAX_WS];
};

// DDResult.status
constexpr int ST_OK = 0, ST_CUTOFF = 1, ST_ERR_CAPACITY = -3, ST_ERR_INTERNAL = -5, ST_NOT_RUN = 77;
constexpr int ST_SKIPPED = 79; // IN_MUST_EXPLORE: the cache says this sub-problem need not be explored (cache.rs:32-39)
// a DD of the layer-keeping mode outgrew its slot's pool of kept layers / of arcs (EngineParams::lpool_nodes / apool_arcs): a capacity
// error that enlarging the output arena does not cure (DDO_HIP_LPOOL_M / DDO_HIP_APOOL_M do)
constexpr int ST_ERR_LPOOL = ST_ERR_CAPACITY - 2100, ST_ERR_APOOL = ST_ERR_CAPACITY - 2200;
// The SHARED OUTPUT ARENA of a launch was full: the one capacity error a re-run with a larger arena (or fewer compiles per launch)
// cures.  Every other status <= -100 names a per-slot workspace (node slots, dedup table, work / cut lists, pools): running the
// compile again gives the same answer, so it goes to the caller at once.
constexpr int ST_ERR_ARENA = ST_ERR_CAPACITY - 700;
constexpr int ST_RETRY = 78;   // capacity tier: the DD outgrew this tier's node slots (host: compile it on the next tier)

/// Everything observable about one compiled DD (clean.rs:237-266).  Variable
/// sized data lives in the output arena at `arena_off` (8-byte units):
///   best path        : best_len     x u32   (variable << 1 | value), terminal first
///   best exact path  : exact_len    x u32
///   cut-set states   : n_cutset x ws x u64  (row major)
///   cut-set value    : n_cutset x i32
///   cut-set ub       : n_cutset x i32
///   cut-set paths    : n_cutset x lel x u32 (node first, towards the root)
struct DDResult {
    int32_t status;
    int32_t comp_type;
    int32_t is_exact;              // lel.is_none()                      (clean.rs:635)
    int32_t has_exact_best_path;   // EBPO                               (clean.rs:636)
    int32_t has_best;              // best_node.is_some()
    int32_t has_best_exact;
    int32_t best_value;
    int32_t best_exact_value;
    int32_t n_layers;              // layers.len()
    int32_t lel;                   // index of the last exact layer, -1 when none
    int32_t n_cutset;
    int32_t best_len;
    int32_t exact_len;
    int32_t exact_same_as_best;    // best exact path == best path (not stored twice)
    uint32_t recycled_merges;      // how many times clean.rs:830 found a recycled node
    uint32_t max_width_seen;       // widest layer that was expanded
    uint64_t arena_off;            // byte offset of this DD's block in the arena
    uint64_t arena_bytes;
    uint64_t nodes_expanded;
    uint64_t arcs;
    uint64_t layers;
    uint64_t path_off, exact_off, cs_state_off, cs_value_off, cs_ub_off, cs_path_off;  // byte offsets from arena_off
    uint64_t pool_off;             // IN_POOL_OUT: byte offset of the cut-set block in the node pool
    uint64_t cs_depth_off;         // frontier cut-set: n_cutset x i32, layer of every node below the DD's root (0: all at `lel`)
    int32_t cs_path_stride;        // u32 words per row of the cut-set paths (lel, or n_layers - 1 for a frontier cut-set)
    uint32_t cache_hits;           // nodes removed by _filter_with_cache
    uint64_t cs_lvar_off;          // 0: cut-set paths are u32 rows (above).  Else (IN_PATH_BITS) the rows at cs_path_off are bit strings --
                                   // (cs_path_stride + 63) / 64 u64 words per node, bit tr = decision of transition tr -- and the variables
                                   // branched on, u32[cs_path_stride], lie at this offset
    // LAST: only downloaded when DDO_HIP_STATS asks for the clocks (the records of a launch cross PCIe: 176 instead of 432 bytes)
    uint64_t phase_clk[32];        // shader-clock ticks per phase [0..8), per code mark [8..24), thread-0 probes inside expand [24..32) (profiling aid; engine 2)
};

/// Kernel arguments: model tables, capacities, per-slot workspace and batch I/O.
struct EngineParams {
    // ---- model (MISP: examples/misp/main.rs:37-51)
    int32_t n;                 // nb_vars
    int32_t ws;                // words per state
    int32_t unit_weights;      // all weights == 1 -> rub = popcount (main.rs:191-193)
    int32_t npad;              // n rounded up to 64
    const uint64_t* adj;       // [n][ws] complement-adjacency rows
    const int32_t* weight;     // [n]
    // ---- capacities
    int32_t capN;              // max nodes per layer (width + 2)
    int32_t capC1;             // 2*capN + 1 candidate slots (last one = merged node)
    int32_t max_layers;        // n + 2
    int32_t table_cap;         // hash table slots available (power of two)
    int32_t table_in_lds;      // 1: table lives in LDS, 0: in HBM (gtable)
    int32_t nslots;
    // ---- per-slot workspace (slot = workgroup)
    uint64_t* cstate;          // [slot][2][ws][capC1]   SoA candidate states (word major)
    uint64_t* ckey;            // [slot][2][capC1]       (biased value << 32) | best candidate
    uint32_t* cpop;            // [slot][2][capC1]       popcount of the state
    uint32_t* cflags;          // [slot][2][capC1]       NF_* bits
    uint32_t* ctarget;         // [slot][2*capN]         arc -> winner candidate (dedup result)
    uint32_t* keep;            // [slot][capN]           node position -> candidate index
    uint32_t* posmap;          // [slot][capC1]          candidate -> node position
    uint8_t* cls;              // [slot][capC1]          0 none / 1 kept / 2 deleted
    uint32_t* ninfo;           // [slot][max_layers][capN]   best arc + flags per node per layer
    uint32_t* arct;            // [slot][max_layers][2*capN] arc -> child position (relaxed, below LEL)
    int32_t* arcc;             // [slot][max_layers][2*capN] arc cost (as relaxed when the arc was redirected to a merged node)
    int32_t* nlayer;           // [slot][max_layers]     nodes per layer
    // kept layers and arc arrays as per-slot POOLS (layer-keeping mode of the D-ary models, round 4): 0 = fixed strides
    uint64_t lpool_nodes;      // node records per slot shared by all kept layers of a DD (ninfo / lstate / lval / lrub / lvb / lth)
    uint64_t apool_arcs;       // arc records per slot (arct / arcc)
    uint64_t* lbase;           // [slot][max_layers + 1] first node record of every layer
    uint64_t* abase;           // [slot][max_layers + 1] first arc record of the arcs entering every layer
    int32_t* lvar;             // [slot][max_layers]     variable branched below each layer
    int32_t* ldup;             // [slot][max_layers][2]  recycled-merge duplicate (from pos, to pos)
    uint64_t* cs_state;        // [slot][ws][capN]       copy of the last exact layer
    int32_t* cs_value;         // [slot][capN]
    uint32_t* cs_pop;          // [slot][capN]
    uint32_t* gtable;          // [slot][table_cap] when !table_in_lds
    // ---- batch I/O
    const DDInput* inputs;
    DDResult* results;         // [nbatch][2]  (index 1 only used by IN_FUSED)
    int32_t nbatch;
    int32_t phase_clocks;      // 1: engine 2 accounts shader-clock ticks per phase (one extra barrier per phase)
    int32_t* work_counter;
    const uint32_t* order;     // in-place engine: the k-th draw of the work counter compiles inputs[order[k]] (longest first, ddo_hip_engine.hip:
                               // lpt_order_kernel); nullptr = input order
    int32_t* done;             // in-place engine, split launches (nullptr: not split): draw k < nbatch compiles the RESTRICTED decision diagram
                               // of sub-problem order[k] and raises done[order[k]], draw nbatch + k waits for that flag and compiles the
                               // RELAXED one -- twice as many work items of half the size: the launch's tail is half as long
    unsigned long long* arena_head;
    uint8_t* arena;
    uint64_t arena_cap;
    const int32_t* cutoff_flag; // device-visible flag, polled once per layer
    // ---- in-place engine (misp_dd_inplace.hpp): persistent node slots instead of per-layer candidate arrays
    int32_t capS;              // node slots per DD (2*width + 8)
    int32_t capW;              // work-list capacity (width + 4)
    int32_t tab2_cap;          // dedup table slots (power of two, LDS)
    int32_t vbase_off;         // lowest reachable value relative to the residual value (sum of negative weights)
    int32_t model_kind;        // MODEL_MISP | MODEL_KNAPSACK
    int32_t lists_in_lds;      // in-place engine, narrow capacity tiers: work / free lists live in LDS behind the shared block
    int32_t keys_global;       // in-place engine: 1 = ranking keys live in HBM packed with the hashes (s_hash holds key32 << 32 | h32), 0 = keys in LDS
    const int32_t* kp_weight;  // knapsack: item weights [n]   (`weight` holds the profits)
    const int32_t* kp_order;   // knapsack: items by decreasing profit / weight [n]
    // maximum cut (examples/mcp): states are n signed benefits packed two per word + a depth word
    const int32_t* vgraph;     // [n][n] edge weights
    const int32_t* vest;       // [n+1]  estimates (relax.rs:58-80)
    const int32_t* vnk;        // [n+1]  nk        (relax.rs:83-106)
    int32_t vr;                // initial value (sum of the negative edge weights)
    int32_t pad3;
    int32_t* lddelta;          // [slot][max_layers] relax delta of the node re-added by a recycled merge
    // MAX2SAT (examples/max2sat): clause weights per variable pair and polarity, unit clauses, branching order
    const int32_t* m2_wtt;     // [n][n] weight(t(k), t(l))
    const int32_t* m2_wtf;     // [n][n] weight(t(k), f(l))
    const int32_t* m2_wft;     // [n][n] weight(f(k), t(l))
    const int32_t* m2_wff;     // [n][n] weight(f(k), f(l))
    const int32_t* m2_order;   // [n]    vars_by_sum_of_clause_weights (model.rs:138-140)
    const int32_t* m2_rankpos; // [n]    position of each variable in that order
    int32_t lex_cap;           // tie lists up to this size (<= 1024) are split by rank counting in LDS, longer ones by radix rounds
    int32_t pad1;
    uint64_t* s_state;         // [slot][ws][capS]  node states, word major (streaming scan copy)
    uint64_t* s_rec;           // [slot][capS][RW]  node records: the state words, RW = 8*ceil((ws+1)/8) words (whole 64-byte lines)
    uint64_t* s_ptree;         // [slot][ev_cap/4]  path tree: per event record, parent path id | layer << 32 (misp_dd_inplace.hpp: PID_NONE)
    uint64_t* s_pvr;           // [slot][ev_cap/4]  pooled engine with a cache: value_top | rub << 32 of the node of every event record (nullptr otherwise)
    uint64_t* s_pst;           // [slot][pst_cap][ws]  ... and the state it had then (the cache update of an exact node needs it); nullptr otherwise
    uint32_t pst_cap;          // event records per slot whose states are kept (a decision diagram with more of them ends with a capacity error)
    uint64_t* s_hash;          // [slot][capS]      keys_global: key32 << 32 | h32 per node; else the h32 values as a u32 array (streamed by the
                               //                   select sweeps and the per-layer table rebuild)
    uint16_t* s_wl;            // [slot][2*capW]    work list + free list
    uint32_t* s_ev;            // [slot][ev_cap]    per-transition event records for the backward pass
    uint32_t* s_evoff;         // [slot][max_layers+1][4] offsets / counts per transition
    uint64_t ev_cap;
    uint32_t* s_cs_slot;       // [slot][capW]      node slots of the last exact layer (cut-set)
    uint64_t* s_cs_path;       // [slot][ws][capW]  their path bits
    // ---- device node pool (fringe payload stays in HBM)
    uint8_t* pool;
    uint64_t pool_cap;
    unsigned long long* pool_head;
    // ---- capacity tiers (engine.hpp): a tier engine has node slots for narrow decision diagrams only and never
    // squashes (its layer capacity is below every width it is asked for): a DD that outgrows it reports ST_RETRY and is
    // compiled again by the next tier.  hist_bins < 2048 shrinks the LDS area only the squash phases use.
    // TSPTW (examples/tsptw): distances [n][n], time windows, cheapest entering edge (dd_tsptw.hpp)
    const int32_t *tw_dist, *tw_early, *tw_late, *tw_cheap, *tw_order;   // tw_order: the nodes by increasing tw_cheap
    int32_t tw_lds;            // != 0: the workgroup keeps a copy of the five tables in LDS behind its shared block (n (n + 4) words: tw_lds_words)
    // TSPTW dominance (examples/tsptw/dominance.rs:26-60): best value per (depth, position, must_visit), same table layout as the cache
    uint64_t* dkey_tab;
    uint64_t dkey_cap;
    unsigned long long* dkey_stats;
    int32_t fan;               // children per node of the layer-rebuilding engine: 0 / 2 binary models, nb_nodes for TSPTW
    int32_t dbits;             // bits of a decision index in arc words and path words (1 for binary models)
    int32_t hist_bins;         // 0 = 2048
    int32_t tier;              // 0 = full-width engine, 1 = capacity tier
    // ---- frontier cut-set / thresholds / cache (engine 1, dd_thresholds.hpp): every layer of the DD is kept
    int32_t tmode;             // 1: the per-layer arrays below exist
    int32_t lstride;           // nodes per layer in ninfo / lstate / lval / ... (capN, or 2 * capN + 2 with tmode)
    uint64_t* lstate;          // [slot][max_layers][ws][lstride]
    int32_t* lval;             // [slot][max_layers][lstride]  value_top
    int32_t* lrub;             // [slot][max_layers][lstride]  rough upper bound (INT32_MAX: never computed)
    int32_t* lvb;              // [slot][max_layers][lstride]  value_bot (VB_UNMARKED: not marked)
    int32_t* lth;              // [slot][max_layers][lstride]  theta (TH_NONE / TH_INF / value); then [slot][capC1] scratch
    int32_t* lntot;            // [slot][max_layers] nodes per layer including the ones the cache pruned
    // SimpleCache (cache/simple.rs:36-73) as one open-addressing table in HBM shared by every compile of a solver / mdd:
    // entry = [tag | lock, packed threshold, depth, state words...]
    uint64_t* cache_tab;
    uint64_t cache_cap;        // entries (power of two); 0: no cache (EmptyCache)
    int32_t cache_stride;      // u64 words per entry
    int32_t pad4;
    unsigned long long* cache_stats;   // [0] entries in use, [1] insertions refused (table full)
    // SimpleDominanceChecker (dominance/simple.rs:37-117) for models whose dominance key is the depth and whose states have
    // ONE coordinate besides the value (knapsack: KPDominance, examples/knapsack/main.rs:198-218): per depth the set of
    // non-dominated (coordinate, value) pairs -- a Pareto front kept sorted by coordinate -- in HBM, one spin lock per depth
    uint64_t* dom_coord;       // [max_layers][dom_cap]
    int32_t* dom_value;        // [max_layers][dom_cap]
    uint32_t* dom_count;       // [max_layers]
    uint32_t* dom_lock;        // [max_layers]
    uint32_t dom_cap;          // entries per depth; 0: EmptyDominanceChecker
    uint32_t pad5;
    unsigned long long* dom_stats;     // [0] insertions dropped (front full)
};

// node flag bits (node_flags.rs:48-185 restricted to what the device needs)
constexpr uint32_t NF_INEXACT = 1u;   // !F_EXACT
constexpr uint32_t NF_RELAXED = 2u;   // F_RELAXED
constexpr uint32_t NF_CACHE = 8u;     // F_CACHE
constexpr uint32_t NF_DOM = 16u;      // dominated (dominance/simple.rs:67-111)
constexpr uint32_t NF_OKPATH = 4u;    // signed-vector models: the best arc comes from a node with an exact best path
// ninfo word: bits 0..24 best arc (parent position << 1 | decision), 25 dominated, 26 pruned by the cache, 27 frontier cut-set,
// 28 = no arc (root), 29 ok best path, 30/31 flags
constexpr uint32_t NI_ARC_MASK = 0x01FFFFFFu;
constexpr uint32_t NI_DOM = 0x02000000u;      // removed by _filter_with_dominance (kept in the layer with its threshold, never expanded)
constexpr uint32_t NI_CACHE = 0x04000000u;    // F_CACHE: pruned by _filter_with_cache (kept in the layer, never expanded)
constexpr uint32_t NI_CUTSET = 0x08000000u;   // F_CUTSET of a frontier cut-set
constexpr uint32_t NI_NOARC = 0x10000000u;
constexpr uint32_t NI_OKPATH = 0x20000000u;
constexpr uint32_t NI_INEXACT = 0x40000000u;
constexpr uint32_t NI_RELAXED = 0x80000000u;

}  // namespace ddo_hip
