/* =============================================================================
 * ddo_hip.h -- C ABI of the MI355X-native MDD compilation engine.
 *
 * This is the drop-in boundary for ddo's hot path.  Every entry point below
 * replaces one item of the reference's `DecisionDiagram` trait
 * (/root/reference/ddo/src/abstraction/mdd.rs:75-114) or of its `Solver` trait
 * (abstraction/solver.rs:32-97); a Rust shim `impl DecisionDiagram for HipMdd`
 * binds them 1:1 through `extern "C"`.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  All functions are
 * thread-compatible: one ddo_mdd is used by one host thread at a time (the
 * reference's `&mut self`, parallel.rs:580), distinct ddo_mdd objects may be
 * used concurrently.  Functions returning `int` return DDO_OK (0) or a
 * negative DDO_ERR_* code unless stated otherwise; ddo_last_error() gives text.
 *
 * The engine FAILS LOUDLY (DDO_ERR_NO_DEVICE) when no HIP device is present:
 * there is no CPU fallback behind this ABI.
 * ========================================================================== */
#ifndef DDO_HIP_H
#define DDO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------- */
#define DDO_OK 0
/** compile() was interrupted by the cutoff flag == Err(Reason::CutoffOccurred), common.rs:108-111; a solver whose
 *  TimeBudget ran out (parallel.rs:479-489).  Distinct from the 1 that ddo_solver_step returns while work remains. */
#define DDO_CUTOFF 2
/** ddo_mdd_compile on an mdd bound to a capacity tier (DDO_MDD_ENGINE_*): the decision diagram outgrew the tier's node slots
 *  (or, on a capacity tier, needed a squash); nothing was produced -- compile it again on an mdd of the full-width engine.
 *  This is what the lazy solver does between its tiers (host_solver.cpp: dispatch). */
#define DDO_HANDED_UP 3
#define DDO_ERR_NO_DEVICE (-1)   /* no HIP device / HIP runtime error            */
#define DDO_ERR_INVALID (-2)     /* bad argument                                 */
#define DDO_ERR_CAPACITY (-3)    /* width / layer / output exceeds the capacity  */
#define DDO_ERR_UNSUPPORTED (-4) /* feature not built into the device engine     */
#define DDO_ERR_INTERNAL (-5)    /* device-side invariant violated               */

/* ---- mdd.rs:41-48  enum CompilationType --------------------------------- */
#define DDO_EXACT 0
#define DDO_RELAXED 1
#define DDO_RESTRICTED 2

/* ---- mdd.rs:24-28  cut-set types ---------------------------------------- */
#define DDO_LAST_EXACT_LAYER 1
#define DDO_FRONTIER 2
/** OR into the cutset_type of ddo_mdd_create: compile() may be handed a ddo_cache (every layer of the DD is then kept on the
 *  device for _compute_thresholds, clean.rs:478-545; costs memory, not needed with the EmptyCache) */
#define DDO_MDD_CACHING 0x10
/** OR into the cutset_type of ddo_mdd_create: the mdd is a `Pooled` decision diagram (implementation/mdd/pooled.rs:117-823; the `D` of
 *  four *SolverPooled aliases, solver/mod.rs:34, :38, :43, :47): the layer of a variable holds the pool nodes it impacts (Problem::
 *  is_impacted_by, dp.rs:68-70), width and ranking apply to those, the cut-set is the frontier, a sub-problem's depth is the layer at
 *  which its node was expanded and its path holds one decision per expanded ancestor.  MISP models (the reference's only model that
 *  implements is_impacted_by, misp/main.rs:145-147).  With DDO_MDD_CACHING compile() may be handed a ddo_cache: the impacted nodes of
 *  every layer but the first are filtered by it (_filter_with_cache, pooled.rs:635, 662-680) and the thresholds of the finished
 *  decision diagram are written back over its long arcs (_compute_thresholds, _maybe_update_cache, :467-535).  A pool that outgrows
 *  its node slots (about 14 000 nodes) ends the compile with DDO_ERR_CAPACITY. */
#define DDO_MDD_POOLED 0x20
/** OR into the cutset_type of ddo_mdd_create (MISP, DDO_LAST_EXACT_LAYER, no DDO_MDD_CACHING): binds the mdd to ONE of the
 *  kernels the lazy solver spreads its sub-problems over, so that each of them can be driven -- and checked against the
 *  reference's results -- compile by compile.  0 = ddo_mdd_create picks: MISP at widths of 2048 and more compiles on the DENSE
 *  kernel and, for the rare decision diagram that outgrows its dedup table, again on the FULL one (the caller sees one compile);
 *  everything else runs on the one engine of its (model, device, width).  FULL = the full-width kernel only (one decision diagram
 *  per CU).  DENSE = misp_compile_kernel2_dense (full layer capacity, 512 threads, two decision diagrams per CU, the kernel
 *  the headline benchmark spends its time in); TIER0 / TIER1 = misp_compile_kernel2_tier with layers of at most 256 / 1024
 *  nodes (64 / 128 threads, 12 / 5 decision diagrams per CU): these never squash and answer DDO_HANDED_UP for a decision
 *  diagram that does not fit. */
#define DDO_MDD_ENGINE_DENSE 0x100
#define DDO_MDD_ENGINE_TIER0 0x200
#define DDO_MDD_ENGINE_TIER1 0x300
#define DDO_MDD_ENGINE_FULL 0x400
#define DDO_MDD_ENGINE_MASK 0x700

/** common.rs:58-61  struct Decision { variable: Variable, value: isize } */
typedef struct ddo_decision {
    int64_t variable;
    int64_t value;
} ddo_decision;

/** common.rs:75-87  struct SubProblem<T>.  `state` is the fixed-width state of
 *  the model: for MISP ceil(n/64) little-endian 64-bit words, bit i of word
 *  i/64 <=> vertex i still eligible (examples/misp/main.rs:37-51). */
typedef struct ddo_subproblem {
    const uint64_t* state;
    size_t state_words;
    int64_t value;
    int64_t ub;
    size_t depth;
    const ddo_decision* path; /* decisions from the problem root, may be NULL when path_len == 0 */
    size_t path_len;
} ddo_subproblem;

typedef struct ddo_cache ddo_cache;
typedef struct ddo_dominance ddo_dominance;

/** mdd.rs:51-71  struct CompilationInput.  problem / relaxation / ranking are
 *  fixed by the ddo_model the mdd was created from (the device cannot call
 *  `&dyn Problem`); dominance is the EmptyDominanceChecker.
 *  `cutoff` points to a host flag polled between launches (heuristics.rs:100-105
 *  Cutoff::must_stop); NULL == NoCutoff.
 *  `cache`: NULL == EmptyCache (cache/empty.rs), else a SimpleCache living in device memory (ddo_cache_create); the
 *  mdd must have been created with DDO_MDD_CACHING. */
typedef struct ddo_compile_input {
    int comp_type; /* DDO_EXACT | DDO_RELAXED | DDO_RESTRICTED */
    size_t max_width;
    int64_t best_lb;
    ddo_subproblem residual;
    const volatile int* cutoff;
    ddo_cache* cache;
    ddo_dominance* dominance; /* NULL == EmptyDominanceChecker (dominance/empty.rs); else see ddo_dominance_create */
} ddo_compile_input;

/** common.rs:115-121  struct Completion { is_exact, best_value: Option<isize> } */
typedef struct ddo_completion {
    int is_exact;
    int has_best_value;
    int64_t best_value;
} ddo_completion;

/** Counters the reference lacks.  nodes_expanded = iterations of
 *  the loop at clean.rs:360, arcs = `_branch_on` calls (clean.rs:367). */
typedef struct ddo_counters {
    uint64_t nodes_expanded;
    uint64_t arcs;
    uint64_t layers;
    uint64_t compiles;
} ddo_counters;

typedef struct ddo_model ddo_model;
typedef struct ddo_mdd ddo_mdd;
typedef struct ddo_solver ddo_solver;

const char* ddo_last_error(void);
/** Number of visible HIP devices (0 when none / runtime missing). */
int ddo_device_count(void);

/* ---- model descriptors: the closed set of `Problem + Relaxation + StateRanking` ----------- */
/** MISP (examples/misp/main.rs:37-209).  `compl_adj_rows`: n rows of ceil(n/64) words, row i =
 *  COMPLEMENT adjacency of vertex i (bit i itself may be set, main.rs:280-310); `weights`: n values. */
ddo_model* ddo_model_create_misp(int n, const uint64_t* compl_adj_rows, const int64_t* weights);
/** Reads a DIMACS-like .clq file exactly as examples/misp/main.rs:258-317 does. */
ddo_model* ddo_model_read_misp(const char* path);
/** Knapsack (examples/knapsack/main.rs:53-194: `Knapsack`, `KPRelax`, `KPRanking`).  The state is two words,
 *  `KnapsackState { capacity, depth }` (main.rs:44-50); variables are branched in decreasing
 *  profit/weight order (main.rs:66-70, 118-125); decision 1 = TAKE_IT, 0 = LEAVE_IT_OUT; merge = largest capacity;
 *  fast_upper_bound = fractional bound over the remaining items (main.rs:158-184). */
ddo_model* ddo_model_create_knapsack(int n, int64_t capacity, const int64_t* profit, const int64_t* weight);
/** Reads an instance exactly as examples/knapsack/main.rs:267-303 does ("n capacity", then n x "profit weight"). */
ddo_model* ddo_model_read_knapsack(const char* path);
/** Maximum cut (examples/mcp/{graph,model,relax}.rs: `Mcp`, `McpRelax`, `McpRanking`).  `adj_matrix`: n x n symmetric
 *  edge weights (graph.rs:30-46).  The state is `McpState { benef, depth }` (model.rs:27-31): n signed 32-bit benefits,
 *  two per word (variable v in half v & 1 of word v / 2), followed by one word holding the depth; n <= 142.  Variables in
 *  natural order (model.rs:88-96); decision +1 = side S, -1 = side T (model.rs:33-35). */
ddo_model* ddo_model_create_mcp(int n, const int64_t* adj_matrix);
/** Reads an instance exactly as examples/mcp/graph.rs:48-79 does ("c " comments, "<vertices> <edges>", "<src> <dst> <w>"). */
ddo_model* ddo_model_read_mcp(const char* path);
/** Weighted MAX2SAT (examples/max2sat/{data,model,relax,heuristics}.rs: `Max2Sat`, `Max2SatRelax`, `Max2SatRanking`).
 *  Clause k is (lit_a[k] OR lit_b[k]) with weight[k]; literals are +-(1 + variable); a unit clause has lit_a == lit_b; a clause
 *  listed twice keeps its last weight (data.rs:31-62, 96-110).  State as for max-cut: `State { depth, substates }`
 *  (model.rs:55-60) = n signed 32-bit benefits, two per word, then one depth word; n <= 142 (frb15-9-x: n = 135).  Variables are branched from
 *  the end of `vars_by_sum_of_clause_weights` (model.rs:138-140, 330-346); decision +1 = true, -1 = false. */
ddo_model* ddo_model_create_max2sat(int n, size_t nb_clauses, const int64_t* lit_a, const int64_t* lit_b, const int64_t* weight);
/** Reads a .wcnf file exactly as examples/max2sat/data.rs:67-116 does. */
ddo_model* ddo_model_read_max2sat(const char* path);
/** TSPTW (examples/tsptw/{instance,state,model,relax,heuristics,dominance}.rs: `Tsptw`, `TsptwRelax`, `TsptwRanking`).
 *  `distances`: nb_nodes x nb_nodes travel times, `earliest` / `latest`: one time window per node, all in the reference's
 *  fixed-point unit (1/10000, instance.rs:87-98); node 0 is the depot; 2 <= nb_nodes <= 256 (the reference's Set256).
 *  Variable k is the k-th move of the tour, its decision value the node visited (up to nb_nodes children per DD node; the last
 *  move returns to 0); the objective is MINUS the arrival time back at the depot (travel + waiting).  The state is
 *  `TsptwState { position, elapsed, must_visit, maybe_visit, depth }` (state.rs:34-69) in 3K + 2 words, K = words of one
 *  node set = 1 up to 64 nodes, 2 up to 128 nodes, 4 up to 256 nodes (ddo_model_state_words: 5 / 8 / 14):
 *    [0 .. K)   Position::Virtual(set) as a node bit mask (0 for Position::Node)
 *    [K .. 2K)  must_visit                    [2K .. 3K)  maybe_visit (0 when None)
 *    [3K]       elapsed: earliest | latest << 32 (ElapsedTime::FixedAt(d): both d)
 *    [3K + 1]   node of Position::Node | bit 16 position is virtual | bit 17 elapsed is fuzzy | bit 18 maybe_visit is Some | depth << 32 */
ddo_model* ddo_model_create_tsptw(int nb_nodes, const int64_t* distances, const int64_t* earliest, const int64_t* latest);
/** Reads an instance exactly as examples/tsptw/instance.rs:52-109 does (`(f32 * 10000.0) as usize` per number). */
ddo_model* ddo_model_read_tsptw(const char* path);
void ddo_model_destroy(ddo_model* model);
int ddo_model_nb_variables(const ddo_model* model);
int ddo_model_state_words(const ddo_model* model);
/** Problem::initial_state / initial_value (dp.rs:43-48) */
int ddo_model_initial_state(const ddo_model* model, uint64_t* state_out);
int64_t ddo_model_initial_value(const ddo_model* model);
/** StateRanking::compare (heuristics.rs:69-77) evaluated on the host: <0, 0, >0 */
int ddo_model_compare_states(const ddo_model* model, const uint64_t* a, const uint64_t* b);
/** Copies the descriptor back (n rows * words, n weights); buffers may be NULL. */
int ddo_model_export_misp(const ddo_model* model, uint64_t* compl_adj_rows, int64_t* weights);

/* ---- Cache (abstraction/cache.rs:27-57; implementation/cache/simple.rs:36-73) ------------------------------------ */
/** SimpleCache for the states of `model`, as one hash table of at least `capacity_entries` (depth, state) -> Threshold
 *  entries in the memory of `device`.  It is shared by every compile() that names it and by the solver that owns it;
 *  a full table drops new thresholds (sound: less pruning), see ddo_cache_stats. */
ddo_cache* ddo_cache_create(const ddo_model* model, int device, size_t capacity_entries);
void ddo_cache_destroy(ddo_cache* cache);
/** Cache::clear (cache.rs:56) */
int ddo_cache_clear(ddo_cache* cache);
/** entries in use / thresholds dropped because the table was full */
int ddo_cache_stats(const ddo_cache* cache, uint64_t* used, uint64_t* dropped);
/** Cache::get_threshold (cache.rs:44): 1 and (*value, *explored) when (state, depth) has a threshold, else 0.
 *  Cache::update_threshold (cache.rs:47): keeps the larger of the stored and the given threshold.
 *  Host-side views of the device table (tests, debugging): one small kernel launch each. */
int ddo_cache_get_threshold(const ddo_cache* cache, const uint64_t* state, size_t depth, int64_t* value, int* explored);
int ddo_cache_update_threshold(ddo_cache* cache, const uint64_t* state, size_t depth, int64_t value, int explored);

/* ---- DominanceChecker (abstraction/dominance.rs:100-124; implementation/dominance/simple.rs:37-117) ---------------- */
/** SimpleDominanceChecker in device memory for the dominance relation of `model`.  The device checker covers relations
 *  whose key is the depth and whose states have one coordinate besides the value (use_value = true): the knapsack's
 *  KPDominance (examples/knapsack/main.rs:198-218: more remaining capacity and more value dominate).  Per depth it keeps
 *  the non-dominated (capacity, value) pairs, at most `capacity_per_depth` of them (a full set drops new pairs: sound).
 *  For a TSPTW model it is TsptwDominance (examples/tsptw/dominance.rs:26-60: same position and must_visit set, larger
 *  value dominates): one hash table of `capacity_per_depth` (depth, position, must_visit) keys in all.
 *  Handed to compile() through ddo_compile_input.dominance (the mdd must have been created with DDO_MDD_CACHING: dominated
 *  nodes keep their threshold in the kept layers) or owned by a solver (ddo_solver_config.dominance). */
ddo_dominance* ddo_dominance_create(const ddo_model* model, int device, size_t capacity_per_depth);
void ddo_dominance_destroy(ddo_dominance* dominance);
int ddo_dominance_clear(ddo_dominance* dominance);

/* ---- DecisionDiagram (mdd.rs:75-114) -------------------------------------------------------- */
/** == `D::default()` bound to a model and a device.  `max_width` is the largest width any
 *  compile() on this object will ask for (sizes the HBM workspace). */
ddo_mdd* ddo_mdd_create(const ddo_model* model, int device, int cutset_type, size_t max_width);
void ddo_mdd_destroy(ddo_mdd* mdd);
/** mdd.rs:83  fn compile(&mut self, input) -> Result<Completion, Reason>.
 *  Returns DDO_OK, DDO_CUTOFF, or DDO_ERR_*. */
int ddo_mdd_compile(ddo_mdd* mdd, const ddo_compile_input* input, ddo_completion* out);
/** Extension (not in the reference): B independent compiles in one device launch.
 *  statuses[i] receives the per-compile DDO_OK / DDO_CUTOFF / error.  inputs[i].cutoff is polled per compile like
 *  Cutoff::must_stop (clean.rs:352): a compile is stopped by ITS flag only -- the device-visible flag belongs to the launch, so
 *  compiles that a neighbour's flag cut are run again before the call returns. */
int ddo_mdd_compile_batch(ddo_mdd* const* mdds, const ddo_compile_input* inputs, ddo_completion* outs,
                          int* statuses, size_t count);
/** mdd.rs:86 */
int ddo_mdd_is_exact(const ddo_mdd* mdd);
/** mdd.rs:89 / :97 -- return 1 and write *value when Some, 0 when None */
int ddo_mdd_best_value(const ddo_mdd* mdd, int64_t* value);
int ddo_mdd_best_exact_value(const ddo_mdd* mdd, int64_t* value);
/** mdd.rs:92 / :100 -- residual path followed by the DD's best-edge chain (terminal first),
 *  same order as clean.rs:329-343.  *len: in = capacity of buf, out = number of decisions.
 *  Return 1 when Some, 0 when None, DDO_ERR_CAPACITY when buf is too small (*len = needed). */
int ddo_mdd_best_solution(const ddo_mdd* mdd, ddo_decision* buf, size_t* len);
int ddo_mdd_best_exact_solution(const ddo_mdd* mdd, ddo_decision* buf, size_t* len);
/** mdd.rs:107-113 drain_cutset: calls cb once per cut-set sub-problem; pointers inside the
 *  ddo_subproblem are valid during the callback only.  May be called once per compile. */
typedef void (*ddo_cutset_cb)(const ddo_subproblem* node, void* user);
int ddo_mdd_drain_cutset(ddo_mdd* mdd, ddo_cutset_cb cb, void* user);
/** Bulk form of drain_cutset for hosts that keep their fringe behind ONE lock.  The reference drains a cut-set INSIDE the solver's
 *  critical section (parallel.rs:456-469: `enqueue_cutset` holds the mutex across `mdd.drain_cutset(..)`), so whatever a shim does
 *  per node inside that call -- building the state, the `Arc`, the `Vec<Decision>` -- is serialised over all worker threads.  This
 *  call hands over the whole cut-set as flat rows; the shim turns them into `SubProblem`s at the END of its `compile()` -- in the
 *  worker's own thread, before the solver takes its lock -- and its `drain_cutset` only moves them into the closure.  Rows whose
 *  `ub <= ub_above` are left out (pass INT64_MIN for all of them): with `ub_above = input.best_lb` those are exactly the nodes both
 *  reference solvers drop in their closures (`if cutset_node.ub > best_lb`, parallel.rs:461-463, sequential.rs:372-376; the bound
 *  only rises between the compile and the drain).  Row order = the order of ddo_mdd_drain_cutset's callbacks.  The arrays belong
 *  to the mdd and stay valid until its next compile / drain / destroy.  Like ddo_mdd_drain_cutset: once per compile. */
typedef struct ddo_cutset_rows {
    size_t count;               /* rows */
    size_t state_words;
    size_t path_stride;         /* decisions per row of `paths` */
    const uint64_t* states;     /* [count][state_words] */
    const int64_t* values;      /* [count] */
    const int64_t* ubs;         /* [count] */
    const size_t* depths;       /* [count] depth of the node = residual.depth + its layer in the DD */
    const size_t* path_lens;    /* [count] decisions of row i: paths[i * path_stride .. i * path_stride + path_lens[i]) */
    const ddo_decision* paths;  /* the DD's part of the path only, node first (clean.rs:329-343); the residual's own path
                                   (ddo_compile_input.residual.path, which the caller holds) goes in FRONT of it */
} ddo_cutset_rows;
int ddo_mdd_drain_cutset_rows(ddo_mdd* mdd, int64_t ub_above, ddo_cutset_rows* rows);
/** Nodes in the cut-set of the latest compile that have not been drained (0 after a drain, for an exact DD, or without a compile). */
size_t ddo_mdd_cutset_count(const ddo_mdd* mdd);
/** Counters of the latest compile on this object. */
int ddo_mdd_last_counters(const ddo_mdd* mdd, ddo_counters* out);
/** Measurement support (no counterpart in the reference).  The reference runs one DecisionDiagram per worker thread and calls
 *  compile() from all of them concurrently (parallel.rs:576-602); here concurrent ddo_mdd_compile calls on mdds of one (model,
 *  device, width, engine) share device launches (one workgroup per decision diagram: see DESIGN.md, "combining layer").  Device
 *  launches and compiles that went through that layer on this mdd's engine since it was created (requests / launches = mean
 *  decision diagrams per launch) and the HIP-event time of those launches' kernels in milliseconds.  Any pointer may be NULL. */
int ddo_mdd_combine_stats(const ddo_mdd* mdd, uint64_t* launches, uint64_t* requests, double* kernel_ms);

/* ---- Solver (solver.rs:32-97; parallel.rs:287-641) ------------------------------------------ */
#define DDO_WIDTH_FIXED 0         /* width.rs:166-171 FixedWidth(w)            */
#define DDO_WIDTH_NB_UNASSIGNED 1 /* width.rs:397-402 NbUnassignedWidth(n)     */
#define DDO_WIDTH_TSPTW 2         /* examples/tsptw/heuristics.rs:38-52 TsptwWidth: nb_vars * (depth + 1) * factor,
                                     factor = ddo_solver_config.width */

/* Fringe implementations (abstraction/fringe.rs:26-45) */
#define DDO_FRINGE_NODUP 0 /* fringe/no_duplicate.rs:52-324: one entry per state, host resident (exact ddo order) */
#define DDO_FRINGE_LAZY 1  /* fringe/simple.rs:35-62 semantics (MaxUB order, no de-duplication): cut-sets stay in the
                              device node pool, the host orders whole cut-set blocks lazily by (ub, value)        */

typedef struct ddo_solver_config {
    int device;            /* HIP device ordinal                                                   */
    int width_policy;      /* DDO_WIDTH_FIXED | DDO_WIDTH_NB_UNASSIGNED                             */
    size_t width;          /* FixedWidth value (ignored for NB_UNASSIGNED)                          */
    int nb_concurrent;     /* sub-problems compiled concurrently on the device == the reference's
                              nb_threads (parallel.rs:328); 1 = the one-thread ParallelSolver         */
    double time_budget_s;  /* <= 0: NoCutoff; else TimeBudget (cutoff.rs:302-323)                   */
    int rank;              /* fringe shard owned by this solver ...                                 */
    int world_size;        /* ... out of this many (1 = whole problem); see ddo_solver_step        */
    int fringe;            /* DDO_FRINGE_NODUP | DDO_FRINGE_LAZY                                    */
    int sequential;        /* 1 (with nb_concurrent 1, DDO_FRINGE_NODUP): SequentialSolver's bookkeeping
                              (sequential.rs:433-461: `explored` counts every popped node, also the ones
                              skipped because ub <= best_lb) instead of ParallelSolver's (parallel.rs:531-553) */
    int cutset_type;       /* 0 / DDO_LAST_EXACT_LAYER | DDO_FRONTIER: the `D` of ParallelSolver<State, D, C>
                              (DefaultMDDLEL / DefaultMDDFC, mdd/mod.rs:42-49); DDO_FRONTIER needs DDO_FRINGE_NODUP */
    size_t cache_entries;  /* 0: EmptyCache; else a SimpleCache of at least that many entries in device memory
                              (the `C` of the solver: DefaultCachingSolver, solver/mod.rs); needs DDO_FRINGE_NODUP */
    size_t dominance_entries; /* 0: EmptyDominanceChecker; else a SimpleDominanceChecker with that many pairs per depth
                              (knapsack models: KPDominance, knapsack/main.rs:325); needs DDO_FRINGE_NODUP */
    size_t width_times;    /* 0: none; else the decorator Times(k, inner) (width.rs:636-642): max(1, k * inner), inner = the policy
                              above.  (Times(0, inner) is the constant 1 == FixedWidth(1).)                              */
    size_t width_div_by;   /* 0: none; else DivBy(k, inner) (width.rs:875-881): max(1, inner / k); with both set the width is
                              DivBy(div_by, Times(times, inner))                                                         */
    int pooled;            /* 1: `D` = Pooled (mdd/pooled.rs; Par / SeqNoCachingSolverPooled, solver/mod.rs:34, :43, and with
                              cache_entries > 0 Par / SeqCachingSolverPooled, :38, :47): MISP models, DDO_FRINGE_NODUP; see
                              DDO_MDD_POOLED                                                                             */
} ddo_solver_config;

ddo_solver* ddo_solver_create(const ddo_model* model, const ddo_solver_config* cfg);
/** WidthHeuristic::max_width (width.rs:166-171, 397-402, 636-642, 875-881; tsptw/heuristics.rs:38-52) of a sub-problem `depth`
 *  decisions below the root of a problem with `nb_vars` variables, as the solver evaluates it for `cfg`.  Pure host arithmetic
 *  (no device needed). */
size_t ddo_width_heuristic(const ddo_solver_config* cfg, size_t nb_vars, size_t depth);
void ddo_solver_destroy(ddo_solver* s);
/** solver.rs:37 maximize() -> Completion */
int ddo_solver_maximize(ddo_solver* s, ddo_completion* out);
/** solver.rs:70-96 */
int ddo_solver_best_value(const ddo_solver* s, int64_t* value);
int ddo_solver_best_solution(const ddo_solver* s, ddo_decision* buf, size_t* len);
int64_t ddo_solver_best_lower_bound(const ddo_solver* s);
int64_t ddo_solver_best_upper_bound(const ddo_solver* s);
int ddo_solver_set_primal(ddo_solver* s, int64_t value, const ddo_decision* solution, size_t len);
double ddo_solver_gap(const ddo_solver* s);
uint64_t ddo_solver_explored(const ddo_solver* s);
int ddo_solver_counters(const ddo_solver* s, ddo_counters* out);

/** Stepwise driving (one step = pop up to nb_concurrent sub-problems, restricted + relaxed compile,
 *  enqueue cut-sets), used by multi-GPU hosts that exchange the incumbent between steps.
 *  Returns 1 while work remains on this shard, 0 when its fringe is exhausted, DDO_CUTOFF on
 *  cutoff, <0 on error. */
int ddo_solver_step(ddo_solver* s);
/** With DDO_FRINGE_LAZY a step leaves its launch in flight and folds the results in during the next step;
 *  flush waits for it and absorbs the results (counters, incumbent and fringe are then up to date). */
int ddo_solver_flush(ddo_solver* s);
/** Measurement support (extension, DDO_FRINGE_LAZY only; no counterpart in the reference).  freeze: pops sub-problems
 *  in fringe order, keeps every `stride`-th one until `nbatches` x nb_concurrent are kept (or the fringe runs dry) and
 *  freezes them with the incumbent; returns how many batches were taken or DDO_ERR_*; bench_frozen = sub-problems kept.  bench_step: compiles frozen batch (k mod nbatches)
 *  exactly as ddo_solver_step would (same launch, same pipelining, same host work on the results) without folding the
 *  cut-sets into the fringe, so a timed region is the same work whatever its length.  Counters / explored /
 *  device_time accumulate as usual; ddo_solver_flush waits for the step in flight. */
int ddo_solver_bench_freeze(ddo_solver* s, int nbatches, int stride);
uint64_t ddo_solver_bench_frozen(const ddo_solver* s);
int ddo_solver_bench_step(ddo_solver* s);
/** One epoch of a sharded search in ONE call (one rank per GPU): `in` = the MAX-reduced vector of the previous
 *  epoch (NULL the first time; in[0] = the global incumbent, imported before the steps), then search steps until `min_ms` ms have passed
 *  (at least one, at most `max_steps`), then `out[7]` = this rank's contribution to the next MAX all-reduce: [incumbent, work remains,
 *  cut off, open nodes, -open nodes, best open bound, -best open bound].  Returns the last step's return value. */
int ddo_solver_epoch(ddo_solver* s, const int64_t* in, int64_t* out, int max_steps, double min_ms);
/** Lower bound seen by the next step (max-reduced across ranks by the caller, parallel.rs:439-453). */
int ddo_solver_import_lower_bound(ddo_solver* s, int64_t best_lb);
/** Work hand-over between the solvers of a sharded search (one rank per GPU): export pops up to
 *  `max_count` open sub-problems in fringe order as self-contained records -- `state_words` words each, value, ub, depth
 *  and the decisions from the problem root (paths[path_off[i] .. path_off[i+1]), path_off has max_count + 1 entries; the
 *  export stops early when fewer than nb_variables of the `path_cap` decisions are left).  The nodes leave this solver;
 *  import hands them to another solver of the same model (any fringe kind, any device), which explores them as its
 *  own.  The caller moves the bytes (torch.distributed broadcast in ddo_amd/distributed.py). */
int ddo_solver_export_subproblems(ddo_solver* s, size_t max_count, uint64_t* states, int64_t* value, int64_t* ub, int64_t* depth,
                                  uint64_t* path_off, ddo_decision* paths, size_t path_cap, size_t* count);
int ddo_solver_import_subproblems(ddo_solver* s, size_t count, const uint64_t* states, const int64_t* value, const int64_t* ub,
                                  const int64_t* depth, const uint64_t* path_off, const ddo_decision* paths);
/** Open sub-problems on this shard (fringe length), for termination detection (parallel.rs:512). */
uint64_t ddo_solver_fringe_len(const ddo_solver* s);
/** Largest upper bound left on this shard's fringe (INT64_MIN when empty). */
int64_t ddo_solver_fringe_best_ub(const ddo_solver* s);
/** Milliseconds spent inside device compile launches so far (HIP events on the engine's stream)
 *  and the number of launches. */
/* Per-tier accounting of the lazy solver (no reference counterpart: the reference has one engine).  Tier 0.. are the
 * capacity tiers (narrow decision diagrams, several per CU), then the dense tier (full width, two per CU) when it fits, the
 * last tier is the full-width engine.  kernel_ms = HIP-event time of the tier's launches. */
typedef struct ddo_tier_stats {
    double kernel_ms;
    uint64_t launches, subproblems, retried, nodes_expanded, lds_bytes;
    int32_t layer_capacity, threads, slots, dense;
} ddo_tier_stats;
int ddo_solver_tier_count(const ddo_solver* s);
int ddo_solver_tier_stats(const ddo_solver* s, int tier, ddo_tier_stats* out);
int ddo_solver_device_time(const ddo_solver* s, double* kernel_ms, uint64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* DDO_HIP_H */
