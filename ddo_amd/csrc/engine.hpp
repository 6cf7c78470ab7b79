// =============================================================================
// engine.hpp -- host-side objects behind include/ddo_hip.h: model descriptors,
// the per-(model, device, width) device engine and its batch interface.
// =============================================================================
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ddo_hip.h"
#include "dd_types.h"

namespace ddo_hip {

void set_error(const std::string& msg);
const char* get_error();

/// MISP model descriptor (examples/misp/main.rs:37-51): the closed-form stand-in for
/// `&dyn Problem + &dyn Relaxation + &dyn StateRanking`.
struct Model {
    int n = 0;
    int ws = 0;                      // ceil(n / 64): words per state at the ABI
    int wsT = 0;                     // words per state in the device kernel (template instance)
    std::vector<uint64_t> adj;       // [n][ws]   complement adjacency rows
    std::vector<int64_t> weight;     // [n]   MISP: vertex weights; knapsack: item profits
    bool unit_weights = true;
    int64_t weight_abs_sum = 0;
    // knapsack (examples/knapsack/main.rs:53-72): `Knapsack` + `KPRelax` + `KPRanking`; the state is one word, the
    // remaining capacity (all nodes of a layer share their depth)
    int kind = MODEL_MISP;
    int64_t kp_capacity = 0;
    std::vector<int64_t> kp_weight;  // [n]
    std::vector<int32_t> kp_order;   // [n]   items by decreasing profit / weight (main.rs:66-70)
    // maximum cut (examples/mcp/{graph,model,relax}.rs): `Mcp` + `McpRelax` + `McpRanking`; the state is the vector of n
    // signed benefits, two per word, followed by a depth word
    std::vector<int32_t> vgraph;     // [n][n] symmetric edge weights
    std::vector<int32_t> vest, vnk;  // [n+1] each (relax.rs:58-106)
    int64_t initial_value = 0;       // Problem::initial_value (MCP: sum of the negative edge weights; MAX2SAT: tautologies)
    // MAX2SAT (examples/max2sat/{data,model,relax,heuristics}.rs): same state layout, merge, relax and ranking as MCP
    std::vector<int32_t> m2_w[4];    // [n][n] each: weight(t(k),t(l)), weight(t(k),f(l)), weight(f(k),t(l)), weight(f(k),f(l))
    std::vector<int32_t> m2_order, m2_rankpos;   // vars_by_sum_of_clause_weights and its inverse
    void initial_state(uint64_t* out) const;   // Problem::initial_state
    /// decision value of the device's decision bit: MISP / knapsack 0 | 1, MCP +1 (side S) | -1 (side T)
    int64_t decision_value(uint32_t bit) const { return (kind == MODEL_MCP || kind == MODEL_MAX2SAT) ? (bit ? -1 : 1) : (int64_t)bit; }   // TSPTW: the node index itself
    /// decisions on the device wire: (variable << dbits) | decision index (binary models: 1 bit; TSPTW: the node, 6 bits up to 64 nodes, 8 bits above)
    int dbits = 1;
    ddo_decision path_decision(uint32_t x) const { return ddo_decision{(int64_t)(x >> dbits), decision_value(x & ((1u << dbits) - 1u))}; }
    uint32_t path_word(const ddo_decision& d) const {
        const uint32_t idx = kind == MODEL_TSPTW ? (uint32_t)d.value : (decision_value(1) == d.value ? 1u : 0u);
        return ((uint32_t)d.variable << dbits) | idx;
    }
    // TSPTW (examples/tsptw): distance matrix and time windows in 1/10000 units (instance.rs:87-98), cheapest entering edges
    std::vector<int32_t> tw_dist, tw_early, tw_late, tw_cheap, tw_order;

    std::mutex mtx;
    std::map<std::pair<int, long>, std::weak_ptr<class Engine>> engines;  // (device, max_width)

    /// MispRanking::compare (main.rs:205-208): popcount, then BitSet::cmp
    int compare_states(const uint64_t* a, const uint64_t* b) const;
    int popcount(const uint64_t* a) const;
};

/// SimpleCache (cache/simple.rs:36-73) on the device: one open-addressing table in HBM (dd_thresholds.hpp) that every
/// compile launched with it reads (_filter_with_cache, must_explore) and updates (_maybe_update_cache).
struct CacheTable {
    int device = 0;
    uint64_t* tab = nullptr;
    uint64_t cap = 0;                  // entries, power of two
    int stride = 0;                    // u64 words per entry: tag, threshold, depth, state words
    unsigned long long* stats = nullptr;   // device: [0] entries in use, [1] updates dropped because the table was full
    ~CacheTable();
    static CacheTable* create(const Model* model, int device, size_t capacity_entries);
    int clear();
    int read_stats(uint64_t* used, uint64_t* dropped) const;
};

/// SimpleDominanceChecker (dominance/simple.rs:37-117) on the device, for models whose dominance key is the depth and whose
/// states carry one coordinate besides the value (knapsack): per depth a sorted Pareto front in HBM (misp_dd_core.hpp).
struct DominanceTable {
    int device = 0;
    uint64_t* coord = nullptr;
    int32_t* value = nullptr;
    uint32_t *count = nullptr, *lock = nullptr;
    unsigned long long* stats = nullptr;
    uint32_t cap = 0;
    int depths = 0;
    // TSPTW (TsptwDominance, examples/tsptw/dominance.rs:26-60): best value per (depth, position, must_visit) -- a hash table with
    // the cache's entry layout, 3 key words
    uint64_t* dkey = nullptr;
    uint64_t dkey_cap = 0;
    int dkey_stride = 6;   // words per entry: 3 + the key (2K + 1 words, K = words of a node set, dd_tsptw.hpp)
    ~DominanceTable();
    static DominanceTable* create(const Model* model, int device, size_t capacity_per_depth);
    int clear();
};

/// One decoded compile() result living on the host.
struct HostResult {
    DDResult hdr{};
    std::vector<uint32_t> best_path;    // (var << 1 | value), terminal first
    std::vector<uint32_t> exact_path;
    int n_cutset = 0;
    int cs_path_len = 0;
    std::vector<uint64_t> cs_state;     // n_cutset x ws (ABI words)
    std::vector<int32_t> cs_value, cs_ub;
    std::vector<uint32_t> cs_path;      // n_cutset x cs_path_len, node first
    // IN_PATH_BITS (DDResult::cs_lvar_off): the paths as the device wrote them -- per node cs_pw words of decision bits (bit tr =
    // decision of transition tr) and the variable of every transition once; expanded when the cut-set is drained
    std::vector<uint64_t> cs_pbits;     // n_cutset x cs_pw
    std::vector<uint32_t> cs_lvar;      // cs_path_len
    int cs_pw = 0;
    std::vector<int32_t> cs_depth;      // frontier cut-set: layer of every node below the DD's root (empty: all at cs_path_len)
    std::vector<int32_t> cs_plen;       // Pooled: decisions on the path of every node (<= its depth: Engine::pooled_fixup); empty: the depth
    uint64_t pool_off = ~0ULL;          // IN_POOL_OUT: the cut-set block stayed in the device node pool
    bool valid = false;
    void clear() {
        valid = false;
        best_path.clear();
        exact_path.clear();
        n_cutset = 0;
        cs_path_len = 0;
        cs_state.clear();
        cs_value.clear();
        cs_ub.clear();
        cs_path.clear();
        cs_pbits.clear();
        cs_lvar.clear();
        cs_pw = 0;
        cs_depth.clear();
        cs_plen.clear();
    }
};

/// Device engine: HBM workspace for `nslots` concurrent decision diagrams + batch launcher.
class Engine {
  public:
    /// features: ENGINE_KEEP_LAYERS = every layer of a DD is kept (frontier cut-set, thresholds, cache): layer-rebuilding engine
    static constexpr int ENGINE_KEEP_LAYERS = 1;
    /// ENGINE_POOLED = the in-place engine compiles Pooled decision diagrams (mdd/pooled.rs:117-823; MISP): a slot is sized for a POOL of as
    /// many nodes as the LDS dedup table admits (about 14 000), whatever the width the layers are squashed to
    static constexpr int ENGINE_POOLED = 2;
    bool is_pooled() const { return pooled_; }
    /// Pooled results name one decision BIT per layer; the reference's paths hold a decision only where the variable impacted the
    /// path's node (pooled.rs:316-334: one edge per EXPANDED ancestor).  Replays the paths of `out` from the residual state of `in`
    /// and drops the other layers: best / exact path and cut-set rows become (variable << 1 | bit) lists, cs_plen their lengths.
    void pooled_fixup(const DDInput& in, HostResult& out) const;
    static std::shared_ptr<Engine> get(Model* model, int device, long max_width, int features = 0);
    bool keeps_layers() const { return P_.tmode != 0; }
    /// an engine of its own (not shared through the model): needed by owners of the device node pool
    static std::shared_ptr<Engine> create_private(Model* model, int device, long max_width);
    /// A capacity tier of `owner` (in-place engine only): node slots for decision diagrams whose layers stay within
    /// `cap_width` nodes, `threads` per workgroup, many workgroups per CU.  It compiles the same sub-problems with the
    /// same width semantics (widths up to owner->max_width()) out of the owner's node pool, never squashes, and reports
    /// ST_RETRY for a DD that outgrows it.  The owner must outlive the tier.
    static std::shared_ptr<Engine> create_tier(Model* model, int device, Engine* owner, int cap_width, int threads);
    /// The engine behind an mdd created with DDO_MDD_ENGINE_* (`selector` = that flag): a tier (DENSE: max_width / 512 threads,
    /// TIER0: 256 / 64, TIER1: 1024 / 128) of a private full-width owner, which the tier keeps alive.  Shared per
    /// (model, device, max_width, selector) like Engine::get.
    static std::shared_ptr<Engine> get_selected(Model* model, int device, long max_width, int selector);
    ~Engine();

    /// Runs `count` work items in one launch.  results: 2 per item ([1] used by IN_FUSED).
    /// Returns DDO_OK or a negative error.  Thread-safe (serialised internally).
    /// `stop_flags` (optional, `nflags` of them, entries may be null): host flags of the callers' Cutoff objects
    /// (`ddo_compile_input.cutoff`).  While the launch runs they are polled; when one is raised the device-visible cutoff flag
    /// goes up and the running compiles end with ST_CUTOFF at their next layer -- Cutoff::must_stop inside the layer loop
    /// (clean.rs:352) for a TimeBudget that expires mid-compile.
    int run_batch(const DDInput* inputs, int count, std::vector<HostResult>& results, const CacheTable* cache = nullptr,
                  const DominanceTable* dom = nullptr, const volatile int* const* stop_flags = nullptr, int nflags = 0);
    /// the kernel of the launch in flight has finished (non-blocking)
    bool kernel_done();
    /// One work item on its own, for a compile that found the shared output arena of its batch full: runs it, and while its
    /// own output does not fit, enlarges the arena (4x per attempt up to 8 GB) and runs it again -- all under the batch lock,
    /// so that no other host thread's launch gets between the growth and the retry.  A compile that fails on the arena
    /// leaves nothing in the cache (misp_dd_core.hpp: thresholds follow the reservation), so repeating it is sound.
    int run_solo_growing(const DDInput& input, std::vector<HostResult>& results, const CacheTable* cache = nullptr,
                         const DominanceTable* dom = nullptr);
    /// ---- the combining layer under ddo_mdd_compile ------------------------------------------------------------------------------
    /// The reference's threading contract is one DecisionDiagram per worker thread, compile() called CONCURRENTLY from all of them
    /// (parallel.rs:576-602: `let mut mdd = D::default();` in every thread, process_one_node in a loop).  One compile is one
    /// workgroup; a launch per caller would run the 256 CUs one workgroup at a time.  So concurrent compile() calls on the mdds of
    /// an engine RENDEZVOUS (flat combining): a caller queues its request; if no launch is being put together it becomes the
    /// leader, waits a short window for the other worker threads (until as many requests as there are mdds on the engine -- or
    /// node slots -- have arrived; the window is a fraction of the previous launch's length), takes everything that is queued and
    /// sends it out as ONE launch; callers that arrive while that launch runs queue up for the next one, whose leader is the
    /// first of them.  Every caller decodes its own result out of the pinned output arena (in parallel, while the next launch
    /// already runs out of the other buffer set).  Per-compile semantics stay per compile: a request cut by ANOTHER caller's
    /// cutoff flag goes back into the queue (Cutoff::must_stop is per compile, clean.rs:352), one that found the shared output
    /// arena full is compiled again (the arena grows when a compile does not fit it on its own).
    struct Waiter;
    struct CompileReq {
        DDInput in{};
        const volatile int* stop = nullptr;   // the caller's Cutoff flag (ddo_compile_input.cutoff)
        const CacheTable* cache = nullptr;
        const DominanceTable* dom = nullptr;
        bool route_ok = false;                // the mdd left the choice of the kernel to ddo_mdd_create (no DDO_MDD_ENGINE_* selector): a small
                                              // launch of the dense engine may run on its owner, the full-width engine
        HostResult* out = nullptr;            // decoded result (two records when in.flags has IN_FUSED: out[0], out[1])
        int rc = DDO_OK;                      // launch-level error (DDO_ERR_*): nothing was decoded
        // filled by the combiner
        Waiter* waiter = nullptr;
        int state = 0;                        // 0 queued / in a launch, 2 raw result ready (decode, then release the buffer set), 3 done
        DDResult hdr{};
        const uint8_t* arena = nullptr;
        size_t arena_used = 0;
        int set = -1;
        Engine* raw_eng = nullptr;            // the engine that buffer set belongs to (release_set)
    };
    /// Compiles `count` requests through the combining layer; returns when all of them are finished (their `out` decoded or `rc` set).
    int compile_combined(CompileReq* const* reqs, int count);
    /// mdds bound to this engine (= worker threads that may call compile concurrently): what a leader waits for
    void add_user(int d) { users_.fetch_add(d); }
    /// launches / requests that went through the combining layer (tests, bench: mean decision diagrams per launch)
    void combine_stats(uint64_t* launches, uint64_t* requests, double* kernel_ms) const {
        if (launches) *launches = cq_launches_.load();
        if (requests) *requests = cq_requests_.load();
        if (kernel_ms) *kernel_ms = (double)cq_kernel_us_.load() * 1e-3;
    }
    /// decodes one result record of a fetched batch (a record whose block lies beyond `arena_used` becomes a capacity error)
    void decode_checked(const DDResult& r, const uint8_t* arena, size_t arena_used, HostResult& out) const;

    /// The two halves of run_batch: launch() enqueues upload + kernel + download of the result headers and
    /// returns at once; collect() waits, fetches the arena and decodes.  One launch may be in flight.
    int launch(const DDInput* inputs, int count, const CacheTable* cache = nullptr, const DominanceTable* dom = nullptr);
    int collect(std::vector<HostResult>& results);   // == wait() + fetch()
    /// wait(): the launch in flight has left the device (its result headers are on the host).  After it a new
    /// launch() may be issued at once -- it uses the other buffer set -- and fetch() then downloads and decodes
    /// the finished batch's output arena on a second stream while the new kernel runs.
    int wait();
    int fetch(std::vector<HostResult>& results);
    /// fetch() without the decoding: the result records (2 per item) and the output arena of the finished launch as they
    /// lie in pinned host memory.  The pointers stay valid until the launch AFTER THE NEXT ONE of this engine (two buffer sets).
    /// The lazy solver reads them in place: building a HostResult (seven vectors) per record cost the proof search tens of
    /// seconds of host time for its 86 million sub-problem launches.
    struct RawBatch {
        const DDResult* hdr = nullptr;
        const uint8_t* arena = nullptr;
        int count = 0;
        size_t arena_used = 0;
        int set = -1;   // buffer set the batch lies in
    };
    int fetch_raw(RawBatch& out);
    /// pinned staging buffer for the inputs of the next launch (`count` records): fill it in place, then launch(nullptr, count)
    DDInput* stage_inputs(int count);
    bool in_flight() const { return pending_ > 0; }

    int device() const { return device_; }
    long max_width() const { return max_width_; }
    int nslots() const { return nslots_; }
    int cap_width() const { return cap_width_; }
    bool is_tier() const { return owner_ != nullptr; }
    std::shared_ptr<Engine> owner_shared() const { return owner_ref_; }   // get_selected: the full-width owner of this tier
    bool is_dense() const { return dense_; }
    int threads() const { return threads_; }
    int engine_kind() const { return engine_kind_; }
    size_t lds_bytes() const { return lds_bytes_; }
    double kernel_ms() const { return kernel_ms_; }
    uint64_t launches() const { return launches_; }
    double last_kernel_ms() const { return last_kernel_ms_; }
    /// device-visible cutoff flag (set asynchronously by the host to abort running compiles)
    void set_cutoff(bool on);
    /// device node pool (cut-set blocks that never leave HBM)
    bool has_pool() const { return P_.pool != nullptr; }
    uint64_t pool_capacity() const { return P_.pool_cap; }
    int pool_reset();
    /// a solver that will keep `count` sub-problems in flight: the worst case of their cut-set blocks (twice: the launch in flight and the
    /// one being fetched) is mapped by the pool's mapper thread while the search compiles its first, small steps
    void pool_expect(int count);
    /// bench support: the next launch() first rewinds the pool's bump allocator to `head` (on the engine stream, ahead
    /// of the kernel), so that a frozen batch can be compiled again and again without growing the pool
    void set_pool_rewind(uint64_t head) { rewind_ = (long long)head; }
    /// current head of the pool's bump allocator (synchronous: call with no launch in flight)
    int read_pool_head(uint64_t* out);
    /// appends a host-built block to the node pool (imported sub-problems); call with no launch in flight
    int pool_append(const void* data, size_t bytes, uint64_t* off);
    int read_pool(uint64_t off, void* dst, size_t bytes);
    /// `count` 8-byte words that lie `stride_bytes` apart (one row of a word-major matrix in a pool block) with ONE copy
    int read_pool_strided(uint64_t off, uint64_t* dst, size_t count, size_t stride_bytes);
    int words_per_state_device() const { return P_.ws; }

  private:
    Engine() = default;
    int init(Model* model, int device, long max_width, bool want_pool, Engine* owner = nullptr, int cap_width = 0, int tier_threads = 0,
             int features = 0);
    Engine* pool_owner() { return owner_ ? owner_ : this; }
    Engine* owner_ = nullptr;        // capacity tier: the engine whose node pool / cutoff flag this one shares
    std::shared_ptr<Engine> owner_ref_;   // get_selected: the tier owns its owner
    int cap_width_ = 0;              // capacity tier: layer capacity (0 = full-width engine)
    void* kernel_ = nullptr;         // the __global__ entry picked in init(): launch() must use the very same one
    bool pooled_ = false;            // ENGINE_POOLED
    bool dense_ = false;             // two 512-thread workgroups per CU (kernels_inplace_tier.hip: misp_compile_kernel2_dense)
    bool mid_ = false;               // wide capacity tier: four 256-thread workgroups per CU on the same kernel build
    void decode(const DDResult& r, const uint8_t* arena, HostResult& out) const;
  public:
    /// After wait(), before fetch(): which sub-problems of the finished launch ended with ST_RETRY (a capacity tier hands them
    /// up).  Lets the host launch the next tier before it decodes this one.
    int peek_retry(std::vector<uint8_t>& retry);
  private:
  public:
    /// Enlarge the output arena (no launch may be in flight): one sub-problem's cut-set did not fit.
    int grow_arena(size_t bytes);
    size_t arena_capacity() const { return arena_cap_; }
  private:

    Model* model_ = nullptr;
    int device_ = 0;
    long max_width_ = 0;
    int nslots_ = 0;
    int threads_ = 0;
    size_t lds_bytes_ = 0;
    bool table_lds_ = true;
    bool keys_global_ = false;
    int engine_kind_ = 2;            // 1: per-layer rebuild (misp_dd_core.hpp), 2: in-place layers (misp_dd_inplace.hpp)
    EngineParams P_{};
    std::vector<void*> allocs_;
    void* stream_ = nullptr;
    void* ev0_ = nullptr;
    void* ev1_ = nullptr;
    /// ping-pong I/O buffers: batch k+1 is uploaded and launched while batch k's arena is still being downloaded
    struct IoSet {
        void* d_inputs = nullptr;
        void* d_results = nullptr;
        int in_cap = 0;
        uint8_t* d_cnt = nullptr;      // [0] work counter (i32), [8..16) arena head (u64)
        uint8_t* h_arena = nullptr;    // pinned host memory, written directly by the kernel (zero-copy output)
        size_t h_arena_cap = 0;
        DDResult* h_results = nullptr; // pinned: a pageable destination would make the "async" download block launch()
        unsigned long long* h_head = nullptr;   // pinned (same allocation)
        DDInput* h_inputs = nullptr;            // pinned staging copy of the batch (same allocation)
        int count = 0;
    };
    IoSet io_[2];
    int io_reserve(IoSet& io, int count);   // input / result buffers of a buffer set for `count` sub-problems (pinned staging + device)
    int next_set_ = 0;       // set the next launch() uses
    int pending_set_ = -1;   // set of the launch in flight
    int fetch_set_ = -1;     // set whose kernel finished (wait() done) but whose arena is not fetched yet
    void* copy_stream_ = nullptr;
    void* d_counters_ = nullptr;   // [16] cutoff flag, [32..40) pool head
    // Growable node pool (default; DDO_HIP_POOL_VMM=0 selects one fixed allocation): the address range is reserved once (hipMemAddressReserve), physical chunks
    // are mapped as the search fills it -- creating a solver costs milliseconds instead of the seconds a 64 GB hipMalloc
    // takes, co-resident solvers only hold what they use, and one search can grow into all of the 288 GB.
    uint8_t* vm_base_ = nullptr;
    size_t vm_reserved_ = 0, vm_chunk_ = 0;
    std::atomic<size_t> vm_mapped_{0};
    std::mutex vm_mtx_;                // mapping chunks: the mapper thread or a launch that cannot wait for it
    std::condition_variable vm_cv_;
    std::thread vm_thread_;
    size_t vm_want_ = 0;
    bool vm_stop_ = false;
    void pool_want(size_t bytes);      // asks the mapper thread to have `bytes` of the pool mapped
    int pool_grow_locked(size_t target);
    std::vector<void*> vm_handles_;
    uint64_t pool_head_bound_ = 0;     // upper bound of the pool head over every result fetched so far
    size_t pool_unfetched_worst_ = 0;  // worst-case growth of the launches whose results were not fetched yet
    int pool_grow(size_t target);
    void pool_release();
    size_t arena_cap_ = 0;
    int pending_ = 0;
    double kernel_ms_ = 0, last_kernel_ms_ = 0;
    uint64_t launches_ = 0;
    std::mutex mtx_;
    long long rewind_ = -1;
    unsigned long long rewind_val_ = 0;
    std::mutex batch_mtx_;   // held across launch + wait + fetch of run_batch / of a combined launch: one launch of this engine at a time
    std::mutex cut_mtx_;     // set_cutoff (must not wait for mtx_, which wait() holds while the driver sleeps on a long kernel)
    // combining layer (compile_combined)
    std::mutex cq_mtx_;
    std::condition_variable cq_cv_;        // the leader's window: woken when enough requests have arrived
    std::deque<CompileReq*> cq_;
    bool cq_leader_ = false;               // a caller is putting a launch together / has one in flight
    int cq_batch_cap_ = 1 << 20;           // requests per launch (shrinks when the output arena, at its largest, overflows)
    double cq_last_ms_ = 0;                // length of the previous combined launch (window = a fraction of it)
    int cq_active_ = 0;                    // callers expected at the next launch: the ones the previous launch released + the ones already queued
    std::atomic<int> users_{0};
    std::atomic<uint64_t> cq_launches_{0}, cq_requests_{0}, cq_kernel_us_{0};
    std::mutex dec_mtx_;
    std::condition_variable dec_cv_;
    int decoders_[2] = {0, 0};             // callers still reading results / arena of buffer set k
    void wait_decoders(int set, Waiter* me = nullptr);   // -1: both sets; me: the waiting leader decodes what it is handed meanwhile
    void decode_ready(Waiter* me);
    void release_set(int set);
    struct HandOut;
    double cq_bytes_per_req_ = 0;          // output-arena bytes per compile of the recent launches (sizes the arena ahead of a launch)
    void lead(Waiter* me);
    void hand_out(HandOut& ho, Waiter* me);
    int combined_launch(std::vector<CompileReq*>& batch, std::vector<CompileReq*>& again, Waiter* me, HandOut& ho);
    void finish_req(CompileReq* r, int state);
};

/// Reads a DIMACS-like .clq file the way examples/misp/main.rs:258-317 does.
/// Returns false (and sets the error text) on IO / format errors.
bool read_misp_clq(const std::string& path, int& n, std::vector<uint64_t>& rows, std::vector<int64_t>& weights);
/// Reads a knapsack instance the way examples/knapsack/main.rs:267-303 does ("n capacity", then n lines "profit weight";
/// lines starting with 'c' are comments).
bool read_knapsack(const std::string& path, int64_t& capacity, std::vector<int64_t>& profit, std::vector<int64_t>& weight);
/// Reads a max-cut instance the way examples/mcp/graph.rs:48-79 does ("c " comments, "<vertices> <edges>", "<src> <dst> <w>").
bool read_mcp(const std::string& path, int& n, std::vector<int64_t>& adj);
/// Reads a weighted MAX2SAT instance the way examples/max2sat/data.rs:67-116 does: clause k = (lit_a[k], lit_b[k], weight[k]),
/// unit clauses with lit_a == lit_b, in file order (a clause listed twice keeps its last weight: the caller applies that).
bool read_max2sat(const std::string& path, int& n, std::vector<int64_t>& lit_a, std::vector<int64_t>& lit_b, std::vector<int64_t>& weight);
/// Reads a TSPTW instance the way examples/tsptw/instance.rs:52-109 does: nb_nodes, the distance matrix, one time window per node;
/// every number is `(f32 * 10000.0) as usize`.
bool read_tsptw(const std::string& path, int& n, std::vector<int64_t>& dist, std::vector<int64_t>& earliest, std::vector<int64_t>& latest);

}  // namespace ddo_hip

// opaque ABI types
struct ddo_model {
    ddo_hip::Model m;
};
