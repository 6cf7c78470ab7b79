// =============================================================================
// ddo_hip_engine.hip -- the device engine (workspace, launches) + the model /
// mdd half of the C ABI (include/ddo_hip.h).  The gfx950 kernels are compiled in
// kernels_inplace*.hip / kernels_core*.hip (separate translation units: the build
// runs them in parallel).
//
// Execution model: one persistent workgroup per concurrently compiled decision
// diagram ("slot").  A launch of min(batch, nslots) workgroups drains a batch of
// sub-problems through a device work counter; every workgroup runs the complete
// layer loop of its DD (misp_dd_core.hpp) out of its own HBM slot + LDS, so a
// compile needs no host round trip and no inter-workgroup synchronisation.
// =============================================================================
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <thread>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <type_traits>

#include "../../include/ddo_hip.h"
#include "engine.hpp"
#include "kernels.hpp"
#include "misp_dd_inplace.hpp"   // LDS layout sizes (dd_lds_bytes / dd2_lds_bytes); the kernels themselves are not instantiated here

namespace ddo_hip {

// host seconds inside launch(), by section (DDO_HIP_TIMES prints them when an engine goes away): checks + staging, inputs to the device,
// launch order, pool growth, kernel launch, result copies
static std::atomic<uint64_t> g_launch_ns[8];   // (engines are launched from several host threads)
static std::atomic<uint64_t> g_launch_n{0};
static const bool g_times_on = std::getenv("DDO_HIP_TIMES") != nullptr;
// combining layer (Engine::lead / combined_launch), host seconds by section: window, staging + launch, kernel wait, hand-out, idle
// between the end of a launch and the start of the next leader's window
static std::atomic<uint64_t> g_cq_ns[6];
static std::atomic<int64_t> g_cq_last_end{0};
static inline int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }


// Launch order of a batch on the in-place engine: sub-problems by decreasing number of vertices left in their residual state -- a
// counting sort over the popcounts, ties in no particular order.  order[k] = index of the k-th DD to be drawn.
//   lpt_count_kernel   (n threads)       key[i] = vertices left in sub-problem i, bins[key[i]] += 1   (bins zeroed by the caller;
//                                         a histogram per workgroup in LDS first)
//   lpt_order_kernel   (one workgroup)   start of every bin (the fullest states first), then order[start[key[i]]++] = i
constexpr int LPT_BINS = 64 * MAX_WS + 1;
struct LptBuffers {   // behind the inputs of an I/O set (Engine::launch)
    uint32_t* order;
    uint32_t* key;
    uint32_t* bins;
    int32_t* done;    // split launches: one flag per sub-problem (EngineParams::done)
    static size_t bytes(int cap) { return (size_t)cap * 12 + (size_t)LPT_BINS * 4; }
    LptBuffers(void* d_inputs, int cap) {
        order = (uint32_t*)((uint8_t*)d_inputs + (size_t)cap * sizeof(DDInput));
        key = order + cap;
        bins = key + cap;
        done = (int32_t*)(bins + LPT_BINS);
    }
};

__global__ void __launch_bounds__(256) lpt_count_kernel(const DDInput* __restrict__ in, int n, const uint8_t* __restrict__ pool, int ws, int nbins, uint32_t* __restrict__ key,
                                                        uint32_t* __restrict__ bins) {
    __shared__ uint32_t mine[LPT_BINS];   // this workgroup's share of the histogram (most states of a launch fall into a few dozen bins:
                                          // straight into HBM that is thousands of atomics on the same few addresses)
    for (int b = threadIdx.x; b < nbins; b += blockDim.x) mine[b] = 0;
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const DDInput& d = in[i];
        int pc = 0;
        if (d.src_off != NO_POOL_SRC) {   // a row of a cut-set block in the device pool (word-major rows)
            const PoolBlockHeader* h = (const PoolBlockHeader*)(pool + d.src_off);
            const uint64_t* rows = (const uint64_t*)(pool + d.src_off + h->off_states);
            const int hw = (int)h->ws < ws ? (int)h->ws : ws;
            const size_t stride = h->rows;
            for (int k = 0; k < hw; ++k) pc += __popcll(rows[(size_t)k * stride + d.src_row]);
        } else {
            for (int k = 0; k < ws; ++k) pc += __popcll(d.state[k]);
        }
        pc = pc < nbins ? pc : nbins - 1;
        key[i] = (uint32_t)pc;
        atomicAdd(&mine[pc], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += blockDim.x)
        if (mine[b]) atomicAdd(&bins[b], mine[b]);
}

__global__ void __launch_bounds__(1024) lpt_order_kernel(int n, int nbins, const uint32_t* __restrict__ key, const uint32_t* __restrict__ bins, uint32_t* __restrict__ order) {
    __shared__ uint32_t start[LPT_BINS];
    for (int b = threadIdx.x; b < nbins; b += blockDim.x) start[b] = bins[b];
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t at = 0;
        for (int b = nbins - 1; b >= 0; --b) {
            const uint32_t cnt = start[b];
            start[b] = at;
            at += cnt;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) order[atomicAdd(&start[key[i]], 1u)] = (uint32_t)i;
}

static int pick_ws(int ws) {
    const int opts[] = {1, 2, 4, 7, 8, 16, 32, 72};
    for (int o : opts)
        if (ws <= o) return o;
    return -1;
}

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
const char* get_error() { return g_error.c_str(); }

#define HIP_TRY(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                        \
            return DDO_ERR_NO_DEVICE;                                                            \
        }                                                                                        \
    } while (0)

// ---------------------------------------------------------------------------
// Model
// ---------------------------------------------------------------------------
int Model::popcount(const uint64_t* a) const {
    int c = 0;
    for (int k = 0; k < ws; ++k) c += __builtin_popcountll(a[k]);
    return c;
}
void Model::initial_state(uint64_t* out) const {
    for (int k = 0; k < ws; ++k) out[k] = 0;
    if (kind == MODEL_MCP || kind == MODEL_MAX2SAT) return;                     // mcp/model.rs:51-53, max2sat/model.rs:259-264
    if (kind == MODEL_TSPTW) {   // tsptw/model.rs:36-47: at the depot (node 0) at time 0, every other node still to visit
        const int K = tw_set_words(n);
        for (int i = 1; i < n; ++i) out[K + i / 64] |= 1ULL << (i % 64);
        return;
    }
    if (kind == MODEL_KNAPSACK) out[0] = (uint64_t)kp_capacity;                 // knapsack/main.rs:100-102
    else for (int i = 0; i < n; ++i) out[i / 64] |= 1ULL << (i % 64);           // misp/main.rs:69-71
}
int Model::compare_states(const uint64_t* a, const uint64_t* b) const {
    if (kind == MODEL_KNAPSACK) return a[0] < b[0] ? -1 : (a[0] > b[0] ? 1 : 0);   // KPRanking (knapsack/main.rs:187-194)
    if (kind == MODEL_TSPTW) {   // TsptwRanking (tsptw/heuristics.rs:29-36): the depth; ties: packed words (shared tie-break)
        const uint64_t da = (a[ws - 1] >> 32) & 0xFFFF, db = (b[ws - 1] >> 32) & 0xFFFF;
        if (da != db) return da < db ? -1 : 1;
        for (int k = 0; k < ws; ++k)
            if (a[k] != b[k]) return a[k] < b[k] ? -1 : 1;
        return 0;
    }
    if (kind == MODEL_MCP || kind == MODEL_MAX2SAT) {              // McpRanking (mcp/model.rs:154-163), Max2SatRanking
        auto rank = [&](const uint64_t* s) {
            int64_t r = 0;
            for (int v = 0; v < n; ++v) {
                const int32_t x = (int32_t)(uint32_t)(s[v >> 1] >> (32 * (v & 1)));
                r += x < 0 ? -(int64_t)x : x;
            }
            return r;
        };
        const int64_t ra = rank(a), rb = rank(b);
        if (ra != rb) return ra < rb ? -1 : 1;
        // Deterministic tie-break shared with the device kernels (lexkey = raw word, misp_dd_core.hpp) and with the
        // CPU restatement used by the tests (compare_signed_vectors): the reference leaves ties of this ranking to the
        // iteration order of its hash map; here the packed state words decide, compared from word 0, larger first.
        for (int k = 0; k < ws; ++k)
            if (a[k] != b[k]) return a[k] < b[k] ? -1 : 1;
        return 0;
    }
    int pa = popcount(a), pb = popcount(b);
    if (pa != pb) return pa < pb ? -1 : 1;
    // equal popcounts: at the lowest differing member, the set owning it is the smaller one
    for (int k = 0; k < ws; ++k) {
        uint64_t x = a[k] ^ b[k];
        if (x) {
            uint64_t t = x & (~x + 1);
            return (a[k] & t) ? -1 : 1;
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------
// Engine
// ---------------------------------------------------------------------------
std::shared_ptr<Engine> Engine::get(Model* model, int device, long max_width, int features) {
    std::lock_guard<std::mutex> g(model->mtx);
    auto key = std::make_pair(device, max_width * 8 + features);
    auto it = model->engines.find(key);
    if (it != model->engines.end()) {
        if (auto sp = it->second.lock()) return sp;
    }
    std::shared_ptr<Engine> e(new Engine());
    if (e->init(model, device, max_width, false, nullptr, 0, 0, features) != DDO_OK) return nullptr;
    model->engines[key] = e;
    return e;
}

std::shared_ptr<Engine> Engine::create_private(Model* model, int device, long max_width) {
    std::shared_ptr<Engine> e(new Engine());
    if (e->init(model, device, max_width, true) != DDO_OK) return nullptr;
    return e;
}

std::shared_ptr<Engine> Engine::create_tier(Model* model, int device, Engine* owner, int cap_width, int threads) {
    // cap_width == max_width: the DENSE tier -- full layer capacity (it squashes like the owner), half the dedup table, 512
    // threads, two workgroups per CU; a layer whose table would overflow hands the DD to the owner (ST_RETRY)
    const bool dense = owner && (long)cap_width == owner->max_width() && threads == 512;
    if (!owner || owner->engine_kind() != 2 || !owner->has_pool() || cap_width < 8 ||
        (!dense && (2 * (long)cap_width > owner->max_width() || threads < 64 || threads > 256 || threads % 64))) {
        set_error("Engine::create_tier: needs an in-place owner engine with a node pool, 8 <= cap_width <= max_width / 2, 64..256 threads (or cap_width == max_width with 512 threads: dense tier)");
        return nullptr;
    }
    std::shared_ptr<Engine> e(new Engine());
    if (e->init(model, device, owner->max_width(), false, owner, cap_width, threads) != DDO_OK) return nullptr;
    return e;
}

std::shared_ptr<Engine> Engine::get_selected(Model* model, int device, long max_width, int selector) {
    std::lock_guard<std::mutex> g(model->mtx);
    if (selector == 0) selector = DDO_MDD_ENGINE_DENSE;   // ddo_mdd_create without a selector: dense first, its owner as the fallback
    auto key = std::make_pair(device, max_width * 8 + (long)(4 + (selector >> 8)));   // (features of Engine::get stay below 4)
    auto it = model->engines.find(key);
    if (it != model->engines.end()) {
        if (auto sp = it->second.lock()) return sp;
    }
    int cap = 0, threads = 0;
    if (selector == DDO_MDD_ENGINE_DENSE && max_width < 8) max_width = 8;   // (the smallest layer capacity a tier is built for)
    if (selector == DDO_MDD_ENGINE_DENSE) cap = (int)max_width, threads = 512;
    else if (selector == DDO_MDD_ENGINE_TIER0) cap = 256, threads = 64;
    else if (selector == DDO_MDD_ENGINE_TIER1) cap = 1024, threads = 128;
    else {
        set_error("ddo_mdd_create: unknown DDO_MDD_ENGINE_* selector");
        return nullptr;
    }
    if (model->kind != MODEL_MISP || (selector != DDO_MDD_ENGINE_DENSE && 2 * (long)cap > max_width)) {
        set_error("ddo_mdd_create: DDO_MDD_ENGINE_* needs a MISP model; TIER0 / TIER1 need max_width >= 512 / 2048");
        return nullptr;
    }
    std::shared_ptr<Engine> owner(new Engine());
    if (owner->init(model, device, max_width, true) != DDO_OK) return nullptr;
    auto e = create_tier(model, device, owner.get(), cap, threads);
    if (!e) return nullptr;
    e->owner_ref_ = owner;
    model->engines[key] = e;
    return e;
}

template <class T>
static int dev_alloc(std::vector<void*>& allocs, T*& ptr, size_t count) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) {
        set_error(std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
        return DDO_ERR_NO_DEVICE;
    }
    allocs.push_back(p);
    ptr = (T*)p;
    // diagnosis (tools/diag): hipMalloc does not clear; DDO_HIP_ALLOC_FILL=<byte> fills every workspace array with that byte, so that a
    // kernel that reads memory it never wrote behaves the same whatever the allocator handed out (and differently from byte to byte)
    static const char* fill = std::getenv("DDO_HIP_ALLOC_FILL");
    if (fill && hipMemset(p, (int)std::strtol(fill, nullptr, 0) & 0xFF, bytes) != hipSuccess) {
        set_error("hipMemset (DDO_HIP_ALLOC_FILL) failed");
        return DDO_ERR_NO_DEVICE;
    }
    static const bool trace = std::getenv("DDO_HIP_ALLOC_TRACE") != nullptr;   // diagnosis: which array a faulting address lies in
    if (trace) std::fprintf(stderr, "[ddo alloc] #%zu %p .. %p (%zu bytes, %zu x %zu)\n", allocs.size() - 1, p, (void*)((uint8_t*)p + bytes), bytes, count, sizeof(T));
    return DDO_OK;
}

int Engine::init(Model* model, int device, long max_width, bool want_pool, Engine* owner, int cap_width, int tier_threads, int features) {
    model_ = model;
    owner_ = owner;
    cap_width_ = owner ? cap_width : 0;
    device_ = device;
    max_width_ = max_width;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device is visible: the MDD engine has no CPU fallback");
        return DDO_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) {
        set_error("invalid device ordinal");
        return DDO_ERR_INVALID;
    }
    if (max_width < 1 || max_width > 400000) {
        set_error("max_width out of range [1, 400000]");
        return DDO_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));

    EngineParams& P = P_;
    std::memset(&P, 0, sizeof(P));
    P.n = model->n;
    P.ws = model->wsT;
    P.unit_weights = model->unit_weights ? 1 : 0;
    P.npad = (model->n + 63) / 64 * 64;
    // the terminal layer is never squashed (clean.rs:608-618): MISP ends in at most one node, a knapsack DD in up to 2W
    pooled_ = (features & ENGINE_POOLED) != 0;
    if (pooled_ && (model->kind != MODEL_MISP || owner || (features & ENGINE_KEEP_LAYERS))) {
        set_error("Pooled decision diagrams: MISP models only (Problem::is_impacted_by, misp/main.rs:145-147); their cache is a ddo_cache handed to compile, not kept layers");
        return DDO_ERR_UNSUPPORTED;
    }
    // Pooled: a slot holds a POOL -- every node that waits for a variable that impacts it -- which the width does not bound (only
    // the layers are squashed).  It is sized for what the LDS dedup table admits: 32 768 entries at 7/8 full, 14 000 work-list entries.
    long pool_nodes = 14000;
    if (const char* env = std::getenv("DDO_HIP_POOLED_NODES")) pool_nodes = std::max(64L, std::min(30000L, std::atol(env)));
    const long slot_width = pooled_ ? std::max(pool_nodes, max_width) : owner ? (long)cap_width : max_width;   // what the node slots are sized for
    P.capN = model->kind == MODEL_TSPTW ? std::max((int)slot_width, model->n) + 2      // layers are squashed to the width, the last one has one child per node
             : model->kind != MODEL_MISP ? 2 * (int)slot_width + 3 : (int)slot_width + 2;
    P.fan = model->kind == MODEL_TSPTW ? model->n : 2;
    P.dbits = model->dbits;
    P.capC1 = P.fan * P.capN + 1;
    P.max_layers = model->n + 2;
    int tc = 1024;
    while (tc * 2 < P.capC1 * 3 || tc < 2 * P.capN) tc <<= 1;
    P.table_cap = tc;
    threads_ = max_width >= 2048 ? 1024 : 256;
    dense_ = owner && (long)cap_width == max_width && tier_threads == 512;
    if (owner) threads_ = tier_threads;
    else if (pooled_) threads_ = 512;
    else if (const char* env = std::getenv("DDO_HIP_THREADS")) {
        int t = std::atoi(env);
        if (t >= 256 && t <= 1024 && t % 64 == 0) threads_ = t;
    }
    const size_t lds_max = 160 * 1024;
    size_t lds_with_table = dd_lds_bytes(P.table_cap, P.npad, threads_);
    table_lds_ = lds_with_table <= lds_max;
    lds_bytes_ = table_lds_ ? lds_with_table : dd_lds_bytes(0, P.npad, threads_);
    P.table_in_lds = table_lds_ ? 1 : 0;
    P.tw_lds = 0;
    if (model->kind == MODEL_TSPTW && !std::getenv("DDO_HIP_TW_GLOBAL")) {   // the TSPTW tables next to the shared block when LDS has the room
        const size_t with_tw = dd_lds_bytes(table_lds_ ? P.table_cap : 0, P.npad, threads_, tw_lds_words(model->n));
        if (with_tw <= lds_max) {
            P.tw_lds = 1;
            lds_bytes_ = with_tw;
        }
    }
    // ---- engine 2 (in-place layers) when its LDS footprint fits and values fit the packed 21-bit key
    P.capS = 2 * (int)slot_width + 8;
    P.tier = dense_ ? 2 : owner ? 1 : 0;
    P.hist_bins = dense_ ? 256 : owner ? 64 : 2048;   // a capacity tier never squashes: the histogram / tie-break area of LDS is not needed
    P.capW = P.capN;
    int t2 = 1024;
    while (t2 < 3 * P.capW) t2 <<= 1;
    P.tab2_cap = t2;
    if (dense_) {   // the largest table that leaves room for two workgroups per CU
        const char* env = std::getenv("DDO_HIP_DENSE_TABLE");
        P.tab2_cap = std::min(t2, env ? std::max(64, std::atoi(env)) : 16384);
    }
    // A WIDE capacity tier (layers of up to some thousand nodes, 256 threads): decision diagrams that are too wide for the
    // narrow tiers but far from the full width spend their layers waiting on memory round trips with most of a 512-thread
    // workgroup idle -- four of them per CU instead of two.  Like the dense tier it gets a table smaller than 3 x its layer
    // capacity (a layer whose nodes and YES-children would fill more than 7/8 of it hands the DD up) and runs the 4-waves-per-SIMD
    // build of the kernel; like every capacity tier it never squashes.
    mid_ = owner && !dense_ && tier_threads == 256 && cap_width >= 2048;
    if (mid_) P.tab2_cap = std::min(t2, 8192);
    // A 128-thread capacity tier (layers of up to 1 024 nodes) is held to five workgroups per CU by its LDS -- 29.8 KB, more than half of
    // it the 4 096-entry dedup table -- where its registers admit six (3 waves per SIMD).  A narrow decision diagram is all latency: what
    // a CU delivers follows the number it overlaps.  With 2 048 entries (a layer whose nodes and YES-children would fill more than 7/8
    // of them hands the DD up, as in the dense tier: about n > 1 500 at the usual share of branching nodes, i.e. never below the layer
    // capacity) a workgroup takes 21.6 KB.  DDO_HIP_TIER1_TABLE=4096 restores the full table.
    if (owner && !dense_ && !mid_ && tier_threads == 64) {   // experiment knob: the one-wave tier's table (default: 3 x its layer capacity)
        if (const char* env = std::getenv("DDO_HIP_TIER0_TABLE")) P.tab2_cap = std::min(t2, std::max(512, std::atoi(env)));
    }
    if (owner && !dense_ && !mid_ && tier_threads == 128) {
        const char* env = std::getenv("DDO_HIP_TIER1_TABLE");
        P.tab2_cap = std::min(t2, env ? std::max(1024, std::atoi(env)) : 2048);
    }
    if (pooled_) P.tab2_cap = std::min(t2, 32768);   // (a pool that would fill more than 7/8 of it ends with a capacity error)
    if (mid_ || dense_ || pooled_) P.capS = std::min(P.capS, (int)((long)P.tab2_cap * 7 / 8) + 16);
    if (pooled_) P.hist_bins = 256;                  // 8-bit select digits: the LDS goes to the table and the bitmaps
    long long neg = 0;
    for (int i = 0; i < model->n; ++i)
        if (model->weight[i] < 0) neg += model->weight[i];
    P.vbase_off = (int32_t)neg;
    P.phase_clocks = std::getenv("DDO_HIP_STATS") ? 1 : 0;
    P.lex_cap = 1024;
    if (const char* env = std::getenv("DDO_HIP_LEX_CAP")) P.lex_cap = std::max(1, std::min(1024, std::atoi(env)));   // tests: force the radix path
    if (!dense_) P.lex_cap = std::max(1, std::min(P.lex_cap, pooled_ ? 512 : P.hist_bins / 2));   // the tie-break keys (8 bytes each) share the histogram area
    if (dense_) {
        // A dense-tier DD whose layer would hold more than 7/8 of the table's entries is handed up, so no layer ever needs more
        // node slots than that: fewer slots = smaller LDS bitmaps, and the room goes to the tie-break keys -- ties of a few
        // hundred nodes are the rule at width 10 000, and ranking them out of LDS costs a fraction of the radix rounds through
        // HBM scratch (round 3: 95 of the 320 kcycles a squashed layer spends on its squash).
        const char* lenv = std::getenv("DDO_HIP_LEX_CAP");
        int lc = lenv ? P.lex_cap : 1024;
        while (lc > 128 && dd2_lds_bytes(P.capS, P.tab2_cap, P.npad, threads_, false, P.hist_bins, lc, model->wsT) > lds_max / 2) lc -= 32;
        P.lex_cap = std::max(1, std::min(lc, lenv ? P.lex_cap : 1024));
    }
    P.ev_cap = ((uint64_t)P.max_layers * (uint64_t)(5 * P.capW + 16) + 2ull * P.capW + 64 + 3) & ~3ull;   // 16-byte event records stay aligned per slot
    engine_kind_ = 2;
    if (const char* env = std::getenv("DDO_HIP_ENGINE")) engine_kind_ = std::atoi(env) == 1 ? 1 : 2;
    size_t lds2 = dd2_lds_bytes(P.capS, P.tab2_cap, P.npad, threads_, true, P.hist_bins, P.lex_cap, model->wsT);
    const size_t lds2g = dd2_lds_bytes(P.capS, P.tab2_cap, P.npad, threads_, false, P.hist_bins, P.lex_cap, model->wsT);
    // the dedup table always lives in LDS; the ranking keys join it when both fit, else they stay in HBM (L2-hot)
    keys_global_ = lds2 > (dense_ ? lds_max / 2 : mid_ ? lds_max / 4 : lds_max);   // (room for 2 resp. 4 workgroups per CU)
    if (const char* env = std::getenv("DDO_HIP_KEYS_GLOBAL")) keys_global_ = std::atoi(env) != 0;
    if (keys_global_) lds2 = lds2g;
    if (lds2 > lds_max || model->weight_abs_sum >= (1 << 20) || P.capS >= 65535 || model->n > 2047) engine_kind_ = 1;
    if (model->kind != MODEL_MISP) engine_kind_ = 1;   // scalar-state models run on the layer-rebuilding engine
    if (features & ENGINE_KEEP_LAYERS) engine_kind_ = 1;   // frontier cut-set / thresholds / cache need every layer of the DD
    if (pooled_ && engine_kind_ != 2) {
        set_error("Pooled decision diagrams: the model does not fit the in-place engine (LDS, value range or variables)");
        return DDO_ERR_UNSUPPORTED;
    }
    P.tmode = (features & ENGINE_KEEP_LAYERS) ? 1 : 0;
    // a kept layer holds its surviving nodes AND the cache-pruned / dominated ones (their thresholds flow to the parents):
    // up to every distinct child of the layer above
    P.lstride = P.tmode ? P.capC1 + 1 : P.capN;
    if (owner && engine_kind_ != 2) {
        set_error("Engine::create_tier: the tier does not fit the in-place engine");
        return DDO_ERR_UNSUPPORTED;
    }
    if (engine_kind_ == 2) lds_bytes_ = lds2;
    if (engine_kind_ == 2 && owner && !dense_ && !mid_ && P.capW <= 512 && !std::getenv("DDO_HIP_LISTS_GLOBAL")) {
        P.lists_in_lds = 1;
        lds_bytes_ += ((size_t)4 * P.capW + 15) & ~(size_t)15;
    }
    if (dense_ && (lds_bytes_ > lds_max / 2 || (long)P.tab2_cap * 7 / 8 < P.capW + 8)) {   // (the sweep alone inserts up to capW nodes)
        set_error("Engine::create_tier: the dense tier does not fit two workgroups per CU at this width");
        return DDO_ERR_UNSUPPORTED;
    }

    // ---- how many DDs in flight: residency of the kernel, then HBM
    int blocks_per_cu = (int)std::min<size_t>(lds_max / lds_bytes_, (size_t)(2048 / threads_));
    blocks_per_cu = std::max(1, std::min(blocks_per_cu, 8));
    if (dense_) blocks_per_cu = std::max(1, (int)std::min<size_t>(lds_max / lds_bytes_, 2));   // 2 x 8 waves at 128 VGPRs
    else if (mid_) blocks_per_cu = std::max(1, (int)std::min<size_t>(lds_max / lds_bytes_, 4));   // 4 x 4 waves at 128 VGPRs
    else if (owner)   // tier kernel: 3 waves per SIMD = 12 waves per CU (kernels_inplace_tier.hip)
        blocks_per_cu = std::max(1, (int)std::min<size_t>(lds_max / lds_bytes_, (size_t)(4 * tier_waves_per_simd() / (threads_ / 64))));
    int nslots = prop.multiProcessorCount * blocks_per_cu;
    if (P.tmode) nslots = std::min(nslots, 64);   // every layer of every DD in flight is kept: hundreds of MB per slot at large widths
    if (const char* env = std::getenv(owner ? "DDO_HIP_TIER_SLOTS" : "DDO_HIP_SLOTS")) {
        int s = std::atoi(env);
        if (s > 0) nslots = s;
    }
    const size_t capC1 = P.capC1, capN = P.capN, ml = P.max_layers, wsT = model->wsT;
    size_t per_slot = engine_kind_ == 1
                          ? 2 * wsT * capC1 * 8 + 2 * capC1 * 8 + 2 * capC1 * 4 * 2 + 2 * capN * 4 + capN * 4 + capC1 * 4 + capC1 +
                                ml * capN * 4 + 2 * ml * (size_t)P.fan * capN * 4 + ml * 4 * 4 + wsT * capN * 8 + capN * 8 +
                                (table_lds_ ? 0 : (size_t)P.table_cap * 4)
                          : wsT * (size_t)P.capS * 8 + (size_t)P.capS * 8 * (((wsT + 8) / 8) * 8 + ((wsT + 7) / 8) * 8) +
                                (size_t)P.capS * 12 + capN * 4 + P.ev_cap * 4 + P.ev_cap * 2 + ml * 8 * 4 + ml * 4 * 4 + capN * 4 + 2 * wsT * capN * 8 + capN * 8;
    if (P.tmode) {
        // Kept layers at fixed strides cost ml * lstride node records per slot -- (nb_vars + 2) * nb_vars^3 of them under TsptwWidth,
        // 25 GB per slot at 126 nodes (round 3: four DDs at a time).  Beyond 1 GB per slot the kept layers and the arc arrays become
        // per-slot POOLS that a DD fills layer by layer with what it really holds (run_dd: lbase / abase); a DD that outgrows its
        // pools ends with a capacity error.  DDO_HIP_LPOOL_M / DDO_HIP_APOOL_M: pool sizes in millions of records.
        const size_t node_b = wsT * 8 + 6 * 4;
        const size_t fixed_nodes = ml * (size_t)P.lstride, fixed_arcs = ml * (size_t)P.fan * capN;
        if (model->kind == MODEL_TSPTW && fixed_nodes * node_b > (1ull << 30) && !std::getenv("DDO_HIP_FIXED_LAYERS")) {
            size_t ln = std::min<size_t>(fixed_nodes, 8ull << 20), an = std::min<size_t>(fixed_arcs, 96ull << 20);
            if (const char* env = std::getenv("DDO_HIP_LPOOL_M")) ln = std::min<size_t>(fixed_nodes, (size_t)std::max(1, std::atoi(env)) << 20);
            if (const char* env = std::getenv("DDO_HIP_APOOL_M")) an = std::min<size_t>(fixed_arcs, (size_t)std::max(1, std::atoi(env)) << 20);
            P.lpool_nodes = ln;
            P.apool_arcs = an;
            per_slot -= 2 * ml * (size_t)P.fan * capN * 4 + ml * capN * 4;   // (the fixed arc arrays and ninfo counted above)
            per_slot += ln * node_b + an * 8 + capC1 * 8 + 2 * (ml + 1) * 8;
            nslots = std::min(nslots, (int)std::max<size_t>(4, (160ull << 30) / per_slot));
        } else {
            per_slot += fixed_nodes * node_b + capC1 * 8;
            nslots = std::min(nslots, (int)std::max<size_t>(4, (8ull << 30) / per_slot));   // D-ary models: up to a GB per slot
        }
    }
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    // output arena: restricted/relaxed results of one batch (cut-set rows + paths), sized by the slots that can run
    size_t arena_mb = std::min<size_t>(1024, std::max<size_t>(64, ((size_t)nslots * (size_t)P.capN * 256) >> 20));
    if (owner && !dense_) arena_mb = std::min<size_t>(arena_mb, 256);
    if (const char* env = std::getenv("DDO_HIP_ARENA_MB")) arena_mb = (size_t)std::max(16, std::atoi(env));
    arena_cap_ = arena_mb << 20;
    if (const char* env = std::getenv("DDO_HIP_ARENA_KB")) arena_cap_ = (size_t)std::max(1, std::atoi(env)) << 10;   // tests: force the overflow paths
    size_t budget = free_b > (arena_cap_ + (2ull << 30)) ? (size_t)((free_b - arena_cap_ - (1ull << 30)) * 0.8) : 0;
    if (budget / per_slot < (size_t)nslots) nslots = (int)(budget / per_slot);
    if (nslots < 1) {
        set_error("not enough device memory for one decision-diagram slot");
        return DDO_ERR_CAPACITY;
    }
    nslots_ = nslots;
    P.nslots = nslots;
    const size_t S = nslots;

    // ---- model tables
    std::vector<uint64_t> adjT((size_t)model->n * wsT, 0);
    if (model->kind == MODEL_MISP)
        for (int i = 0; i < model->n; ++i)
            for (int k = 0; k < model->ws; ++k) adjT[(size_t)i * wsT + k] = model->adj[(size_t)i * model->ws + k];
    std::vector<int32_t> w32(model->n);
    for (int i = 0; i < model->n; ++i) w32[i] = (int32_t)model->weight[i];
    uint64_t* d_adj = nullptr;
    int32_t* d_w = nullptr;
    int rc;
    if ((rc = dev_alloc(allocs_, d_adj, adjT.size()))) return rc;
    if ((rc = dev_alloc(allocs_, d_w, w32.size()))) return rc;
    HIP_TRY(hipMemcpy(d_adj, adjT.data(), adjT.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_w, w32.data(), w32.size() * 4, hipMemcpyHostToDevice));
    P.adj = d_adj;
    P.weight = d_w;
    P.model_kind = model->kind;
    if (model->kind == MODEL_KNAPSACK) {
        std::vector<int32_t> kw(model->n);
        for (int i = 0; i < model->n; ++i) kw[i] = (int32_t)model->kp_weight[i];
        int32_t *d_kw = nullptr, *d_ko = nullptr;
        if ((rc = dev_alloc(allocs_, d_kw, kw.size()))) return rc;
        if ((rc = dev_alloc(allocs_, d_ko, kw.size()))) return rc;
        HIP_TRY(hipMemcpy(d_kw, kw.data(), kw.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_ko, model->kp_order.data(), kw.size() * 4, hipMemcpyHostToDevice));
        P.kp_weight = d_kw;
        P.kp_order = d_ko;
    }
    if (model->kind == MODEL_MCP) {
        int32_t *d_g = nullptr, *d_e = nullptr, *d_k = nullptr;
        if ((rc = dev_alloc(allocs_, d_g, model->vgraph.size()))) return rc;
        if ((rc = dev_alloc(allocs_, d_e, model->vest.size()))) return rc;
        if ((rc = dev_alloc(allocs_, d_k, model->vnk.size()))) return rc;
        HIP_TRY(hipMemcpy(d_g, model->vgraph.data(), model->vgraph.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_e, model->vest.data(), model->vest.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_k, model->vnk.data(), model->vnk.size() * 4, hipMemcpyHostToDevice));
        P.vgraph = d_g;
        P.vest = d_e;
        P.vnk = d_k;
        P.vr = (int32_t)model->initial_value;
    }
    if (model->kind == MODEL_TSPTW) {
        auto upi = [&](const std::vector<int32_t>& v, const int32_t*& dst) -> int {
            int32_t* d = nullptr;
            int r = dev_alloc(allocs_, d, v.size());
            if (r) return r;
            if (hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return DDO_ERR_INTERNAL;
            dst = d;
            return DDO_OK;
        };
        if ((rc = upi(model->tw_dist, P.tw_dist)) || (rc = upi(model->tw_early, P.tw_early)) || (rc = upi(model->tw_late, P.tw_late)) ||
            (rc = upi(model->tw_cheap, P.tw_cheap)) || (rc = upi(model->tw_order, P.tw_order)))
            return rc;
    }
    if (model->kind == MODEL_MAX2SAT) {
        auto up = [&](const std::vector<int32_t>& v, const int32_t*& dst) -> int {
            int32_t* d = nullptr;
            int r = dev_alloc(allocs_, d, v.size());
            if (r) return r;
            if (hipMemcpy(d, v.data(), v.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return DDO_ERR_INTERNAL;
            dst = d;
            return DDO_OK;
        };
        if ((rc = up(model->m2_w[0], P.m2_wtt)) || (rc = up(model->m2_w[1], P.m2_wtf)) || (rc = up(model->m2_w[2], P.m2_wft)) ||
            (rc = up(model->m2_w[3], P.m2_wff)) || (rc = up(model->m2_order, P.m2_order)) ||
            (rc = up(model->m2_rankpos, P.m2_rankpos)) || (rc = up(model->vest, P.vest)) || (rc = up(model->vnk, P.vnk)))
            return rc;
        P.vr = (int32_t)model->initial_value;
    }

    // ---- workspace
    if (engine_kind_ == 1) {
        if ((rc = dev_alloc(allocs_, P.cstate, S * 2 * wsT * capC1))) return rc;
        if ((rc = dev_alloc(allocs_, P.ckey, S * 2 * capC1))) return rc;
        if ((rc = dev_alloc(allocs_, P.cpop, S * 2 * capC1))) return rc;
        if ((rc = dev_alloc(allocs_, P.cflags, S * 2 * capC1))) return rc;
        if ((rc = dev_alloc(allocs_, P.ctarget, S * (size_t)P.fan * capN))) return rc;
        if ((rc = dev_alloc(allocs_, P.keep, S * (P.tmode ? capC1 : capN)))) return rc;
        if ((rc = dev_alloc(allocs_, P.posmap, S * capC1))) return rc;
        if ((rc = dev_alloc(allocs_, P.cls, S * capC1))) return rc;
        if ((rc = dev_alloc(allocs_, P.ninfo, S * (P.lpool_nodes ? (size_t)P.lpool_nodes : ml * (size_t)P.lstride)))) return rc;
        if (P.tmode) {
            const size_t lsz = S * (P.lpool_nodes ? (size_t)P.lpool_nodes : ml * (size_t)P.lstride);
            if (P.lpool_nodes) {
                if ((rc = dev_alloc(allocs_, P.lbase, S * (ml + 1)))) return rc;
                if ((rc = dev_alloc(allocs_, P.abase, S * (ml + 1)))) return rc;
            }
            if ((rc = dev_alloc(allocs_, P.lstate, lsz * wsT))) return rc;
            if ((rc = dev_alloc(allocs_, P.lval, lsz))) return rc;
            if ((rc = dev_alloc(allocs_, P.lrub, lsz))) return rc;
            if ((rc = dev_alloc(allocs_, P.lvb, lsz))) return rc;
            if ((rc = dev_alloc(allocs_, P.lth, lsz + S * capC1))) return rc;
        }
        if ((rc = dev_alloc(allocs_, P.arct, S * (P.lpool_nodes ? (size_t)P.apool_arcs : ml * (size_t)P.fan * capN)))) return rc;
        if ((rc = dev_alloc(allocs_, P.arcc, S * (P.lpool_nodes ? (size_t)P.apool_arcs : ml * (size_t)P.fan * capN)))) return rc;
        if ((rc = dev_alloc(allocs_, P.lddelta, S * ml))) return rc;
    } else {
        const size_t capS = P.capS, capW = P.capW;
        const size_t RW = ((wsT + 1 + 7) / 8) * 8;
#if defined(DDO_WORD_MAJOR)
        if ((rc = dev_alloc(allocs_, P.s_state, S * wsT * capS))) return rc;   // experiment: word-major copy of the states for the work-list sweep
#endif
        if ((rc = dev_alloc(allocs_, P.s_rec, S * capS * RW))) return rc;
        P.keys_global = keys_global_ ? 1 : 0;
        if ((rc = dev_alloc(allocs_, P.s_ptree, S * (size_t)(P.ev_cap / 4)))) return rc;
        P.s_pvr = nullptr;
        // Pooled behind a SimpleCache (Par / SeqCachingSolverPooled): (value_top, rough upper bound) per event record for the threshold pass
        if (pooled_ && (rc = dev_alloc(allocs_, P.s_pvr, S * (size_t)(P.ev_cap / 4)))) return rc;
        P.s_pst = nullptr;
        P.pst_cap = 0;
        if (pooled_) {   // the states of the first pst_cap event records of a DD (1 M by default: far beyond what a pool of 28 000 nodes expands)
            long m = 1;
            if (const char* env = std::getenv("DDO_HIP_POOLED_STATES_M")) m = std::max(1L, std::min(64L, std::atol(env)));
            P.pst_cap = (uint32_t)std::min<uint64_t>(P.ev_cap / 4, (uint64_t)m << 20);
            if ((rc = dev_alloc(allocs_, P.s_pst, S * (size_t)P.pst_cap * wsT))) return rc;
        }
        if ((rc = dev_alloc(allocs_, P.s_hash, S * capS))) return rc;
        if ((rc = dev_alloc(allocs_, P.s_wl, S * 2 * capW))) return rc;
        if ((rc = dev_alloc(allocs_, P.s_ev, S * P.ev_cap))) return rc;
        if ((rc = dev_alloc(allocs_, P.s_evoff, S * ml * 8))) return rc;
        if ((rc = dev_alloc(allocs_, P.s_cs_slot, S * capW))) return rc;
        if ((rc = dev_alloc(allocs_, P.s_cs_path, S * wsT * capW))) return rc;
    }
    if ((rc = dev_alloc(allocs_, P.nlayer, S * ml))) return rc;
    if ((rc = dev_alloc(allocs_, P.lntot, S * ml))) return rc;
    if ((rc = dev_alloc(allocs_, P.lvar, S * ml))) return rc;
    if ((rc = dev_alloc(allocs_, P.ldup, S * ml * 2))) return rc;
    if ((rc = dev_alloc(allocs_, P.cs_state, S * wsT * capN))) return rc;
    if ((rc = dev_alloc(allocs_, P.cs_value, S * capN))) return rc;
    if ((rc = dev_alloc(allocs_, P.cs_pop, S * capN))) return rc;
    if (engine_kind_ == 1 && !table_lds_) {
        if ((rc = dev_alloc(allocs_, P.gtable, S * (size_t)P.table_cap))) return rc;
    }
    for (int k = 0; k < 2; ++k) {
        if ((rc = dev_alloc(allocs_, io_[k].d_cnt, 64))) return rc;
        HIP_TRY(hipMemset(io_[k].d_cnt, 0, 64));
    }
    P.arena_cap = arena_cap_;
    uint8_t* cnt = nullptr;
    if ((rc = dev_alloc(allocs_, cnt, 64))) return rc;
    d_counters_ = cnt;
    HIP_TRY(hipMemset(cnt, 0, 64));
    P.cutoff_flag = (const int32_t*)(cnt + 16);
    P.pool_head = (unsigned long long*)(cnt + 32);
    if (owner) {   // the tier allocates its cut-set blocks in the owner's node pool and obeys the owner's cutoff flag
        P.cutoff_flag = owner->P_.cutoff_flag;
        P.pool_head = owner->P_.pool_head;
        P.pool = owner->P_.pool;
        P.pool_cap = owner->P_.pool_cap;
    }
    if (engine_kind_ == 2 && want_pool) {   // node pool: whatever HBM is left (capped), for cut-sets that stay on the device
        size_t free2 = 0, total2 = 0;
        HIP_TRY(hipMemGetInfo(&free2, &total2));
        size_t want = (size_t)64 << 30;
        if (const char* env = std::getenv("DDO_HIP_POOL_GB")) want = (size_t)std::max(0, std::atoi(env)) << 30;
        size_t avail = free2 > ((size_t)6 << 30) ? free2 - ((size_t)6 << 30) : 0;
        size_t pool_bytes = std::min(want, avail);
        const char* vmm = std::getenv("DDO_HIP_POOL_VMM");
        if (!(vmm && std::atoi(vmm) == 0) && avail >= ((size_t)1 << 30)) {   // DDO_HIP_POOL_VMM=0: one fixed hipMalloc instead
            // reserve the address range (everything that is free unless DDO_HIP_POOL_GB caps it), map the first chunks
            hipMemAllocationProp prop{};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = device;
            size_t gran = 0;
            if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) == hipSuccess && gran > 0) {
                const size_t chunk = ((((size_t)2 << 30) + gran - 1) / gran) * gran;
                const size_t cap = std::getenv("DDO_HIP_POOL_GB") ? std::min(want, avail) : avail;
                const size_t reserve = cap / chunk * chunk;
                void* base = nullptr;
                if (reserve >= chunk && hipMemAddressReserve(&base, reserve, gran, nullptr, 0) == hipSuccess && base) {
                    vm_base_ = (uint8_t*)base;
                    vm_reserved_ = reserve;
                    vm_chunk_ = chunk;
                    if (pool_grow(std::min(reserve, 4 * chunk)) == DDO_OK && vm_mapped_ >= chunk) {
                        P.pool = vm_base_;
                        P.pool_cap = vm_mapped_;
                    } else {
                        pool_release();
                    }
                }
            }
            (void)hipGetLastError();   // a failed attempt falls back to the plain allocation below
        }
        if (!P.pool && pool_bytes >= ((size_t)1 << 28)) {
            uint8_t* pool = nullptr;
            if (dev_alloc(allocs_, pool, pool_bytes) == DDO_OK) {
                P.pool = pool;
                P.pool_cap = pool_bytes;
            }
        }
    }

    {   // The output arena of the first buffer set -- pinned host memory the kernels write into -- is allocated here, with the rest of
        // the workspace: pinning costs about 0.2 ms per MB, and left to the first launch it was part of every search's wall time
        // (round 5, DDO_HIP_TIMES: 13 of the 21 ms of an MCP n = 30 search, 87 of the 239 ms of config C3).
        // The layer-rebuilding engine and the capacity tiers get the SECOND set's arena and both sets' input / result buffers here as
        // well: launches alternate between the two sets, and what the first use of the second set allocated was still inside every
        // search (DDO_HIP_TIMES: 45 of the 131 ms of config C3, 5-7 of the 10 ms of an MCP n = 30 search).  The full-width in-place
        // engine keeps its second arena for the first overlapped launch: next to capacity tiers it may never see one.
        const int eager = (engine_kind_ == 1 || owner) ? 2 : 1;
        for (int k = 0; k < eager; ++k) {
            void* hp = nullptr;
            HIP_TRY(hipHostMalloc(&hp, arena_cap_, hipHostMallocDefault));
            io_[k].h_arena = (uint8_t*)hp;
            io_[k].h_arena_cap = arena_cap_;
        }
        for (int k = 0; k < 2; ++k) {
            const int rc = io_reserve(io_[k], std::min(nslots_, 1024));
            if (rc != DDO_OK) return rc;
        }
    }
    hipStream_t st, st2;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    stream_ = st;
    HIP_TRY(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    copy_stream_ = st2;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    ev0_ = e0;
    ev1_ = e1;

    kernel_fn fn = pooled_ ? pick_kernel2_pooled(model->wsT)
                   : (dense_ || mid_) ? pick_kernel2_dense(model->wsT)
                   : owner ? pick_kernel2_tier(model->wsT)
                   : engine_kind_ == 2 ? pick_kernel2(model->wsT, threads_)
                                       : pick_kernel(model->wsT, table_lds_);
    if (!fn) {
        set_error("unsupported state width");
        return DDO_ERR_UNSUPPORTED;
    }
    kernel_ = (void*)fn;
    {   // several engines (tiers, widths) share a kernel: the attribute must cover the largest of them
        static std::mutex attr_mtx;
        static std::map<std::pair<int, const void*>, size_t> attr_max;
        std::lock_guard<std::mutex> g(attr_mtx);
        size_t& m = attr_max[{device, (const void*)fn}];
        if (lds_bytes_ > m) {
            HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_));
            m = lds_bytes_;
        }
    }
    return DDO_OK;
}

Engine::~Engine() {
    if (g_times_on && g_launch_n.load() > 0) {
        auto sec = [](int k) { return (double)g_launch_ns[k].exchange(0) * 1e-9; };
        const unsigned long long nl = g_launch_n.exchange(0);
        const double t0 = sec(0), t1 = sec(1), t2 = sec(2), t3 = sec(3), t4 = sec(4), t5 = sec(5), t6 = sec(6), t7 = sec(7);
        std::fprintf(stderr, "[ddo times] launch() host s over %llu launches: checks %.3f, inputs %.3f, launch order %.3f, parameters + rewind %.3f, first event %.3f, pool growth %.3f, kernel %.3f, result copies %.3f\n",
                     nl, t0, t1, t2, t6, t7, t3, t4, t5);
    }
    if (g_times_on && cq_launches_.load() > 0) {
        auto sec = [](int k) { return (double)g_cq_ns[k].exchange(0) * 1e-9; };
        const double a = sec(0), b = sec(1), c = sec(2), d = sec(3), e = sec(4);
        std::fprintf(stderr, "[ddo times] combining layer, %llu launches of %llu compiles: window %.3f s, staging + launch %.3f, kernel wait %.3f, hand-out %.3f, no leader %.3f\n",
                     (unsigned long long)cq_launches_.load(), (unsigned long long)cq_requests_.load(), a, b, c, d, e);
    }
    if (device_ >= 0) (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize((hipStream_t)stream_);
    for (void* p : allocs_) (void)hipFree(p);
    pool_release();
    for (int k = 0; k < 2; ++k) {
        if (io_[k].d_inputs) (void)hipFree(io_[k].d_inputs);
        if (io_[k].d_results) (void)hipFree(io_[k].d_results);
        if (io_[k].h_arena) (void)hipHostFree(io_[k].h_arena);
        if (io_[k].h_results) (void)hipHostFree(io_[k].h_results);   // (h_head and h_inputs live in the same allocation)
    }
    if (copy_stream_) (void)hipStreamDestroy((hipStream_t)copy_stream_);
    if (ev0_) (void)hipEventDestroy((hipEvent_t)ev0_);
    if (ev1_) (void)hipEventDestroy((hipEvent_t)ev1_);
    if (stream_) (void)hipStreamDestroy((hipStream_t)stream_);
}

void Engine::set_cutoff(bool on) {
    std::lock_guard<std::mutex> g(cut_mtx_);   // (not mtx_: wait() holds that while the driver sleeps on a long kernel -- ADVICE r04)
    int32_t v = on ? 1 : 0;
    (void)hipSetDevice(device_);
    (void)hipMemcpy((uint8_t*)d_counters_ + 16, &v, 4, hipMemcpyHostToDevice);
}

int Engine::pool_reset() {
    std::lock_guard<std::mutex> g(mtx_);
    HIP_TRY(hipSetDevice(device_));
    HIP_TRY(hipMemset((uint8_t*)d_counters_ + 32, 0, 8));   // pool head
    pool_head_bound_ = 0;
    pool_unfetched_worst_ = 0;
    return DDO_OK;
}

/// maps physical chunks until `target` bytes of the reserved range are backed (or memory runs out: the pool then stays
/// at its size and the kernel reports ST_ERR_CAPACITY when it is full, as with the fixed pool)
/// The pool grows OFF the critical path: a mapper thread keeps two chunks mapped beyond what the launches can take (round 4 mapped
/// them on demand inside launch(): 2.5 s of the 76 s of a brock400_1 proof, 60 ms per 2 GB chunk).  launch() still maps itself when
/// the thread has not caught up.
void Engine::pool_want(size_t bytes) {
    if (!vm_base_) return;
    bytes = std::min(bytes, vm_reserved_);
    if (bytes <= vm_mapped_.load()) return;
    {
        std::lock_guard<std::mutex> g(vm_mtx_);
        if (bytes > vm_want_) vm_want_ = bytes;
        if (!vm_thread_.joinable()) {
            vm_stop_ = false;
            vm_thread_ = std::thread([this] {
                (void)hipSetDevice(device_);
                std::unique_lock<std::mutex> lk(vm_mtx_);
                for (;;) {
                    vm_cv_.wait(lk, [&] { return vm_stop_ || vm_want_ > vm_mapped_.load(); });
                    if (vm_stop_) return;
                    const size_t before = vm_mapped_.load();
                    pool_grow_locked(before + vm_chunk_);   // one chunk per turn: a launch that needs the lock waits for one chunk at most
                    if (vm_mapped_.load() == before) vm_want_ = before;   // (out of memory: stop trying)
                    lk.unlock();
                    lk.lock();
                }
            });
        }
    }
    vm_cv_.notify_one();
}

void Engine::pool_expect(int count) {
    if (!vm_base_ || count <= 0) return;
    const size_t worst = (size_t)count * (size_t)pool_block_bytes((uint32_t)P_.capW, (uint32_t)model_->wsT, (uint32_t)P_.max_layers);
    // (bounded: 24 GB ahead, and never more than a quarter of what is free right now -- several solvers may share the device, and the
    // engines created after this one size their slots by the memory they find)
    size_t target = std::min<size_t>(2 * worst + 2 * vm_chunk_, (size_t)24 << 30);
    size_t free_b = 0, total_b = 0;
    (void)hipSetDevice(device_);
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return;
    const size_t mapped = vm_mapped_.load();
    if (target > mapped && target - mapped > free_b / 4) target = mapped + free_b / 4 / vm_chunk_ * vm_chunk_;
    pool_want(target);
}

int Engine::pool_grow(size_t target) {
    std::lock_guard<std::mutex> g(vm_mtx_);
    return pool_grow_locked(target);
}

int Engine::pool_grow_locked(size_t target) {
    if (!vm_base_) return DDO_OK;
    target = std::min(target, vm_reserved_);
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device_;
    while (vm_mapped_ < target) {
        hipMemGenericAllocationHandle_t h{};
        if (hipMemCreate(&h, vm_chunk_, &prop, 0) != hipSuccess) break;
        if (hipMemMap(vm_base_ + vm_mapped_, vm_chunk_, 0, h, 0) != hipSuccess) {
            (void)hipMemRelease(h);
            break;
        }
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (hipMemSetAccess(vm_base_ + vm_mapped_, vm_chunk_, &acc, 1) != hipSuccess) {
            (void)hipMemUnmap(vm_base_ + vm_mapped_, vm_chunk_);
            (void)hipMemRelease(h);
            break;
        }
        vm_handles_.push_back((void*)h);
        vm_mapped_ += vm_chunk_;
    }
    (void)hipGetLastError();
    return DDO_OK;
}
void Engine::pool_release() {
    if (vm_thread_.joinable()) {
        {
            std::lock_guard<std::mutex> g(vm_mtx_);
            vm_stop_ = true;
        }
        vm_cv_.notify_one();
        vm_thread_.join();
    }
    if (!vm_base_) return;
    for (size_t i = 0; i < vm_handles_.size(); ++i) {
        (void)hipMemUnmap(vm_base_ + i * vm_chunk_, vm_chunk_);
        (void)hipMemRelease((hipMemGenericAllocationHandle_t)vm_handles_[i]);
    }
    vm_handles_.clear();
    (void)hipMemAddressFree(vm_base_, vm_reserved_);
    vm_base_ = nullptr;
    vm_reserved_ = vm_mapped_ = 0;
}
int Engine::read_pool_head(uint64_t* out) {
    std::lock_guard<std::mutex> g(mtx_);
    HIP_TRY(hipSetDevice(device_));
    unsigned long long h = 0;
    HIP_TRY(hipMemcpy(&h, (uint8_t*)d_counters_ + 32, 8, hipMemcpyDeviceToHost));
    *out = h;
    return DDO_OK;
}
int Engine::pool_append(const void* data, size_t bytes, uint64_t* off) {
    uint64_t head = 0;
    int rc = read_pool_head(&head);
    if (rc != DDO_OK) return rc;
    std::lock_guard<std::mutex> g(mtx_);
    head = (head + 63) & ~63ull;
    const size_t padded = (bytes + 63) & ~(size_t)63;
    if (vm_base_ && head + padded > vm_mapped_) {
        pool_grow(head + padded + vm_chunk_);
        P_.pool_cap = vm_mapped_;
    }
    if (!P_.pool || head + padded > P_.pool_cap) {
        set_error("pool_append: the device node pool is full");
        return DDO_ERR_CAPACITY;
    }
    HIP_TRY(hipSetDevice(device_));
    HIP_TRY(hipMemcpy(P_.pool + head, data, bytes, hipMemcpyHostToDevice));
    const unsigned long long nh = head + padded;
    HIP_TRY(hipMemcpy((void*)P_.pool_head, &nh, 8, hipMemcpyHostToDevice));
    pool_head_bound_ = std::max<uint64_t>(pool_head_bound_, nh);
    *off = head;
    return DDO_OK;
}
int Engine::read_pool(uint64_t off, void* dst, size_t bytes) {
    std::lock_guard<std::mutex> g(mtx_);
    if (!P_.pool || off + bytes > P_.pool_cap) {
        set_error("read_pool: out of range");
        return DDO_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(device_));
    HIP_TRY(hipMemcpy(dst, P_.pool + off, bytes, hipMemcpyDeviceToHost));
    return DDO_OK;
}

int Engine::read_pool_strided(uint64_t off, uint64_t* dst, size_t count, size_t stride_bytes) {
    std::lock_guard<std::mutex> g(mtx_);
    if (count == 0) return DDO_OK;
    if (!P_.pool || off + (count - 1) * stride_bytes + 8 > P_.pool_cap) {
        set_error("read_pool_strided: out of range");
        return DDO_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(device_));
    HIP_TRY(hipMemcpy2D(dst, 8, P_.pool + off, stride_bytes, 8, count, hipMemcpyDeviceToHost));
    return DDO_OK;
}

DominanceTable* DominanceTable::create(const Model* model, int device, size_t capacity_per_depth) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error("ddo_dominance_create: no such HIP device (the checker lives in device memory: there is no CPU fallback)");
        return nullptr;
    }
    if (model->kind != MODEL_KNAPSACK && model->kind != MODEL_TSPTW) {
        set_error("ddo_dominance_create: the device checker covers the dominance relations of the knapsack (KPDominance) and TSPTW "
                  "(TsptwDominance) examples");
        return nullptr;
    }
    DominanceTable* t = new DominanceTable();
    t->device = device;
    if (model->kind == MODEL_TSPTW) {
        size_t cap = 1024;
        while (cap < capacity_per_depth) cap <<= 1;
        t->dkey_cap = cap;
        t->dkey_stride = 3 + 2 * tw_set_words(model->n) + 1;
        if (hipSetDevice(device) != hipSuccess || hipMalloc((void**)&t->dkey, cap * (size_t)t->dkey_stride * 8) != hipSuccess ||
            hipMalloc((void**)&t->stats, 64) != hipSuccess || t->clear() != DDO_OK) {
            set_error("ddo_dominance_create: could not allocate device memory");
            delete t;
            return nullptr;
        }
        return t;
    }
    t->cap = (uint32_t)std::max<size_t>(64, capacity_per_depth);
    t->depths = model->n + 2;
    const size_t nd = (size_t)t->depths;
    if (hipSetDevice(device) != hipSuccess || hipMalloc((void**)&t->coord, nd * t->cap * 8) != hipSuccess ||
        hipMalloc((void**)&t->value, nd * t->cap * 4) != hipSuccess || hipMalloc((void**)&t->count, nd * 4) != hipSuccess ||
        hipMalloc((void**)&t->lock, nd * 4) != hipSuccess || hipMalloc((void**)&t->stats, 64) != hipSuccess || t->clear() != DDO_OK) {
        set_error("ddo_dominance_create: could not allocate device memory");
        delete t;
        return nullptr;
    }
    return t;
}
DominanceTable::~DominanceTable() {
    (void)hipSetDevice(device);
    for (void* p : {(void*)coord, (void*)value, (void*)count, (void*)lock, (void*)stats, (void*)dkey})
        if (p) (void)hipFree(p);
}
int DominanceTable::clear() {
    HIP_TRY(hipSetDevice(device));
    if (dkey) {
        HIP_TRY(hipMemset(dkey, 0, dkey_cap * (size_t)dkey_stride * 8));
        HIP_TRY(hipMemset(stats, 0, 64));
        return DDO_OK;
    }
    HIP_TRY(hipMemset(count, 0, (size_t)depths * 4));
    HIP_TRY(hipMemset(lock, 0, (size_t)depths * 4));
    HIP_TRY(hipMemset(stats, 0, 64));
    return DDO_OK;
}

CacheTable* CacheTable::create(const Model* model, int device, size_t capacity_entries) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        set_error("ddo_cache_create: no such HIP device (the cache lives in device memory: there is no CPU fallback)");
        return nullptr;
    }
    size_t cap = 1024;
    while (cap < 2 * capacity_entries) cap <<= 1;   // (the table takes new entries up to HALF its slots: dd_thresholds.hpp, cache_update)
    CacheTable* t = new CacheTable();
    t->device = device;
    t->cap = cap;
    t->stride = 3 + model->wsT;
    if (hipSetDevice(device) != hipSuccess || hipMalloc((void**)&t->tab, cap * (size_t)t->stride * 8) != hipSuccess ||
        hipMalloc((void**)&t->stats, 64) != hipSuccess || t->clear() != DDO_OK) {
        set_error("ddo_cache_create: could not allocate the table in device memory");
        delete t;
        return nullptr;
    }
    return t;
}
CacheTable::~CacheTable() {
    (void)hipSetDevice(device);
    if (tab) (void)hipFree(tab);
    if (stats) (void)hipFree(stats);
}
int CacheTable::clear() {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipMemset(tab, 0, cap * (size_t)stride * 8));
    HIP_TRY(hipMemset(stats, 0, 64));
    return DDO_OK;
}
int CacheTable::read_stats(uint64_t* used, uint64_t* dropped) const {
    unsigned long long v[2] = {0, 0};
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(v, stats, 16, hipMemcpyDeviceToHost));
    if (used) *used = v[0];
    if (dropped) *dropped = v[1];
    return DDO_OK;
}

int Engine::grow_arena(size_t bytes) {
    std::lock_guard<std::mutex> g(mtx_);
    if (pending_ > 0 || fetch_set_ >= 0) {
        set_error("Engine::grow_arena: a launch is in flight");
        return DDO_ERR_INVALID;
    }
    if (bytes <= arena_cap_) return DDO_OK;
    HIP_TRY(hipSetDevice(device_));
    for (int k = 0; k < 2; ++k)
        if (io_[k].h_arena) {
            HIP_TRY(hipHostFree(io_[k].h_arena));
            io_[k].h_arena = nullptr;
            io_[k].h_arena_cap = 0;
        }
    arena_cap_ = bytes;
    P_.arena_cap = arena_cap_;
    return DDO_OK;
}

void Engine::decode(const DDResult& r, const uint8_t* arena, HostResult& out) const {
    out.clear();
    out.hdr = r;
    out.valid = true;
    out.pool_off = r.pool_off;
    if (r.status != ST_OK) return;
    const int ws = model_->ws, wsT = model_->wsT;
    const uint8_t* base = arena + r.arena_off;
    if (r.best_len) {
        const uint32_t* p = (const uint32_t*)(base + r.path_off);
        out.best_path.assign(p, p + r.best_len);
    }
    if (r.exact_len) {
        const uint32_t* p = (const uint32_t*)(base + r.exact_off);
        out.exact_path.assign(p, p + r.exact_len);
    }
    out.n_cutset = r.n_cutset;
    out.cs_path_len = r.cs_path_stride;
    if (r.n_cutset && r.pool_off != NO_POOL_SRC) {
        const int32_t* v = (const int32_t*)(base + r.cs_value_off);
        out.cs_value.assign(v, v + r.n_cutset);
        const int32_t* u = (const int32_t*)(base + r.cs_ub_off);
        out.cs_ub.assign(u, u + r.n_cutset);
    } else if (r.n_cutset) {
        const uint64_t* s = (const uint64_t*)(base + r.cs_state_off);
        out.cs_state.resize((size_t)r.n_cutset * ws);
        for (int i = 0; i < r.n_cutset; ++i)
            for (int k = 0; k < ws; ++k) out.cs_state[(size_t)i * ws + k] = s[(size_t)i * wsT + k];
        const int32_t* v = (const int32_t*)(base + r.cs_value_off);
        out.cs_value.assign(v, v + r.n_cutset);
        const int32_t* u = (const int32_t*)(base + r.cs_ub_off);
        out.cs_ub.assign(u, u + r.n_cutset);
        if (r.cs_lvar_off) {   // IN_PATH_BITS
            out.cs_pw = (out.cs_path_len + 63) / 64;
            const uint64_t* pb = (const uint64_t*)(base + r.cs_path_off);
            out.cs_pbits.assign(pb, pb + (size_t)r.n_cutset * out.cs_pw);
            const uint32_t* lv = (const uint32_t*)(base + r.cs_lvar_off);
            out.cs_lvar.assign(lv, lv + out.cs_path_len);
        } else {
            const uint32_t* pth = (const uint32_t*)(base + r.cs_path_off);
            out.cs_path.assign(pth, pth + (size_t)r.n_cutset * out.cs_path_len);
        }
        if (r.cs_depth_off) {
            const int32_t* dp = (const int32_t*)(base + r.cs_depth_off);
            out.cs_depth.assign(dp, dp + r.n_cutset);
        }
    }
}

bool Engine::kernel_done() {
    std::lock_guard<std::mutex> g(mtx_);
    if (pending_ <= 0) return true;
    (void)hipSetDevice(device_);
    return hipEventQuery((hipEvent_t)ev1_) != hipErrorNotReady;
}

int Engine::run_batch(const DDInput* inputs, int count, std::vector<HostResult>& results, const CacheTable* cache, const DominanceTable* dom,
                      const volatile int* const* stop_flags, int nflags) {
    results.resize((size_t)std::max(count, 0) * 2);
    if (count <= 0) return DDO_OK;
    // One engine is shared by every ddo_mdd of a (model, device, width) and has a single launch in flight: the three
    // halves below must not interleave between host threads (mdd.rs: one DecisionDiagram per worker thread,
    // parallel.rs:576-602).  The asynchronous launch()/wait()/fetch() path belongs to the lazy solver, which owns a
    // private engine.
    std::lock_guard<std::mutex> batch_guard(batch_mtx_);
    wait_decoders(-1);   // (callers of the combining layer may still read the previous launches' buffers)
    int rc = launch(inputs, count, cache, dom);
    if (rc != DDO_OK) return rc;
    bool raised = false;
    if (stop_flags && nflags > 0) {
        bool any = false;
        for (int i = 0; i < nflags; ++i) any |= stop_flags[i] != nullptr;
        while (any && !raised && !kernel_done()) {
            for (int i = 0; i < nflags && !raised; ++i)
                if (stop_flags[i] && *stop_flags[i]) raised = true;
            if (raised) pool_owner()->set_cutoff(true);
            else std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    rc = collect(results);
    if (raised) pool_owner()->set_cutoff(false);
    return rc;
}

int Engine::run_solo_growing(const DDInput& input, std::vector<HostResult>& results, const CacheTable* cache, const DominanceTable* dom) {
    // only the shared output arena grows: every other capacity status names a per-slot workspace and would come back unchanged
    auto capacity = [](const HostResult& r) { return r.hdr.status == ST_ERR_ARENA; };
    std::lock_guard<std::mutex> batch_guard(batch_mtx_);
    wait_decoders(-1);
    for (;;) {
        results.assign(2, HostResult());
        int rc = launch(&input, 1, cache, dom);
        if (rc == DDO_OK) rc = collect(results);
        if (rc != DDO_OK) return rc;
        if (!(capacity(results[0]) || capacity(results[1])) || arena_cap_ >= (8ull << 30)) return DDO_OK;
        if ((rc = grow_arena(arena_cap_ * 4)) != DDO_OK) return rc;
    }
}

// ---------------------------------------------------------------------------
// Combining layer under ddo_mdd_compile (engine.hpp: compile_combined)
// ---------------------------------------------------------------------------
struct Engine::Waiter {   // one per compile_combined call
    std::mutex m;
    std::condition_variable cv;
    int remaining = 0;    // requests of this call that are not finished
    int ready = 0;        // requests of this call in state 2: raw result delivered, to be decoded by the caller (who then releases the buffer set)
    bool lead = false;    // "you are the leader now"
    CompileReq* const* reqs = nullptr;   // the call's requests (the leader decodes its own delivered results before it waits for a buffer set)
    int count = 0;
};

/// A caller holding several requests decodes each one AS SOON AS it is delivered (finish_req wakes it for every state-2 request,
/// not only for the last): a delivered, undecoded result pins its buffer set (decoders_), and the leader waits for the sets --
/// a caller that sat on one result while another of its requests ran again would block the launch that re-runs it.
void Engine::decode_ready(Waiter* me) {
    std::vector<CompileReq*> todo;
    {
        std::lock_guard<std::mutex> lk(me->m);
        if (me->ready == 0) return;
        for (int i = 0; i < me->count; ++i)
            if (me->reqs[i]->state == 2) {
                me->reqs[i]->state = 4;   // being decoded
                todo.push_back(me->reqs[i]);
            }
        me->ready = 0;
    }
    for (CompileReq* r : todo) {
        decode_checked(r->hdr, r->arena, r->arena_used, r->out[0]);
        if (pooled_) pooled_fixup(r->in, r->out[0]);
        r->raw_eng->release_set(r->set);
    }
    std::lock_guard<std::mutex> lk(me->m);
    for (CompileReq* r : todo) r->state = 3;
}

/// `me` != nullptr: the caller is the LEADER and may itself hold delivered results (the previous leader names its successor first and
/// hands the results out afterwards): it decodes them while it waits, so that a leader never waits for itself.
void Engine::wait_decoders(int set, Waiter* me) {
    auto free_now = [&] { return set < 0 ? decoders_[0] + decoders_[1] == 0 : decoders_[set] == 0; };
    if (!me) {
        std::unique_lock<std::mutex> lk(dec_mtx_);
        dec_cv_.wait(lk, free_now);
        return;
    }
    for (;;) {
        decode_ready(me);
        std::unique_lock<std::mutex> lk(dec_mtx_);
        if (dec_cv_.wait_for(lk, std::chrono::microseconds(200), free_now)) return;
    }
}
void Engine::release_set(int set) {
    std::lock_guard<std::mutex> lk(dec_mtx_);
    if (--decoders_[set] == 0) dec_cv_.notify_all();
}

void Engine::pooled_fixup(const DDInput& in, HostResult& out) const {
    if (!out.valid || out.hdr.status != ST_OK) return;
    const int wsT = model_->wsT;
    const uint64_t* adj = model_->adj.data();
    const int ws = model_->ws;
    std::vector<uint64_t> cur((size_t)wsT);
    auto has = [&](int v) { return (cur[(size_t)(v >> 6)] >> (v & 63)) & 1ULL; };
    auto apply = [&](int v, bool yes) {   // Problem::transition (misp/main.rs:77-85)
        cur[(size_t)(v >> 6)] &= ~(1ULL << (v & 63));
        if (yes)
            for (int k = 0; k < ws; ++k) cur[(size_t)k] &= adj[(size_t)v * ws + k];
    };
    // best / exact path: one (variable << 1 | bit) word per layer, terminal first -> the layers whose variable impacted the node
    auto filter = [&](std::vector<uint32_t>& p) {
        for (int k = 0; k < wsT; ++k) cur[(size_t)k] = in.state[k];
        std::vector<uint32_t> kept;
        for (size_t i = p.size(); i-- > 0;) {   // root first
            const int v = (int)(p[i] >> 1);
            if (!has(v)) continue;
            kept.push_back(p[i]);
            apply(v, (p[i] & 1u) != 0);
        }
        p.assign(kept.rbegin(), kept.rend());
    };
    filter(out.best_path);
    filter(out.exact_path);
    if (out.n_cutset > 0 && out.cs_pw > 0) {
        const int stride = out.cs_path_len;
        out.cs_path.assign((size_t)out.n_cutset * (size_t)stride, 0u);
        out.cs_plen.assign((size_t)out.n_cutset, 0);
        std::vector<uint32_t> kept;
        for (int i = 0; i < out.n_cutset; ++i) {
            for (int k = 0; k < wsT; ++k) cur[(size_t)k] = in.state[k];
            const uint64_t* pb = out.cs_pbits.data() + (size_t)i * out.cs_pw;
            const int depth = out.cs_depth.empty() ? stride : out.cs_depth[(size_t)i];
            kept.clear();
            for (int tr = 0; tr < depth; ++tr) {
                const int v = (int)out.cs_lvar[(size_t)tr];
                if (!has(v)) continue;
                const bool yes = ((pb[tr >> 6] >> (tr & 63)) & 1ULL) != 0;
                kept.push_back(((uint32_t)v << 1) | (yes ? 1u : 0u));
                apply(v, yes);
            }
            out.cs_plen[(size_t)i] = (int32_t)kept.size();
            for (size_t k = 0; k < kept.size(); ++k) out.cs_path[(size_t)i * stride + k] = kept[kept.size() - 1 - k];   // node first
        }
        out.cs_pbits.clear();
        out.cs_lvar.clear();
        out.cs_pw = 0;
        // The device lists its cut-set rows in the order its threads won an atomic counter: different from run to run.  The hosts keep
        // the FIRST of two equal-ranked sub-problems (NoDupFringe, heap ties by insertion), so the rows leave in a fixed order: by the
        // layer of expansion, then by state (a state occurs once per layer).
        std::vector<int> ord((size_t)out.n_cutset);
        for (int i = 0; i < out.n_cutset; ++i) ord[(size_t)i] = i;
        const size_t w = (size_t)ws;
        std::sort(ord.begin(), ord.end(), [&](int x, int y) {
            const int dx = out.cs_depth.empty() ? 0 : out.cs_depth[(size_t)x], dy = out.cs_depth.empty() ? 0 : out.cs_depth[(size_t)y];
            if (dx != dy) return dx < dy;
            for (size_t k = 0; k < w; ++k) {
                const uint64_t a2 = out.cs_state[(size_t)x * w + k], b2 = out.cs_state[(size_t)y * w + k];
                if (a2 != b2) return a2 < b2;
            }
            return x < y;
        });
        auto permute = [&](auto& v, size_t per) {
            if (v.empty()) return;
            std::remove_reference_t<decltype(v)> t(v.size());
            for (size_t i = 0; i < ord.size(); ++i)
                for (size_t k = 0; k < per; ++k) t[i * per + k] = v[(size_t)ord[i] * per + k];
            v.swap(t);
        };
        permute(out.cs_state, w);
        permute(out.cs_value, 1);
        permute(out.cs_ub, 1);
        permute(out.cs_depth, 1);
        permute(out.cs_plen, 1);
        permute(out.cs_path, (size_t)stride);
    }
}

void Engine::decode_checked(const DDResult& r, const uint8_t* arena, size_t arena_used, HostResult& out) const {
    if (r.status == ST_NOT_RUN) {
        out.clear();
        out.hdr = r;
        return;
    }
    if (r.status == ST_OK && r.arena_off + r.arena_bytes > arena_used) {
        out.clear();
        out.hdr = r;
        out.hdr.status = ST_ERR_ARENA;
        out.valid = true;
        return;
    }
    decode(r, arena, out);
}

void Engine::finish_req(CompileReq* r, int state) {
    Waiter* w = r->waiter;
    std::lock_guard<std::mutex> lk(w->m);   // (notify under the lock: the waiter lives on the caller's stack)
    r->state = state;
    if (state == 2) ++w->ready;
    if (--w->remaining == 0 || state == 2) w->cv.notify_one();
}

/// What a finished launch leaves to be handed to the callers: done AFTER the next leader has been named, so that waking a
/// thousand threads overlaps the next launch instead of delaying it.
struct Engine::HandOut {
    Engine* eng = nullptr;                            // the engine whose buffer set `raw` lies in (a small launch of the dense engine runs on its owner)
    RawBatch raw;
    std::vector<std::pair<CompileReq*, int>> done;   // (request, its index in the launch)
    int own = 0;                                      // requests of the leader itself among them
};

/// Hands the finished requests to their callers, who decode them out of the pinned buffers of the launch's set (in parallel; the
/// set is not launched into again before they are done: decoders_).  The leader's own requests are decoded right here: a caller
/// that leads never sits on undecoded results, so waiting for a set's decoders cannot wait for the leader itself.
void Engine::hand_out(HandOut& ho, Waiter* me) {
    for (auto& [r, i] : ho.done) {
        const DDResult* h = ho.raw.hdr + (size_t)i * 2;
        if (r->in.flags & IN_FUSED) {
            decode_checked(h[1], ho.raw.arena, ho.raw.arena_used, r->out[1]);
            if (pooled_) pooled_fixup(r->in, r->out[1]);
        }
        if (r->waiter == me) {
            decode_checked(h[0], ho.raw.arena, ho.raw.arena_used, r->out[0]);
            if (pooled_) pooled_fixup(r->in, r->out[0]);
            ho.eng->release_set(ho.raw.set);
            finish_req(r, 3);
            continue;
        }
        r->hdr = h[0];
        r->arena = ho.raw.arena;
        r->arena_used = ho.raw.arena_used;
        r->set = ho.raw.set;
        r->raw_eng = ho.eng;
        finish_req(r, 2);
    }
    ho.done.clear();
    ho.own = 0;
}

/// One launch for `batch` (compatible requests: same cache and dominance tables).  Finished requests end up in `ho` (the caller
/// hands them out), requests that must run again (cut by another caller's flag, output arena full) come back in `again`.
int Engine::combined_launch(std::vector<CompileReq*>& batch, std::vector<CompileReq*>& again, Waiter* me, HandOut& ho) {
    const int n = (int)batch.size();
    auto fail_all = [&](int rc) {
        for (CompileReq* r : batch) {
            r->rc = rc;
            finish_req(r, 3);
        }
        return rc;
    };
    std::vector<DDInput> din((size_t)n);
    for (int i = 0; i < n; ++i) din[(size_t)i] = batch[(size_t)i]->in;
    std::unique_lock<std::mutex> bg(batch_mtx_);
    // The compiles of a launch share the output arena.  What the previous launches wrote per compile predicts what this one
    // needs: an arena that would overflow grows BEFORE the launch (once, to the next power of two, up to 8 GB per buffer set)
    // instead of after a launch whose compiles then run twice.
    if (cq_bytes_per_req_ > 0 && arena_cap_ < (8ull << 30)) {
        const double need = (double)n * cq_bytes_per_req_ * 1.25;
        if (need > (double)arena_cap_) {
            size_t cap = arena_cap_;
            while ((double)cap < need && cap < (8ull << 30)) cap *= 2;
            wait_decoders(-1, me);   // both arenas are freed and allocated again at the next launches
            const int grc = grow_arena(cap);
            if (grc != DDO_OK) return fail_all(grc);
        }
    }
    int set;
    {
        std::lock_guard<std::mutex> g(mtx_);
        set = next_set_;
    }
    wait_decoders(set, me);   // the launch before the previous one lies in this buffer set: its callers must have decoded
    const int64_t tq0 = g_times_on ? now_ns() : 0;
    int rc = launch(din.data(), n, batch[0]->cache, batch[0]->dom);
    if (rc != DDO_OK) return fail_all(rc);
    const int64_t tq1 = g_times_on ? now_ns() : 0;
    bool raised = false, any = false;
    for (CompileReq* r : batch) any |= r->stop != nullptr;
    while (any && !raised && !kernel_done()) {
        for (int i = 0; i < n && !raised; ++i)
            if (batch[(size_t)i]->stop && *batch[(size_t)i]->stop) raised = true;
        if (raised) pool_owner()->set_cutoff(true);
        else std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    rc = wait();
    if (raised) pool_owner()->set_cutoff(false);
    if (rc != DDO_OK) return fail_all(rc);
    const int64_t tq2 = g_times_on ? now_ns() : 0;
    cq_last_ms_ = last_kernel_ms_;
    RawBatch& raw = ho.raw;
    ho.eng = this;
    if ((rc = fetch_raw(raw)) != DDO_OK || raw.count != n) return fail_all(rc != DDO_OK ? rc : DDO_ERR_INTERNAL);
    cq_launches_.fetch_add(1);
    cq_requests_.fetch_add((uint64_t)n);
    cq_kernel_us_.fetch_add((uint64_t)(last_kernel_ms_ * 1000.0));
    // The shared output arena was full: the ONE capacity outcome a second run cures.  Every other capacity status (node slots, dedup
    // table, work / cut lists, the pools of kept layers, a Pooled DD's pool) is the compile's own and goes to its caller at once.
    auto overflow = [&](const DDResult& r) {
        if (r.status == ST_OK) return r.arena_off + r.arena_bytes > raw.arena_used;
        return r.status == ST_ERR_ARENA;
    };
    bool foreign_cut = false;
    int n_over = 0, n_ok = 0;
    std::vector<CompileReq*> over;
    uint64_t bytes = 0;
    for (int i = 0; i < n; ++i) {
        CompileReq* r = batch[(size_t)i];
        const DDResult& h0 = raw.hdr[(size_t)i * 2];
        const DDResult& h1 = raw.hdr[(size_t)i * 2 + 1];
        const bool fused = (r->in.flags & IN_FUSED) != 0;
        bytes += h0.arena_bytes + (fused && h1.status != ST_NOT_RUN ? h1.arena_bytes : 0);
        const bool cut = h0.status == ST_CUTOFF || (fused && h1.status == ST_CUTOFF);
        if (cut && !(r->stop && *r->stop)) {   // stopped by somebody else's flag: not this compile's business
            foreign_cut |= !raised;
            again.push_back(r);
            continue;
        }
        if (overflow(h0) || (fused && h1.status != ST_NOT_RUN && overflow(h1))) {
            ++n_over;
            over.push_back(r);
            continue;
        }
        ++n_ok;
        ho.done.emplace_back(r, i);
        ho.own += r->waiter == me;
    }
    cq_bytes_per_req_ = std::max(0.9 * cq_bytes_per_req_, (double)bytes / (double)n);
    {
        std::lock_guard<std::mutex> lk(dec_mtx_);
        decoders_[raw.set] += (int)ho.done.size();   // (the leader's own included: it decodes them in hand_out, after the next leader may have started)
    }
    if (n_over) {
        // Requests that did not fit the arena run again.  The arena grows (x2; x4 for a compile that was alone; up to 8 GB per
        // buffer set) while launches overflow it; beyond that the launches get smaller.  (A compile that failed on the arena
        // left nothing in the cache: misp_dd_core.hpp, thresholds follow the reservation.)
        if (arena_cap_ < (8ull << 30)) {
            hand_out(ho, me);
            wait_decoders(-1, me);   // both arenas are freed and allocated again at the next launches
            const int grc = grow_arena(std::min<size_t>(arena_cap_ * (n == 1 ? 4 : 2), 8ull << 30));
            if (grc != DDO_OK) {     // no larger arena to be had: the compiles that needed it fail (running them again would loop)
                for (CompileReq* r : over) {
                    r->rc = grc;
                    finish_req(r, 3);
                }
                over.clear();
            }
        } else if (n > 1) {
            cq_batch_cap_ = std::max(1, std::min(cq_batch_cap_, std::max(n_ok, n / 2)));
        } else {   // one compile alone does not fit 8 GB: its caller gets the capacity error
            CompileReq* r = over.back();
            over.pop_back();
            ho.done.emplace_back(r, 0);
            ho.own += r->waiter == me;
            std::lock_guard<std::mutex> lk(dec_mtx_);
            decoders_[raw.set] += 1;
        }
    } else if (cq_batch_cap_ < (1 << 20)) {
        cq_batch_cap_ = std::min(1 << 20, cq_batch_cap_ * 2);   // clean launches win the room back that overflowing ones gave up
    }
    again.insert(again.end(), over.begin(), over.end());
    if (g_times_on) {
        const int64_t tq3 = now_ns();
        g_cq_ns[1] += (uint64_t)(tq1 - tq0), g_cq_ns[2] += (uint64_t)(tq2 - tq1), g_cq_ns[3] += (uint64_t)(tq3 - tq2);
        g_cq_last_end.store(tq3);
    }
    bg.unlock();
    if (foreign_cut) std::this_thread::sleep_for(std::chrono::microseconds(200));   // another engine's caller holds the shared flag up
    return DDO_OK;
}

void Engine::lead(Waiter* me) {
    for (;;) {
        std::vector<CompileReq*> batch, again;
        {
            std::unique_lock<std::mutex> lk(cq_mtx_);
            if (cq_.empty()) {
                cq_leader_ = false;
                return;
            }
            // How many requests to wait for: the callers the previous launch released are on their way back (they decode, drain
            // their cut-set, pick their next sub-problem) and join the ones that queued up meanwhile -- cq_active_, set when that
            // launch ended -- but never more than there are mdds on the engine, node slots to fill, or (first launch) all mdds.
            // A launch of fewer decision diagrams than slots lasts as long as its longest one whatever their number, so waiting
            // for the stragglers beats launching halves; the wait is bounded by a quarter of the previous launch (50 us .. 5 ms;
            // 200 us before the first launch), and a caller that does not return (no work left) costs that once: the estimate
            // follows what the launches really carried.
            const int active = cq_active_ > 0 ? cq_active_ : users_.load();
            const size_t expect = (size_t)std::max(1, std::min(std::min(users_.load(), active), std::min(nslots_, cq_batch_cap_)));
            const int64_t tw0 = g_times_on ? now_ns() : 0;
            if (g_times_on && g_cq_last_end.load()) g_cq_ns[4] += (uint64_t)std::max<int64_t>(0, tw0 - g_cq_last_end.exchange(0));
            if (cq_.size() < expect) {
                const double us = cq_last_ms_ > 0 ? std::min(5000.0, std::max(50.0, cq_last_ms_ * 250.0)) : 200.0;
                cq_cv_.wait_for(lk, std::chrono::microseconds((long)us), [&] { return cq_.size() >= expect; });
                if (g_times_on) g_cq_ns[0] += (uint64_t)(now_ns() - tw0);
            }
            const CacheTable* cache = cq_.front()->cache;
            const DominanceTable* dom = cq_.front()->dom;
            for (auto it = cq_.begin(); it != cq_.end() && (int)batch.size() < cq_batch_cap_;) {
                if ((*it)->cache == cache && (*it)->dom == dom) {
                    batch.push_back(*it);
                    it = cq_.erase(it);
                } else {
                    ++it;
                }
            }
        }
        HandOut ho;
        // A launch of fewer decision diagrams than the full-width engine has node slots (one per CU) runs THERE: every DD gets a
        // 1 024-thread workgroup and a CU of its own instead of a 512-thread one on a half-empty CU (the reference's default
        // nb_threads is the host's core count: 64 callers are 64 decision diagrams per launch; micro grid, n = 400 / W = 10 000,
        // batches of 16: 1.2e9 nodes/s on the full-width kernel, 0.8-0.9e9 on the dense one).  Nothing is handed up from there.
        static const bool route_small = [] { const char* e = std::getenv("DDO_HIP_ROUTE_SMALL"); return !(e && std::atoi(e) == 0); }();
        bool route_ok = route_small && dense_ && owner_ && !pooled_ && (int)batch.size() <= owner_->nslots_;
        for (CompileReq* r : batch) route_ok = route_ok && r->route_ok;   // (an mdd bound to THIS kernel by DDO_MDD_ENGINE_DENSE stays on it)
        Engine* target = route_ok ? owner_ : this;
        if (target != this) {
            uint64_t l0 = target->cq_launches_.load(), r0 = target->cq_requests_.load(), k0 = target->cq_kernel_us_.load();
            target->combined_launch(batch, again, me, ho);
            cq_launches_.fetch_add(target->cq_launches_.load() - l0);   // (ddo_mdd_combine_stats reads the engine the mdd is bound to)
            cq_requests_.fetch_add(target->cq_requests_.load() - r0);
            cq_kernel_us_.fetch_add(target->cq_kernel_us_.load() - k0);
            cq_last_ms_ = target->cq_last_ms_;
        } else {
            combined_launch(batch, again, me, ho);
        }
        bool handed_over = false;
        {
            std::unique_lock<std::mutex> lk(cq_mtx_);
            for (auto it = again.rbegin(); it != again.rend(); ++it) cq_.push_front(*it);
            cq_active_ = (int)(ho.done.size() + cq_.size());   // callers that come back + callers already waiting
            int mine_left;
            {
                std::lock_guard<std::mutex> wl(me->m);
                mine_left = me->remaining - ho.own;   // (the leader's finished requests are still to be handed out, below)
            }
            if (mine_left == 0) {   // the first caller in the queue leads the next launch -- named BEFORE this launch's callers are woken
                handed_over = true;
                if (cq_.empty()) {
                    cq_leader_ = false;
                } else {
                    Waiter* w = cq_.front()->waiter;
                    std::lock_guard<std::mutex> wl(w->m);
                    w->lead = true;
                    w->cv.notify_one();
                }
            }
        }
        hand_out(ho, me);
        if (handed_over) return;   // (else one of this caller's requests runs again: it keeps leading)
    }
}

int Engine::compile_combined(CompileReq* const* reqs, int count) {
    if (count <= 0) return DDO_OK;
    Waiter me;
    me.remaining = count;
    me.reqs = reqs;
    me.count = count;
    bool lead_now = false;
    {
        std::lock_guard<std::mutex> lk(cq_mtx_);
        for (int i = 0; i < count; ++i) {
            reqs[i]->waiter = &me;
            reqs[i]->state = 0;
            reqs[i]->rc = DDO_OK;
            cq_.push_back(reqs[i]);
        }
        if (!cq_leader_) {
            cq_leader_ = true;
            lead_now = true;
        } else {
            cq_cv_.notify_one();   // (the leader's window re-checks the queue length)
        }
    }
    for (;;) {
        if (lead_now) {
            lead(&me);
            lead_now = false;
        }
        bool all_done;
        {
            std::unique_lock<std::mutex> lk(me.m);
            me.cv.wait(lk, [&] { return me.lead || me.ready > 0 || me.remaining == 0; });
            lead_now = me.lead;
            me.lead = false;
            all_done = me.remaining == 0;
        }
        decode_ready(&me);   // every delivered result is decoded at once (and its buffer set released), also before this caller leads
        if (lead_now) continue;
        if (all_done) break;
    }
    int worst = DDO_OK;
    for (int i = 0; i < count; ++i)
        if (reqs[i]->rc != DDO_OK) worst = reqs[i]->rc;
    return worst;
}

int Engine::io_reserve(IoSet& io, int count) {
    if (io.d_inputs) HIP_TRY(hipFree(io.d_inputs));
    if (io.d_results) HIP_TRY(hipFree(io.d_results));
    if (io.h_results) HIP_TRY(hipHostFree(io.h_results));
    io.d_inputs = io.d_results = nullptr;
    io.h_results = nullptr;
    io.in_cap = 0;
    const int cap = std::max(count, 256);
    void* hp = nullptr;
    HIP_TRY(hipHostMalloc(&hp, (size_t)cap * 2 * sizeof(DDResult) + 64 + (size_t)cap * sizeof(DDInput), hipHostMallocDefault));
    io.h_results = (DDResult*)hp;
    io.h_head = (unsigned long long*)((uint8_t*)hp + (size_t)cap * 2 * sizeof(DDResult));
    io.h_inputs = (DDInput*)((uint8_t*)hp + (size_t)cap * 2 * sizeof(DDResult) + 64);
    HIP_TRY(hipMalloc(&io.d_inputs, (size_t)cap * sizeof(DDInput) + LptBuffers::bytes(cap)));   // + the launch order (lpt_order_kernel)
    HIP_TRY(hipMalloc(&io.d_results, (size_t)cap * 2 * sizeof(DDResult)));
    io.in_cap = cap;
    return DDO_OK;
}

int Engine::launch(const DDInput* inputs, int count, const CacheTable* cache, const DominanceTable* dom) {
    std::lock_guard<std::mutex> g(mtx_);
    auto lt0 = std::chrono::steady_clock::now();
    auto tick = [&](int k) {
        if (!g_times_on) return;
        const auto now = std::chrono::steady_clock::now();
        g_launch_ns[k].fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(now - lt0).count(), std::memory_order_relaxed);
        lt0 = now;
    };
    if (pending_ > 0) {
        set_error("Engine::launch: a batch is already in flight");
        return DDO_ERR_INVALID;
    }
    if (count <= 0) return DDO_OK;
    if (fetch_set_ == next_set_) {
        set_error("Engine::launch: the previous results of this buffer set were not fetched");
        return DDO_ERR_INVALID;
    }
    HIP_TRY(hipSetDevice(device_));
    hipStream_t st = (hipStream_t)stream_;
    IoSet& io = io_[next_set_];
    const bool staged = inputs == nullptr;
    if (staged && (count > io.in_cap || !io.h_inputs)) {
        set_error("Engine::launch: no staged inputs (stage_inputs(count) first)");
        return DDO_ERR_INVALID;
    }
    if (staged) inputs = io.h_inputs;   // filled in place (no growth below: count <= in_cap)
    if (count > io.in_cap) {
        const int rc = io_reserve(io, count);
        if (rc != DDO_OK) return rc;
    }
    for (int i = 0; i < count; ++i) {
        // (a capacity tier never squashes: a DD whose width is below the tier's layer capacity and that needs a squash is handed up)
        if ((!owner_ && inputs[i].width + 2 > P_.capN) || inputs[i].width < 1 || inputs[i].width > max_width_) {
            set_error("compile width exceeds the max_width the mdd was created with (or is < 1)");
            return DDO_ERR_CAPACITY;
        }
    }
    if (!io.h_arena) {
        // The output arena is PINNED HOST memory the kernel writes straight into (write-only, coalesced rows): a
        // device->host copy issued while the persistent kernel of the next batch occupies every CU would wait for
        // that kernel (measured: 290-470 ms for 50 MB), zero-copy output costs nothing on the host side.
        void* hp = nullptr;
        HIP_TRY(hipHostMalloc(&hp, arena_cap_, hipHostMallocDefault));
        io.h_arena = (uint8_t*)hp;
        io.h_arena_cap = arena_cap_;
    }
    if (std::getenv("DDO_HIP_ALLOC_TRACE"))
        std::fprintf(stderr, "[ddo alloc] launch: inputs %p (+%zu) results %p (+%zu) host results %p arena %p .. %p cnt %p\n", io.d_inputs,
                     (size_t)io.in_cap * sizeof(DDInput) + LptBuffers::bytes(io.in_cap), io.d_results, (size_t)io.in_cap * 2 * sizeof(DDResult), (void*)io.h_results,
                     (void*)io.h_arena, (void*)(io.h_arena + arena_cap_), (void*)io.d_cnt);
    tick(0);
    if (!staged) std::memcpy(io.h_inputs, inputs, (size_t)count * sizeof(DDInput));
    HIP_TRY(hipMemcpyAsync(io.d_inputs, io.h_inputs, (size_t)count * sizeof(DDInput), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(io.d_cnt, 0, 16, st));  // work counter + arena head of this buffer set
    tick(1);
    EngineParams P = P_;
    P.inputs = (const DDInput*)io.d_inputs;
    P.results = (DDResult*)io.d_results;
    P.nbatch = count;
    P.work_counter = (int32_t*)io.d_cnt;
    P.order = nullptr;
    static const bool lpt = [] { const char* e = std::getenv("DDO_HIP_LPT"); return !(e && std::atoi(e) == 0); }();
    if (lpt && engine_kind_ == 2 && count > nslots_) {
        // More DDs than slots: the workgroups draw them from a counter, and the launch lasts until the LAST one is done.  In input
        // order a slot that draws a large DD late sets the length of the launch; drawn largest first the launch ends within one DD
        // of the mean load of a slot (brock400_1, 1024 root sub-problems on 512 slots, in nodes: 1.26 x the mean load in the host's
        // order by ub - value, 1.58 x in cut-set order, 1.20 x here; in time -8.5 %: the last DDs of a launch run faster than the
        // first).  The number of vertices left in the residual state predicts the nodes of its DDs (correlation 0.995,
        // tools/tail_predict.py), and the states are on the device.
        const LptBuffers lb(io.d_inputs, io.in_cap);
        const int nbins = std::min(LPT_BINS, 64 * model_->wsT + 1);   // a state of wsT words has at most 64 wsT vertices left
        HIP_TRY(hipMemsetAsync(lb.bins, 0, (size_t)nbins * 4, st));
        hipLaunchKernelGGL(lpt_count_kernel, dim3((count + 255) / 256), dim3(256), 0, st, P.inputs, count, (const uint8_t*)P_.pool, model_->wsT, nbins, lb.key, lb.bins);
        hipLaunchKernelGGL(lpt_order_kernel, dim3(1), dim3(1024), 0, st, count, nbins, (const uint32_t*)lb.key, (const uint32_t*)lb.bins, lb.order);
        HIP_TRY(hipGetLastError());
        P.order = lb.order;
    }
    // ... and draws the two decision diagrams of a sub-problem (restricted, then relaxed: IN_FUSED) as two work items -- all the
    // restricted ones first, each raising a flag its relaxed twin waits for (run_work_item2).  Twice as many items of half the size:
    // the launch ends within half a sub-problem of the mean load of a slot instead of a whole one.
    static const bool split = [] { const char* e = std::getenv("DDO_HIP_SPLIT"); return !(e && std::atoi(e) == 0); }();
    P.done = nullptr;
    bool split_now = false;
    // Only where a slot sees few, large work items: a capacity tier's decision diagrams are tiny (a second draw, a flag and a
    // re-read result record per sub-problem cost tier 0 a third of its time: 7.7 -> 10.0 s of a brock400_1 proof), and a launch
    // of many items per slot has no tail to speak of.
    if (split && engine_kind_ == 2 && (!owner_ || dense_) && count > nslots_ && count <= 4 * nslots_) {
        const LptBuffers lb(io.d_inputs, io.in_cap);
        HIP_TRY(hipMemsetAsync(lb.done, 0, (size_t)count * 4, st));
        P.done = lb.done;
        split_now = true;
    }
    tick(2);
    P.arena_head = (unsigned long long*)(io.d_cnt + 8);
    P.arena = io.h_arena;
    if (cache && (P_.tmode || pooled_) && cache->device == device_) {   // (kept layers of the layer-rebuilding engine, or a Pooled DD's event replay)
        P.cache_tab = cache->tab;
        P.cache_cap = cache->cap;
        P.cache_stride = cache->stride;
        P.cache_stats = cache->stats;
    } else {
        P.cache_tab = nullptr;
        P.cache_cap = 0;
    }
    P.dom_cap = 0;
    P.dkey_cap = 0;
    if (dom && P_.tmode && dom->device == device_ && dom->dkey) {
        P.dkey_tab = dom->dkey;
        P.dkey_cap = dom->dkey_cap;
        P.dkey_stats = dom->stats;
    } else if (dom && P_.tmode && dom->device == device_ && dom->depths >= P_.max_layers - 1) {
        P.dom_coord = dom->coord;
        P.dom_value = dom->value;
        P.dom_count = dom->count;
        P.dom_lock = dom->lock;
        P.dom_stats = dom->stats;
        P.dom_cap = dom->cap;
    }
    const int grid = std::min(split_now ? 2 * count : count, nslots_);
    kernel_fn fn = (kernel_fn)kernel_;
    if (rewind_ >= 0) {   // bench: the frozen batch overwrites the blocks of its previous run
        rewind_val_ = (unsigned long long)rewind_;
        rewind_ = -1;
        HIP_TRY(hipMemcpyAsync((void*)P_.pool_head, &rewind_val_, 8, hipMemcpyHostToDevice, st));
        pool_owner()->pool_head_bound_ = rewind_val_;
    }
    tick(6);
    HIP_TRY(hipEventRecord((hipEvent_t)ev0_, st));
    tick(7);
    if (Engine* po = pool_owner(); po->vm_base_) {
        // the kernel allocates blocks with one atomic on the pool head: back everything the launches that are not
        // fetched yet (the previous one may still run) and this one can possibly take.  The pool belongs to the owner
        // engine; a tier's launches are accounted there as well (fetch() takes them off again).
        const size_t worst = (size_t)count * (size_t)pool_block_bytes((uint32_t)P_.capW, (uint32_t)model_->wsT, (uint32_t)P_.max_layers);
        const size_t need = (size_t)po->pool_head_bound_ + po->pool_unfetched_worst_ + worst;
        if (need > po->vm_mapped_) po->pool_grow(need + 2 * po->vm_chunk_);   // (the mapper thread fell behind)
        po->pool_want(need + 2 * po->vm_chunk_);
        po->pool_unfetched_worst_ += worst;
        po->P_.pool_cap = po->vm_mapped_;
        P_.pool_cap = po->vm_mapped_;
        P.pool_cap = po->vm_mapped_;
    }
    tick(3);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(threads_), lds_bytes_, st, P);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord((hipEvent_t)ev1_, st));
    tick(4);
    if (P_.phase_clocks)
        HIP_TRY(hipMemcpyAsync(io.h_results, io.d_results, (size_t)count * 2 * sizeof(DDResult), hipMemcpyDeviceToHost, st));
    else   // every record without its clocks
        HIP_TRY(hipMemcpy2DAsync(io.h_results, sizeof(DDResult), io.d_results, sizeof(DDResult), offsetof(DDResult, phase_clk), (size_t)count * 2,
                                 hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(io.h_head, io.d_cnt + 8, 8, hipMemcpyDeviceToHost, st));
    tick(5);
    if (g_times_on) g_launch_n.fetch_add(1, std::memory_order_relaxed);
    io.count = count;
    pending_ = count;
    pending_set_ = next_set_;
    next_set_ ^= 1;
    return DDO_OK;
}

DDInput* Engine::stage_inputs(int count) {
    std::lock_guard<std::mutex> g(mtx_);
    if (pending_ > 0 || count <= 0 || fetch_set_ == next_set_) return nullptr;
    if (hipSetDevice(device_) != hipSuccess) return nullptr;
    IoSet& io = io_[next_set_];
    if (count > io.in_cap) {
        if (io.d_inputs) (void)hipFree(io.d_inputs);
        if (io.d_results) (void)hipFree(io.d_results);
        if (io.h_results) (void)hipHostFree(io.h_results);
        io.d_inputs = io.d_results = nullptr;
        io.h_results = nullptr;
        io.in_cap = 0;
        const int cap = std::max(count + count / 4, 256);
        void* hp = nullptr;
        if (hipHostMalloc(&hp, (size_t)cap * 2 * sizeof(DDResult) + 64 + (size_t)cap * sizeof(DDInput), hipHostMallocDefault) != hipSuccess) return nullptr;
        io.h_results = (DDResult*)hp;
        io.h_head = (unsigned long long*)((uint8_t*)hp + (size_t)cap * 2 * sizeof(DDResult));
        io.h_inputs = (DDInput*)((uint8_t*)hp + (size_t)cap * 2 * sizeof(DDResult) + 64);
        if (hipMalloc(&io.d_inputs, (size_t)cap * sizeof(DDInput) + LptBuffers::bytes(cap)) != hipSuccess || hipMalloc(&io.d_results, (size_t)cap * 2 * sizeof(DDResult)) != hipSuccess) return nullptr;
        io.in_cap = cap;
    }
    return io.h_inputs;
}

int Engine::wait() {
    {   // poll WITHOUT the engine lock: other host threads' calls are not locked out for the length of the kernel, and after a
        // bounded spin the waiting thread sleeps in the driver instead of burning a core
        void* st = nullptr;
        {
            std::lock_guard<std::mutex> g(mtx_);
            if (pending_ <= 0) return DDO_OK;
            st = stream_;
        }
        static const bool block0 = [] { const char* e = std::getenv("DDO_HIP_SYNC"); return e && std::string(e) == "block"; }();
        if (!block0) {
            (void)hipSetDevice(device_);
            const auto t0 = std::chrono::steady_clock::now();
            while (hipStreamQuery((hipStream_t)st) == hipErrorNotReady) {
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;   // a launch this long: let the driver wait
                std::this_thread::yield();
            }
        }
    }
    std::lock_guard<std::mutex> g(mtx_);
    if (pending_ <= 0) return DDO_OK;
    HIP_TRY(hipSetDevice(device_));
    // Polling instead of hipStreamSynchronize: the blocking wait wakes up milliseconds after the stream has drained (kernel
    // trace of the live search, round 3: 11 ms of idle device after every launch of a step), and the search synchronises three
    // times per step.  DDO_HIP_SYNC=block restores the blocking wait.
    static const bool block = [] { const char* e = std::getenv("DDO_HIP_SYNC"); return e && std::string(e) == "block"; }();
    (void)block;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream_));   // (drained already unless the spin above gave up)
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, (hipEvent_t)ev0_, (hipEvent_t)ev1_));
    last_kernel_ms_ = ms;
    kernel_ms_ += ms;
    launches_ += 1;
    fetch_set_ = pending_set_;
    pending_set_ = -1;
    pending_ = 0;
    return DDO_OK;
}

int Engine::peek_retry(std::vector<uint8_t>& retry) {
    std::lock_guard<std::mutex> g(mtx_);
    retry.clear();
    if (fetch_set_ < 0) return DDO_OK;
    const IoSet& io = io_[fetch_set_];
    retry.resize((size_t)io.count);
    for (int i = 0; i < io.count; ++i)
        retry[(size_t)i] = io.h_results[(size_t)i * 2].status == ST_RETRY || io.h_results[(size_t)i * 2 + 1].status == ST_RETRY;
    return DDO_OK;
}

int Engine::fetch(std::vector<HostResult>& results) {
    std::lock_guard<std::mutex> g(mtx_);
    if (fetch_set_ < 0) {
        results.clear();
        return DDO_OK;
    }
    IoSet& io = io_[fetch_set_];
    fetch_set_ = -1;
    const int count = io.count;
    results.resize((size_t)count * 2);
    HIP_TRY(hipSetDevice(device_));
    const size_t used = (size_t)std::min<unsigned long long>(*io.h_head, arena_cap_);
    for (int i = 0; i < count; ++i) {
        for (int k = 0; k < 2; ++k) {
            const DDResult& r = io.h_results[(size_t)i * 2 + k];
            HostResult& out = results[(size_t)i * 2 + k];
            if (r.status == ST_NOT_RUN) {
                out.clear();
                out.hdr = r;
                continue;
            }
            if (r.status == ST_OK && r.arena_off + r.arena_bytes > used) {
                out.clear();
                out.hdr = r;
                out.hdr.status = ST_ERR_ARENA;
                out.valid = true;
                continue;
            }
            decode(r, io.h_arena, out);
            if (pooled_) pooled_fixup(io.h_inputs[i], out);
            if (r.n_cutset > 0 && r.pool_off != NO_POOL_SRC)
                pool_owner()->pool_head_bound_ = std::max<uint64_t>(pool_owner()->pool_head_bound_,
                                                                    r.pool_off + pool_block_bytes((uint32_t)r.n_cutset, (uint32_t)model_->wsT, (uint32_t)(r.lel > 0 ? r.lel : 0)));
        }
    }
    if (pool_owner()->vm_base_) {   // this batch is accounted for exactly now
        Engine* po = pool_owner();
        const size_t worst = (size_t)count * (size_t)pool_block_bytes((uint32_t)P_.capW, (uint32_t)model_->wsT, (uint32_t)P_.max_layers);
        po->pool_unfetched_worst_ = po->pool_unfetched_worst_ > worst ? po->pool_unfetched_worst_ - worst : 0;
    }
    return DDO_OK;
}

int Engine::fetch_raw(RawBatch& out) {
    std::lock_guard<std::mutex> g(mtx_);
    out = RawBatch();
    if (fetch_set_ < 0) return DDO_OK;
    IoSet& io = io_[fetch_set_];
    out.set = fetch_set_;
    fetch_set_ = -1;
    out.hdr = io.h_results;
    out.arena = io.h_arena;
    out.count = io.count;
    out.arena_used = (size_t)std::min<unsigned long long>(*io.h_head, arena_cap_);
    Engine* po = pool_owner();
    uint64_t bound = po->pool_head_bound_;
    for (int i = 0; i < 2 * io.count; ++i) {
        const DDResult& r = io.h_results[i];
        if (r.status == ST_OK && r.n_cutset > 0 && r.pool_off != NO_POOL_SRC)
            bound = std::max<uint64_t>(bound, r.pool_off + pool_block_bytes((uint32_t)r.n_cutset, (uint32_t)model_->wsT, (uint32_t)(r.lel > 0 ? r.lel : 0)));
    }
    po->pool_head_bound_ = bound;
    if (po->vm_base_) {   // this batch is accounted for exactly now
        const size_t worst = (size_t)io.count * (size_t)pool_block_bytes((uint32_t)P_.capW, (uint32_t)model_->wsT, (uint32_t)P_.max_layers);
        po->pool_unfetched_worst_ = po->pool_unfetched_worst_ > worst ? po->pool_unfetched_worst_ - worst : 0;
    }
    return DDO_OK;
}

int Engine::collect(std::vector<HostResult>& results) {
    int rc = wait();
    if (rc != DDO_OK) return rc;
    return fetch(results);
}

}  // namespace ddo_hip

// ---------------------------------------------------------------------------
// host views of the device cache table (Cache::get_threshold / update_threshold for tests and tools)
// ---------------------------------------------------------------------------
namespace ddo_hip {
struct CacheView {
    uint64_t* cache_tab;
    uint64_t cache_cap;
    int cache_stride;
    unsigned long long* cache_stats;
};
template <int WS>
__global__ void cache_probe_kernel(CacheView v, const uint64_t* state, int depth, int op, long long packed_in, long long* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint64_t s[WS];
    for (int k = 0; k < WS; ++k) s[k] = state[k];
    if (op == 0) {
        int64_t packed = 0;
        out[0] = cache_get<WS>(v, s, depth, &packed) ? 1 : 0;
        out[1] = packed;
    } else {
        cache_update<WS>(v, s, depth, (int64_t)packed_in);
    }
}
static int cache_probe(const CacheTable* t, int wsT, const uint64_t* state, int ws, int depth, int op, long long packed_in, long long* out2) {
    HIP_TRY(hipSetDevice(t->device));
    uint64_t* d_state = nullptr;
    long long* d_out = nullptr;
    HIP_TRY(hipMalloc((void**)&d_state, (size_t)wsT * 8));
    HIP_TRY(hipMalloc((void**)&d_out, 16));
    std::vector<uint64_t> padded((size_t)wsT, 0);
    for (int k = 0; k < ws; ++k) padded[(size_t)k] = state[k];
    HIP_TRY(hipMemcpy(d_state, padded.data(), (size_t)wsT * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(d_out, 0, 16));
    CacheView v{t->tab, t->cap, t->stride, t->stats};
    switch (wsT) {
        case 1: hipLaunchKernelGGL(cache_probe_kernel<1>, dim3(1), dim3(64), 0, 0, v, d_state, depth, op, packed_in, d_out); break;
        case 2: hipLaunchKernelGGL(cache_probe_kernel<2>, dim3(1), dim3(64), 0, 0, v, d_state, depth, op, packed_in, d_out); break;
        case 4: hipLaunchKernelGGL(cache_probe_kernel<4>, dim3(1), dim3(64), 0, 0, v, d_state, depth, op, packed_in, d_out); break;
        case 7: hipLaunchKernelGGL(cache_probe_kernel<7>, dim3(1), dim3(64), 0, 0, v, d_state, depth, op, packed_in, d_out); break;
        case 8: hipLaunchKernelGGL(cache_probe_kernel<8>, dim3(1), dim3(64), 0, 0, v, d_state, depth, op, packed_in, d_out); break;
        case 16: hipLaunchKernelGGL(cache_probe_kernel<16>, dim3(1), dim3(64), 0, 0, v, d_state, depth, op, packed_in, d_out); break;
        case 32: hipLaunchKernelGGL(cache_probe_kernel<32>, dim3(1), dim3(64), 0, 0, v, d_state, depth, op, packed_in, d_out); break;
        case 72: hipLaunchKernelGGL(cache_probe_kernel<72>, dim3(1), dim3(64), 0, 0, v, d_state, depth, op, packed_in, d_out); break;
        default: set_error("cache_probe: unsupported state width"); return DDO_ERR_UNSUPPORTED;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    if (out2) HIP_TRY(hipMemcpy(out2, d_out, 16, hipMemcpyDeviceToHost));
    (void)hipFree(d_state);
    (void)hipFree(d_out);
    return DDO_OK;
}
}  // namespace ddo_hip

// =============================================================================
// C ABI: models and decision diagrams
// =============================================================================
using namespace ddo_hip;

struct ddo_cache {
    CacheTable* t = nullptr;
    Model* model = nullptr;
    ~ddo_cache() { delete t; }
};

struct ddo_dominance {
    DominanceTable* t = nullptr;
    Model* model = nullptr;
    ~ddo_dominance() { delete t; }
};

struct ddo_mdd {
    Model* model = nullptr;
    std::shared_ptr<Engine> engine;
    int cutset_type = DDO_LAST_EXACT_LAYER;
    bool caching = false;
    std::shared_ptr<Engine> fallback;   // engine picked by ddo_mdd_create (dense kernel): the full-width engine that takes what it hands up
    bool registered = false;            // counted among the engine's users (Engine::add_user)
    bool pooled = false;                // DDO_MDD_POOLED
    ~ddo_mdd() {
        if (registered && engine) engine->add_user(-1);
    }
    HostResult res;
    // residual of the latest compile (clean.rs:149 path_to_root, :398 depth)
    std::vector<ddo_decision> path_to_root;
    size_t depth = 0;
    bool drained = false;
    // ddo_mdd_drain_cutset_rows: the rows handed to the caller (valid until the next compile / drain)
    std::vector<uint64_t> rows_state;
    std::vector<int64_t> rows_value, rows_ub;
    std::vector<size_t> rows_depth, rows_plen;
    std::vector<ddo_decision> rows_path;
};

extern "C" {

const char* ddo_last_error(void) { return get_error(); }

int ddo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

ddo_model* ddo_model_create_misp(int n, const uint64_t* rows, const int64_t* weights) {
    if (n < 1 || !rows || !weights) {
        set_error("ddo_model_create_misp: invalid arguments");
        return nullptr;
    }
    int ws = (n + 63) / 64;
    int wsT = pick_ws(ws);
    if (wsT < 0 || n > 1024) {
        set_error("ddo_model_create_misp: at most 1024 variables are supported");
        return nullptr;
    }
    ddo_model* m = new ddo_model();
    m->m.n = n;
    m->m.ws = ws;
    m->m.wsT = wsT;
    m->m.adj.assign(rows, rows + (size_t)n * ws);
    // bits beyond n never belong to a state
    if (n % 64) {
        uint64_t mask = (1ULL << (n % 64)) - 1;
        for (int i = 0; i < n; ++i) m->m.adj[(size_t)i * ws + ws - 1] &= mask;
    }
    m->m.weight.assign(weights, weights + n);
    m->m.unit_weights = true;
    m->m.weight_abs_sum = 0;
    for (int i = 0; i < n; ++i) {
        m->m.unit_weights &= weights[i] == 1;
        m->m.weight_abs_sum += weights[i] < 0 ? -weights[i] : weights[i];
    }
    if (m->m.weight_abs_sum >= (1LL << 30)) {
        set_error("ddo_model_create_misp: sum of |weights| must stay below 2^30 (device values are int32)");
        delete m;
        return nullptr;
    }
    return m;
}

ddo_model* ddo_model_create_knapsack(int n, int64_t capacity, const int64_t* profit, const int64_t* weight) {
    if (n < 1 || n > 4095 || capacity < 0 || !profit || !weight) {
        set_error("ddo_model_create_knapsack: invalid arguments (1 <= n <= 4095, capacity >= 0)");
        return nullptr;
    }
    int64_t psum = 0;
    for (int i = 0; i < n; ++i) {
        if (profit[i] < 0 || weight[i] < 1 || weight[i] >= (1LL << 31)) {
            set_error("ddo_model_create_knapsack: profits must be >= 0 and weights in [1, 2^31)");
            return nullptr;
        }
        psum += profit[i];
    }
    if (psum >= (1LL << 30) || capacity >= (1LL << 62)) {
        set_error("ddo_model_create_knapsack: sum of profits must stay below 2^30 (device values are int32)");
        return nullptr;
    }
    ddo_model* m = new ddo_model();
    m->m.kind = MODEL_KNAPSACK;
    m->m.n = n;
    m->m.ws = 2;      // KnapsackState { capacity, depth } (main.rs:44-50): the depth keeps equal capacities of
    m->m.wsT = 2;     // different levels apart in the fringe; on the device it is constant over a layer
    m->m.kp_capacity = capacity;
    m->m.weight.assign(profit, profit + n);
    m->m.kp_weight.assign(weight, weight + n);
    m->m.unit_weights = false;
    m->m.weight_abs_sum = psum;
    // main.rs:66-70: items by increasing -profit/weight; ties keep index order (the reference's unstable sort leaves
    // them unspecified: documented deviation shared with the oracle)
    m->m.kp_order.resize(n);
    for (int i = 0; i < n; ++i) m->m.kp_order[i] = i;
    std::stable_sort(m->m.kp_order.begin(), m->m.kp_order.end(), [&](int32_t a, int32_t b) {
        return -(double)profit[a] / (double)weight[a] < -(double)profit[b] / (double)weight[b];
    });
    return m;
}

ddo_model* ddo_model_create_mcp(int n, const int64_t* adj_matrix) {
    const int ws = (n + 1) / 2 + 1;   // two benefits per word + the depth word
    if (n < 1 || !adj_matrix || ws > MAX_WS) {
        set_error("ddo_model_create_mcp: 1 <= n <= 142 vertices are supported (two benefits per word, 72 words per state)");
        return nullptr;
    }
    int64_t abs_sum = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (adj_matrix[(size_t)i * n + j] != adj_matrix[(size_t)j * n + i]) {
                set_error("ddo_model_create_mcp: the adjacency matrix must be symmetric");
                return nullptr;
            }
            abs_sum += std::llabs(adj_matrix[(size_t)i * n + j]);
        }
    if (abs_sum >= (1LL << 28)) {
        set_error("ddo_model_create_mcp: sum of |weights| must stay below 2^27 (device values are int32)");
        return nullptr;
    }
    ddo_model* m = new ddo_model();
    Model& M = m->m;
    M.kind = MODEL_MCP;
    M.n = n;
    M.ws = ws;
    M.wsT = pick_ws(ws);
    M.unit_weights = false;
    M.weight.assign(n, 0);
    M.weight_abs_sum = abs_sum;
    M.vgraph.resize((size_t)n * n);
    int64_t neg = 0;
    for (size_t i = 0; i < M.vgraph.size(); ++i) {
        M.vgraph[i] = (int32_t)adj_matrix[i];
        if (adj_matrix[i] < 0) neg += adj_matrix[i];
    }
    M.initial_value = neg / 2;                        // graph.rs:37-42: every edge sits twice in the matrix
    auto w = [&](int a, int b) { return (int64_t)M.vgraph[(size_t)a * n + b]; };
    M.vest.assign(n + 1, 0);                          // relax.rs:58-80: positive edges among the vertices >= depth
    M.vnk.assign(n + 1, 0);                           // relax.rs:83-106: negative edges among the vertices < depth
    for (int d = 0; d <= n; ++d) {
        int64_t e = 0, k = 0;
        for (int a = d; a < n; ++a)
            for (int b = a + 1; b < n; ++b)
                if (w(a, b) > 0) e += w(a, b);
        for (int j = 0; j < d; ++j)
            for (int i = 0; i < j; ++i)
                if (w(i, j) < 0) k += w(i, j);
        M.vest[d] = (int32_t)e;
        M.vnk[d] = (int32_t)k;
    }
    return m;
}

ddo_model* ddo_model_create_max2sat(int n, size_t nb_clauses, const int64_t* lit_a, const int64_t* lit_b, const int64_t* weight) {
    const int ws = (n + 1) / 2 + 1;   // two benefits per word + the depth word
    if (n < 1 || ws > MAX_WS || (nb_clauses && (!lit_a || !lit_b || !weight))) {
        set_error("ddo_model_create_max2sat: 1 <= n <= 142 variables are supported (two benefits per word, 72 words per state)");
        return nullptr;
    }
    // data.rs:31-46 + FxHashMap::insert: clause = (min literal, max literal); a repeated clause keeps its LAST weight
    const size_t N2 = 2 * (size_t)n;
    auto mk_lit = [](int64_t x) { const size_t a = (size_t)((x < 0 ? -x : x) - 1); return a + a + (x > 0 ? 1 : 0); };   // model.rs:108-113
    std::vector<int64_t> W(N2 * N2, 0);
    std::vector<char> seen(N2 * N2, 0);
    auto offset = [&](int64_t x, int64_t y) { const int64_t a = std::min(x, y), b = std::max(x, y); return mk_lit(a) * N2 + mk_lit(b); };
    int64_t abs_sum = 0;
    for (size_t k = 0; k < nb_clauses; ++k) {
        const int64_t a = lit_a[k], b = lit_b[k];
        if (a == 0 || b == 0 || std::llabs(a) > n || std::llabs(b) > n) {
            set_error("ddo_model_create_max2sat: literals must be in [-n, -1] or [1, n]");
            return nullptr;
        }
        W[offset(a, b)] = weight[k];
        seen[offset(a, b)] = 1;
    }
    ddo_model* m = new ddo_model();
    Model& M = m->m;
    M.kind = MODEL_MAX2SAT;
    M.n = n;
    M.ws = ws;
    M.wsT = pick_ws(ws);
    M.unit_weights = false;
    M.weight.assign(n, 0);
    // model.rs:126-137: sum of clause weights per variable, tautologies into the initial value
    std::vector<int64_t> socw(n, 0);
    int64_t initial = 0;
    for (size_t la = 0; la < N2; ++la)
        for (size_t lb = 0; lb < N2; ++lb) {
            if (!seen[la * N2 + lb]) continue;
            const int64_t w = W[la * N2 + lb];
            abs_sum += std::llabs(w);
            const size_t va = la / 2, vb = lb / 2;
            socw[va] += w;
            if (la != lb) socw[vb] += w;                 // !is_unit
            if (va == vb && la != lb) initial += w;      // is_tautology: a == -b
        }
    if (abs_sum >= (1LL << 27)) {
        set_error("ddo_model_create_max2sat: sum of |weights| must stay below 2^27 (device values are int32)");
        delete m;
        return nullptr;
    }
    M.weight_abs_sum = abs_sum;
    M.initial_value = initial;
    // model.rs:138-140: worst variable first; ties keep index order (the reference's unstable sort leaves them unspecified:
    // documented deviation shared with the oracle)
    M.m2_order.resize(n);
    for (int i = 0; i < n; ++i) M.m2_order[i] = i;
    std::stable_sort(M.m2_order.begin(), M.m2_order.end(), [&](int32_t a, int32_t b) { return socw[a] < socw[b]; });
    M.m2_rankpos.assign(n, 0);
    for (int i = 0; i < n; ++i) M.m2_rankpos[M.m2_order[i]] = i;
    auto lit_t = [](int v) { return (int64_t)(1 + v); };
    auto lit_f = [](int v) { return -(int64_t)(1 + v); };
    auto wgt = [&](int64_t x, int64_t y) { return W[offset(x, y)]; };
    for (int q = 0; q < 4; ++q) M.m2_w[q].assign((size_t)n * n, 0);
    for (int k = 0; k < n; ++k)
        for (int l = 0; l < n; ++l) {
            M.m2_w[0][(size_t)k * n + l] = (int32_t)wgt(lit_t(k), lit_t(l));
            M.m2_w[1][(size_t)k * n + l] = (int32_t)wgt(lit_t(k), lit_f(l));
            M.m2_w[2][(size_t)k * n + l] = (int32_t)wgt(lit_f(k), lit_t(l));
            M.m2_w[3][(size_t)k * n + l] = (int32_t)wgt(lit_f(k), lit_f(l));
        }
    // model.rs:172-229: nk[k] = tautologies of the k first variables of the order; estimates[k] = best the clauses among the
    // variables order[k..] can still yield
    M.vest.assign(n + 1, 0);
    M.vnk.assign(n + 1, 0);
    for (int k = 0; k < n; ++k) {
        int64_t sum = 0;
        for (int i = 0; i < k; ++i) sum += wgt(lit_t(M.m2_order[i]), lit_f(M.m2_order[i]));
        M.vnk[k] = (int32_t)sum;
        int64_t est = 0;
        for (int i = k; i < n; ++i) {
            const int vi = M.m2_order[i];
            for (int j = i + 1; j < n; ++j) {
                const int vj = M.m2_order[j];
                const int64_t tt = wgt(lit_t(vi), lit_t(vj)), tf = wgt(lit_t(vi), lit_f(vj));
                const int64_t ft = wgt(lit_f(vi), lit_t(vj)), ff = wgt(lit_f(vi), lit_f(vj));
                est += std::max(std::max(tt + tf + ft, tt + tf + ff), std::max(tt + ft + ff, tf + ft + ff));
            }
            est += wgt(lit_t(vi), lit_f(vi)) + std::max(wgt(lit_t(vi), lit_t(vi)), wgt(lit_f(vi), lit_f(vi)));
        }
        M.vest[k] = (int32_t)est;
    }
    return m;
}

ddo_model* ddo_model_create_tsptw(int n, const int64_t* distances, const int64_t* earliest, const int64_t* latest) {
    if (n < 2 || n > 256 || !distances || !earliest || !latest) {
        set_error("ddo_model_create_tsptw: 2 <= nb_nodes <= 256 (the reference's Set256, examples/tsptw/state.rs:34-69)");
        return nullptr;
    }
    int64_t worst = 0;
    for (int i = 0; i < n; ++i) {
        int64_t row = 0;
        for (int j = 0; j < n; ++j) {
            const int64_t d = distances[(size_t)i * n + j];
            if (d < 0 || d >= (1LL << 29)) {
                set_error("ddo_model_create_tsptw: distances must be in [0, 2^29)");
                return nullptr;
            }
            row = std::max(row, d);
        }
        worst += row;
        if (earliest[i] < 0 || latest[i] < 0 || latest[i] >= (1LL << 31) || earliest[i] >= (1LL << 31)) {
            set_error("ddo_model_create_tsptw: time windows must be in [0, 2^31)");
            return nullptr;
        }
        worst = std::max(worst, latest[i]);
    }
    if (worst >= (1LL << 30)) {
        set_error("ddo_model_create_tsptw: tour lengths must stay below 2^30 (device values are int32)");
        return nullptr;
    }
    ddo_model* m = new ddo_model();
    Model& M = m->m;
    M.kind = MODEL_TSPTW;
    M.n = n;
    M.ws = tw_state_words(n);                   // 3K + 2 words, K words per node set (dd_tsptw.hpp): 5 / 8 / 14
    M.wsT = M.ws == 5 ? 7 : (M.ws == 8 ? 8 : 16);   // the widths tw_k_of_ws maps back to K = 1 / 2 / 4
    M.dbits = n <= 64 ? 6 : 8;
    M.unit_weights = false;
    M.weight.assign(n, 0);
    M.weight_abs_sum = worst;
    M.initial_value = 0;
    M.tw_dist.resize((size_t)n * n);
    M.tw_early.resize(n);
    M.tw_late.resize(n);
    M.tw_cheap.resize(n);
    for (size_t i = 0; i < M.tw_dist.size(); ++i) M.tw_dist[i] = (int32_t)distances[i];
    for (int i = 0; i < n; ++i) {
        M.tw_early[i] = (int32_t)earliest[i];
        M.tw_late[i] = (int32_t)latest[i];
        int64_t c = INT32_MAX;                               // relax.rs:50-63: cheapest edge entering node i
        for (int j = 0; j < n; ++j)
            if (j != i) c = std::min<int64_t>(c, distances[(size_t)j * n + i]);
        M.tw_cheap[i] = (int32_t)c;
    }
    M.tw_order.resize(n);
    for (int i = 0; i < n; ++i) M.tw_order[i] = i;
    std::stable_sort(M.tw_order.begin(), M.tw_order.end(), [&](int32_t a, int32_t b) { return M.tw_cheap[a] < M.tw_cheap[b]; });
    return m;
}

ddo_model* ddo_model_read_tsptw(const char* path) {
    int n = 0;
    std::vector<int64_t> d, e, l;
    if (!path || !read_tsptw(path, n, d, e, l)) return nullptr;
    return ddo_model_create_tsptw(n, d.data(), e.data(), l.data());
}

ddo_model* ddo_model_read_max2sat(const char* path) {
    int n = 0;
    std::vector<int64_t> a, b, w;
    if (!path || !read_max2sat(path, n, a, b, w)) return nullptr;
    return ddo_model_create_max2sat(n, a.size(), a.data(), b.data(), w.data());
}

ddo_model* ddo_model_read_mcp(const char* path) {
    int n = 0;
    std::vector<int64_t> adj;
    if (!path || !read_mcp(path, n, adj)) return nullptr;
    return ddo_model_create_mcp(n, adj.data());
}

ddo_model* ddo_model_read_knapsack(const char* path) {
    int64_t capacity = 0;
    std::vector<int64_t> profit, weight;
    if (!path || !read_knapsack(path, capacity, profit, weight)) return nullptr;
    return ddo_model_create_knapsack((int)profit.size(), capacity, profit.data(), weight.data());
}

ddo_model* ddo_model_read_misp(const char* path) {
    int n = 0;
    std::vector<uint64_t> rows;
    std::vector<int64_t> weights;
    if (!path || !read_misp_clq(path, n, rows, weights)) return nullptr;
    return ddo_model_create_misp(n, rows.data(), weights.data());
}

void ddo_model_destroy(ddo_model* model) { delete model; }
int ddo_model_nb_variables(const ddo_model* model) { return model ? model->m.n : DDO_ERR_INVALID; }
int ddo_model_state_words(const ddo_model* model) { return model ? model->m.ws : DDO_ERR_INVALID; }

int ddo_model_initial_state(const ddo_model* model, uint64_t* out) {
    if (!model || !out) return DDO_ERR_INVALID;
    model->m.initial_state(out);
    return DDO_OK;
}
int64_t ddo_model_initial_value(const ddo_model* model) { return model ? model->m.initial_value : 0; }  // Problem::initial_value
int ddo_model_compare_states(const ddo_model* model, const uint64_t* a, const uint64_t* b) {
    return model->m.compare_states(a, b);
}
int ddo_model_export_misp(const ddo_model* model, uint64_t* rows, int64_t* weights) {
    if (!model) return DDO_ERR_INVALID;
    if (rows) std::memcpy(rows, model->m.adj.data(), model->m.adj.size() * 8);
    if (weights) std::memcpy(weights, model->m.weight.data(), model->m.weight.size() * 8);
    return DDO_OK;
}

ddo_mdd* ddo_mdd_create(const ddo_model* model, int device, int cutset_type, size_t max_width) {
    if (!model) {
        set_error("ddo_mdd_create: null model");
        return nullptr;
    }
    const bool caching = (cutset_type & DDO_MDD_CACHING) != 0;
    const int selector = cutset_type & DDO_MDD_ENGINE_MASK;
    const bool pooled = (cutset_type & DDO_MDD_POOLED) != 0;
    cutset_type &= ~(DDO_MDD_CACHING | DDO_MDD_ENGINE_MASK | DDO_MDD_POOLED);
    if (pooled && cutset_type == 0) cutset_type = DDO_FRONTIER;   // (a Pooled DD has one kind of cut-set: the frontier, pooled.rs:543-566)
    if (cutset_type != DDO_LAST_EXACT_LAYER && cutset_type != DDO_FRONTIER) {
        set_error("ddo_mdd_create: cutset_type must be DDO_LAST_EXACT_LAYER or DDO_FRONTIER (optionally | DDO_MDD_CACHING)");
        return nullptr;
    }
    Model* m = const_cast<Model*>(&model->m);
    const bool keep = !pooled && (caching || cutset_type == DDO_FRONTIER);   // both need every layer of the DD on the device
    if (selector && selector != DDO_MDD_ENGINE_FULL && keep) {
        set_error("ddo_mdd_create: DDO_MDD_ENGINE_* selects a kernel of the in-place engine (DDO_LAST_EXACT_LAYER, no DDO_MDD_CACHING)");
        return nullptr;
    }
    // No selector: MISP decision diagrams of width 2048 and more start on the dense kernel (two decision diagrams per CU, the kernel
    // the solver spends its time in) and what it hands up -- a layer beyond its dedup table: 0 of 6.1 million compiles of the
    // brock400_1 search -- runs on the full-width engine; everything else on the one engine of its (model, device, width).
    std::shared_ptr<Engine> eng, fallback;
    if (pooled) {
        if (selector) {   // (DDO_MDD_CACHING: Pooled behind a SimpleCache, pooled.rs:467-535, 662-680 -- the cache of ddo_compile_input.cache)
            set_error("ddo_mdd_create: DDO_MDD_POOLED takes no engine selector");
            return nullptr;
        }
        eng = Engine::get(m, device, (long)max_width, Engine::ENGINE_POOLED);
    } else if (selector == DDO_MDD_ENGINE_FULL || keep) eng = Engine::get(m, device, (long)max_width, keep ? Engine::ENGINE_KEEP_LAYERS : 0);
    else if (selector) eng = Engine::get_selected(m, device, (long)max_width, selector);
    else {
        if (m->kind == MODEL_MISP && max_width >= 2048 && !std::getenv("DDO_HIP_NO_AUTO_DENSE")) {
            const std::string keep_err = get_error();
            eng = Engine::get_selected(m, device, (long)max_width, 0);   // (dense tier of a private full-width owner)
            if (eng) fallback = eng->owner_shared();
            else set_error(keep_err);   // (widths the dense tier is not built for: the plain engine below)
        }
        if (!eng) eng = Engine::get(m, device, (long)max_width, 0);
    }
    if (!eng) return nullptr;
    ddo_mdd* d = new ddo_mdd();
    d->model = m;
    d->engine = eng;
    d->fallback = fallback;
    d->pooled = pooled;
    eng->add_user(1);
    d->registered = true;
    d->cutset_type = cutset_type;
    d->caching = caching;
    return d;
}
void ddo_mdd_destroy(ddo_mdd* mdd) { delete mdd; }

static int fill_input(const Model& m, const ddo_compile_input* in, DDInput& out, uint32_t flags) {
    if (!in || !in->residual.state) return DDO_ERR_INVALID;
    if (in->comp_type != DDO_EXACT && in->comp_type != DDO_RELAXED && in->comp_type != DDO_RESTRICTED) return DDO_ERR_INVALID;
    if (in->residual.state_words != 0 && (int)in->residual.state_words != m.ws) return DDO_ERR_INVALID;
    std::memset(&out, 0, sizeof(out));
    out.src_off = NO_POOL_SRC;
    out.comp_type = in->comp_type;
    out.flags = flags;
    // Exact compiles use the whole workspace as "width"
    out.width = (int32_t)std::min<size_t>(in->max_width, 1u << 30);
    const int64_t lim = (1LL << 30);
    if (in->residual.value <= -lim || in->residual.value >= lim) return DDO_ERR_INVALID;
    out.value = (int32_t)in->residual.value;
    out.depth = (int32_t)in->residual.depth;
    out.best_lb = in->best_lb;
    for (int k = 0; k < m.ws; ++k) out.state[k] = in->residual.state[k];
    return DDO_OK;
}

int ddo_mdd_compile_batch(ddo_mdd* const* mdds, const ddo_compile_input* inputs, ddo_completion* outs, int* statuses,
                          size_t count) {
    if (!mdds || !inputs || count == 0) return DDO_ERR_INVALID;
    Engine* eng = mdds[0]->engine.get();
    std::vector<Engine::CompileReq> reqs(count);
    std::vector<Engine::CompileReq*> active;
    std::vector<size_t> active_i;
    const ddo_cache* cache = inputs[0].cache;
    const ddo_dominance* dom = inputs[0].dominance;
    for (size_t i = 0; i < count; ++i) {
        if (!mdds[i] || mdds[i]->engine.get() != eng) {
            set_error("ddo_mdd_compile_batch: all mdds must come from the same model, device and max_width");
            return DDO_ERR_INVALID;
        }
        if (inputs[i].dominance != dom || (dom && (!mdds[i]->caching || dom->model != mdds[i]->model))) {
            set_error("ddo_mdd_compile_batch: one ddo_dominance (of the same model) per batch, and only for mdds created with DDO_MDD_CACHING");
            return DDO_ERR_INVALID;
        }
        if (inputs[i].cache != cache || (cache && (!mdds[i]->caching || cache->model != mdds[i]->model))) {
            set_error("ddo_mdd_compile_batch: one ddo_cache (of the same model) per batch, and only for mdds created with DDO_MDD_CACHING");
            return DDO_ERR_INVALID;
        }
        mdds[i]->res.clear();
        mdds[i]->drained = false;
        if (inputs[i].cutoff && *inputs[i].cutoff) {  // Cutoff::must_stop() polled before the first layer (clean.rs:352)
            if (statuses) statuses[i] = DDO_CUTOFF;
            if (outs) outs[i] = ddo_completion{0, 0, 0};
            continue;
        }
        Engine::CompileReq& rq = reqs[i];
        int rc = fill_input(*mdds[i]->model, &inputs[i], rq.in,
                            IN_WANT_PATHS | IN_PATH_BITS | ((mdds[i]->cutset_type == DDO_FRONTIER && !mdds[i]->pooled) ? IN_FRONTIER : 0u) | (cache ? IN_CACHE : 0u) | (dom ? IN_DOMINANCE : 0u));
        if (rc != DDO_OK) {
            set_error("ddo_mdd_compile: invalid compile input");
            return rc;
        }
        if (inputs[i].comp_type == DDO_EXACT) rq.in.width = (int32_t)eng->max_width();
        if (rq.in.width > eng->max_width() || rq.in.width < 1) {
            set_error("ddo_mdd_compile: max_width must be in [1, width the mdd was created with]");
            return DDO_ERR_CAPACITY;
        }
        rq.stop = inputs[i].cutoff;
        rq.route_ok = mdds[i]->fallback != nullptr;
        rq.cache = cache ? cache->t : nullptr;
        rq.dom = dom ? dom->t : nullptr;
        rq.out = &mdds[i]->res;
        active.push_back(&rq);
        active_i.push_back(i);
    }
    // Concurrent callers -- the reference's worker threads, one mdd each (parallel.rs:576-602) -- meet in the engine's combining layer
    // and share launches (Engine::compile_combined); so do the compiles of this batch.  Per-compile semantics are kept there: a
    // compile cut by ANOTHER compile's cutoff flag runs again, one that found the shared output arena full runs again (the arena
    // grows for a cut-set that does not fit it on its own).
    int rc = eng->compile_combined(active.data(), (int)active.size());
    if (rc != DDO_OK) return rc;
    // an mdd whose engine ddo_mdd_create picked (dense kernel first): what the dense kernel hands up runs on the full-width engine
    if (Engine* fb = mdds[0]->fallback.get()) {
        std::vector<Engine::CompileReq*> up;
        for (Engine::CompileReq* r : active)
            if (r->out->hdr.status == ST_RETRY) up.push_back(r);
        if (!up.empty() && (rc = fb->compile_combined(up.data(), (int)up.size())) != DDO_OK) return rc;
    }
    int worst = DDO_OK;
    for (size_t a = 0; a < active.size(); ++a) {
        const size_t i = active_i[a];
        ddo_mdd* d = mdds[i];
        d->depth = inputs[i].residual.depth;
        d->path_to_root.assign(inputs[i].residual.path, inputs[i].residual.path + inputs[i].residual.path_len);
        int st = d->res.hdr.status;
        int code = st == ST_OK ? DDO_OK : (st == ST_CUTOFF ? DDO_CUTOFF : (st == ST_RETRY ? DDO_HANDED_UP : (st <= -100 ? DDO_ERR_CAPACITY : st)));
        if (statuses) statuses[i] = code;
        if (code < 0 && worst >= 0) {
            worst = code;
            if (st == ST_ERR_LPOOL || st == ST_ERR_APOOL)
                set_error("device compile failed: a decision diagram outgrew the per-slot pools of kept layers / arcs "
                          "(DDO_HIP_LPOOL_M, DDO_HIP_APOOL_M: pool sizes in millions of records)");
            else
                set_error("device compile failed (capacity or internal error), see per-item status");
        }
        if (code != DDO_OK) d->res.valid = false;
        if (outs) {
            outs[i].is_exact = ddo_mdd_is_exact(d);
            outs[i].has_best_value = d->res.valid ? d->res.hdr.has_best : 0;
            outs[i].best_value = outs[i].has_best_value ? d->res.hdr.best_value : 0;
        }
    }
    return worst;
}

int ddo_mdd_compile(ddo_mdd* mdd, const ddo_compile_input* input, ddo_completion* out) {
    int status = DDO_OK;
    int rc = ddo_mdd_compile_batch(&mdd, input, out, &status, 1);
    if (rc < 0) return rc;
    return status;
}

int ddo_mdd_is_exact(const ddo_mdd* mdd) {  // clean.rs:241-243
    return mdd && mdd->res.valid && (mdd->res.hdr.is_exact || mdd->res.hdr.has_exact_best_path) ? 1 : 0;
}
int ddo_mdd_best_value(const ddo_mdd* mdd, int64_t* value) {
    if (!mdd || !mdd->res.valid || !mdd->res.hdr.has_best) return 0;
    if (value) *value = mdd->res.hdr.best_value;
    return 1;
}
int ddo_mdd_best_exact_value(const ddo_mdd* mdd, int64_t* value) {
    if (!mdd || !mdd->res.valid || !mdd->res.hdr.has_best_exact) return 0;
    if (value) *value = mdd->res.hdr.best_exact_value;
    return 1;
}
static int emit_path(const ddo_mdd* mdd, const std::vector<uint32_t>& p, ddo_decision* buf, size_t* len) {
    size_t need = mdd->path_to_root.size() + p.size();
    if (!len) return DDO_ERR_INVALID;
    if (!buf || *len < need) {
        *len = need;
        return DDO_ERR_CAPACITY;
    }
    size_t k = 0;
    for (const ddo_decision& d : mdd->path_to_root) buf[k++] = d;
    for (uint32_t x : p) buf[k++] = mdd->model->path_decision(x);
    *len = need;
    return 1;
}
int ddo_mdd_best_solution(const ddo_mdd* mdd, ddo_decision* buf, size_t* len) {
    if (!mdd || !mdd->res.valid || !mdd->res.hdr.has_best) return 0;
    return emit_path(mdd, mdd->res.best_path, buf, len);
}
int ddo_mdd_best_exact_solution(const ddo_mdd* mdd, ddo_decision* buf, size_t* len) {
    if (!mdd || !mdd->res.valid || !mdd->res.hdr.has_best_exact) return 0;
    return emit_path(mdd, mdd->res.hdr.exact_same_as_best ? mdd->res.best_path : mdd->res.exact_path, buf, len);
}
size_t ddo_mdd_cutset_count(const ddo_mdd* mdd) {
    if (!mdd || !mdd->res.valid || mdd->drained) return 0;
    return (size_t)mdd->res.n_cutset;
}

int ddo_mdd_drain_cutset_rows(ddo_mdd* mdd, int64_t ub_above, ddo_cutset_rows* rows) {
    if (!mdd || !rows) return DDO_ERR_INVALID;
    *rows = ddo_cutset_rows{};
    const int ws = mdd->model->ws;
    rows->state_words = (size_t)ws;
    if (!mdd->res.valid || mdd->drained) return DDO_OK;
    mdd->drained = true;
    const HostResult& r = mdd->res;
    const bool bits = r.cs_pw > 0;
    size_t keep = 0, stride = 0;
    for (int i = 0; i < r.n_cutset; ++i) {
        if ((int64_t)r.cs_ub[(size_t)i] <= ub_above) continue;
        ++keep;
        const int ldepth = r.cs_depth.empty() ? r.cs_path_len : r.cs_depth[(size_t)i];
        stride = std::max(stride, (size_t)(r.cs_plen.empty() ? ldepth : r.cs_plen[(size_t)i]));
    }
    mdd->rows_state.resize(keep * (size_t)ws);
    mdd->rows_value.resize(keep);
    mdd->rows_ub.resize(keep);
    mdd->rows_depth.resize(keep);
    mdd->rows_plen.resize(keep);
    mdd->rows_path.resize(keep * stride);
    size_t o = 0;
    for (int i = 0; i < r.n_cutset; ++i) {
        if ((int64_t)r.cs_ub[(size_t)i] <= ub_above) continue;
        const int ldepth = r.cs_depth.empty() ? r.cs_path_len : r.cs_depth[(size_t)i];   // frontier cut-set: nodes of several layers
        const int plen = r.cs_plen.empty() ? ldepth : r.cs_plen[(size_t)i];             // Pooled: a decision per EXPANDED ancestor
        std::memcpy(mdd->rows_state.data() + o * (size_t)ws, r.cs_state.data() + (size_t)i * (size_t)ws, (size_t)ws * 8);
        mdd->rows_value[o] = r.cs_value[(size_t)i];
        mdd->rows_ub[o] = r.cs_ub[(size_t)i];
        mdd->rows_depth[o] = mdd->depth + (size_t)ldepth;
        mdd->rows_plen[o] = (size_t)plen;
        ddo_decision* out = mdd->rows_path.data() + o * stride;
        if (bits) {   // IN_PATH_BITS: node first, towards the DD's root (clean.rs:329-343)
            const uint64_t* pb = r.cs_pbits.data() + (size_t)i * (size_t)r.cs_pw;
            for (int k = 0; k < plen; ++k) {
                const int tr = plen - 1 - k;
                out[k] = mdd->model->path_decision((r.cs_lvar[(size_t)tr] << 1) | (uint32_t)((pb[tr >> 6] >> (tr & 63)) & 1ULL));
            }
        } else {
            for (int k = 0; k < plen; ++k) out[k] = mdd->model->path_decision(r.cs_path[(size_t)i * (size_t)r.cs_path_len + (size_t)k]);
        }
        ++o;
    }
    rows->count = keep;
    rows->path_stride = stride;
    rows->states = mdd->rows_state.data();
    rows->values = mdd->rows_value.data();
    rows->ubs = mdd->rows_ub.data();
    rows->depths = mdd->rows_depth.data();
    rows->path_lens = mdd->rows_plen.data();
    rows->paths = mdd->rows_path.data();
    return DDO_OK;
}

int ddo_mdd_drain_cutset(ddo_mdd* mdd, ddo_cutset_cb cb, void* user) {
    if (!mdd || !cb) return DDO_ERR_INVALID;
    if (!mdd->res.valid || mdd->drained) return DDO_OK;
    mdd->drained = true;
    const HostResult& r = mdd->res;
    const int ws = mdd->model->ws;
    std::vector<ddo_decision> path = mdd->path_to_root;
    const size_t root_len = path.size();
    const bool bits = r.cs_pw > 0;
    for (int i = 0; i < r.n_cutset; ++i) {
        const int ldepth = r.cs_depth.empty() ? r.cs_path_len : r.cs_depth[i];   // frontier cut-set: nodes of several layers
        const int plen = r.cs_plen.empty() ? ldepth : r.cs_plen[i];             // Pooled: a decision per EXPANDED ancestor (pooled.rs:316-334)
        path.resize(root_len + (size_t)plen);
        if (bits) {   // IN_PATH_BITS: node first, towards the DD's root (clean.rs:329-343)
            const uint64_t* pb = r.cs_pbits.data() + (size_t)i * r.cs_pw;
            for (int k = 0; k < plen; ++k) {
                const int tr = plen - 1 - k;
                path[root_len + (size_t)k] = mdd->model->path_decision((r.cs_lvar[(size_t)tr] << 1) | (uint32_t)((pb[tr >> 6] >> (tr & 63)) & 1ULL));
            }
        } else {
            for (int k = 0; k < plen; ++k) path[root_len + (size_t)k] = mdd->model->path_decision(r.cs_path[(size_t)i * r.cs_path_len + k]);
        }
        ddo_subproblem sp;
        sp.state = r.cs_state.data() + (size_t)i * ws;
        sp.state_words = (size_t)ws;
        sp.value = r.cs_value[i];
        sp.ub = r.cs_ub[i];
        sp.depth = mdd->depth + (size_t)ldepth;
        sp.path = path.data();
        sp.path_len = path.size();
        cb(&sp, user);
    }
    return DDO_OK;
}
ddo_dominance* ddo_dominance_create(const ddo_model* model, int device, size_t capacity_per_depth) {
    if (!model) {
        set_error("ddo_dominance_create: null model");
        return nullptr;
    }
    DominanceTable* t = DominanceTable::create(&model->m, device, capacity_per_depth);
    if (!t) return nullptr;
    ddo_dominance* d = new ddo_dominance();
    d->t = t;
    d->model = const_cast<Model*>(&model->m);
    return d;
}
void ddo_dominance_destroy(ddo_dominance* d) { delete d; }
int ddo_dominance_clear(ddo_dominance* d) { return d ? d->t->clear() : DDO_ERR_INVALID; }

ddo_cache* ddo_cache_create(const ddo_model* model, int device, size_t capacity_entries) {
    if (!model || capacity_entries < 1) {
        set_error("ddo_cache_create: invalid arguments");
        return nullptr;
    }
    CacheTable* t = CacheTable::create(&model->m, device, capacity_entries);
    if (!t) return nullptr;
    ddo_cache* c = new ddo_cache();
    c->t = t;
    c->model = const_cast<Model*>(&model->m);
    return c;
}
void ddo_cache_destroy(ddo_cache* cache) { delete cache; }
int ddo_cache_clear(ddo_cache* cache) { return cache ? cache->t->clear() : DDO_ERR_INVALID; }
int ddo_cache_stats(const ddo_cache* cache, uint64_t* used, uint64_t* dropped) {
    return cache ? cache->t->read_stats(used, dropped) : DDO_ERR_INVALID;
}

int ddo_cache_get_threshold(const ddo_cache* cache, const uint64_t* state, size_t depth, int64_t* value, int* explored) {
    if (!cache || !state) return DDO_ERR_INVALID;
    long long out[2] = {0, 0};
    int rc = cache_probe(cache->t, cache->model->wsT, state, cache->model->ws, (int)depth, 0, 0, out);
    if (rc != DDO_OK) return rc;
    if (!out[0]) return 0;
    const int32_t tv = th_value((int64_t)out[1]);
    if (value) *value = tv == TH_INF ? INT64_MAX : (int64_t)tv;   // isize::MAX and everything saturating arithmetic derives from it
    if (explored) *explored = th_explored((int64_t)out[1]) ? 1 : 0;
    return 1;
}
int ddo_cache_update_threshold(ddo_cache* cache, const uint64_t* state, size_t depth, int64_t value, int explored) {
    if (!cache || !state) return DDO_ERR_INVALID;
    const int32_t tv = value >= (int64_t)TH_INF ? TH_INF : (value < INT32_MIN + 2 ? INT32_MIN + 2 : (int32_t)value);
    return cache_probe(cache->t, cache->model->wsT, state, cache->model->ws, (int)depth, 1, (long long)th_pack(tv, explored != 0), nullptr);
}

int ddo_mdd_combine_stats(const ddo_mdd* mdd, uint64_t* launches, uint64_t* requests, double* kernel_ms) {
    if (!mdd || !mdd->engine) return DDO_ERR_INVALID;
    mdd->engine->combine_stats(launches, requests, kernel_ms);
    return DDO_OK;
}

int ddo_mdd_last_counters(const ddo_mdd* mdd, ddo_counters* out) {
    if (!mdd || !out) return DDO_ERR_INVALID;
    out->nodes_expanded = mdd->res.hdr.nodes_expanded;
    out->arcs = mdd->res.hdr.arcs;
    out->layers = mdd->res.hdr.layers;
    out->compiles = mdd->res.valid ? 1 : 0;
    return DDO_OK;
}

}  // extern "C"
