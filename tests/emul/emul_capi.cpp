// =============================================================================
// emul_capi.cpp -- TEST INFRASTRUCTURE ONLY.
// Builds ddo_amd/csrc/misp_dd_core.hpp as a lock-step host emulation
// (-DDDO_HOST_EMULATION: every PAR block is a sequential loop over thread ids) so
// that the workgroup logic of the device kernel can be checked against the CPU
// oracle inside this GPU-less container.  It is NOT part of the product library,
// is never loaded by ddo_amd/, and proves nothing about the HIP build beyond
// shared control flow -- the `-m gpu` tests are the parity tests proper.
// =============================================================================
#define DDO_HOST_EMULATION 1
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../ddo_amd/csrc/misp_dd_inplace.hpp"
#include "../../ddo_amd/csrc/engine.hpp"   // host-side Model: emul_create_model reads the product's model descriptors

using namespace ddo_hip;

namespace {
struct Emul {
    EngineParams P;
    std::vector<uint64_t> adj;
    std::vector<int32_t> weight;
    std::vector<unsigned char> mem;   // workspace
    std::vector<unsigned char> lds;
    std::vector<unsigned char> arena;
    unsigned long long arena_head = 0;
    int32_t work_counter = 0;
    int nthreads = 256;
    int wsT = 0;  // template WS used
    int engine = 1;
    bool pooled = false;
    std::vector<unsigned char> mem2;
    std::vector<unsigned char> lds2;
    std::vector<int32_t> aux[2], lddelta;   // knapsack tables, per-layer relax deltas
    std::vector<unsigned char> mem3;         // kept layers (frontier cut-set / thresholds / cache)
    std::vector<uint64_t> lbase, abase;      // DDO_EMUL_LPOOL: kept layers and arcs as pools (run_dd: dynl)
    std::vector<uint64_t> cache_tab;
    std::vector<uint64_t> pvr;       // pooled engine behind a cache: (value, rub) per event record (EngineParams::s_pvr)
    std::vector<uint64_t> pst;       // ... and the state it was expanded with (EngineParams::s_pst)
    std::vector<uint64_t> dom_coord;
    std::vector<int32_t> dom_value;
    std::vector<uint32_t> dom_count, dom_lock;
    unsigned long long dom_stats[8] = {0};
    unsigned long long cache_stats[8] = {0};
};

template <class T>
T* carve(unsigned char*& p, size_t count) {
    T* r = (T*)p;
    p += (count * sizeof(T) + 15) & ~(size_t)15;
    return r;
}

int pick_ws(int ws) {
    const int opts[] = {1, 2, 4, 7, 8, 16, 32, 72};
    for (int o : opts)
        if (ws <= o) return o;
    return -1;
}

template <int WS>
void run(Emul& e, const DDInput& in, DDResult* res2) {
    if (e.engine == 2) {
        DD2Ctx<WS> c2;
        dd2_bind<WS>(c2, e.P, 0, e.lds2.data(), e.nthreads);
        if (e.pooled) run_work_item2<WS, 0, 1>(c2, in, res2);   // Pooled decision diagrams (mdd/pooled.rs) out of the same node slots
        else run_work_item2<WS>(c2, in, res2);
        return;
    }
    DDCtx<WS> c;
    dd_bind<WS, true>(c, e.P, 0, e.lds.data(), e.nthreads);
    dd_stage_tables<WS>(c, e.P);
    run_work_item<WS>(c, in, res2);
}
}  // namespace

/// what freshly "allocated" workspace memory holds (hipMalloc does not clear): 0xCD by default; DDO_EMU_POISON picks another byte so that a
/// read of memory the kernel never wrote shows whatever its sign or size (tools/diag/emu_order.sh runs the suites with 0x00 / 0x01 / 0x7F / 0xFF)
static unsigned char emu_poison() {
    static const int v = [] { const char* e = std::getenv("DDO_EMU_POISON"); return e ? (int)std::strtol(e, nullptr, 0) & 0xFF : 0xCD; }();
    return (unsigned char)v;
}
static void* emul_finish(Emul* e, int n, int wsT, bool unit, int max_width, int nthreads, uint64_t arena_bytes, int capN_,
                         const int64_t* weights, int fan = 2);

extern "C" {

void* emul_create(int n, const uint64_t* adj_rows, const int64_t* weights, int max_width, int nthreads,
                  uint64_t arena_bytes, int engine) {
    Emul* e = new Emul();
    std::memset(&e->P, 0, sizeof(e->P));
    e->engine = engine;
    int ws = (n + 63) / 64;
    int wsT = pick_ws(ws);
    if (wsT < 0) return nullptr;
    e->wsT = wsT;
    e->nthreads = nthreads;
    e->adj.assign((size_t)n * wsT, 0);
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < ws; ++k) e->adj[(size_t)i * wsT + k] = adj_rows[(size_t)i * ws + k];
    e->weight.resize(n);
    bool unit = true;
    for (int i = 0; i < n; ++i) {
        e->weight[i] = (int32_t)weights[i];
        unit &= weights[i] == 1;
    }
    void* h = emul_finish(e, n, wsT, unit, max_width, nthreads, arena_bytes, max_width + 2, weights);
    if (h && engine == 2 && std::getenv("DDO_EMUL_TIER") && !std::getenv("DDO_EMUL_DENSE")) e->P.tier = 1;   // capacity tier: max_width is the layer capacity, widths above it are accepted
    return h;
}

/// workspace and capacities for one slot (what Engine::init does on the device side)
}  // extern "C"

static void* emul_finish(Emul* e, int n, int wsT, bool unit, int max_width, int nthreads, uint64_t arena_bytes, int capN_,
                         const int64_t* weights, int fan) {
    EngineParams& P = e->P;
    P.n = n;
    P.ws = wsT;
    P.unit_weights = unit;
    P.npad = (n + 63) / 64 * 64;
    P.adj = e->adj.data();
    P.weight = e->weight.data();
    P.capN = capN_;
    P.fan = fan;
    P.dbits = fan > 2 ? 6 : 1;
    P.capC1 = fan * P.capN + 1;
    P.max_layers = n + 2;
    int tc = 1024;
    while (tc * 2 < P.capC1 * 3 || tc < 2 * P.capN) tc <<= 1;
    P.table_cap = tc;
    P.table_in_lds = 1;
    P.nslots = 1;
    const size_t capC1 = P.capC1, capN = P.capN, ml = P.max_layers;
    const size_t LSm = capC1 + 1;   // nodes per kept layer (emul_set_keep_layers switches P.tmode / P.lstride on)
    size_t bytes = 2 * wsT * capC1 * 8 + 2 * capC1 * 8 + 2 * capC1 * 4 * 2 + (size_t)fan * capN * 4 + capC1 * 4 + capC1 * 4 + capC1 +
                   ml * LSm * 4 + 2 * ml * (size_t)fan * capN * 4 + ml * 5 * 4 + wsT * capN * 8 + capN * 8 + 64 * 32;
    e->mem.assign(bytes, emu_poison());  // poison
    unsigned char* p = e->mem.data();
    P.cstate = carve<uint64_t>(p, 2 * wsT * capC1);
    P.ckey = carve<uint64_t>(p, 2 * capC1);
    P.cpop = carve<uint32_t>(p, 2 * capC1);
    P.cflags = carve<uint32_t>(p, 2 * capC1);
    P.ctarget = carve<uint32_t>(p, (size_t)fan * capN);
    P.keep = carve<uint32_t>(p, capC1);
    P.posmap = carve<uint32_t>(p, capC1);
    P.cls = carve<uint8_t>(p, capC1);
    P.ninfo = carve<uint32_t>(p, ml * LSm);
    P.arct = carve<uint32_t>(p, ml * (size_t)fan * capN);
    P.arcc = carve<int32_t>(p, ml * (size_t)fan * capN);
    P.nlayer = carve<int32_t>(p, ml);
    P.lntot = carve<int32_t>(p, ml);
    P.lvar = carve<int32_t>(p, ml);
    P.ldup = carve<int32_t>(p, ml * 2);
    P.cs_state = carve<uint64_t>(p, wsT * capN);
    P.cs_value = carve<int32_t>(p, capN);
    P.cs_pop = carve<uint32_t>(p, capN);
    if ((size_t)(p - e->mem.data()) > bytes) {
        std::fprintf(stderr, "emul: workspace overflow\n");
        std::abort();
    }
    {
        size_t b3 = ml * LSm * ((size_t)wsT * 8 + 4 * 4) + capC1 * 4 + 64 * 8;
        e->mem3.assign(b3, emu_poison());
        unsigned char* r = e->mem3.data();
        P.lstate = carve<uint64_t>(r, ml * LSm * wsT);
        P.lval = carve<int32_t>(r, ml * LSm);
        P.lrub = carve<int32_t>(r, ml * LSm);
        P.lvb = carve<int32_t>(r, ml * LSm);
        P.lth = carve<int32_t>(r, ml * LSm + capC1);
        if ((size_t)(r - e->mem3.data()) > b3) {
            std::fprintf(stderr, "emul: workspace3 overflow\n");
            std::abort();
        }
        P.tmode = 0;
        P.lstride = (int32_t)LSm;
    }
    e->lds.assign(dd_lds_bytes(P.table_cap, P.npad, nthreads, P.tw_lds ? tw_lds_words(n) : 0), 0xEE);
    e->arena.assign(arena_bytes, 0);
    P.arena = e->arena.data();
    P.arena_cap = arena_bytes;
    P.arena_head = &e->arena_head;
    P.work_counter = &e->work_counter;
    P.cutoff_flag = nullptr;
    // ---- in-place engine workspace
    P.capS = 2 * max_width + 8;
    P.capW = P.capN;
    int t2 = 1024;
    while (t2 < 3 * P.capW) t2 <<= 1;
    P.tab2_cap = t2;
    long long neg = 0;
    for (int i = 0; i < n; ++i) if (weights[i] < 0) neg += weights[i];
    P.vbase_off = (int32_t)neg;
    P.lex_cap = 1024;
    if (const char* env = std::getenv("DDO_EMUL_LEX_CAP")) P.lex_cap = std::max(1, std::min(1024, std::atoi(env)));
    P.hist_bins = 2048;
    if (const char* env = std::getenv("DDO_EMUL_DENSE")) {   // dense tier (two workgroups per CU): small table, 8-bit select digits, short tie lists in LDS
        P.tier = 2;
        P.hist_bins = 256;
        P.lex_cap = std::min(P.lex_cap, 128);
        P.tab2_cap = std::min(P.tab2_cap, std::max(64, std::atoi(env)));
        if ((long)P.tab2_cap * 7 / 8 < P.capW + 8) {
            std::fprintf(stderr, "emul: dense table too small for the layer capacity\n");
            std::abort();
        }
    }
    if (const char* env = std::getenv("DDO_EMUL_KEYS_GLOBAL")) P.keys_global = std::atoi(env) != 0;   // keys packed with the hashes in "HBM" (the dense / full-width kernels at large widths)
    P.ev_cap = ((uint64_t)P.max_layers * (uint64_t)(5 * P.capW + 16) + 2ull * P.capW + 64 + 3) & ~3ull;
    {
        const size_t capS = P.capS, capW = P.capW, mlz = P.max_layers;
        const size_t RW = ((wsT + 1 + 7) / 8) * 8, PR = ((wsT + 7) / 8) * 8;
        size_t b2 = wsT * capS * 8 + capS * (RW + PR + 1) * 8 + capW * 4 + P.ev_cap * 4 + P.ev_cap * 2 + 64 + mlz * 8 * 4 + capW * 4 +
                    wsT * capW * 8 + 64 * 16;
        e->mem2.assign(b2, emu_poison());
        unsigned char* q = e->mem2.data();
        P.s_state = carve<uint64_t>(q, wsT * capS);
        P.s_rec = carve<uint64_t>(q, capS * RW);
        P.s_ptree = carve<uint64_t>(q, P.ev_cap / 4 + 1);
        P.s_hash = carve<uint64_t>(q, capS);
        P.s_wl = carve<uint16_t>(q, 2 * capW);
        P.s_ev = carve<uint32_t>(q, P.ev_cap);
        P.s_evoff = carve<uint32_t>(q, mlz * 8);
        P.s_cs_slot = carve<uint32_t>(q, capW);
        P.s_cs_path = carve<uint64_t>(q, wsT * capW);
        if ((size_t)(q - e->mem2.data()) > b2) {
            std::fprintf(stderr, "emul: workspace2 overflow\n");
            std::abort();
        }
        e->lds2.assign(dd2_lds_bytes(P.capS, P.tab2_cap, P.npad, nthreads, true, P.hist_bins, P.lex_cap, wsT), 0xEE);
    }
    return e;
}

extern "C" {
/// Engine-1 emulation for ANY model descriptor of the product library (`ddo_model*` from ddo_model_create_*): the tables are
/// read straight out of the host-side Model, like Engine::init uploads them.
void* emul_create_model(const void* model_handle, int max_width, int nthreads, uint64_t arena_bytes) {
    const ddo_hip::Model& M = ((const ddo_model*)model_handle)->m;
    Emul* e = new Emul();
    std::memset(&e->P, 0, sizeof(e->P));
    e->engine = 1;
    e->wsT = M.wsT;
    e->nthreads = nthreads;
    e->adj.assign((size_t)M.n * M.wsT, 0);
    if (M.kind == MODEL_MISP)
        for (int i = 0; i < M.n; ++i)
            for (int k = 0; k < M.ws; ++k) e->adj[(size_t)i * M.wsT + k] = M.adj[(size_t)i * M.ws + k];
    e->weight.resize(M.n);
    for (int i = 0; i < M.n; ++i) e->weight[i] = (int32_t)M.weight[i];
    EngineParams& P = e->P;
    P.model_kind = M.kind;
    if (M.kind == MODEL_KNAPSACK) {
        e->aux[0].assign(M.kp_weight.begin(), M.kp_weight.end());
        e->aux[1].assign(M.kp_order.begin(), M.kp_order.end());
        P.kp_weight = e->aux[0].data();
        P.kp_order = e->aux[1].data();
    }
    if (M.kind == MODEL_MCP || M.kind == MODEL_MAX2SAT) {
        P.vest = M.vest.data();
        P.vnk = M.vnk.data();
        P.vr = (int32_t)M.initial_value;
    }
    if (M.kind == MODEL_MCP) P.vgraph = M.vgraph.data();
    if (M.kind == MODEL_TSPTW) {
        P.tw_dist = M.tw_dist.data();
        P.tw_early = M.tw_early.data();
        P.tw_late = M.tw_late.data();
        P.tw_cheap = M.tw_cheap.data();
        P.tw_order = M.tw_order.data();
        P.tw_lds = std::getenv("EMUL_TW_GLOBAL") ? 0 : 1;   // the tables in (emulated) LDS, as Engine::init chooses when they fit
    }
    if (M.kind == MODEL_MAX2SAT) {
        P.m2_wtt = M.m2_w[0].data();
        P.m2_wtf = M.m2_w[1].data();
        P.m2_wft = M.m2_w[2].data();
        P.m2_wff = M.m2_w[3].data();
        P.m2_order = M.m2_order.data();
        P.m2_rankpos = M.m2_rankpos.data();
    }
    e->lddelta.assign((size_t)M.n + 2, 0);
    P.lddelta = e->lddelta.data();
    const int capN = M.kind == MODEL_TSPTW ? std::max(max_width, M.n) + 2
                     : M.kind != MODEL_MISP ? 2 * max_width + 3 : max_width + 2;   // the terminal layer is never squashed
    return emul_finish(e, M.n, M.wsT, M.unit_weights, max_width, nthreads, arena_bytes, capN, M.weight.data(), M.kind == MODEL_TSPTW ? M.n : 2);
}
/// keep every layer (frontier cut-set, thresholds, cache) on / off; cache_entries > 0 (re)creates an EMPTY cache table
void emul_set_keep_layers(void* h, int on, uint64_t cache_entries) {
    Emul* e = (Emul*)h;
    e->P.tmode = on ? 1 : 0;
    e->P.lpool_nodes = e->P.apool_arcs = 0;
    if (on && std::getenv("DDO_EMUL_LPOOL")) {   // the same arrays, used as pools: layer X starts where layer X - 1 ended
        const size_t ml = (size_t)e->P.max_layers;
        e->P.lpool_nodes = ml * (size_t)e->P.lstride;
        e->P.apool_arcs = ml * (size_t)(e->P.fan > 2 ? e->P.fan : 2) * (size_t)e->P.capN;
        // tests of the overflow paths: pools SMALLER than what a DD needs (the arrays behind them keep their full size)
        if (const char* v = std::getenv("DDO_EMUL_LPOOL_NODES")) e->P.lpool_nodes = std::min<uint64_t>(e->P.lpool_nodes, std::strtoull(v, nullptr, 10));
        if (const char* v = std::getenv("DDO_EMUL_APOOL_ARCS")) e->P.apool_arcs = std::min<uint64_t>(e->P.apool_arcs, std::strtoull(v, nullptr, 10));
        e->lbase.assign(ml + 1, 0);
        e->abase.assign(ml + 1, 0);
        e->P.lbase = e->lbase.data();
        e->P.abase = e->abase.data();
    }
    e->P.cache_cap = 0;
    e->P.cache_tab = nullptr;
    if (on && cache_entries) {
        uint64_t cap = 1024;
        while (cap < 2 * cache_entries) cap <<= 1;   // (new entries up to half the slots: CacheTable::create does the same)
        e->P.cache_stride = 3 + e->wsT;
        e->cache_tab.assign(cap * (size_t)e->P.cache_stride, 0);
        e->P.cache_tab = e->cache_tab.data();
        e->P.cache_cap = cap;
        e->cache_stats[0] = e->cache_stats[1] = 0;
        e->P.cache_stats = e->cache_stats;
    }
}
uint64_t emul_cache_used(void* h) { return ((Emul*)h)->cache_stats[0]; }
/// a fresh, empty SimpleDominanceChecker (per depth `cap` pairs); cap 0 removes it
void emul_set_dominance(void* h, uint32_t cap) {
    Emul* e = (Emul*)h;
    e->P.dom_cap = 0;
    e->P.dkey_cap = 0;
    if (!cap) return;
    if (e->P.model_kind == MODEL_TSPTW) {   // TsptwDominance: one hash table of (depth, position, must_visit) keys
        uint64_t c2 = 1024;
        while (c2 < cap) c2 <<= 1;
        e->dom_coord.assign(c2 * (size_t)(3 + 2 * tw_set_words(e->P.n) + 1), 0);   // entry = 3 words + the (2K + 1)-word key
        e->P.dkey_tab = e->dom_coord.data();
        e->P.dkey_cap = c2;
        e->dom_stats[0] = e->dom_stats[1] = 0;
        e->P.dkey_stats = e->dom_stats;
        return;
    }
    e->P.dom_cap = cap;
    const size_t nd = (size_t)e->P.max_layers;
    e->dom_coord.assign(nd * cap, 0);
    e->dom_value.assign(nd * cap, 0);
    e->dom_count.assign(nd, 0);
    e->dom_lock.assign(nd, 0);
    e->P.dom_coord = e->dom_coord.data();
    e->P.dom_value = e->dom_value.data();
    e->P.dom_count = e->dom_count.data();
    e->P.dom_lock = e->dom_lock.data();
    e->dom_stats[0] = 0;
    e->P.dom_stats = e->dom_stats;
}
/// engine 2 only: compile Pooled decision diagrams (run_dd2<WS, DEEP, POOLED = 1>)
void emul_set_pooled(void* h, int on) { ((Emul*)h)->pooled = on != 0; }
/// Pooled decision diagrams behind a fresh, empty SimpleCache of at least `entries` entries (0: EmptyCache again)
void emul_set_pooled_cache(void* h, uint64_t entries) {
    Emul* e = (Emul*)h;
    e->P.cache_cap = 0;
    e->P.cache_tab = nullptr;
    e->P.s_pvr = nullptr;
    e->P.s_pst = nullptr;
    e->P.pst_cap = 0;
    if (!entries) return;
    uint64_t cap = 1024;
    while (cap < 2 * entries) cap <<= 1;
    e->P.cache_stride = 3 + e->wsT;
    e->cache_tab.assign(cap * (size_t)e->P.cache_stride, 0);
    e->P.cache_tab = e->cache_tab.data();
    e->P.cache_cap = cap;
    e->cache_stats[0] = e->cache_stats[1] = 0;
    e->P.cache_stats = e->cache_stats;
    e->pvr.assign((size_t)(e->P.ev_cap / 4 + 1), 0x0101010101010101ULL * emu_poison());
    e->P.s_pvr = e->pvr.data();
    e->P.pst_cap = (uint32_t)std::min<uint64_t>(e->P.ev_cap / 4 + 1, 1u << 18);
    e->pst.assign((size_t)e->P.pst_cap * (size_t)e->wsT, 0x0101010101010101ULL * emu_poison());
    e->P.s_pst = e->pst.data();
}
void emul_destroy(void* h) { delete (Emul*)h; }
int emul_state_words(void* h) { return ((Emul*)h)->wsT; }
uint64_t emul_lds_bytes(void* h) { return ((Emul*)h)->lds.size(); }

/// Compiles one sub-problem.  res2: two DDResult records; the arena is reset first and its
/// base pointer returned through *arena_out.
int emul_compile(void* h, const DDInput* in, DDResult* res2, const uint8_t** arena_out) {
    Emul* e = (Emul*)h;
    e->arena_head = 0;
    std::memset(res2, 0, 2 * sizeof(DDResult));
    if (in->width + 2 > e->P.capN && !e->P.tier) return -3;
    switch (e->wsT) {
        case 1: run<1>(*e, *in, res2); break;
        case 2: run<2>(*e, *in, res2); break;
        case 4: run<4>(*e, *in, res2); break;
        case 7: run<7>(*e, *in, res2); break;
        case 8: run<8>(*e, *in, res2); break;
        case 16: run<16>(*e, *in, res2); break;
        case 32: run<32>(*e, *in, res2); break;
        case 72: run<72>(*e, *in, res2); break;
        default: return -2;
    }
    *arena_out = e->arena.data();
    return 0;
}
uint64_t emul_sizeof_input(void) { return sizeof(DDInput); }
uint64_t emul_sizeof_result(void) { return sizeof(DDResult); }
}
