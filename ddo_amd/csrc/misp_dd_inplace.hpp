// =============================================================================
// misp_dd_inplace.hpp -- second-generation device engine: IN-PLACE layers.
//
// Same contract and same results as misp_dd_core.hpp (clean.rs:345-876 with the
// MISP callbacks of examples/misp/main.rs:62-209), different data structure.
//
// Observation (rocprof, profiles/r01): `next_variable` (main.rs:109-143) picks the
// vertex that occurs in the FEWEST states of the layer, so ~93 % of the nodes of a
// layer do not contain the branching vertex: their only child is the NO-child with
// the very same state, value and best path (transition(s, NO) == s, cost 0).  The
// first engine copied them anyway (read 56 B, write 56 B, re-hash, re-select: 8.5x
// the algorithmic HBM traffic).  Here a node lives in a persistent SLOT:
//   * an unaffected node is not touched at all: it simply stays alive;
//   * an affected node s is updated in place (NO-child = s minus the vertex: one word
//     rewritten, hash patched incrementally) and spawns its YES-child into a free slot;
//   * the dedup table lives in LDS and is refilled per layer by the same sweep that builds
//     the work list; children are de-duplicated right where they are created;
//   * the ranking keys (value, popcount) are one u32 per slot (LDS, or L2 at width 10 000
//     where the table needs the LDS); the exact top-K selection of _restrict/_relax is a
//     radix select over them plus a rank-counting tie-break on member order;
//   * the best path to a node is a bit string (one decision bit per layer) stored with
//     the node, copied only when a node is created or its best parent changes -- no
//     per-layer parent arrays, no path walks;
//   * the backward pass for local bounds (clean.rs:448-475) replays per-transition EVENT
//     lists (affected parents, deleted nodes, merged node) instead of per-layer arc arrays:
//     an unaffected node keeps its value_bot without any work.
// Written in the same PAR_BEGIN/PAR_END phase style, so tests/ can run it as a lock-step
// host emulation.
// =============================================================================
#pragma once
#include <cstddef>
#include "misp_dd_core.hpp"

namespace ddo_hip {

constexpr uint32_t T2_EMPTY = 0xFFFFFFFFu;
constexpr uint32_t T2_TOMB = 0xFFFFFFFEu;
constexpr uint32_t EV_CREATED = 0x80000000u;  // flag on a YES target: the slot was created by this arc
constexpr uint32_t EV_RAISED = 0x40000000u;   // flag on a target: the arc raised its target's key (see the event records of run_dd2)
constexpr uint32_t EV_SLOT_MASK = 0x000FFFFFu;
constexpr uint32_t EV_PINEX = 0x40000000u;    // POOLED, flag on the PARENT word of a record: the parent was inexact when it was expanded
constexpr uint32_t EV_PCUT = 0x20000000u;     // POOLED, backward pass: the parent belongs to the frontier cut-set
constexpr int KEY_POP_BITS = 11;              // key32 = (value - vbase) << 11 | popcount
constexpr uint32_t KEY_POP_MASK = (1u << KEY_POP_BITS) - 1;

#if defined(DDO_HOST_EMULATION)
#define LDS_OR_U32(p, v) emu_atomic_or<uint32_t>((p), (v))
#define LDS_AND_U32(p, v) emu_atomic_and<uint32_t>((p), (v))
#define LDS_MAX_U32(p, v) emu_atomic_max<uint32_t>((p), (v))
#define GLB_ST_U32(p, v) (*(p) = (v))
#else
#define LDS_OR_U32(p, v) __hip_atomic_fetch_or((p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LDS_AND_U32(p, v) __hip_atomic_fetch_and((p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LDS_MAX_U32(p, v) __hip_atomic_fetch_max((p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define GLB_ST_U32(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif

#if defined(DDO_HOST_EMULATION)
#define DD_UNIFORM(x) (x)
#define DD_UNIFORM64(x) (x)
#else
// Workgroup-uniform scalars read from LDS land in VGPRs; at the 128-VGPR cap of 1024-thread workgroups every
// long-lived one costs a spill somewhere.  readfirstlane moves them to SGPRs (and makes the loads they index scalar).
#define DD_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
DDO_DEV uint64_t dd_uniform64(uint64_t x) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
}
#define DD_UNIFORM64(x) dd_uniform64(x)
#endif
// phase ids of DDResult::phase_clk
constexpr int PH_VAR = 0, PH_SELECT = 1, PH_VICTIMS = 3, PH_WORKLIST = 0, PH_FREELIST = 0, PH_EXPAND = 7, PH_FINAL = 0, PH_BACKWARD = 0;
constexpr int PH_SELLEX = 2, PH_EXP1 = 4, PH_TABLE = 5, PH_EXP2 = 6;

// Ranking key and state hash of a node travel together: key32 = (value - vbase) << 11 | popcount, h32 = the 32 bits of the
// state hash the dedup table needs (slot and tag).  The keys live in LDS when they fit, else -- to run two workgroups per CU
// at large widths -- in HBM (L2-resident), PACKED with the hash into one 64-bit word per node (key32 << 32 | h32): a node
// that is created or changed in place then costs ONE 8-byte store for both (rocprof, round 2: the expand phase is bound by the
// NUMBER of partial-line write requests, not by bytes), and the atomicMax of a twin's key is a 64-bit one (equal states have
// equal hashes, so the low half never decides).  With the keys in LDS the hashes are a u32 array in HBM.  All accesses go
// through the wrappers below (agent-scope loads/stores so that L2 atomics and plain accesses never mix in the L1).
#if defined(DDO_HOST_EMULATION)
#define KH_LD64(p) (*(p))
#define KH_ST64(p, v) (*(p) = (v))
#define KH_MAX64(p, v) emu_atomic_max<uint64_t>((p), (v))
#define KL_LD32(p) (*(p))
#define KL_ST32(p, v) (*(p) = (v))
#define KL_MAX32(p, v) emu_atomic_max<uint32_t>((p), (v))
#else
#define KH_LD64(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define KH_ST64(p, v) __hip_atomic_store((p), (uint64_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define KH_MAX64(p, v) __hip_atomic_fetch_max((p), (uint64_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define KL_LD32(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define KL_ST32(p, v) __hip_atomic_store((p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define KL_MAX32(p, v) __hip_atomic_fetch_max((p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#endif
#define K32(c, i) k32_ld((c), (i))
#define KH(c, i) kh_ld((c), (i))
#define KH_ST(c, i, key, h32v) kh_st((c), (i), (key), (h32v))
#define K32_ST(c, i, key) k32_st((c), (i), (key))
#define K32_MAX(c, i, key, h32v) k32_max((c), (i), (key), (h32v))

struct DD2Shared {
    int32_t work, status, cutoff;
    uint32_t varkey;
    int32_t nlive;
    int32_t nwl, nwl2, nrec, nnew, nvict, nfl;
    int32_t npruned, nyes, ndup;
    int32_t scan_total, sel_digit, sel_above, sel_bucket, sel_need;
    int32_t tab_used;
    int32_t hiw;            // every live node sits in a slot below hiw (recomputed per layer: the sweeps stop there)
    int32_t merged_slot, recycled, xslot, free_slot;
    int32_t ncut, ncut2;
    uint32_t recycled_merges;
    int32_t maxn;
    uint32_t kbits_and[2], kbits_or[2];   // AND / OR of the keys of layer L in slot L & 1: gathered while layer L is built (sweep + expand of L - 1)
    uint32_t pivKey;
    uint64_t land, lor;
    uint32_t gs[64];
    uint64_t pivLex[1];
    uint64_t mergedKey;     // key32 << 32 | slot of the best victim
    uint64_t bestKey, bestExactKey;
    uint64_t nodes, arcs;
    uint64_t arena_off;
    uint64_t ev_pos;
    uint64_t clk[8];
    uint64_t clk_last;
    uint64_t mk[24];        // statistics and auxiliary tick slots (DD2_STAT, DD2_TICK2), [16..24): DD2_PROBE
    int32_t xcand[64];
    uint64_t merged[MAX_WS];   // LAST: a workgroup only allocates the WS words it uses (dd2_shared_bytes)
};

/// LDS bytes of the shared block of a WS-word engine
DD_HD inline size_t dd2_shared_bytes(int ws) { return (offsetof(DD2Shared, merged) + (size_t)ws * 8 + 15) & ~(size_t)15; }

// DD2_STAT(k, v): profiling statistics (DDO_HIP_STATS) accumulated next to the code marks; workgroup-uniform context
#define DD2_STAT(k, v)                                      \
    if (c.clocks) {                                         \
        PAR_BEGIN                                           \
        if (tid == 0) sh->mk[k] += (uint64_t)(v);           \
        PAR_END                                             \
    }
// DD2_TICK2(k): like DD2_TICK but the elapsed ticks are ALSO charged to the auxiliary slot mk[k] (finer breakdown)
#define DD2_TICK2(ph, k)                                    \
    if (c.clocks) {                                         \
        PAR_BEGIN                                           \
        if (tid == 0) {                                     \
            const uint64_t _t = dd_clock();                 \
            sh->clk[ph] += _t - sh->clk_last;               \
            sh->mk[k] += _t - sh->clk_last;                 \
            sh->clk_last = _t;                              \
        }                                                   \
        PAR_END                                             \
    }
// DD2_PROBE(k): thread 0 times its own node inside expand (all its memory operations drained first): the dependent chain
#if defined(DDO_HOST_EMULATION)
#define DD_DRAIN() (void)0
#else
#define DD_DRAIN() __builtin_amdgcn_s_waitcnt(0)
#endif
// Compiled in only with -DDDO_HIP_PROBES (make PROBES=1): the probe registers cost 6 % at the 128-VGPR cap.
#if defined(DDO_HIP_PROBES)
#define DD2_PROBE(k)                                        \
    if (probing) {                                          \
        DD_DRAIN();                                         \
        const uint64_t _t = dd_clock();                     \
        probe[k] = _t - probe_t;                            \
        probe_t = _t;                                       \
    }
#else
#define DD2_PROBE(k)
#endif
#define DD2_TICK(ph)                                        \
    if (c.clocks) {                                         \
        PAR_BEGIN                                           \
        if (tid == 0) {                                     \
            const uint64_t _t = dd_clock();                 \
            sh->clk[ph] += _t - sh->clk_last;               \
            sh->clk_last = _t;                              \
        }                                                   \
        PAR_END                                             \
    }

// Pointers into HBM carry their address space like the LDS ones: the context lives in LDS (round 4), and a pointer loaded
// from memory whose address space is unknown turns every access through it into a FLAT instruction.
#if defined(DDO_HOST_EMULATION)
#define GLB_PTR(T) T*
#else
#define GLB_PTR(T) __attribute__((address_space(1))) T*
#endif

template <int WS>
struct DD2Ctx {
    int n, npad, unit_weights;
    GLB_PTR(const uint64_t) adj;
    GLB_PTR(const int32_t) weight;
    int capS, capW, max_layers, nbw;
    // HBM, per engine slot
#if defined(DDO_WORD_MAJOR)
    GLB_PTR(uint64_t) st;      // [ws][capS]   word-major copy of the states: only the streaming "who contains v" scan reads it
#endif
    GLB_PTR(uint64_t) rec;     // [capS][RW]   node records, one (or two) 64-byte lines each: state words, then the cached hash
    GLB_PTR(uint64_t) pt;      // [ev_cap / 4]  path tree: one entry per event record, parent path id | layer << 32 (see PID_NONE)
    GLB_PTR(uint64_t) keyh;    // [capS]       key32 << 32 | h32 of every node when the keys live in HBM (else nullptr): streamed by the
                       //              select sweeps and by the per-layer table rebuild
    GLB_PTR(uint32_t) h32;     // [capS]       h32 of every node when the keys live in LDS (else nullptr)
    int RW;
    LDS_PTR(uint32_t) tab;
    int tab_cap;
    GLB_PTR(uint32_t) ev;
    uint64_t ev_cap;
    GLB_PTR(uint32_t) evoff;   // [max_layers][8]: aff_off_lo, aff_off_hi, n_aff, del_off_lo, del_off_hi, n_del, merged|dup.., var
    GLB_PTR(int32_t) lvar;
    GLB_PTR(int32_t) ldup;     // [max_layers][2] (dup from, dup to)
    GLB_PTR(int32_t) lmerge;   // [max_layers]   merged slot (-1 none)
    GLB_PTR(uint32_t) cs_slot;
    GLB_PTR(uint64_t) cs_state;
    GLB_PTR(uint32_t) cs_pid;  // [capW]        path ids of the snapshot's nodes
    GLB_PTR(int32_t) cs_value;
    GLB_PTR(uint32_t) cs_pop;
    // LDS
    LDS_PTR(uint32_t) key32;   // capS   keys in LDS (nullptr when they are packed into keyh); the backward pass reuses the key storage for value_bot
    LDS_PTR(uint32_t) live;    // nbw
    LDS_PTR(uint32_t) inex;    // nbw
    LDS_PTR(uint32_t) okb;     // nbw
    LDS_PTR(uint32_t) fresh;   // nbw
    LDS_PTR(int32_t) cnt;      // npad
    LDS_PTR(uint32_t) hist;    // 2048
    uint16_t* wl;      // capW
    uint16_t* fl;      // capW   (wl+fl together are reused as int32 tmp[capW] in the backward pass)
    LDS_PTR(int32_t) tcount;   // NT
    LDS_PTR(int32_t) tcount2;  // NT
    LDS_PTR(DD2Shared) sh;
    GLB_PTR(uint8_t) arena;
    uint64_t arena_cap;
    GLB_PTR(unsigned long long) arena_head;
    GLB_PTR(const int32_t) cutoff_flag;
    GLB_PTR(uint8_t) pool;
    uint64_t pool_cap;
    GLB_PTR(unsigned long long) pool_head;
    int vbase_off;
    int NT;
    int lex_cap;
    int hist_bins;   // bins of the select histogram: 2048 = digits of 10/11/11 bits, less = 8-bit digits
    int tab_limit;   // entries the dedup table may hold (a dense tier's table is smaller than 3 x the layer capacity)
    int clocks;        // per-phase shader-clock accounting (DDO_HIP_STATS): costs one barrier per phase
    int tier;          // capacity tier: no squash phases (their LDS is not there), capacity errors mean ST_RETRY
    // Pooled decision diagrams behind a SimpleCache (pooled.rs:467-535, 662-680; dd_thresholds.hpp): the table shared by the compiles of
    // a solver, and per event record the (value_top, rough upper bound) its node had when it was expanded (the threshold pass needs both
    // for nodes whose state is gone by then)
    uint64_t* cache_tab;
    uint64_t cache_cap;
    unsigned long long* cache_stats;
    GLB_PTR(uint64_t) pvr;     // [ev_cap / 4]  value_top (low half) | rub (high half) per event record; nullptr without a cache
    GLB_PTR(uint64_t) pst;     // [pst_cap][WS] the state every expanded node had WHEN it was expanded (its slot is rewritten in place): what
    uint32_t pst_cap;          //              _maybe_update_cache writes for the exact ones; records beyond pst_cap end the compile (capacity)
    int cache_stride;
    int depth0;                // depth of the residual sub-problem (node.depth = depth0 + layer)
};

template <int WS> DDO_DEV uint32_t k32_ld(const DD2Ctx<WS>& c, int i) {
    return c.keyh ? (uint32_t)(KH_LD64(&c.keyh[i]) >> 32) : KL_LD32(&c.key32[i]);
}
template <int WS> DDO_DEV uint64_t kh_ld(const DD2Ctx<WS>& c, int i) {
    return c.keyh ? KH_LD64(&c.keyh[i]) : (((uint64_t)KL_LD32(&c.key32[i]) << 32) | (uint64_t)c.h32[i]);
}
/// a node is created, or changed in place: key and hash
template <int WS> DDO_DEV void kh_st(const DD2Ctx<WS>& c, int i, uint32_t key, uint32_t h32v) {
    if (c.keyh) KH_ST64(&c.keyh[i], ((uint64_t)key << 32) | (uint64_t)h32v);
    else {
        KL_ST32(&c.key32[i], key);
        c.h32[i] = h32v;
    }
}
/// the key of an unchanged state
template <int WS> DDO_DEV void k32_st(const DD2Ctx<WS>& c, int i, uint32_t key) {
    if (c.keyh) KH_ST64(&c.keyh[i], ((uint64_t)key << 32) | (KH_LD64(&c.keyh[i]) & 0xFFFFFFFFULL));
    else KL_ST32(&c.key32[i], key);
}
/// a twin arrives at node i (same state, hence same h32): the key keeps the maximum; returns the old key
template <int WS> DDO_DEV uint32_t k32_max(const DD2Ctx<WS>& c, int i, uint32_t key, uint32_t h32v) {
    if (c.keyh) return (uint32_t)(KH_MAX64(&c.keyh[i], ((uint64_t)key << 32) | (uint64_t)h32v) >> 32);
    return KL_MAX32(&c.key32[i], key);
}

template <class BP> DDO_DEV bool bm_test(BP bm, int s) { return (bm[s >> 5] >> (s & 31)) & 1u; }
template <class BP> DDO_DEV void bm_set(BP bm, int s) { LDS_OR_U32(&bm[s >> 5], 1u << (s & 31)); }
template <class BP> DDO_DEV void bm_clr(BP bm, int s) { LDS_AND_U32(&bm[s >> 5], ~(1u << (s & 31))); }
template <class BP> DDO_DEV void bm_put(BP bm, int s, bool v) { if (v) bm_set(bm, s); else bm_clr(bm, s); }

/// per-word mix of the incremental state hash: H(state) = XOR_k mixw(word_k, k)
DDO_DEV uint64_t mixw(uint64_t w, int k) {
    uint64_t x = w + 0x9E3779B97F4A7C15ULL * (uint64_t)(k + 1);
    x ^= x >> 30;
    x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27;
    x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    return x;
}
template <int WS>
DDO_DEV uint64_t hash2_state(const uint64_t* s) {
    uint64_t h = 0;
#pragma unroll
    for (int k = 0; k < WS; ++k) h ^= mixw(s[k], k);
    return h;
}
/// the 32 bits of the hash that are kept per node: XOR-linear in H, so a one-word change patches it incrementally
DDO_DEV uint32_t fold32(uint64_t h) { return (uint32_t)(h ^ (h >> 32)); }
template <int WS>
DDO_DEV uint32_t hash32_state(const uint64_t* s) { return fold32(hash2_state<WS>(s)); }

/// Node records are array-of-structures: a random access to a node costs one 64-byte line instead of one line
/// per state word (rocprof showed the expand phase bound by random 8-byte requests, not by bytes).
struct alignas(16) U64x2 {
    uint64_t a, b;
};
struct alignas(16) U32x4 {
    uint32_t x, y, z, w;
};
/// state words (+ the cached hash) of a node with 16-byte loads: records are 64-byte aligned, and one thread reading
/// its record with 8-byte loads costs 2.5x more than with 16-byte ones (tools/micro/recload.hip: 22.5 vs 8.9 kcycles
/// per 1024 records)
/// Best paths are kept BY REFERENCE.  In-place layers make a path a very regular thing: a node that stays (the NO-child) keeps
/// its path, a new node (the YES-child created by the arc of event record e) has its parent's path plus decision 1 at the
/// layer of e.  So a path is a chain of event records -- the layers at which its 1-decisions were taken -- and a node only
/// stores the id of the last one, in the spare word of its record line: `pt[e]` = (path id of the parent when e was written)
/// | layer << 32.  Event records are append-only within a DD, so a chain is never overwritten.  Creating a child, or handing
/// a better path over to a twin, costs 4 bytes that travel with stores made anyway; round 2 copied a 64-byte bit string per
/// child and per hand-over (a line read plus a line written: a third of the expand phase's memory time,
/// tools/micro/expand_parts.hip).  Bit strings are built only for what leaves the DD: the best terminal node(s) and the
/// cut-set nodes that survive the local bounds.  A MISP path has at most value-many 1-decisions, so chains are short.
constexpr uint32_t PID_NONE = 0xFFFFFFFFu;   // the empty path (the DD's root)
template <int WS>
DDO_DEV uint32_t pid_ld(const DD2Ctx<WS>& c, int slot) { return (uint32_t)c.rec[(size_t)slot * c.RW + WS]; }
template <int WS>
DDO_DEV void pid_st(const DD2Ctx<WS>& c, int slot, uint32_t pid) { c.rec[(size_t)slot * c.RW + WS] = (uint64_t)pid; }
/// ORs the 1-decisions of path `pid` into bits[] (bit L = the decision taken at layer L of this DD)
template <int WS>
DDO_DEV void path_bits(const DD2Ctx<WS>& c, uint32_t pid, uint64_t* bits) {
    int guard = c.max_layers + 1;
    while (pid != PID_NONE && guard-- > 0) {
        const uint64_t e = c.pt[pid];
        const int L = (int)(e >> 32);
#pragma unroll
        for (int k = 0; k < WS; ++k)
            if (k == (L >> 6)) bits[k] |= 1ULL << (L & 63);
        pid = (uint32_t)e;
    }
}

template <int WS>
DDO_DEV void ld_state_p(const DD2Ctx<WS>& c, int slot, uint64_t* s, uint32_t& pid) {
    const U64x2* r2 = (const U64x2*)(c.rec + (size_t)slot * c.RW);
    constexpr int NP = (WS + 2) / 2;   // pairs covering the WS state words and the path id
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const U64x2 v = r2[q];
        if (2 * q < WS) s[2 * q] = v.a; else if (2 * q == WS) pid = (uint32_t)v.a;
        if (2 * q + 1 < WS) s[2 * q + 1] = v.b; else if (2 * q + 1 == WS) pid = (uint32_t)v.b;
    }
}
template <int WS>
DDO_DEV void ld_state(const DD2Ctx<WS>& c, int slot, uint64_t* s) {
    const U64x2* r2 = (const U64x2*)(c.rec + (size_t)slot * c.RW);
    constexpr int NP = (WS + 1) / 2;   // pairs covering the WS state words
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const U64x2 v = r2[q];
        s[2 * q] = v.a;
        if (2 * q + 1 < WS) s[2 * q + 1] = v.b;
    }
}
template <int WS>
DDO_DEV uint64_t ld_word(const DD2Ctx<WS>& c, int slot, int k) { return c.rec[(size_t)slot * c.RW + k]; }
/// full (re)write of a node's record line -- state words and path id -- with 16-byte stores.  Also used for the NO-child
/// that changes ONE word in place: a store that covers its whole 64-byte line is cheaper for the memory system than an
/// 8-byte one into a line that is not cached (tools/micro/expand_parts.hip: 8.4 vs 14.0 kcycles per 512 nodes).
template <int WS>
DDO_DEV void st_node(const DD2Ctx<WS>& c, int slot, const uint64_t* s, uint32_t pid) {
    U64x2* r2 = (U64x2*)(c.rec + (size_t)slot * c.RW);
    constexpr int NP = (WS + 2) / 2;   // pairs covering the WS state words and the path id
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        U64x2 v;
        v.a = 2 * q < WS ? s[2 * q] : (2 * q == WS ? (uint64_t)pid : 0);
        v.b = 2 * q + 1 < WS ? s[2 * q + 1] : (2 * q + 1 == WS ? (uint64_t)pid : 0);
        r2[q] = v;
    }
}

/// The dedup table lives in LDS and is rebuilt for every layer (clear, stream all unchanged live nodes in by their
/// cached hash, then insert the changed / new nodes with duplicate detection): no tombstones, no removals, and no
/// HBM round trip per probe (rocprof: the HBM-resident table accounted for about half of the L2 misses).
/// insert node `x` (hash h, state s) -> x when new, else the node already holding the same state
template <int WS>
DDO_DEV int tab2_insert(const DD2Ctx<WS>& c, int x, uint32_t h, const uint64_t* s) {
    const uint32_t mask = (uint32_t)c.tab_cap - 1;
    const uint32_t tag = h >> 20;
    const uint32_t mine = (tag << 20) | (uint32_t)x;
    uint32_t slot = h & mask;
    for (uint32_t probes = 0; probes <= mask; ++probes) {
        uint32_t e = LD_U32(&c.tab[slot]);
        if (e == T2_EMPTY) {
            e = TAB_CAS(&c.tab[slot], T2_EMPTY, mine);
            if (e == T2_EMPTY) return x;
        }
        if ((e >> 20) == tag) {
            const int w = (int)(e & 0xFFFFFu);
            // the holder may have been written by another wave a moment ago (expand publishes a node right after
            // storing it): agent-scope loads bypass this CU's vector L1, which other waves' stores do not refresh
            GLB_PTR(const uint64_t) o = c.rec + (size_t)w * c.RW;
            bool eq = true;
#pragma unroll
            for (int k = 0; k < WS; ++k) eq &= LD_U64(&o[k]) == s[k];
            if (eq) return w;
        }
        slot = (slot + 1) & mask;
    }
    c.sh->status = ST_ERR_INTERNAL;
    return x;
}
/// insert without duplicate check (all unchanged live states are distinct)
template <int WS>
DDO_DEV void tab2_insert_unique(const DD2Ctx<WS>& c, int x, uint32_t h) {
    const uint32_t mask = (uint32_t)c.tab_cap - 1;
    const uint32_t mine = ((h >> 20) << 20) | (uint32_t)x;
    uint32_t slot = h & mask;
    for (uint32_t probes = 0; probes <= mask; ++probes) {
        if (TAB_CAS(&c.tab[slot], T2_EMPTY, mine) == T2_EMPTY) return;
        slot = (slot + 1) & mask;
    }
    c.sh->status = ST_ERR_INTERNAL;
}
/// node holding state s (hash h), or -1
template <int WS>
DDO_DEV int tab2_find(const DD2Ctx<WS>& c, uint32_t h, const uint64_t* s) {
    const uint32_t mask = (uint32_t)c.tab_cap - 1;
    const uint32_t tag = h >> 20;
    uint32_t slot = h & mask;
    for (uint32_t probes = 0; probes <= mask; ++probes) {
        uint32_t e = LD_U32(&c.tab[slot]);
        if (e == T2_EMPTY) return -1;
        if ((e >> 20) == tag) {
            const int w = (int)(e & 0xFFFFFu);
            uint64_t o[WS];
            ld_state<WS>(c, w, o);
            bool eq = true;
#pragma unroll
            for (int k = 0; k < WS; ++k) eq &= o[k] == s[k];
            if (eq) return w;
        }
        slot = (slot + 1) & mask;
    }
    return -1;
}


/// Inserts the two children of one node together: x1 (hash h1, state s1) and -- when x2 >= 0 -- x2 (h2, s2); r1 / r2 receive
/// the node that holds each state (the child itself when it is new).  The probing runs in two alternating phases: an LDS phase
/// that walks both probe sequences up to a claimed empty entry or the first entry with a matching tag, and a global phase that
/// compares BOTH candidates' records at once.  A wave whose lanes meet their twins after different numbers of probes then
/// pays one memory round trip per phase instead of one per probe step (thread-0 probes, round 3: the two separate inserts
/// took 7.5 + 10.8 kcycles of a 46 kcycle chain, nearly all of it waves waiting for a few lanes' twin compares, one after
/// the other).
template <int WS>
DDO_DEV void tab2_insert2(const DD2Ctx<WS>& c, int x1, uint32_t h1, const uint64_t* s1, int x2, uint32_t h2, const uint64_t* s2, int& r1, int& r2) {
    const uint32_t mask = (uint32_t)c.tab_cap - 1;
    const uint32_t tag1 = h1 >> 20, tag2 = h2 >> 20;
    const uint32_t mine1 = (tag1 << 20) | (uint32_t)x1, mine2 = (tag2 << 20) | (uint32_t)x2;
    uint32_t p1 = h1 & mask, p2 = h2 & mask;
    r1 = -1;
    r2 = x2 >= 0 ? -1 : -2;   // -2: there is no second child
    uint32_t probes = 0;
    for (;;) {
        int c1 = -1, c2 = -1;   // holders whose tag matches: to be compared
        while (r1 == -1 && c1 < 0 && probes <= 2 * mask + 2) {
            const uint32_t e = TAB_CAS(&c.tab[p1], T2_EMPTY, mine1);   // (one LDS round trip: the old entry comes back either way)
            if (e == T2_EMPTY) {
                r1 = x1;
                break;
            }
            if ((e >> 20) == tag1) c1 = (int)(e & 0xFFFFFu);
            else p1 = (p1 + 1) & mask;
            ++probes;
        }
        while (r2 == -1 && c2 < 0 && probes <= 2 * mask + 2) {
            const uint32_t e = TAB_CAS(&c.tab[p2], T2_EMPTY, mine2);   // (one LDS round trip: the old entry comes back either way)
            if (e == T2_EMPTY) {
                r2 = x2;
                break;
            }
            if ((e >> 20) == tag2) c2 = (int)(e & 0xFFFFFu);
            else p2 = (p2 + 1) & mask;
            ++probes;
        }
        if (c1 < 0 && c2 < 0) break;
        // the holders may have been written by other waves a moment ago (a node is published right after it is stored):
        // agent-scope loads bypass this CU's vector L1, which other waves' stores do not refresh
        GLB_PTR(const uint64_t) o1 = c.rec + (size_t)(c1 >= 0 ? c1 : 0) * c.RW;
        GLB_PTR(const uint64_t) o2 = c.rec + (size_t)(c2 >= 0 ? c2 : 0) * c.RW;
        uint64_t a[WS], b[WS];
#pragma unroll
        for (int k = 0; k < WS; ++k) a[k] = c1 >= 0 ? LD_U64(&o1[k]) : 0;
#pragma unroll
        for (int k = 0; k < WS; ++k) b[k] = c2 >= 0 ? LD_U64(&o2[k]) : 0;
        if (c1 >= 0) {
            bool eq = true;
#pragma unroll
            for (int k = 0; k < WS; ++k) eq &= a[k] == s1[k];
            if (eq) r1 = c1;
            else p1 = (p1 + 1) & mask;
        }
        if (c2 >= 0) {
            bool eq = true;
#pragma unroll
            for (int k = 0; k < WS; ++k) eq &= b[k] == s2[k];
            if (eq) r2 = c2;
            else p2 = (p2 + 1) & mask;
        }
    }
    if (r1 == -1 || r2 == -1) {   // probe budget exhausted: the table is full (cannot happen: tab_limit)
        c.sh->status = ST_ERR_INTERNAL;
        if (r1 == -1) r1 = x1;
        if (r2 == -1) r2 = x2;
    }
}

/// OR / MAX over the lanes of a wave (every lane of the wave must call; lanes without a value pass 0)
#if defined(DDO_HOST_EMULATION)
DDO_DEV uint64_t wave_or_u64(uint64_t x) { return x; }
DDO_DEV uint64_t wave_max_u64(uint64_t x) { return x; }
DDO_DEV bool wave_leader(int) { return true; }
#else
DDO_DEV uint64_t wave_or_u64(uint64_t x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x |= (uint64_t)__shfl_xor((unsigned long long)x, off, 64);
    return x;
}
DDO_DEV uint64_t wave_max_u64(uint64_t x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint64_t y = (uint64_t)__shfl_xor((unsigned long long)x, off, 64);
        x = y > x ? y : x;
    }
    return x;
}
DDO_DEV bool wave_leader(int tid) { return (tid & 63) == 0; }
#endif

template <int WS>
DDO_DEV int32_t rub2_of(const DD2Ctx<WS>& c, const uint64_t* s) {
    int32_t sum = 0;
#pragma unroll
    for (int k = 0; k < WS; ++k) {
        uint64_t x = s[k];
        while (x) {
            int b = dd_ctz(x);
            sum += c.weight[k * 64 + b];
            x &= x - 1;
        }
    }
    return sum;
}

/// (value, popcount) part of the exact top-K selection (clean.rs:802-824 with main.rs:205-208): MSD radix select over
/// key32.  Result: sh->pivKey, and sh->sel_need = how many nodes with key == pivKey stay in the layer, or -1 when all of
/// them stay (then the unresolved low digits of pivKey are zero and "kept <=> key >= pivKey").
/// Keys may live in HBM (L2) at large widths: every sweep fetches KB keys per thread before it touches them, so the
/// sweep costs hi / (NT * KB) dependent round trips instead of hi / NT.
/// DEEP (workgroups of up to 512 threads: 256 VGPRs per lane) doubles the loads in flight per thread in every sweep.
template <int WS, int DEEP = 0, int POOLED = 0>
DDO_DEV void select_key2(DD2Ctx<WS>& c, int K, int L) {
    constexpr int KB = DEEP ? 16 : 8;
    DD_TID_SETUP(c)
    LDS_PTR(DD2Shared) sh = c.sh;
    LDS_PTR(uint32_t) member = POOLED ? c.fresh : c.live;   // POOLED: the layer = the pool nodes the variable impacts (marked in `fresh`)
    const int hi = DD_UNIFORM(sh->hiw);
    // Which key bits vary at all?  The AND / OR of the keys was gathered while the layer was built -- by the work-list sweep of the
    // previous transition (every node it saw) and by expand (both children of every branching node): a superset of the layer's
    // keys, which is all the skipping of constant digits needs, and one sweep over the keys less per squash.
    int need = K;
    bool done = false;
    const uint32_t kand = sh->kbits_and[L & 1], kor = sh->kbits_or[L & 1];
    const uint32_t diff = kand ^ kor;
    uint32_t piv = 0;
    // digits of key32, most significant first: bits 22..31, 11..21, 0..10
    // (a dense tier has 256 bins: four 8-bit digits)
    const bool wide_digits = c.hist_bins >= 2048;
    const int nd = wide_digits ? 3 : 4;
    const int dshift[4] = {wide_digits ? 22 : 24, wide_digits ? 11 : 16, wide_digits ? 0 : 8, 0};
    const int dbits[4] = {wide_digits ? 10 : 8, wide_digits ? 11 : 8, wide_digits ? 11 : 8, 8};
    for (int d = 0; d < nd && !done; ++d) {
        const int shift = dshift[d];
        const uint32_t dmask = (1u << dbits[d]) - 1;
        if (((diff >> shift) & dmask) == 0) {
            piv |= kand & (dmask << shift);
            continue;
        }
        const int nb = 1 << dbits[d];
        PAR_BEGIN
        for (int i = tid; i < nb; i += NT) c.hist[i] = 0;
        PAR_END
        PAR_BEGIN
        const int up = shift + dbits[d];
        for (int base = 0; base < hi; base += NT * KB) {
            uint32_t kk[KB];
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const int s = base + b * NT + tid;
                kk[b] = s < hi ? K32(c, s) : 0u;
            }
#pragma unroll
            for (int b = 0; b < KB; ++b) {
                const int s = base + b * NT + tid;
                if (s < hi && bm_test(member, s)) {
                    const uint32_t k = kk[b];
                    const bool active = up >= 32 || (k >> up) == (piv >> up);
                    if (active) LDS_ADD_U32(&c.hist[(k >> shift) & dmask], 1u);
                }
            }
        }
        PAR_END
        PAR_BEGIN  // group sums: 64 groups of nb/64 bins
        if (tid < 64) {
            const int g = nb >> 6;
            uint32_t sum = 0;
            for (int i = 0; i < g; ++i) sum += c.hist[tid * g + i];
            sh->gs[tid] = sum;
        }
        PAR_END
        PAR_BEGIN
        const int g = nb >> 6;
        for (int b = tid; b < nb; b += NT) {
            const int mine = (int)c.hist[b];
            if (mine == 0) continue;   // an empty bin never holds the K-th node
            const int grp = b / g;
            int above = 0;
            for (int x = grp + 1; x < 64; ++x) above += (int)sh->gs[x];
            for (int x = b + 1; x < (grp + 1) * g; ++x) above += (int)c.hist[x];
            if (above < need && need <= above + mine) {
                sh->sel_digit = b;
                sh->sel_above = above;
                sh->sel_bucket = mine;
            }
        }
        PAR_END
        piv |= (uint32_t)sh->sel_digit << shift;
        need -= sh->sel_above;
        if (need == sh->sel_bucket) done = true;
    }
    const int bucket = sh->sel_bucket;
    DD_SYNC();
    PAR_BEGIN
    if (tid == 0) {
        sh->pivKey = piv;
        sh->sel_need = done ? -1 : need;   // not done after the last digit: `bucket` nodes tie on key32, `need` of them stay
        sh->sel_bucket = bucket;
    }
    PAR_END
}

/// Member-order tie-break (BitSet::cmp, main.rs:205-208) among the `m` nodes of the list `tie` that share the pivot
/// (value, popcount): the `need` largest stay, the others are appended to the victim list c.wl (sh->nvict).
/// Per state word the lexicographic keys brev(~word) of the tied nodes are fetched once; the need-th largest key is
/// found by rank counting out of LDS when the list is short (every thread compares its key with all others: one
/// pass, no digit rounds), by 8-bit radix rounds otherwise; then the list shrinks to the nodes still tied.
template <int WS>
DDO_DEV bool lex_split2(DD2Ctx<WS>& c, uint16_t* tie, int m, int need) {
    DD_TID_SETUP(c)
    LDS_PTR(DD2Shared) sh = c.sh;
    const int LCAP = c.lex_cap;                      // lexicographic keys that fit the histogram area of LDS (<= 1024)
    LDS_PTR(uint64_t) lwl = (LDS_PTR(uint64_t))c.hist;
    const uint64_t sbase = (sh->ev_pos + 3) & ~3ULL;   // scratch behind the event records
    if (sbase + 2ull * (uint64_t)m + (uint64_t)(m + 1) / 2 + 8 > c.ev_cap) {
        PAR_BEGIN
        if (tid == 0) sh->status = ST_ERR_CAPACITY - 100 * 1;
        PAR_END
        return false;
    }
    uint64_t* lwg = (uint64_t*)(c.ev + sbase);
    uint16_t* cur = tie;
    uint16_t* nxt = (uint16_t*)(lwg + m);   // second list: the tie list ping-pongs between the caller's and this one
    for (int wj = 0; wj < WS; ++wj) {
        if (need <= 0 || need >= m) break;
        const bool in_lds = m <= LCAP;
        DD2_STAT(3, 1)
        DD2_STAT(8, m)
        PAR_BEGIN
        if (tid == 0) {
            sh->land = ~0ULL;
            sh->lor = 0;
            sh->sel_digit = -1;
            sh->nwl = 0;
        }
        PAR_END
        PAR_BEGIN
        uint64_t a = ~0ULL, o = 0;
        for (int i = tid; i < m; i += NT) {
            const uint64_t v = dd_brev(~ld_word<WS>(c, cur[i], wj));
            if (in_lds) lwl[i] = v; else lwg[i] = v;
            a &= v;
            o |= v;
        }
        if (tid < m) {
            LDS_AND_U64(&sh->land, a);
            LDS_OR_U64(&sh->lor, o);
        }
        PAR_END
        const uint64_t land = sh->land;
        const uint64_t ldiff = land ^ sh->lor;
        DD_SYNC();   // every thread has its copy before thread 0 resets land/lor for the next word
        DD2_TICK2(PH_SELLEX, 17)
        if (ldiff == 0) continue;   // every tied node has the same word: nothing to decide here
        uint64_t pivw = 0;
        bool all_ge_kept = false;   // radix path only: the whole digit bucket stays, pivw's low bytes are unresolved (zero)
        if (m <= LCAP) {
            // rank counting: node i holds the need-th largest word iff #greater < need <= #greater-or-equal.  With a
            // short list the inner loop is split over P threads per node (partial counts meet in LDS).
            const int P = 2 * m <= NT ? NT / m : 1;
            if (P > 1) {
                PAR_BEGIN
                for (int i = tid; i < m; i += NT) {
                    c.tcount[i] = 0;
                    c.tcount2[i] = 0;
                }
                PAR_END
                PAR_BEGIN
                if (tid < m * P) {
                    const int i = tid % m, part = tid / m;
                    const int j0 = (int)((long long)part * m / P), j1 = (int)((long long)(part + 1) * m / P);
                    const uint64_t v = lwl[i];
                    int gt = 0, ge = 0;
#pragma unroll 8
                    for (int j = j0; j < j1; ++j) {   // (unrolled: eight LDS reads in flight instead of one dependent read per step)
                        const uint64_t u = lwl[j];
                        gt += u > v ? 1 : 0;
                        ge += u >= v ? 1 : 0;
                    }
                    if (gt) LDS_ADD_I32(&c.tcount[i], gt);
                    if (ge) LDS_ADD_I32(&c.tcount2[i], ge);
                }
                PAR_END
                PAR_BEGIN
                for (int i = tid; i < m; i += NT) {
                    const int gt = c.tcount[i], ge = c.tcount2[i];
                    if (gt < need && need <= ge) sh->pivLex[0] = lwl[i];
                }
                PAR_END
            } else {
                PAR_BEGIN
                for (int i = tid; i < m; i += NT) {
                    const uint64_t v = lwl[i];
                    int gt = 0, ge = 0;
#pragma unroll 8
                    for (int j = 0; j < m; ++j) {
                        const uint64_t u = lwl[j];
                        gt += u > v ? 1 : 0;
                        ge += u >= v ? 1 : 0;
                    }
                    if (gt < need && need <= ge) sh->pivLex[0] = v;   // every node holding this word agrees
                }
                PAR_END
            }
            pivw = sh->pivLex[0];
        } else {
            uint64_t prefix = 0;
            int nd = need;
            bool rdone = false;
            for (int b = 7; b >= 0 && !rdone; --b) {
                const int shift = 8 * b;
                if (((ldiff >> shift) & 0xFF) == 0) {
                    prefix |= land & (0xFFULL << shift);
                    continue;
                }
                DD2_STAT(2, 1)
                PAR_BEGIN
                if (tid < 256) c.hist[tid] = 0;
                PAR_END
                PAR_BEGIN
                for (int i = tid; i < m; i += NT) {
                    const uint64_t v = lwg[i];
                    if (shift + 8 < 64 && (v >> (shift + 8)) != (prefix >> (shift + 8))) continue;
                    LDS_ADD_U32(&c.hist[(v >> shift) & 0xFF], 1u);
                }
                PAR_END
                PAR_BEGIN
                if (tid < 256) {
                    int above = 0;
                    for (int x = tid + 1; x < 256; ++x) above += (int)c.hist[x];
                    const int mine = (int)c.hist[tid];
                    if (above < nd && nd <= above + mine) {
                        sh->sel_digit = tid;
                        sh->sel_above = above;
                        sh->sel_bucket = mine;
                    }
                }
                PAR_END
                prefix |= (uint64_t)sh->sel_digit << shift;
                nd -= sh->sel_above;
                if (nd == sh->sel_bucket) rdone = true;   // everything still tied below this digit stays
                DD_SYNC();
            }
            pivw = prefix;
            all_ge_kept = rdone;
        }
        DD2_TICK2(PH_SELLEX, 18)
        // ---- partition: larger words stay, smaller ones are victims, equal ones remain tied
        PAR_BEGIN
        if (tid == 0) {
            sh->nrec = 0;   // kept
            sh->nnew = 0;   // still tied
        }
        PAR_END
        PAR_BEGIN
        for (int i = tid; i < m; i += NT) {
            const uint64_t v = in_lds ? lwl[i] : lwg[i];
            const int s = cur[i];
            if (v > pivw || (all_ge_kept && v >= pivw)) {
                LDS_ADD_I32(&sh->nrec, 1);
            } else if (v == pivw) {
                const int k = LDS_ADD_I32(&sh->nnew, 1);
                nxt[k] = (uint16_t)s;
            } else {
                const int k = LDS_ADD_I32(&sh->nvict, 1);
                if (k < c.capW) c.wl[k] = (uint16_t)s;
            }
        }
        PAR_END
        need -= sh->nrec;
        m = sh->nnew;
        DD_SYNC();
        DD2_TICK2(PH_SELLEX, 19)
        uint16_t* t = cur;
        cur = nxt;
        nxt = t;
    }
    if (need > 0 && need < m) {   // cannot happen: two live nodes never hold the same state
        PAR_BEGIN
        if (tid == 0) sh->status = ST_ERR_INTERNAL;
        PAR_END
        return false;
    }
    if (need <= 0 && m > 0) {   // nothing more to keep: the remaining tied nodes are victims
        PAR_BEGIN
        for (int i = tid; i < m; i += NT) {
            const int k = LDS_ADD_I32(&sh->nvict, 1);
            if (k < c.capW) c.wl[k] = cur[i];
        }
        PAR_END
    }
    return true;
}

/// One compile() (clean.rs:345-381) with in-place layers.
///
/// Event records (u32 stream `ev`, per transition L -> L+1, 4 words per affected or pruned parent):
///   [0] parent slot   [1] NO target | flags   [2] YES target | flags   [3] slot allocated for the YES-child
/// flags: EV_CREATED (bit 31, YES target is the new node), EV_RAISED (bit 30, the arc raised its target's key).
/// The squash of layer L (deleted slots, merged slot, re-added duplicate) is stored with iteration L.


/// POOLED = 1: the same engine compiles `Pooled` decision diagrams (mdd/pooled.rs:117-823) -- which is what its node slots are:
/// a pool of nodes that live across layers.  What changes against clean.rs:
///   * the layer of a variable is made of the pool nodes it IMPACTS (Problem::is_impacted_by, dp.rs:68-70; MISP: the states
///     that contain the vertex, main.rs:145-147): width, ranking and merge apply to those only (pooled.rs:614-641, 734-829);
///   * a node's rough upper bound is checked when the node is expanded, not in every layer it waits in (pooled.rs:351-361);
///   * nodes_expanded counts the impacted nodes; a node's depth is the layer at which it is expanded;
///   * the cut-set is the FRONTIER (pooled.rs:543-566): exact nodes with an inexact child, whatever their layer.  A child
///     may turn inexact long after its parent was expanded, so the frontier is found by the backward replay of the event
///     records (the parent's exactness travels in its record, EV_PINEX), and a cut-set node's state, value and rough upper
///     bound are rebuilt from its best path (an exact node's state is the residual state with its path's decisions applied).
/// A node's own threshold (pooled.rs:493-511; the corner cases of the saturating arithmetic as in misp_dd_core.hpp): `th` is what its
/// children left it (TH_NONE: nothing), `rub` INT32_MAX for a node that was never expanded, `vbv` VB_UNMARKED for value_bot == isize::MIN.
DDO_DEV int32_t pooled_own_theta(int32_t th, int32_t val, int32_t rub, bool is_cut, bool exact, int32_t vbv, int32_t bk, bool bk_min) {
    if (rub != INT32_MAX && !bk_min && (int64_t)val + rub <= (int64_t)bk) return bk - rub;
    if (is_cut) {
        const bool locb_le = vbv == VB_UNMARKED ? (!bk_min || val <= 0) : (!bk_min && (int64_t)val + vbv <= (int64_t)bk);
        if (!locb_le) return val;
        const int32_t cand = vbv == VB_UNMARKED ? (bk_min ? 0 : TH_INF) : bk - vbv;
        const int32_t old = th == TH_NONE ? TH_INF : th;
        return cand < old ? cand : old;
    }
    if (exact && th == TH_NONE) return TH_INF;   // large theta for dangling nodes
    return th;
}

template <int WS, int DEEP = 0, int POOLED = 0>
DDO_DEV void run_dd2(DD2Ctx<WS>& c, const DDInput& in, int comp_type, int64_t best_lb, DDResult* res) {
    constexpr int KB = DEEP ? 16 : 8;
    DD_TID_SETUP(c)
    LDS_PTR(DD2Shared) sh = c.sh;
    const int capS = c.capS;
    const int W = in.width;
    const bool relaxed = comp_type == CT_RELAXED;
    const bool restricted = comp_type == CT_RESTRICTED;
    const int32_t vbase = in.value + c.vbase_off;
    // Pooled behind a SimpleCache (pooled.rs:467-535, 662-680): the impacted nodes of every layer but the first are looked up in the
    // cache before the layer is squashed and expanded, and the thresholds of the finished decision diagram are written back.
    const bool pcache = POOLED && c.cache_cap != 0 && c.pvr != nullptr && c.pst != nullptr && (in.flags & IN_CACHE) != 0;
    uint32_t cache_hits = 0;

    // ---------------------------------------------------------------- _clear + _initialize
    PAR_BEGIN
    for (int i = tid; i < c.npad; i += NT) c.cnt[i] = 0;
    for (int i = tid; i < c.nbw; i += NT) {
        c.live[i] = 0;
        c.inex[i] = 0;
        c.okb[i] = 0;
        c.fresh[i] = 0;
    }
    for (int i = tid; i < c.tab_cap; i += NT) c.tab[i] = T2_EMPTY;
    if (tid == 0) {
        sh->status = ST_OK;
        sh->nodes = 0;
        sh->arcs = 0;
        sh->recycled_merges = 0;
        sh->maxn = 0;
        sh->cutoff = 0;
        sh->nlive = 1;
        sh->hiw = 1;
        sh->varkey = 0xFFFFFFFFu;
        sh->kbits_and[0] = sh->kbits_and[1] = 0xFFFFFFFFu;
        sh->kbits_or[0] = sh->kbits_or[1] = 0;
        sh->nwl = 0;
        sh->nwl2 = 0;
        sh->nrec = 0;
        sh->nnew = 0;
        sh->nfl = 0;
        sh->npruned = 0;
        sh->nyes = 0;
        sh->ev_pos = 0;
        for (int k = 0; k < 8; ++k) sh->clk[k] = 0;
        for (int k = 0; k < 24; ++k) sh->mk[k] = 0;
        sh->clk_last = dd_clock();
        int pop = 0;
        uint64_t root[WS];
        if (in.src_off != NO_POOL_SRC) {   // the residual state is a row of a cut-set block kept in the device pool
            const PoolBlockHeader* h = (const PoolBlockHeader*)(c.pool + in.src_off);
            const uint64_t* rows = (const uint64_t*)(c.pool + in.src_off + h->off_states);
            for (int k = 0; k < WS; ++k) root[k] = k < (int)h->ws ? rows[(size_t)k * h->rows + in.src_row] : 0;
        } else {
            for (int k = 0; k < WS; ++k) root[k] = in.state[k];
        }
        for (int k = 0; k < WS; ++k) pop += dd_popc(root[k]);
        st_node<WS>(c, 0, root, PID_NONE);
        KH_ST(c, 0, ((uint32_t)(in.value - vbase) << KEY_POP_BITS) | (uint32_t)pop, hash32_state<WS>(root));
    }
    PAR_END
    PAR_BEGIN
    if (tid == 0) {
        uint64_t root[WS];
        ld_state<WS>(c, 0, root);
        add_bits<WS>(c.cnt, root, +1);
        c.live[0] = 1u;
        c.okb[0] = 1u;
        c.fresh[0] = POOLED ? 0u : 1u;   // (POOLED: `fresh` marks the members of a layer that is being squashed, nothing else)
    }
    PAR_END

    int lel = -1;
    int snapL = -1;  // layer whose snapshot sits in the cut-set buffers
    int ncs = 0;
    int L = 0;
    int var = -1;
    bool failed = false;

    for (;;) {
        // ------------------------------------------------------------ next_variable (main.rs:109-143)
        // (sh->varkey = ~0, sh->hiw = 1 and the work-list counters were reset by the last region of the previous layer /
        // of the initialisation)
        const int nU = DD_UNIFORM(sh->nlive);
        // _squash_if_needed (clean.rs:779-795) is decided up front: a layer that is NOT squashed never probes the dedup
        // table of the current layer again, so the table of the next layer is cleared here and the region that would
        // do it later disappears -- narrow layers are all barrier and round-trip latency, every region counts
        // (POOLED: the layer is what the variable impacts -- cnt[var] nodes, known once the variable is; decided below)
        bool squash = !POOLED && ((restricted && nU > W) || (relaxed && nU > W && L > 1));
        PAR_BEGIN
        if (!squash && !POOLED)
            for (int i = tid; i < c.tab_cap; i += NT) c.tab[i] = T2_EMPTY;
        // Cutoff::must_stop (clean.rs:352) -- polled every 8th layer: the flag lives in host-visible memory and a
        // read is a full round trip on the critical path of the layer
        if (tid == 0 && c.cutoff_flag && (L & 7) == 0) sh->cutoff = LD_I32(c.cutoff_flag);
        for (int i = tid; i < c.n; i += NT) {
            int cv = c.cnt[i];
            if (cv > 0) LDS_MIN_U32(&sh->varkey, ((uint32_t)cv << 12) | (uint32_t)i);
            else if (cv < 0) sh->status = ST_ERR_INTERNAL;
        }
        for (int w = tid; w < c.nbw; w += NT) {   // highest live slot: dead slots above it need not be swept
            const uint32_t lv = c.live[w];
            if (lv) LDS_MAX_I32(&sh->hiw, w * 32 + 32 - dd_clz32(lv));
        }
        PAR_END
        var = DD_UNIFORM(sh->varkey == 0xFFFFFFFFu ? -1 : (int)(sh->varkey & 0xFFFu));
        if (var < 0) break;
        if (sh->cutoff) {
            PAR_BEGIN
            if (tid == 0) sh->status = ST_CUTOFF;
            PAR_END
            failed = true;
            break;
        }
        if (L >= c.max_layers - 1 || sh->status != ST_OK) { failed = true; break; }
        if (POOLED) {
            // pooled.rs:734-749: the width bounds the nodes that are EXPANDED; `layers.len() >= 2` is L > 1 here (every layer so
            // far held the nodes its variable impacts, at least one)
            const int vw0 = var >> 6;
            const uint64_t vbit0 = 1ULL << (var & 63);
            if (pcache && L >= 1) {
                // _filter_with_cache (pooled.rs:662-680; `if !self.layers.is_empty()`, :635): an impacted node whose value does not exceed
                // the threshold the cache holds for (state, depth) leaves the pool without being expanded -- it stays a node of the
                // LAYER (its threshold travels upwards in _compute_thresholds), so it is listed in front of this layer's deleted
                // nodes: (slot | exactness, threshold) pairs at cp_off, their number in evoff[L][6].
                const uint64_t cp_off = DD_UNIFORM64(sh->ev_pos);
                PAR_BEGIN
                if (tid == 0) sh->ncut = 0;
                PAR_END
                PAR_BEGIN
                const int hi = DD_UNIFORM(sh->hiw);
                for (int s = tid; s < hi; s += NT) {
                    if (!bm_test(c.live, s)) continue;
                    if ((c.rec[(size_t)s * c.RW + vw0] & vbit0) == 0) continue;
                    uint64_t st[WS];
                    ld_state<WS>(c, s, st);
                    int64_t packed = 0;
                    if (!cache_get<WS>(c, st, c.depth0 + L, &packed)) continue;
                    const int32_t val = vbase + (int32_t)(K32(c, s) >> KEY_POP_BITS);
                    const int32_t tv = th_value(packed);
                    if (val > tv) continue;   // (`node.value_top > threshold.value`: explored further)
                    add_bits<WS>(c.cnt, st, -1);
                    bm_clr(c.live, s);
                    LDS_ADD_I32(&sh->nlive, -1);
                    const uint64_t k = (uint64_t)LDS_ADD_I32(&sh->ncut, 1);
                    if (cp_off + 2 * k + 2 <= c.ev_cap) {
                        c.ev[cp_off + 2 * k] = (uint32_t)s | (bm_test(c.inex, s) ? EV_PINEX : 0u);
                        c.ev[cp_off + 2 * k + 1] = (uint32_t)tv;
                    }
                }
                PAR_END
                const int n_cp = DD_UNIFORM(sh->ncut);
                if (cp_off + 2ull * (uint64_t)n_cp + 8 > c.ev_cap) {
                    PAR_BEGIN
                    if (tid == 0) sh->status = ST_ERR_CAPACITY - 100 * 2;
                    PAR_END
                    failed = true;
                    break;
                }
                PAR_BEGIN
                if (tid == 0) {
                    sh->ev_pos = cp_off + 2ull * (uint64_t)n_cp;
                    c.evoff[(size_t)L * 8 + 6] = (uint32_t)n_cp;
                }
                PAR_END
                cache_hits += (uint32_t)n_cp;
            } else if (pcache) {
                PAR_BEGIN
                if (tid == 0) c.evoff[(size_t)L * 8 + 6] = 0;
                PAR_END
            }
            const int nimp = c.cnt[var];
            squash = (restricted && nimp > W) || (relaxed && nimp > W && L > 1);
            PAR_BEGIN
            if (!squash) {
                for (int i = tid; i < c.tab_cap; i += NT) c.tab[i] = T2_EMPTY;
            } else {
                // membership of the layer (`fresh` bitmap, otherwise unused by pooled DDs) and the AND / OR of its keys for the select
                uint32_t kb_and = 0xFFFFFFFFu, kb_or = 0;
                const int hi = DD_UNIFORM(sh->hiw);
                for (int s = tid; s < hi; s += NT) {
                    if (!bm_test(c.live, s)) continue;
                    if ((c.rec[(size_t)s * c.RW + vw0] & vbit0) == 0) continue;
                    bm_set(c.fresh, s);
                    const uint32_t k = K32(c, s);
                    kb_and &= k;
                    kb_or |= k;
                }
                if (kb_or != 0 || kb_and != 0xFFFFFFFFu) {
                    LDS_AND_U32(&sh->kbits_and[L & 1], kb_and);
                    LDS_OR_U32(&sh->kbits_or[L & 1], kb_or);
                }
            }
            PAR_END
        }
        DD2_TICK2(PH_VAR, 15)
        // ------------------------------------------------------------ _squash_if_needed (clean.rs:779-795)
        int merged_slot = -1, dup_from = -1, dup_to = -1;
        const uint64_t del_off = DD_UNIFORM64(sh->ev_pos);
        int n_del = 0;
        if (squash && c.tier == 1) {   // a capacity tier has no squash phases (their LDS is not there): the DD is handed up.  The lazy
                                       // solver never gets here (a tier's layer capacity is below its width); an mdd bound to a tier may
            PAR_BEGIN
            if (tid == 0) sh->status = ST_ERR_CAPACITY - 100 * 11;
            PAR_END
            failed = true;
            break;
        }
        if (squash) {
            if (lel < 0) {
                lel = L - 1;   // _maybe_save_lel
                if (!POOLED && relaxed && snapL != lel) {  // cannot happen: see the snapshot rule below
                    PAR_BEGIN
                    if (tid == 0) sh->status = ST_ERR_INTERNAL;
                    PAR_END
                    failed = true;
                    break;
                }
            }
            const int K = restricted ? W : W - 1;
            if (K > 0) select_key2<WS, DEEP, POOLED>(c, K, L);
            DD2_TICK(PH_SELECT)
            PAR_BEGIN
            if (tid == 0) {
                sh->nvict = 0;
                sh->nfl = 0;      // nodes tied with the pivot on (value, popcount)
                sh->mergedKey = 0;
                for (int k = 0; k < WS; ++k) sh->merged[k] = 0;
                sh->recycled = 0;
                sh->xslot = -1;
                sh->merged_slot = -1;
                sh->free_slot = 0x7FFFFFFF;
            }
            PAR_END
            const uint32_t pivKey = (uint32_t)DD_UNIFORM((int)sh->pivKey);
            const int tie_need = DD_UNIFORM(K > 0 ? sh->sel_need : -1);
            PAR_BEGIN   // victims: live nodes ranked below the pivot (clean.rs:810-812 / :851-852); ties are listed apart
            for (int base = 0; base < sh->hiw; base += NT * KB) {
                uint32_t kk[KB];
#pragma unroll
                for (int b = 0; b < KB; ++b) {
                    const int s = base + b * NT + tid;
                    kk[b] = s < sh->hiw ? K32(c, s) : 0u;
                }
#pragma unroll
                for (int b = 0; b < KB; ++b) {
                    const int s = base + b * NT + tid;
                    if (s >= sh->hiw || !bm_test(POOLED ? c.fresh : c.live, s)) continue;
                    const uint32_t key = kk[b];
                    if (K > 0 && key > pivKey) continue;
                    if (K > 0 && key == pivKey) {
                        if (tie_need < 0) continue;
                        const int i = LDS_ADD_I32(&sh->nfl, 1);
                        if (i < c.capW) c.fl[i] = (uint16_t)s;
                        continue;
                    }
                    const int i = LDS_ADD_I32(&sh->nvict, 1);
                    if (i < c.capW) c.wl[i] = (uint16_t)s;
                }
            }
            PAR_END
            DD2_TICK2(PH_SELLEX, 16)
            if (K > 0 && tie_need >= 0) {
                const int m = DD_UNIFORM(sh->nfl);
                DD2_STAT(0, m)
                DD2_STAT(1, 1)
                if (m > c.capW || !lex_split2<WS>(c, c.fl, m, tie_need)) {
                    if (m > c.capW) {
                        PAR_BEGIN
                        if (tid == 0) sh->status = ST_ERR_CAPACITY - 100 * 2;
                        PAR_END
                    }
                    failed = true;
                    break;
                }
            }
            DD2_TICK(PH_SELLEX)
            const int nv = DD_UNIFORM(sh->nvict);
            if (nv > c.capW || sh->ev_pos + (uint64_t)nv + 8 > c.ev_cap) {
                PAR_BEGIN
                if (tid == 0) sh->status = ST_ERR_CAPACITY - 100 * 2;
                PAR_END
                failed = true;
                break;
            }
            PAR_BEGIN
            for (int i = tid; (i & ~63) < nv; i += NT) {   // whole waves: the merged state is OR-reduced per wave first
                const bool valid = i < nv;
                const int s = valid ? (int)c.wl[i] : 0;
                uint64_t st[WS];
                ld_state<WS>(c, s, st);
                if (valid) {
                    add_bits<WS>(c.cnt, st, -1);
                    bm_clr(c.live, s);
                    bm_clr(c.fresh, s);
                    c.ev[del_off + i] = (uint32_t)s;
                }
                if (relaxed) {
                    // MispRelax::merge (main.rs:172-178): union of the merged states; one LDS atomic per wave and
                    // word instead of one per victim (they all hit the same seven addresses)
#pragma unroll
                    for (int k = 0; k < WS; ++k) {
                        const uint64_t u = wave_or_u64(valid ? st[k] : 0ULL);
                        if (wave_leader(tid) && u) LDS_OR_U64(&sh->merged[k], u);
                    }
                    const uint64_t mk = wave_max_u64(valid ? (((uint64_t)K32(c, s) << 32) | (uint32_t)s) : 0ULL);
                    if (wave_leader(tid) && mk) LDS_MAX_U64(&sh->mergedKey, mk);
                }
            }
            if (tid == 0) {
                sh->nlive -= nv;
                sh->ev_pos += (uint64_t)nv;
            }
            PAR_END
            n_del = nv;
            DD2_STAT(6, nv)
            DD2_STAT(9, 1)
            DD2_TICK2(PH_VICTIMS, 10)
            if (relaxed) {
                // ---------------------------------------------------- merged node (clean.rs:826-875)
                PAR_BEGIN
                if (tid == 0) {
                    uint64_t ms[WS];
                    for (int k = 0; k < WS; ++k) ms[k] = sh->merged[k];
                    const int r = tab2_find<WS>(c, hash32_state<WS>(ms), ms);
                    sh->recycled = (r >= 0 && bm_test(c.live, r)) ? 1 : 0;   // clean.rs:830
                    sh->merged_slot = r;
                }
                for (int w = tid; w < c.nbw; w += NT) {   // lowest slot that is not live
                    uint32_t freebits = ~c.live[w];
                    if (freebits) {
                        int s = w * 32 + dd_ctz((uint64_t)freebits);
                        if (s < capS) LDS_MIN_U32((LDS_PTR(uint32_t))&sh->free_slot, (uint32_t)s);
                    }
                }
                PAR_END
                DD2_TICK2(PH_VICTIMS, 12)
                if (sh->recycled) {
                    // clean.rs:868-872: the best-ranked node X of the merged set stays in the layer.  X = the victim with
                    // the largest (value, popcount) key and, among those, the largest member order: the candidates are
                    // filtered word by word against the maximum lexicographic key (one gather per word).
                    const uint32_t topk = (uint32_t)(sh->mergedKey >> 32);
                    PAR_BEGIN
                    if (tid == 0) sh->nfl = 0;
                    PAR_END
                    PAR_BEGIN
                    for (int i = tid; i < nv; i += NT) {
                        const int s = c.wl[i];
                        if (K32(c, s) == topk) {
                            const int k = LDS_ADD_I32(&sh->nfl, 1);
                            c.fl[k] = (uint16_t)s;
                        }
                    }
                    PAR_END
                    int m = sh->nfl;
                    const uint64_t sbase = (sh->ev_pos + 3) & ~3ULL;   // scratch behind the event records
                    if (sbase + 2ull * (uint64_t)m + (uint64_t)(m + 1) / 2 + 8 > c.ev_cap) {
                        PAR_BEGIN
                        if (tid == 0) sh->status = ST_ERR_CAPACITY - 100 * 9;
                        PAR_END
                        failed = true;
                        break;
                    }
                    uint64_t* lwg = (uint64_t*)(c.ev + sbase);
                    uint16_t* cur = c.fl;
                    uint16_t* nxt = (uint16_t*)(lwg + m);
                    for (int wj = 0; wj < WS && m > 1; ++wj) {
                        LDS_PTR(uint64_t) lwl = (LDS_PTR(uint64_t))c.hist;
                        const bool in_lds = m <= c.lex_cap;
                        PAR_BEGIN
                        if (tid == 0) {
                            sh->lor = 0;
                            sh->nnew = 0;
                        }
                        PAR_END
                        PAR_BEGIN
                        uint64_t mx = 0;
                        for (int i = tid; i < m; i += NT) {
                            const uint64_t v = dd_brev(~ld_word<WS>(c, cur[i], wj));
                            if (in_lds) lwl[i] = v; else lwg[i] = v;
                            mx = v > mx ? v : mx;
                        }
                        if (mx) LDS_MAX_U64(&sh->lor, mx);
                        PAR_END
                        const uint64_t top = sh->lor;
                        PAR_BEGIN
                        for (int i = tid; i < m; i += NT)
                            if ((in_lds ? lwl[i] : lwg[i]) == top) nxt[LDS_ADD_I32(&sh->nnew, 1)] = cur[i];
                        PAR_END
                        m = sh->nnew;
                        DD_SYNC();
                        uint16_t* t = cur;
                        cur = nxt;
                        nxt = t;
                    }
                    PAR_BEGIN
                    if (tid == 0) {
                        const int best = cur[0];
                        sh->xslot = best;
                        sh->recycled_merges += 1;
                        const int r = sh->merged_slot;
                        const uint32_t mkey = (uint32_t)(sh->mergedKey >> 32);
                        const int bestv = (int)(uint32_t)sh->mergedKey;
                        if ((mkey >> KEY_POP_BITS) > (K32(c, r) >> KEY_POP_BITS)) {   // the best redirected arc wins
                            K32_ST(c, r, (mkey & ~KEY_POP_MASK) | (K32(c, r) & KEY_POP_MASK));
                            pid_st<WS>(c, r, pid_ld<WS>(c, bestv));
                        }
                        bm_set(c.inex, r);
                        bm_clr(c.okb, r);       // F_RELAXED: best paths through r are not exact
                        uint64_t xs[WS];       // re-add X: it is un-deleted (clean.rs:870-871)
                        ld_state<WS>(c, best, xs);
                        add_bits<WS>(c.cnt, xs, +1);
                        bm_set(c.live, best);
                        bm_set(c.fresh, best);   // its fresh mark was dropped with the victims: check it (again)
                        sh->nlive += 1;
                    }
                    PAR_END
                    PAR_BEGIN
                    {
                        const uint32_t best = (uint32_t)sh->xslot;   // X is not deleted after all
                        for (int i = tid; i < nv; i += NT)
                            if (c.ev[del_off + i] == best) c.ev[del_off + i] = NONE32;
                    }
                    PAR_END
                    merged_slot = sh->merged_slot;
                    dup_from = sh->xslot;
                    dup_to = sh->merged_slot;
                } else {
                    if (sh->free_slot >= capS) {
                        PAR_BEGIN
                        if (tid == 0) sh->status = ST_ERR_CAPACITY - 100 * 3;
                        PAR_END
                        failed = true;
                        break;
                    }
                    PAR_BEGIN
                    {
                        const int m = sh->free_slot;
                        const int bestv = (int)(uint32_t)sh->mergedKey;
                        if (tid < WS) {   // one thread per state word: state, best path, vertex counters
                            const uint64_t w = sh->merged[tid];
#if defined(DDO_WORD_MAJOR)
                            c.st[(size_t)tid * capS + m] = w;
#endif
                            c.rec[(size_t)m * c.RW + tid] = w;
                            if (tid == 0) pid_st<WS>(c, m, pid_ld<WS>(c, bestv));
                            uint64_t x = w;
                            while (x) {
                                int b = dd_ctz(x);
                                LDS_ADD_I32(&c.cnt[tid * 64 + b], 1);
                                x &= x - 1;
                            }
                        }
                        if (tid == WS % NT) {
                            int pop = 0;
                            uint64_t ms[WS];
                            for (int k = 0; k < WS; ++k) {
                                ms[k] = sh->merged[k];
                                pop += dd_popc(ms[k]);
                            }
                            const uint32_t mkey = (uint32_t)(sh->mergedKey >> 32);
                            KH_ST(c, m, (mkey & ~KEY_POP_MASK) | (uint32_t)pop, hash32_state<WS>(ms));
                            bm_set(c.live, m);
                            bm_set(c.inex, m);
                            bm_clr(c.okb, m);
                            bm_set(c.fresh, m);
                            sh->nlive += 1;
                            if (m >= sh->hiw) sh->hiw = m + 1;
                            sh->merged_slot = m;
                        }
                    }
                    PAR_END
                    merged_slot = sh->merged_slot;
                }
            }
        }
        DD2_TICK2(PH_VICTIMS, 11)
        const int n = DD_UNIFORM(sh->nlive);   // |layer L| after squash
        const int naff_bound = c.cnt[var] > 0 ? c.cnt[var] : 0;   // live states containing the variable

        // ------------------------------------------------------------ candidate last exact layer: snapshot
        // Layer L+1 can only be squashed when n + (#YES-children) > W; only then is layer L copied, so the
        // cut-set (clean.rs:566-573) is available although layers are updated in place.
        if (!POOLED && relaxed && lel < 0 && n + naff_bound > W && L >= 1) {
            PAR_BEGIN
            if (tid == 0) sh->ncut = 0;
            PAR_END
            PAR_BEGIN
            for (int s = tid; s < sh->hiw; s += NT) {
                if (!bm_test(c.live, s)) continue;
                int i = LDS_ADD_I32(&sh->ncut, 1);
                if (i >= c.capW) continue;
                c.cs_slot[i] = (uint32_t)s;
                {
                    uint64_t ss[WS];
                    uint32_t spid = PID_NONE;
                    ld_state_p<WS>(c, s, ss, spid);
#pragma unroll
                    for (int k = 0; k < WS; ++k) c.cs_state[(size_t)k * c.capW + i] = ss[k];
                    c.cs_pid[i] = spid;
                }
                const uint32_t key = K32(c, s);
                c.cs_value[i] = vbase + (int32_t)(key >> KEY_POP_BITS);
                c.cs_pop[i] = key & KEY_POP_MASK;
            }
            PAR_END
            ncs = sh->ncut;
            snapL = L;
        }

        // ------------------------------------------------------------ work list: affected or fresh nodes
        if (squash) {   // the squash phases probed the table of this layer and borrowed the counters
            PAR_BEGIN
            for (int i = tid; i < c.tab_cap; i += NT) c.tab[i] = T2_EMPTY;   // dedup table of the next layer (see the sweep)
            if (POOLED)
                for (int i = tid; i < c.nbw; i += NT) c.fresh[i] = 0;   // (the layer's membership marks)
            if (tid == 0) {
                sh->nwl = 0;
                sh->nwl2 = 0;
                sh->nrec = 0;
                sh->nnew = 0;
                sh->nfl = 0;
                sh->npruned = 0;
                sh->nyes = 0;
            }
            PAR_END
        }
        const int vw = var >> 6;
        const uint64_t vbit = 1ULL << (var & 63);

        // One sweep over the live slots does two jobs: (1) the work list -- nodes that contain the variable go to the
        // front of `wl`, fresh nodes that do not (their rough upper bound was never checked) to its back; (2) every
        // other node stays unchanged in the next layer and enters the dedup table right away with its cached hash
        // (fresh survivors follow in expand 1, changed / new nodes in expand 2).  Word `var/64` of every slot and
        // the hashes are contiguous streams; the loads of a batch are in flight before the first is used.
#if defined(DDO_KS)   // experiment knob: slots fetched per thread and batch by the work-list sweep
        constexpr int KS = DDO_KS;
#else
        constexpr int KS = DEEP ? 8 : 4;
#endif
        PAR_BEGIN
        {
            uint32_t kb_and = 0xFFFFFFFFu, kb_or = 0;   // key bits of the nodes seen (for the select of the next layer)
#if defined(DDO_WORD_MAJOR)
            const uint64_t* row = c.st + (size_t)vw * capS;
#else
            // Word var/64 of every live node comes straight out of its record (one 64-byte sector per node and layer -- the
            // read the algorithmic byte count allots to a node anyway).  Round 2 kept a word-major copy of the states for this
            // sweep, a contiguous 8-byte stream, but every YES-child then cost 7 scattered 8-byte stores to keep it up to date,
            // and partial-line write requests are what the memory system handles worst (tools/micro/expand_patterns.hip: 100 of
            // the 155 kcycles a workgroup spends per 512 new nodes; the strided sweep adds 6 cycles per node).
            GLB_PTR(const uint64_t) row = c.rec + vw;
#endif
            const int hi = DD_UNIFORM(sh->hiw);
            for (int base = 0; base < hi; base += NT * KS) {
                uint64_t ww[KS], hh[KS];   // hh = key32 << 32 | h32
#pragma unroll
                for (int b = 0; b < KS; ++b) {
                    const int s = base + b * NT + tid;
#if defined(DDO_WORD_MAJOR)
                    ww[b] = s < hi ? row[s] : 0;
#else
                    ww[b] = (s < hi && bm_test(c.live, s)) ? row[(size_t)s * c.RW] : 0;   // (dead slots: no sector fetched)
#endif
                    hh[b] = s < hi ? KH(c, s) : 0;
                }
                uint32_t pslot[KS], pmine[KS];
                bool pend[KS];
#pragma unroll
                for (int b = 0; b < KS; ++b) {
                    const int s = base + b * NT + tid;
                    pend[b] = false;
                    pslot[b] = (uint32_t)hh[b] & ((uint32_t)c.tab_cap - 1);
                    pmine[b] = (((uint32_t)hh[b] >> 20) << 20) | (uint32_t)s;
                    if (s >= hi || !bm_test(c.live, s)) continue;
                    kb_and &= (uint32_t)(hh[b] >> 32);
                    kb_or |= (uint32_t)(hh[b] >> 32);
                    if ((ww[b] & vbit) != 0) {
                        const int i = LDS_ADD_I32(&sh->nwl, 1);
                        if (i < c.capW) c.wl[i] = (uint16_t)s;
                    } else if (!POOLED && bm_test(c.fresh, s)) {
                        // unit weights: the rough upper bound is the popcount held in the key (main.rs:191-193); a fresh
                        // node that passes the check (clean.rs:362-365) is its own only child and joins the table here
                        const uint32_t key = (uint32_t)(hh[b] >> 32);
                        if (c.unit_weights && (int64_t)(key & KEY_POP_MASK) + (int64_t)(vbase + (int32_t)(key >> KEY_POP_BITS)) > best_lb) {
                            bm_clr(c.fresh, s);
                            pend[b] = true;
                        } else {
                            const int i = LDS_ADD_I32(&sh->nwl2, 1);
                            if (i < c.capW) c.wl[c.capW - 1 - i] = (uint16_t)s;
                        }
                    } else {
                        pend[b] = true;   // unchanged node: into the table (all such states are distinct)
                    }
                }
                // the KS inserts of the batch probe together: their compare-and-swaps are in flight at the same time
                for (int round = 0; round <= c.tab_cap; ++round) {
                    uint32_t got[KS];
                    bool any = false;
#pragma unroll
                    for (int b = 0; b < KS; ++b)
                        if (pend[b]) got[b] = TAB_CAS(&c.tab[pslot[b]], T2_EMPTY, pmine[b]);
#pragma unroll
                    for (int b = 0; b < KS; ++b)
                        if (pend[b]) {
                            if (got[b] == T2_EMPTY) pend[b] = false;
                            else {
                                pslot[b] = (pslot[b] + 1) & ((uint32_t)c.tab_cap - 1);
                                any = true;
                            }
                        }
                    if (!any) break;
                    if (round == c.tab_cap) sh->status = ST_ERR_INTERNAL;
                }
            }
            if (kb_or != 0 || kb_and != 0xFFFFFFFFu) {
                LDS_AND_U32(&sh->kbits_and[(L + 1) & 1], kb_and);
                LDS_OR_U32(&sh->kbits_or[(L + 1) & 1], kb_or);
            }
        }
        PAR_END
        const int nwl = DD_UNIFORM(sh->nwl);      // nodes containing the variable
        const int nwl2 = DD_UNIFORM(sh->nwl2);    // fresh nodes that do not
        // (n + nwl bounds the entries of the next layer's dedup table: every node of this layer and one YES-child per branching node)
        // (POOLED: the pool may hold more nodes than a work list -- up to the node slots and the dedup table; only a layer is bounded)
        if (nwl + nwl2 > c.capW || (!POOLED && n > c.capW) || n + nwl > c.tab_limit || sh->ev_pos + 4ull * (uint64_t)(nwl + nwl2) + 12 > c.ev_cap) {
            PAR_BEGIN
            if (tid == 0) {
                sh->status = ST_ERR_CAPACITY - 100 * (nwl + nwl2 > c.capW ? 41 : ((!POOLED && n > c.capW) ? 42 : (n + nwl > c.tab_limit ? 44 : 43)));
                sh->bestKey = ((uint64_t)(uint32_t)nwl << 32) | (uint32_t)n;   // debugging aid: reported as best_value fields
                sh->nodes = sh->ev_pos;
                sh->arcs = (uint64_t)L;
            }
            PAR_END
            failed = true;
            break;
        }
        DD2_STAT(4, nwl + nwl2)
        DD2_STAT(7, naff_bound)
        DD2_TICK2(PH_WORKLIST, 13)
        // ------------------------------------------------------------ free slots for the YES-children
        // Each thread claims room in the list for the free bits of its bitmap words with one LDS atomic; only as
        // many slots as this transition can consume are listed (at most one YES-child per work item).
        const int need_free = nwl + 1 < c.capW ? nwl + 1 : c.capW;
        // Every slot at or above the high-water mark is dead.  While mark + need stays inside the first batch of the
        // sweeps (NT * KS slots: sweeping them costs the same whatever the mark), the children simply take the slots
        // above the mark and no list is built -- the common case of the many small DDs deep in a search.
        const int hiw_now = DD_UNIFORM(sh->hiw);
        const bool fl_implicit = hiw_now + need_free <= (capS < NT * KS ? capS : NT * KS);
        if (!fl_implicit) {
        PAR_BEGIN
        for (int ww = tid; ww < c.nbw; ww += NT) {
            if (sh->nfl >= need_free) break;
            uint32_t freebits = ~c.live[ww];
            const int base = ww * 32;
            if (base + 32 > capS) freebits &= ((1u << (capS - base)) - 1u);
            if (!freebits) continue;
            int pos = LDS_ADD_I32(&sh->nfl, dd_popc((uint64_t)freebits));
            while (freebits && pos < need_free) {
                int b = dd_ctz((uint64_t)freebits);
                c.fl[pos++] = (uint16_t)(base + b);
                freebits &= freebits - 1;
            }
        }
        PAR_END
        PAR_BEGIN
        if (tid == 0 && sh->nfl > need_free) sh->nfl = need_free;
        PAR_END
        }
        const int nfl_lim = fl_implicit ? need_free : DD_UNIFORM(sh->nfl);
        DD2_TICK2(PH_FREELIST, 14)
        // ------------------------------------------------------------ expand, phase 1 (clean.rs:360-370)
        const uint64_t aff_off = DD_UNIFORM64((sh->ev_pos + 3) & ~3ULL);   // 16-byte aligned records
        if (nwl2 > 0) {
        PAR_BEGIN
        // ---- fresh nodes without the variable: bound check only (clean.rs:362-365); a survivor's only child is the
        // node itself, it joins the dedup table with its cached hash.  With unit weights the rough upper bound is the
        // popcount held in the key, so the record is not even read.
        for (int j = tid; j < nwl2; j += NT) {
            const int s = c.wl[c.capW - 1 - j];
            const uint64_t kh = KH(c, s);
            const uint32_t key = (uint32_t)(kh >> 32);
            const uint32_t hs = (uint32_t)kh;
            const int32_t val = vbase + (int32_t)(key >> KEY_POP_BITS);
            bm_clr(c.fresh, s);
            bool pruned;
            uint64_t st[WS];
            if (c.unit_weights) {
                pruned = (int64_t)(key & KEY_POP_MASK) + (int64_t)val <= best_lb;
                if (pruned) ld_state<WS>(c, s, st);
            } else {
                ld_state<WS>(c, s, st);
                pruned = (int64_t)rub2_of<WS>(c, st) + (int64_t)val <= best_lb;
            }
            if (pruned) {
                add_bits<WS>(c.cnt, st, -1);
                bm_clr(c.live, s);
                LDS_ADD_I32(&sh->nlive, -1);
                const int r = LDS_ADD_I32(&sh->nrec, 1);
                U32x4* rec4 = (U32x4*)(c.ev + aff_off + 4ull * (uint64_t)r);
                *rec4 = U32x4{(uint32_t)s, NONE32, NONE32, NONE32};
                LDS_ADD_I32(&sh->npruned, 1);
            } else {
                tab2_insert_unique<WS>(c, s, hs);
            }
        }
        PAR_END
        }
        // ---- nodes containing the variable: expansion AND dedup (clean.rs:738-775) in one pass.  The table holds every
        // unchanged node of the next layer now, so a thread writes its two children, makes the stores visible to the
        // workgroup and inserts them right away -- the records never have to be read back.
        PAR_BEGIN
        uint64_t adjv[WS];   // (workgroup-uniform: kept in scalar registers)
#pragma unroll
        for (int k = 0; k < WS; ++k) adjv[k] = DD_UNIFORM64(c.adj[(size_t)var * WS + k]);
        const int32_t wv = DD_UNIFORM(c.weight[var]);
#if defined(DDO_HIP_PROBES)
        const bool probing = c.clocks && tid == 0;
        uint64_t probe[7] = {0, 0, 0, 0, 0, 0, 0};
        uint64_t probe_t = probing ? dd_clock() : 0;
#endif
        uint32_t kb_and = 0xFFFFFFFFu, kb_or = 0;   // key bits of the children (for the select of the next layer)
        for (int i = tid; i < nwl; i += NT) {
            const int s = c.wl[i];
            const uint64_t kh = KH(c, s);
            const uint32_t key = (uint32_t)(kh >> 32);
            const uint32_t oldh = (uint32_t)kh;
            const int32_t val = vbase + (int32_t)(key >> KEY_POP_BITS);
            const int pop = (int)(key & KEY_POP_MASK);
            uint64_t st[WS];
            uint32_t ppid = PID_NONE;   // the parent's best path: it stays the NO-child's, the YES-child extends it
            ld_state_p<WS>(c, s, st, ppid);
            DD2_PROBE(0)
            const int32_t rub = c.unit_weights ? pop : rub2_of<WS>(c, st);   // main.rs:191-193
            if (!POOLED) bm_clr(c.fresh, s);
            if ((int64_t)rub + (int64_t)val <= best_lb) {   // clean.rs:364-365: no children
                add_bits<WS>(c.cnt, st, -1);
                bm_clr(c.live, s);
                LDS_ADD_I32(&sh->nlive, -1);
                const int r = LDS_ADD_I32(&sh->nrec, 1);
                U32x4* rec4 = (U32x4*)(c.ev + aff_off + 4ull * (uint64_t)r);
                *rec4 = U32x4{(uint32_t)s | ((POOLED && bm_test(c.inex, s)) ? EV_PINEX : 0u), NONE32, NONE32, NONE32};
                if (pcache) {   // the threshold pass needs what this node was when it was pruned: value, bound, state (its slot is free from here on)
                    const uint32_t eid = (uint32_t)(aff_off >> 2) + (uint32_t)r;
                    c.pt[eid] = (uint64_t)ppid | ((uint64_t)(uint32_t)L << 32);
                    c.pvr[eid] = (uint64_t)(uint32_t)val | ((uint64_t)(uint32_t)rub << 32);
                    if (eid < c.pst_cap) {
#pragma unroll
                        for (int k = 0; k < WS; ++k) c.pst[(size_t)eid * WS + k] = st[k];
                    } else {
                        sh->status = ST_ERR_CAPACITY - 100 * 13;
                    }
                }
                LDS_ADD_I32(&sh->npruned, 1);
                continue;
            }
            bool hasv = false;
#pragma unroll
            for (int k = 0; k < WS; ++k)
                if (k == vw) hasv = (st[k] & vbit) != 0;
            if (!hasv) continue;   // fresh but unaffected: its NO-child is the node itself
            const bool pinex = POOLED && bm_test(c.inex, s);   // (read before the table publishes the slot: a twin folds its flag into it)
            // ---- decision NO, in place (main.rs:77-85)
            uint64_t oldw = 0, neww = 0;
#pragma unroll
            for (int k = 0; k < WS; ++k)
                if (k == vw) {
                    oldw = st[k];
                    st[k] &= ~vbit;
                    neww = st[k];
                }
            const uint32_t newh = oldh ^ fold32(mixw(oldw, vw) ^ mixw(neww, vw));   // (the hash is XOR-linear in the per-word mixes)
            // One word changed, and only that word goes out: the node's line was read a moment ago (by the sweep, then by this
            // thread), so the 8-byte store meets it in the L2 -- one write request instead of four, -0.5 of the 3.8 L1->L2 requests
            // per expanded node, +1.5 % on the bench.  (Round 3 rewrote the whole line after a micro-benchmark that stored into
            // COLD lines, where a partial store costs more than a full one: DDO_NO_CHILD_LINE restores that.)
#if defined(DDO_NO_CHILD_LINE)
            st_node<WS>(c, s, st, ppid);
#else
            c.rec[(size_t)s * c.RW + vw] = neww;
#endif
            const uint32_t kno = key - 1;  // popcount - 1, same value (cost 0)
            kb_and &= kno;
            kb_or |= kno;
            KH_ST(c, s, kno, newh);   // (cnt[var]: no state of the next layer contains the variable, it is zeroed with the layer)
            // ---- decision YES into a free slot (main.rs:95-102)
            const int r = LDS_ADD_I32(&sh->nrec, 1);                       // this node's event record ...
            const uint32_t eid = (uint32_t)(aff_off >> 2) + (uint32_t)r;   // ... is also the path-tree node of its YES arc
            if (pcache) {
                c.pvr[eid] = (uint64_t)(uint32_t)val | ((uint64_t)(uint32_t)rub << 32);
                if (eid < c.pst_cap) {   // the node as it was expanded (st holds its NO-child by now: word vw back to what it was)
#pragma unroll
                    for (int k = 0; k < WS; ++k) c.pst[(size_t)eid * WS + k] = k == vw ? oldw : st[k];
                } else {
                    sh->status = ST_ERR_CAPACITY - 100 * 13;
                }
            }
            const int fi = LDS_ADD_I32(&sh->nnew, 1);
            int ny = -1;
            uint64_t y[WS];
            uint32_t yh = 0;
            uint32_t kyes = 0;
#pragma unroll
            for (int k = 0; k < WS; ++k) y[k] = 0;
            if (fi < nfl_lim) {
                ny = fl_implicit ? hiw_now + fi : (int)c.fl[fi];
                int ypop = 0;
#pragma unroll
                for (int k = 0; k < WS; ++k) {
                    y[k] = st[k] & adjv[k];
                    ypop += dd_popc(y[k]);
                }
                // the child's best path is the parent's plus decision 1 at layer L: path-tree node eid
                c.pt[eid] = (uint64_t)ppid | ((uint64_t)(uint32_t)L << 32);
                yh = hash32_state<WS>(y);
                st_node<WS>(c, ny, y, eid);
                kyes = ((uint32_t)(val + wv - vbase) << KEY_POP_BITS) | (uint32_t)ypop;
                kb_and &= kyes;
                kb_or |= kyes;
                KH_ST(c, ny, kyes, yh);
                bm_put(c.inex, ny, bm_test(c.inex, s));
                bm_put(c.okb, ny, bm_test(c.okb, s));
            } else {
                sh->status = ST_ERR_CAPACITY - 100 * 6;
            }
            FENCE_BLOCK();   // both records are visible to the workgroup before the table publishes their slots
            DD2_PROBE(1)
            // ---- dedup of both children (append_edge_to!, clean.rs:199-220)
            uint32_t e_no = (uint32_t)s, e_yes = NONE32;
            int t0, t1;
            tab2_insert2<WS>(c, s, newh, st, ny, yh, y, t0, t1);
            DD2_PROBE(2)
            // twins: both keys' atomics are in flight together, their results are needed for the event record only
            uint32_t old0 = 0, old1 = 0;
            if (t0 != s) old0 = K32_MAX(c, t0, kno, newh);
            if (ny >= 0 && t1 != ny) old1 = K32_MAX(c, t1, kyes, yh);
            if (t0 == s) {
                if (!POOLED) bm_set(c.fresh, s);   // it stays in the layer; its rub shrank: check it again before it is expanded
            } else {                         // the in-place NO-child dissolves into its twin t0
                if (bm_test(c.inex, s)) bm_set(c.inex, t0);
                add_bits<WS>(c.cnt, st, -1);
                bm_clr(c.live, s);
                LDS_ADD_I32(&sh->nlive, -1);
            }
            DD2_PROBE(5)
            if (ny >= 0) {
                if (t1 == ny) {              // a new node enters the layer
                    bm_set(c.live, ny);
                    if (!POOLED) bm_set(c.fresh, ny);
                    add_bits<WS>(c.cnt, y, +1);
                    LDS_ADD_I32(&sh->nlive, 1);
                    LDS_MAX_I32(&sh->hiw, ny + 1);
                    e_yes = (uint32_t)ny | EV_CREATED;
                } else {
                    if (bm_test(c.inex, ny)) bm_set(c.inex, t1);
                }
            }
            if (t0 != s) e_no = (uint32_t)t0 | (kno > old0 ? EV_RAISED : 0u);
            if (ny >= 0 && t1 != ny) e_yes = (uint32_t)t1 | (kyes > old1 ? EV_RAISED : 0u);
            DD2_PROBE(6)
            U32x4* rec4 = (U32x4*)(c.ev + aff_off + 4ull * (uint64_t)r);
            // parent | NO target | YES target | slot allocated for the YES-child
            *rec4 = U32x4{(uint32_t)s | (pinex ? EV_PINEX : 0u), e_no, e_yes, ny >= 0 ? (uint32_t)ny : NONE32};
            DD2_PROBE(4)
#if defined(DDO_HIP_PROBES)
            if (probing && i == tid) {   // thread 0's first node went the whole way: charge its chain
                for (int q = 0; q < 7; ++q) sh->mk[16 + q] += probe[q];
                sh->mk[23] += 1;
            }
#endif
        }
        if (kb_or != 0 || kb_and != 0xFFFFFFFFu) {
            LDS_AND_U32(&sh->kbits_and[(L + 1) & 1], kb_and);
            LDS_OR_U32(&sh->kbits_or[(L + 1) & 1], kb_or);
        }
        PAR_END
        const int nrec = DD_UNIFORM(sh->nrec);
        if (sh->status != ST_OK) { failed = true; break; }
        DD2_STAT(5, nrec)
        DD2_TICK(PH_EXP1)
        // (the dedup table of the unchanged nodes was filled by the work-list sweep and by the fresh survivors above)
        DD2_TICK(PH_TABLE)

        // (expand phase 2, the dedup, is part of phase 1 now)
        DD2_TICK(PH_EXP2)
        // ------------------------------------------------------------ expand, phase 3: new best parents
        // The arc that RAISED its target's key and still equals the target's final key is the one that set
        // the maximum (`value >= value_top`, clean.rs:215-218): the target inherits the loser's path.  Losers
        // are dead, targets alive, so no path is read and written in the same phase.
        const bool tie_pass = relaxed && lel >= 0;   // see the region after this one
        PAR_BEGIN
        for (int r = tid; r < nrec; r += NT) {
            GLB_PTR(uint32_t) rec = c.ev + aff_off + 4ull * (uint64_t)r;
            if (rec[1] == NONE32) continue;
            for (int which = 0; which < 2; ++which) {
                const uint32_t w = rec[1 + which];
                if (w == NONE32 || !(w & EV_RAISED)) continue;
                const int t = (int)(w & EV_SLOT_MASK);
                const int x = which == 0 ? (int)(rec[0] & EV_SLOT_MASK) : (int)rec[3];
                if (K32(c, t) == K32(c, x)) {
                    // the NO arc carries the parent's path as it was when the record was written, the YES arc that plus its own node
                    const uint32_t eid = (uint32_t)(aff_off >> 2) + (uint32_t)r;
                    pid_st<WS>(c, t, which == 0 ? (uint32_t)c.pt[eid] : eid);
                    bm_put(c.okb, t, bm_test(c.okb, x));
                }
                rec[1 + which] = w & ~EV_RAISED;
            }
        }
        // Equal-valued arcs (order-independent EBPO, same rule as KEY_OK in misp_dd_core.hpp and Problem::canonical_ties in
        // the CPU restatement): when an arc TIES with its target's final value and comes from a node with an exact best
        // path, the target has one too -- whichever thread got there first.  Only inexact layers can disagree (before the
        // first squash every path is exact), so the extra region exists only below the last exact layer of a relaxed DD;
        // the lane that flips the bit copies its path (one winner: no two paths are mixed).
        if (!tie_pass) {
            if (tid == 0) {
                // the bookkeeping of the layer goes out with the last region: nothing reads it before the backward pass
                c.lvar[L] = var;
                c.lmerge[L] = merged_slot;
                c.ldup[2 * L] = dup_from;
                c.ldup[2 * L + 1] = dup_to;
                GLB_PTR(uint32_t) eo = c.evoff + (size_t)L * 8;
                eo[0] = (uint32_t)aff_off;
                eo[1] = (uint32_t)(aff_off >> 32);
                eo[2] = (uint32_t)nrec;
                eo[3] = (uint32_t)del_off;
                eo[4] = (uint32_t)(del_off >> 32);
                eo[5] = (uint32_t)n_del;
                sh->ev_pos = aff_off + 4ull * (uint64_t)nrec;
                sh->nodes += POOLED ? (uint64_t)nwl : (uint64_t)n;   // (POOLED: the nodes that are expanded are the impacted ones, pooled.rs:351)
                if (n > sh->maxn) sh->maxn = n;
                sh->nyes = nrec - sh->npruned;   // every record is a pruned node or a node with a YES-child
                sh->arcs += (POOLED ? (uint64_t)sh->nyes : (uint64_t)(n - sh->npruned)) + (uint64_t)sh->nyes;
                c.cnt[var] = 0;
    #if defined(DDO_HOST_EMULATION)
                if (getenv("DD_TRACE")) std::printf("E2 L=%d var=%d n=%d pruned=%d yes=%d nU_next=%d squash=%d\n", L, var, n, sh->npruned, sh->nyes, sh->nlive, (int)squash);
    #endif
                sh->kbits_and[L & 1] = 0xFFFFFFFFu;   // (this layer's select is over: the slot is filled again while layer L + 1 is built)
                sh->kbits_or[L & 1] = 0;
                sh->varkey = 0xFFFFFFFFu;   // next layer's next_variable / high-water mark / work lists start from scratch
                sh->hiw = 1;
                sh->nwl = 0;
                sh->nwl2 = 0;
                sh->nrec = 0;
                sh->nnew = 0;
                sh->nfl = 0;
                sh->npruned = 0;
            }
        }
        PAR_END
        if (tie_pass) {
            PAR_BEGIN
            for (int r = tid; r < nrec; r += NT) {
                GLB_PTR(const uint32_t) rec = c.ev + aff_off + 4ull * (uint64_t)r;
                if (rec[1] == NONE32) continue;
                for (int which = 0; which < 2; ++which) {
                    const uint32_t w = rec[1 + which];
                    if (w == NONE32) continue;
                    const int t = (int)(w & EV_SLOT_MASK);
                    const int x = which == 0 ? (int)(rec[0] & EV_SLOT_MASK) : (int)rec[3];
                    if (t == x) continue;   // the child is the holder itself
                    if (K32(c, t) == K32(c, x) && bm_test(c.okb, x) && !bm_test(c.okb, t)) {
                        const uint32_t bit = 1u << (t & 31);
                        const uint32_t old = LDS_OR_U32(&c.okb[t >> 5], bit);
                        if (!(old & bit)) {
                            const uint32_t eid = (uint32_t)(aff_off >> 2) + (uint32_t)r;
                            pid_st<WS>(c, t, which == 0 ? (uint32_t)c.pt[eid] : eid);
                        }
                    }
                }
            }
            if (tid == 0) {
                // the bookkeeping of the layer goes out with the last region: nothing reads it before the backward pass
                c.lvar[L] = var;
                c.lmerge[L] = merged_slot;
                c.ldup[2 * L] = dup_from;
                c.ldup[2 * L + 1] = dup_to;
                GLB_PTR(uint32_t) eo = c.evoff + (size_t)L * 8;
                eo[0] = (uint32_t)aff_off;
                eo[1] = (uint32_t)(aff_off >> 32);
                eo[2] = (uint32_t)nrec;
                eo[3] = (uint32_t)del_off;
                eo[4] = (uint32_t)(del_off >> 32);
                eo[5] = (uint32_t)n_del;
                sh->ev_pos = aff_off + 4ull * (uint64_t)nrec;
                sh->nodes += POOLED ? (uint64_t)nwl : (uint64_t)n;   // (POOLED: the nodes that are expanded are the impacted ones, pooled.rs:351)
                if (n > sh->maxn) sh->maxn = n;
                sh->nyes = nrec - sh->npruned;   // every record is a pruned node or a node with a YES-child
                sh->arcs += (POOLED ? (uint64_t)sh->nyes : (uint64_t)(n - sh->npruned)) + (uint64_t)sh->nyes;
                c.cnt[var] = 0;
    #if defined(DDO_HOST_EMULATION)
                if (getenv("DD_TRACE")) std::printf("E2 L=%d var=%d n=%d pruned=%d yes=%d nU_next=%d squash=%d\n", L, var, n, sh->npruned, sh->nyes, sh->nlive, (int)squash);
    #endif
                sh->kbits_and[L & 1] = 0xFFFFFFFFu;   // (this layer's select is over: the slot is filled again while layer L + 1 is built)
                sh->kbits_or[L & 1] = 0;
                sh->varkey = 0xFFFFFFFFu;   // next layer's next_variable / high-water mark / work lists start from scratch
                sh->hiw = 1;
                sh->nwl = 0;
                sh->nwl2 = 0;
                sh->nrec = 0;
                sh->nnew = 0;
                sh->nfl = 0;
                sh->npruned = 0;
            }
            PAR_END
        }
        DD2_TICK(PH_EXPAND)
        L += 1;
    }

    // ==================================================================== _finalize (clean.rs:407-414)
    const int nT = failed ? 0 : sh->nlive;          // terminal layer = the live nodes, never squashed
    const int n_layers = failed ? L : (nT > 0 ? L + 1 : L);
    PAR_BEGIN
    if (tid == 0) {
        sh->bestKey = 0;
        sh->bestExactKey = 0;
        if (L < c.max_layers) {
            c.lmerge[L] = -1;
            c.ldup[2 * L] = -1;
            c.ldup[2 * L + 1] = -1;
            c.evoff[(size_t)L * 8 + 5] = 0;
            c.evoff[(size_t)L * 8 + 6] = 0;   // (no node of a layer that was never built was pruned by the cache: the replay reads the entry)
        }
    }
    PAR_END
    if (!failed && nT > 0) {
        PAR_BEGIN   // _find_best_node (clean.rs:620-632)
        for (int s = tid; s < sh->hiw; s += NT) {
            if (!bm_test(c.live, s)) continue;
            const uint64_t bk = (((uint64_t)K32(c, s) >> KEY_POP_BITS) << 32) | (uint32_t)s;
            LDS_MAX_U64(&sh->bestKey, bk + 1);
            if (!bm_test(c.inex, s)) LDS_MAX_U64(&sh->bestExactKey, bk + 1);
        }
        PAR_END
    }
    const bool is_exact = lel < 0;
    const bool has_best = !failed && sh->bestKey != 0;
    int best_slot = -1, best_value = 0;
    if (has_best) {
        const uint64_t bk = sh->bestKey - 1;
        best_slot = (int)(uint32_t)bk;
        best_value = vbase + (int32_t)(bk >> 32);
    }
    bool has_best_exact = !failed && sh->bestExactKey != 0;
    int exact_slot = -1, exact_value = 0;
    if (has_best_exact) {
        const uint64_t bk = sh->bestExactKey - 1;
        exact_slot = (int)(uint32_t)bk;
        exact_value = vbase + (int32_t)(bk >> 32);
    }
    // EBPO (clean.rs:636-655): ok bit of the best terminal node
    bool ebpo = false;
    if (relaxed && !failed) {
        ebpo = has_best ? bm_test(c.okb, best_slot) : true;
        if (ebpo && has_best) {
            has_best_exact = true;
            exact_slot = best_slot;
            exact_value = best_value;
        }
    }

    DD2_TICK(PH_FINAL)
    if (POOLED) {
        // ============================================================ Pooled: frontier cut-set + local bounds in ONE backward replay
        // (pooled.rs:438-467, 543-566).  ix = `inex` bitmap, replayed backwards: before transition tr is undone it tells, per slot,
        // whether the node living there AFTER the transition ends up inexact; a parent is restored with the flag of its record.
        const bool want_cs = relaxed && !failed && lel >= 0 && has_best;   // (exact DDs have no cut-set: pooled.rs:556; drain needs a best value: :410)
        // _compute_thresholds (pooled.rs:467-515) rides the same replay: th[slot] = theta of the node living there after the transition
        // that is being undone (TH_NONE: none).  `if input.comp_type == Relaxed || self.is_exact` (:469).
        const bool want_th = pcache && !failed && sh->status == ST_OK && (relaxed || is_exact);
        int32_t* vb = c.keyh ? (int32_t*)c.keyh : (int32_t*)c.key32;   // the ranking keys are dead now: reuse their storage
        int32_t* th = c.keyh ? vb + capS : (int32_t*)c.h32;             // (the packed key|hash words hold two int32 per slot; else the hash array)
        int32_t* tmp = (int32_t*)c.wl;                                   // wl + fl = capW x int32
        int32_t* tmp2 = (int32_t*)c.cs_value;                            // (the snapshot arrays of the default DD are idle in a pooled one: capN >= capW int32)
        int64_t bk64 = best_lb;
        if (has_best_exact && (int64_t)exact_value > bk64) bk64 = exact_value;
        const int32_t bk = bk64 < -(1 << 30) ? -(1 << 30) : (int32_t)bk64;   // best_known (:471-475); values are far above: same comparisons
        const bool bk_min = bk64 <= -((int64_t)1 << 39);                      // no bound known: best_known == isize::MIN
        const uint64_t cbase = (sh->ev_pos + 3) & ~3ULL;                 // cut list behind the event records: (record id, value_bot) pairs
        GLB_PTR(uint32_t) cutl = c.ev + cbase;
        const uint64_t cut_cap = c.ev_cap > cbase + 8 ? (c.ev_cap - cbase) / 2 : 0;
        PAR_BEGIN
        if (tid == 0) {
            sh->ncut = 0;
            sh->ncut2 = 0;
        }
        PAR_END
        if (want_cs || want_th) {
            if (want_th) {
                PAR_BEGIN
                for (int i = tid; i < c.nbw; i += NT) c.fresh[i] = 0;
                PAR_END
            }
            PAR_BEGIN   // what is left in the pool is the last layer: value_bot = 0 and MARKED (pooled.rs:441-446)
            for (int s = tid; s < capS; s += NT) {
                const bool lv = s < sh->hiw && bm_test(c.live, s);
                vb[s] = (want_cs && lv) ? 0 : VB_UNMARKED;
                if (want_th) {
                    // :476-482 exact pool nodes start from best_known when there is a best exact node; then their own rule (never expanded:
                    // rub == isize::MAX, no children, not in the cut-set) and _maybe_update_cache -- depth = the layer the pool was left at (:583)
                    int32_t t = TH_NONE;
                    if (lv) {
                        const bool exact = !bm_test(c.inex, s);
                        if (exact && has_best_exact) t = bk;
                        t = pooled_own_theta(t, 0, INT32_MAX, false, exact, VB_UNMARKED, bk, bk_min);
                        // (its cache update waits, like every other, until the output arena is reserved: `fresh` -- idle here -- remembers
                        // the exact nodes of the final pool, the replay below rewrites `inex`)
                        if (exact && t != TH_NONE) bm_set(c.fresh, s);
                    }
                    th[s] = t;
                }
            }
            PAR_END
            for (int tr = L - 1; tr >= 0; --tr) {
                // (1) undo the squash of layer tr+1: arcs into a deleted node were redirected to the merged node (pooled.rs:806-819),
                // which is inexact
                GLB_PTR(const uint32_t) eo1 = c.evoff + (size_t)(tr + 1) * 8;
                const int nd = (int)eo1[5];
                const int m = c.lmerge[tr + 1];
                const uint64_t doff = (uint64_t)eo1[3] | ((uint64_t)eo1[4] << 32);
                if (nd > 0 && m >= 0) {
                    const int dfrom = c.ldup[2 * (tr + 1)], dto = c.ldup[2 * (tr + 1) + 1];
                    PAR_BEGIN
                    const int32_t vm = vb[m];
                    const int32_t tm = want_th ? th[m] : TH_NONE;
                    for (int i = tid; i < nd; i += NT) {
                        const uint32_t d = c.ev[doff + i];
                        if (d != NONE32 && (int)d != m) {
                            vb[d] = vm;
                            if (want_th) th[d] = tm;
                            bm_set(c.inex, (int)d);
                        }
                    }
                    PAR_END
                    if (dfrom >= 0) {
                        PAR_BEGIN
                        if (tid == 0) {
                            int32_t a = vb[dfrom], b = vb[dto];
                            vb[dfrom] = a > b ? a : b;   // X keeps its own arcs and they were also redirected (pooled.rs:821-825)
                            if (want_th) {
                                const int32_t ta = th[dfrom], tb = th[dto];
                                th[dfrom] = ta < tb ? ta : tb;   // (TH_NONE is the largest value: min() ignores it)
                            }
                            bm_set(c.inex, dfrom);       // (for the arcs INTO X: they also reach the relaxed node; X's own flag is in its record)
                        }
                        PAR_END
                    }
                } else if (nd > 0 && want_th) {   // restricted: a dropped node passes nothing on (no merged node took its arcs)
                    PAR_BEGIN
                    for (int i = tid; i < nd; i += NT) {
                        const uint32_t d = c.ev[doff + i];
                        if (d != NONE32) th[d] = TH_NONE;
                    }
                    PAR_END
                }
                // (1b) the nodes of layer tr+1 the cache pruned (:662-680): back in their slots with the cached threshold as theta; they were
                // not expanded (no children: unmarked) and only pass that theta on (pruned_by_cache: no rule of their own, no update)
                if (pcache) {
                    const int ncp = (int)eo1[6];
                    if (ncp > 0) {
                        const uint64_t cpo = doff - 2ull * (uint64_t)ncp;
                        PAR_BEGIN
                        for (int i = tid; i < ncp; i += NT) {
                            const uint32_t w0 = c.ev[cpo + 2 * (uint64_t)i];
                            const int sl = (int)(w0 & EV_SLOT_MASK);
                            vb[sl] = VB_UNMARKED;
                            th[sl] = want_th ? (int32_t)c.ev[cpo + 2 * (uint64_t)i + 1] : TH_NONE;
                            bm_put(c.inex, sl, (w0 & EV_PINEX) != 0);
                        }
                        PAR_END
                    }
                }
                // (2) parents of the transition tr -> tr+1
                GLB_PTR(const uint32_t) eo = c.evoff + (size_t)tr * 8;
                const int na = (int)eo[2];
                const uint64_t aoff = (uint64_t)eo[0] | ((uint64_t)eo[1] << 32);
                const int32_t wv = c.weight[c.lvar[tr]];
                if (na > c.capW) {   // (cannot happen: a work list holds at most capW nodes)
                    PAR_BEGIN
                    if (tid == 0) sh->status = ST_ERR_INTERNAL;
                    PAR_END
                    break;
                }
                PAR_BEGIN
                for (int i = tid; i < na; i += NT) {
                    GLB_PTR(uint32_t) rec = c.ev + aoff + 4ull * (uint64_t)i;
                    int32_t best = VB_UNMARKED;
                    int32_t tch = TH_NONE;   // what the children pass up: min over the arcs of theta(child) - cost (:519-525)
                    bool cix = false;
                    if (rec[1] != NONE32) {
                        const int t = (int)(rec[1] & EV_SLOT_MASK);
                        const int32_t v = vb[t];
                        if (v != VB_UNMARKED) best = v;
                        cix |= bm_test(c.inex, t);
                        if (want_th && th[t] != TH_NONE) tch = th[t];
                    }
                    if (rec[2] != NONE32) {
                        const int t = (int)(rec[2] & EV_SLOT_MASK);
                        const int32_t v = vb[t];
                        if (v != VB_UNMARKED && v + wv > best) best = v + wv;
                        cix |= bm_test(c.inex, t);
                        if (want_th && th[t] != TH_NONE) {
                            const int32_t u = th_sub(th[t], wv);
                            if (u < tch) tch = u;
                        }
                    }
                    tmp[i] = best;
                    if (cix && !(rec[0] & EV_PINEX)) rec[0] |= EV_PCUT;   // exact parent of an inexact node: frontier (pooled.rs:552-560)
                    if (want_th) {
                        const uint32_t p0 = rec[0];
                        const bool exact = !(p0 & EV_PINEX);
                        const bool is_cut = (p0 & EV_PCUT) != 0;
                        const uint32_t eid = (uint32_t)(aoff >> 2) + (uint32_t)i;
                        const uint64_t vr = c.pvr[eid];
                        const int32_t val = (int32_t)(uint32_t)vr, rub = (int32_t)(uint32_t)(vr >> 32);
                        const int32_t t = pooled_own_theta(tch, val, rub, is_cut, exact, want_cs ? best : VB_UNMARKED, bk, bk_min);
                        tmp2[i] = t;
                        // _maybe_update_cache (:527-535): above the cut-set == exact (:548-550), explored unless in the cut-set.  The update
                        // itself waits until the output arena is reserved (a compile that finds the arena full runs again and must not meet
                        // its own thresholds): the record keeps theta | update << 32 | explored << 33 in place of (value, rub)
                        c.pvr[eid] = (uint64_t)(uint32_t)t | ((exact && t != TH_NONE) ? (1ULL << 32) : 0ULL) | (!is_cut ? (1ULL << 33) : 0ULL);
                    }
                }
                PAR_END
                PAR_BEGIN
                for (int i = tid; i < na; i += NT) {
                    GLB_PTR(const uint32_t) rec = c.ev + aoff + 4ull * (uint64_t)i;
                    const uint32_t p0 = rec[0];
                    const int s = (int)(p0 & EV_SLOT_MASK);
                    vb[s] = tmp[i];
                    if (want_th) th[s] = tmp2[i];
                    bm_put(c.inex, s, (p0 & EV_PINEX) != 0);
                    if (want_cs && (p0 & EV_PCUT) && tmp[i] != VB_UNMARKED) {   // drain_cutset keeps the MARKED nodes only (pooled.rs:414)
                        const uint64_t k = (uint64_t)LDS_ADD_I32(&sh->ncut, 1);
                        if (k < cut_cap) {
                            cutl[2 * k] = (uint32_t)(aoff >> 2) + (uint32_t)i;
                            cutl[2 * k + 1] = (uint32_t)tmp[i];
                        }
                    }
                }
                PAR_END
            }
        }
        const int ncut = sh->ncut;
        const bool cut_over = (uint64_t)ncut > cut_cap;
        const bool want_paths = (in.flags & IN_WANT_PATHS) != 0;
        const bool emit_best = has_best && (want_paths || (int64_t)best_value > best_lb);
        const bool emit_exact = has_best_exact && (want_paths || (int64_t)exact_value > best_lb);
        const bool same = emit_best && emit_exact && exact_slot == best_slot;
        const int path_len = L;   // one decision bit per transition; the host drops the layers whose variable did not impact the path's node
        const int best_len = emit_best ? path_len : 0;
        const int exact_len = (emit_exact && !same) ? path_len : 0;
        const uint32_t pw = (uint32_t)((path_len + 63) / 64);
        uint64_t off = 0;
        const uint64_t path_off = off;
        off += ((uint64_t)best_len * 4 + 7) & ~7ULL;
        const uint64_t exact_off = off;
        off += ((uint64_t)exact_len * 4 + 7) & ~7ULL;
        const uint64_t cs_state_off = off;
        off += (uint64_t)ncut * WS * 8;
        const uint64_t cs_value_off = off;
        off += ((uint64_t)ncut * 4 + 7) & ~7ULL;
        const uint64_t cs_ub_off = off;
        off += ((uint64_t)ncut * 4 + 7) & ~7ULL;
        const uint64_t cs_depth_off = off;
        off += ((uint64_t)ncut * 4 + 7) & ~7ULL;
        const uint64_t cs_path_off = off;
        off += (uint64_t)ncut * pw * 8;
        const uint64_t cs_lvar_off = off;
        off += ((uint64_t)path_len * 4 + 7) & ~7ULL;
        const uint64_t total = off;
        PAR_BEGIN
        if (tid == 0) {
            unsigned long long a = total ? GLB_ADD_U64(c.arena_head, (unsigned long long)total) : 0ULL;
            sh->arena_off = a;
            if (a + total > c.arena_cap) sh->status = ST_ERR_ARENA;
            if (cut_over) sh->status = ST_ERR_CAPACITY - 100 * 12;   // the cut list outgrew the room behind the event records
        }
        PAR_END
        const bool arena_ok = sh->status == ST_OK || sh->status == ST_CUTOFF;
        GLB_PTR(uint8_t) base = c.arena + sh->arena_off;
        if (want_th && sh->status == ST_OK) {
            // ---- _maybe_update_cache for every node the replay flagged, with the state logged when the node was expanded (c.pst)
            for (int tr = 0; tr < L; ++tr) {
                GLB_PTR(const uint32_t) eo = c.evoff + (size_t)tr * 8;
                const int na = (int)eo[2];
                const uint64_t aoff = (uint64_t)eo[0] | ((uint64_t)eo[1] << 32);
                PAR_BEGIN
                for (int i = tid; i < na; i += NT) {
                    const uint32_t eid = (uint32_t)(aoff >> 2) + (uint32_t)i;
                    const uint64_t w = c.pvr[eid];
                    if (!(w & (1ULL << 32))) continue;
                    uint64_t st[WS];
#pragma unroll
                    for (int k = 0; k < WS; ++k) st[k] = c.pst[(size_t)eid * WS + k];
                    cache_update<WS>(c, st, c.depth0 + tr, th_pack((int32_t)(uint32_t)w, (w & (1ULL << 33)) != 0));
                }
                PAR_END
            }
            PAR_BEGIN
            for (int s = tid; s < capS; s += NT) {
                if (!bm_test(c.fresh, s)) continue;
                int32_t t = has_best_exact ? bk : TH_NONE;
                t = pooled_own_theta(t, 0, INT32_MAX, false, true, VB_UNMARKED, bk, bk_min);
                uint64_t st[WS];
                ld_state<WS>(c, s, st);
                cache_update<WS>(c, st, c.depth0 + L, th_pack(t, true));
            }
            PAR_END
        }
        if (arena_ok && !failed) {
            LDS_PTR(uint64_t) bbits = (LDS_PTR(uint64_t))sh->merged;
            LDS_PTR(uint64_t) xbits = (LDS_PTR(uint64_t))sh->xcand;   // 64 x int32 = 32 words >= WS
            PAR_BEGIN
            if (tid == 0 && best_len) {
                uint64_t b[WS];
#pragma unroll
                for (int k = 0; k < WS; ++k) b[k] = 0;
                path_bits<WS>(c, pid_ld<WS>(c, best_slot), b);
#pragma unroll
                for (int k = 0; k < WS; ++k) bbits[k] = b[k];
            }
            if (tid == 64 % NT && exact_len) {
                uint64_t b[WS];
#pragma unroll
                for (int k = 0; k < WS; ++k) b[k] = 0;
                path_bits<WS>(c, pid_ld<WS>(c, exact_slot), b);
#pragma unroll
                for (int k = 0; k < WS; ++k) xbits[k] = b[k];
            }
            PAR_END
            PAR_BEGIN
            if (best_len) {
                uint32_t* out = (uint32_t*)(base + path_off);
                for (int i = tid; i < best_len; i += NT) {
                    const int tr = path_len - 1 - i;
                    out[i] = ((uint32_t)c.lvar[tr] << 1) | (uint32_t)((bbits[tr >> 6] >> (tr & 63)) & 1ULL);
                }
            }
            if (exact_len) {
                uint32_t* out = (uint32_t*)(base + exact_off);
                for (int i = tid; i < exact_len; i += NT) {
                    const int tr = path_len - 1 - i;
                    out[i] = ((uint32_t)c.lvar[tr] << 1) | (uint32_t)((xbits[tr >> 6] >> (tr & 63)) & 1ULL);
                }
            }
            if (ncut) {
                uint32_t* o_lvar = (uint32_t*)(base + cs_lvar_off);
                for (int j = tid; j < path_len; j += NT) o_lvar[j] = (uint32_t)c.lvar[j];
                uint64_t* o_state = (uint64_t*)(base + cs_state_off);
                int32_t* o_value = (int32_t*)(base + cs_value_off);
                int32_t* o_ub = (int32_t*)(base + cs_ub_off);
                int32_t* o_depth = (int32_t*)(base + cs_depth_off);
                uint64_t* o_bits = (uint64_t*)(base + cs_path_off);
                for (int q = tid; q < ncut; q += NT) {
                    const uint32_t eid = cutl[2 * (uint64_t)q];
                    const int32_t locb = (int32_t)cutl[2 * (uint64_t)q + 1];
                    const uint64_t pe = c.pt[eid];
                    const int depth = (int)(pe >> 32);   // the layer at which the node was expanded (pooled.rs:622)
                    uint64_t pb[WS], st[WS];
#pragma unroll
                    for (int k = 0; k < WS; ++k) pb[k] = 0;
                    path_bits<WS>(c, (uint32_t)pe, pb);
                    // An exact node's state is the residual state with the decisions of (any of) its paths applied (main.rs:77-85),
                    // its value the residual value plus the weights of the 1-decisions (main.rs:87-93).
                    if (in.src_off != NO_POOL_SRC) {
                        const PoolBlockHeader* h = (const PoolBlockHeader*)(c.pool + in.src_off);
                        const uint64_t* rows = (const uint64_t*)(c.pool + in.src_off + h->off_states);
#pragma unroll
                        for (int k = 0; k < WS; ++k) st[k] = k < (int)h->ws ? rows[(size_t)k * h->rows + in.src_row] : 0;
                    } else {
#pragma unroll
                        for (int k = 0; k < WS; ++k) st[k] = in.state[k];
                    }
                    int64_t v = in.value;
                    for (int j = 0; j < depth; ++j) {
                        const int x = c.lvar[j];
                        const int xw = x >> 6;
                        const uint64_t xb = 1ULL << (x & 63);
                        bool has = false, yes = false;
#pragma unroll
                        for (int k = 0; k < WS; ++k) {
                            if (k == xw) {
                                has = (st[k] & xb) != 0;
                                st[k] &= ~xb;
                            }
                            if (k == (j >> 6)) yes = ((pb[k] >> (j & 63)) & 1ULL) != 0;
                        }
                        if (has && yes) {
#pragma unroll
                            for (int k = 0; k < WS; ++k) st[k] &= c.adj[(size_t)x * WS + k];
                            v += c.weight[x];
                        }
                    }
                    int64_t ub = v + (int64_t)rub2_of<WS>(c, st);
                    if (v + locb < ub) ub = v + locb;
                    if (best_value < ub) ub = best_value;
#pragma unroll
                    for (int k = 0; k < WS; ++k) o_state[(size_t)q * WS + k] = st[k];
                    o_value[q] = (int32_t)v;
                    o_ub[q] = (int32_t)ub;
                    o_depth[q] = depth;
#pragma unroll
                    for (int k = 0; k < WS; ++k)
                        if ((uint32_t)k < pw) o_bits[(size_t)q * pw + k] = pb[k];
                }
            }
            PAR_END
        }
        PAR_BEGIN
        if (tid == 0) {
            DDResult r;
            r.status = sh->status;
            if (c.tier && (sh->status == ST_ERR_CAPACITY || (sh->status <= ST_ERR_CAPACITY - 100 && sh->status != ST_ERR_ARENA &&
                                                             sh->status != ST_ERR_CAPACITY - 800)))
                r.status = ST_RETRY;
            r.comp_type = comp_type;
            r.is_exact = is_exact ? 1 : 0;
            r.has_exact_best_path = ebpo ? 1 : 0;
            r.has_best = has_best ? 1 : 0;
            r.has_best_exact = has_best_exact ? 1 : 0;
            r.best_value = best_value;
            r.best_exact_value = exact_value;
            r.n_layers = n_layers;
            r.lel = lel;
            r.n_cutset = arena_ok ? ncut : 0;
            r.best_len = arena_ok ? best_len : 0;
            r.exact_len = arena_ok ? (same ? best_len : exact_len) : 0;
            r.exact_same_as_best = same ? 1 : 0;
            r.recycled_merges = sh->recycled_merges;
            r.max_width_seen = (uint32_t)sh->maxn;
            r.arena_off = sh->arena_off;
            r.arena_bytes = total;
            r.nodes_expanded = sh->nodes;
            r.arcs = sh->arcs;
            r.layers = (uint64_t)L;
            r.path_off = path_off;
            r.exact_off = same ? path_off : exact_off;
            r.cs_state_off = cs_state_off;
            r.cs_value_off = cs_value_off;
            r.cs_ub_off = cs_ub_off;
            r.cs_path_off = cs_path_off;
            sh->clk[PH_FINAL] += dd_clock() - sh->clk_last;
            for (int k = 0; k < 8; ++k) r.phase_clk[k] = sh->clk[k];
            for (int k = 0; k < 24; ++k) r.phase_clk[8 + k] = sh->mk[k];
            r.pool_off = NO_POOL_SRC;
            r.cs_depth_off = (arena_ok && ncut) ? cs_depth_off : 0;
            r.cs_path_stride = path_len;
            r.cs_lvar_off = (arena_ok && ncut) ? cs_lvar_off : 0;
            r.cache_hits = cache_hits;
            *res = r;
        }
        PAR_END
        return;
    }
    // ---------------------------------------------------------------- local bounds (clean.rs:448-475)
    const bool want_cutset = relaxed && !failed && lel >= 0 && has_best;
    int32_t* vb = c.keyh ? (int32_t*)c.keyh : (int32_t*)c.key32;   // the ranking keys are dead now: reuse their storage (LDS, or the packed words)
    int32_t* tmp = (int32_t*)c.wl;         // wl + fl = capW x int32
    if (want_cutset) {
        PAR_BEGIN   // terminal layer: value_bot = 0 and MARKED; everything else unmarked
        for (int s = tid; s < capS; s += NT) vb[s] = (s < sh->hiw && bm_test(c.live, s)) ? 0 : VB_UNMARKED;
        PAR_END
        for (int tr = L - 1; tr >= lel; --tr) {
            // (1) undo the squash of layer tr+1: arcs into a deleted node were redirected to the merged node
            GLB_PTR(const uint32_t) eo1 = c.evoff + (size_t)(tr + 1) * 8;
            const int nd = (int)eo1[5];
            const int m = c.lmerge[tr + 1];
            if (nd > 0 && m >= 0) {
                const uint64_t doff = (uint64_t)eo1[3] | ((uint64_t)eo1[4] << 32);
                const int dfrom = c.ldup[2 * (tr + 1)], dto = c.ldup[2 * (tr + 1) + 1];
                PAR_BEGIN
                const int32_t vm = vb[m];
                for (int i = tid; i < nd; i += NT) {
                    const uint32_t d = c.ev[doff + i];
                    if (d != NONE32 && (int)d != m) vb[d] = vm;
                }
                PAR_END
                if (dfrom >= 0) {
                    PAR_BEGIN
                    if (tid == 0) {
                        int32_t a = vb[dfrom], b = vb[dto];
                        vb[dfrom] = a > b ? a : b;   // X keeps its own arcs and they were also redirected (clean.rs:851-872)
                    }
                    PAR_END
                }
            }
            // (2) parents of the transition tr -> tr+1
            GLB_PTR(const uint32_t) eo = c.evoff + (size_t)tr * 8;
            const int na = (int)eo[2];
            const uint64_t aoff = (uint64_t)eo[0] | ((uint64_t)eo[1] << 32);
            const int32_t wv = c.weight[c.lvar[tr]];
            for (int base = 0; base < na; base += c.capW) {
                const int cnt_here = na - base < c.capW ? na - base : c.capW;
                PAR_BEGIN
                for (int i = tid; i < cnt_here; i += NT) {
                    GLB_PTR(const uint32_t) rec = c.ev + aoff + 4ull * (uint64_t)(base + i);
                    int32_t best = VB_UNMARKED;
                    if (rec[1] != NONE32) {
                        int32_t v = vb[rec[1] & EV_SLOT_MASK];
                        if (v != VB_UNMARKED) best = v;
                    }
                    if (rec[2] != NONE32) {
                        int32_t v = vb[rec[2] & EV_SLOT_MASK];
                        if (v != VB_UNMARKED && v + wv > best) best = v + wv;
                    }
                    tmp[i] = best;
                }
                PAR_END
                PAR_BEGIN
                for (int i = tid; i < cnt_here; i += NT) {
                    GLB_PTR(const uint32_t) rec = c.ev + aoff + 4ull * (uint64_t)(base + i);
                    vb[rec[0]] = tmp[i];
                }
                PAR_END
            }
        }
    }

    DD2_TICK(PH_BACKWARD)
    // ---------------------------------------------------------------- cut-set (clean.rs:417-445)
    const int ncs_layer = want_cutset ? ncs : 0;
    const bool filter = (in.flags & IN_FILTER_CUTSET) != 0;
    PAR_BEGIN
    if (tid == 0) {
        sh->ncut = 0;
        sh->ncut2 = 0;
    }
    PAR_END
    if (want_cutset) {
        PAR_BEGIN
        int mine = 0;
        for (int i = tid; i < ncs_layer; i += NT) {
            const int32_t vbv = vb[c.cs_slot[i]];
            if (vbv == VB_UNMARKED) continue;
            if (filter) {
                int64_t v = c.cs_value[i];
                int64_t rub = c.cs_pop[i];
                if (!c.unit_weights) {
                    uint64_t s[WS];
                    for (int k = 0; k < WS; ++k) s[k] = c.cs_state[(size_t)k * c.capW + i];
                    rub = rub2_of<WS>(c, s);
                }
                int64_t ub = v + rub;
                if (v + vbv < ub) ub = v + vbv;
                if (best_value < ub) ub = best_value;
                if (ub <= best_lb) continue;
            }
            ++mine;
        }
        if (mine) LDS_ADD_I32(&sh->ncut, mine);
        PAR_END
    }
    const int ncut = sh->ncut;
    const bool want_paths = (in.flags & IN_WANT_PATHS) != 0;
    const bool emit_best = has_best && (want_paths || (int64_t)best_value > best_lb);
    const bool emit_exact = has_best_exact && (want_paths || (int64_t)exact_value > best_lb);
    const bool same = emit_best && emit_exact && exact_slot == best_slot;
    const int path_len = n_layers > 0 ? n_layers - 1 : 0;
    const int best_len = emit_best ? path_len : 0;
    const int exact_len = (emit_exact && !same) ? path_len : 0;
    const int cs_path_len = lel > 0 ? lel : 0;
    const bool pool_out = (in.flags & IN_POOL_OUT) != 0 && c.pool != nullptr;

    uint64_t off = 0;
    const uint64_t path_off = off;
    off += ((uint64_t)best_len * 4 + 7) & ~7ULL;
    const uint64_t exact_off = off;
    off += ((uint64_t)exact_len * 4 + 7) & ~7ULL;
    const uint64_t cs_state_off = off;
    if (!pool_out) off += (uint64_t)ncut * WS * 8;
    const uint64_t cs_value_off = off;
    off += ((uint64_t)ncut * 4 + 7) & ~7ULL;
    const uint64_t cs_ub_off = off;
    off += ((uint64_t)ncut * 4 + 7) & ~7ULL;
    const uint64_t cs_path_off = off;
    const uint32_t pw = (uint32_t)((cs_path_len + 63) / 64);
    // IN_PATH_BITS: a cut-set node's path crosses PCIe as pw words of decision bits instead of cs_path_len u32 words (brock400_1 at
    // width 10 000: 8 bytes instead of 160 per node -- 0.7 instead of 2.2 MB per relaxed DD, written by one store per word)
    const bool bits_out = (in.flags & IN_PATH_BITS) != 0 && !pool_out;
    if (!pool_out) off += bits_out ? (uint64_t)ncut * pw * 8 : (((uint64_t)ncut * cs_path_len * 4 + 7) & ~7ULL);
    const uint64_t cs_lvar_off = off;
    if (bits_out) off += ((uint64_t)cs_path_len * 4 + 7) & ~7ULL;
    const uint64_t total = off;
    const uint64_t pool_bytes = (pool_out && ncut) ? pool_block_bytes((uint32_t)ncut, (uint32_t)WS, (uint32_t)cs_path_len) : 0;

    PAR_BEGIN
    if (tid == 0) {
        unsigned long long a = total ? GLB_ADD_U64(c.arena_head, (unsigned long long)total) : 0ULL;
        sh->arena_off = a;
        if (a + total > c.arena_cap) sh->status = ST_ERR_ARENA;
        sh->ev_pos = 0;   // reused: pool offset of this DD's cut-set block
        if (pool_bytes) {
            unsigned long long pa = GLB_ADD_U64(c.pool_head, (unsigned long long)pool_bytes);
            sh->ev_pos = pa;
            if (pa + pool_bytes > c.pool_cap) sh->status = ST_ERR_CAPACITY - 100 * 8;
        }
    }
    PAR_END
    const uint64_t pool_off = sh->ev_pos;
    GLB_PTR(uint8_t) pblock = pool_bytes ? c.pool + pool_off : (GLB_PTR(uint8_t))nullptr;
    // block-relative offsets (PoolBlockHeader)
    const uint64_t b_lvar = 64;
    const uint64_t b_states = b_lvar + (((uint64_t)cs_path_len * 4 + 7) & ~7ULL);
    const uint64_t b_paths = b_states + (uint64_t)WS * (uint64_t)ncut * 8;
    const uint64_t b_values = b_paths + (uint64_t)pw * (uint64_t)ncut * 8;
    const uint64_t b_ubs = b_values + (((uint64_t)ncut * 4 + 7) & ~7ULL);
    const bool arena_ok = sh->status == ST_OK || sh->status == ST_CUTOFF;
    GLB_PTR(uint8_t) base = c.arena + sh->arena_off;

    if (arena_ok && !failed) {
        // the two best paths as bit strings (LDS scratch: the merged-state words and the candidate list are free now)
        LDS_PTR(uint64_t) bbits = (LDS_PTR(uint64_t))sh->merged;
        LDS_PTR(uint64_t) xbits = (LDS_PTR(uint64_t))sh->xcand;   // 64 x int32 = 32 words >= WS
        PAR_BEGIN
        if (tid == 0 && best_len) {
            uint64_t b[WS];
#pragma unroll
            for (int k = 0; k < WS; ++k) b[k] = 0;
            path_bits<WS>(c, pid_ld<WS>(c, best_slot), b);
#pragma unroll
            for (int k = 0; k < WS; ++k) bbits[k] = b[k];
        }
        if (tid == 64 % NT && exact_len) {
            uint64_t b[WS];
#pragma unroll
            for (int k = 0; k < WS; ++k) b[k] = 0;
            path_bits<WS>(c, pid_ld<WS>(c, exact_slot), b);
#pragma unroll
            for (int k = 0; k < WS; ++k) xbits[k] = b[k];
        }
        PAR_END
        PAR_BEGIN
        // best paths (clean.rs:329-343): one decision per transition, terminal first
        if (best_len) {
            uint32_t* out = (uint32_t*)(base + path_off);
            for (int i = tid; i < best_len; i += NT) {
                const int tr = path_len - 1 - i;
                out[i] = ((uint32_t)c.lvar[tr] << 1) | (uint32_t)((bbits[tr >> 6] >> (tr & 63)) & 1ULL);
            }
        }
        if (exact_len) {
            uint32_t* out = (uint32_t*)(base + exact_off);
            for (int i = tid; i < exact_len; i += NT) {
                const int tr = path_len - 1 - i;
                out[i] = ((uint32_t)c.lvar[tr] << 1) | (uint32_t)((xbits[tr >> 6] >> (tr & 63)) & 1ULL);
            }
        }
        if (want_cutset && ncut && pblock) {
            // ---- cut-set kept in the device node pool: the host only receives (value, ub) per node
            if (tid == 0) {
                PoolBlockHeader* h = (PoolBlockHeader*)pblock;
                h->rows = (uint32_t)ncut;
                h->ws = (uint32_t)WS;
                h->lel = (uint32_t)cs_path_len;
                h->depth = (uint32_t)(in.depth + cs_path_len);
                h->parent_off = in.src_off;
                h->parent_row = in.src_row;
                h->pw = pw;
                h->off_lvar = b_lvar;
                h->off_states = b_states;
                h->off_paths = b_paths;
                h->off_values = b_values;
                h->off_ubs = b_ubs;
            }
            uint32_t* p_lvar = (uint32_t*)(pblock + b_lvar);
            for (int j = tid; j < cs_path_len; j += NT) p_lvar[j] = (uint32_t)c.lvar[j];
            uint64_t* p_state = (uint64_t*)(pblock + b_states);
            uint64_t* p_path = (uint64_t*)(pblock + b_paths);
            int32_t* p_value = (int32_t*)(pblock + b_values);
            int32_t* p_ub = (int32_t*)(pblock + b_ubs);
            int32_t* o_value = (int32_t*)(base + cs_value_off);
            int32_t* o_ub = (int32_t*)(base + cs_ub_off);
            for (int i = tid; i < ncs_layer; i += NT) {
                const int32_t vbv = vb[c.cs_slot[i]];
                if (vbv == VB_UNMARKED) continue;
                uint64_t s[WS];
#pragma unroll
                for (int k = 0; k < WS; ++k) s[k] = c.cs_state[(size_t)k * c.capW + i];
                const int64_t v = c.cs_value[i];
                int64_t ub = v + (c.unit_weights ? (int64_t)c.cs_pop[i] : (int64_t)rub2_of<WS>(c, s));
                if (v + vbv < ub) ub = v + vbv;
                if (best_value < ub) ub = best_value;
                if (filter && ub <= best_lb) continue;
                const int idx = LDS_ADD_I32(&sh->ncut2, 1);
#pragma unroll
                for (int k = 0; k < WS; ++k) p_state[(size_t)k * ncut + idx] = s[k];
                {   // the path of a node that stays in the cut-set is built now, from its chain in the path tree
                    uint64_t pb[WS];
#pragma unroll
                    for (int k = 0; k < WS; ++k) pb[k] = 0;
                    path_bits<WS>(c, c.cs_pid[i], pb);
#pragma unroll
                    for (int k = 0; k < WS; ++k)
                        if ((uint32_t)k < pw) p_path[(size_t)k * ncut + idx] = pb[k];
                }
                p_value[idx] = (int32_t)v;
                p_ub[idx] = (int32_t)ub;
                o_value[idx] = (int32_t)v;
                o_ub[idx] = (int32_t)ub;
            }
        } else if (want_cutset && ncut) {
            if (bits_out) {
                uint32_t* o_lvar = (uint32_t*)(base + cs_lvar_off);
                for (int j = tid; j < cs_path_len; j += NT) o_lvar[j] = (uint32_t)c.lvar[j];
            }
            uint64_t* o_state = (uint64_t*)(base + cs_state_off);
            int32_t* o_value = (int32_t*)(base + cs_value_off);
            int32_t* o_ub = (int32_t*)(base + cs_ub_off);
            uint32_t* o_path = (uint32_t*)(base + cs_path_off);
            for (int i = tid; i < ncs_layer; i += NT) {
                const int32_t vbv = vb[c.cs_slot[i]];
                if (vbv == VB_UNMARKED) continue;
                uint64_t s[WS];
#pragma unroll
                for (int k = 0; k < WS; ++k) s[k] = c.cs_state[(size_t)k * c.capW + i];
                const int64_t v = c.cs_value[i];
                int64_t ub = v + (c.unit_weights ? (int64_t)c.cs_pop[i] : (int64_t)rub2_of<WS>(c, s));
                if (v + vbv < ub) ub = v + vbv;
                if (best_value < ub) ub = best_value;
                if (filter && ub <= best_lb) continue;
                const int idx = LDS_ADD_I32(&sh->ncut2, 1);
#pragma unroll
                for (int k = 0; k < WS; ++k) o_state[(size_t)idx * WS + k] = s[k];
                o_value[idx] = (int32_t)v;
                o_ub[idx] = (int32_t)ub;
                uint64_t pb[WS];
#pragma unroll
                for (int k = 0; k < WS; ++k) pb[k] = 0;
                path_bits<WS>(c, c.cs_pid[i], pb);
                if (bits_out) {
                    uint64_t* o_bits = (uint64_t*)(base + cs_path_off);
#pragma unroll
                    for (int k = 0; k < WS; ++k)
                        if ((uint32_t)k < pw) o_bits[(size_t)idx * pw + k] = pb[k];
                    continue;
                }
                for (int j = 0; j < cs_path_len; ++j) {
                    const int tr = cs_path_len - 1 - j;
                    uint64_t w = 0;
#pragma unroll
                    for (int k = 0; k < WS; ++k)
                        if (k == (tr >> 6)) w = pb[k];
                    o_path[(size_t)idx * cs_path_len + j] = ((uint32_t)c.lvar[tr] << 1) | (uint32_t)((w >> (tr & 63)) & 1ULL);
                }
            }
        }
        PAR_END
    }

    PAR_BEGIN
    if (tid == 0) {
        DDResult r;
        r.status = sh->status;
        // a capacity tier that ran out of node slots / work-list / event space hands the DD to the next tier (the
        // shared output arena and node pool are not the tier's: those stay errors)
        if (c.tier && (sh->status == ST_ERR_CAPACITY || (sh->status <= ST_ERR_CAPACITY - 100 && sh->status != ST_ERR_ARENA &&
                                                         sh->status != ST_ERR_CAPACITY - 800)))
            r.status = ST_RETRY;
        r.comp_type = comp_type;
        r.is_exact = is_exact ? 1 : 0;
        r.has_exact_best_path = ebpo ? 1 : 0;
        r.has_best = has_best ? 1 : 0;
        r.has_best_exact = has_best_exact ? 1 : 0;
        r.best_value = best_value;
        r.best_exact_value = exact_value;
        r.n_layers = n_layers;
        r.lel = lel;
        r.n_cutset = arena_ok ? ncut : 0;
        r.best_len = arena_ok ? best_len : 0;
        r.exact_len = arena_ok ? (same ? best_len : exact_len) : 0;
        r.exact_same_as_best = same ? 1 : 0;
        r.recycled_merges = sh->recycled_merges;
        r.max_width_seen = (uint32_t)sh->maxn;
        r.arena_off = sh->arena_off;
        r.arena_bytes = total;
        r.nodes_expanded = sh->nodes;
        r.arcs = sh->arcs;
        r.layers = (uint64_t)L;
        r.path_off = path_off;
        r.exact_off = same ? path_off : exact_off;
        r.cs_state_off = cs_state_off;
        r.cs_value_off = cs_value_off;
        r.cs_ub_off = cs_ub_off;
        r.cs_path_off = cs_path_off;
        sh->clk[PH_FINAL] += dd_clock() - sh->clk_last;
        for (int k = 0; k < 8; ++k) r.phase_clk[k] = sh->clk[k];
        for (int k = 0; k < 24; ++k) r.phase_clk[8 + k] = sh->mk[k];
        r.pool_off = pool_bytes ? pool_off : NO_POOL_SRC;
        r.cs_depth_off = 0;
        r.cs_path_stride = cs_path_len;
        r.cs_lvar_off = (bits_out && arena_ok && ncut) ? cs_lvar_off : 0;
        r.cache_hits = 0;
        *res = r;
    }
    PAR_END
}

/// restricted, then (when inexact) relaxed: the device half of process_one_node (parallel.rs:391-437).
/// ONE call site of run_dd2 (a loop over the two compilations): the function is inlined once, not three times -- the kernel's
/// code is a third of what it was (round 3: 48 000 lines of ISA, several times the instruction cache two CUs share).
template <int WS, int DEEP = 0, int POOLED = 0>
DDO_DEV void run_work_item2(DD2Ctx<WS>& c, const DDInput& in, DDResult* res2, int only = -1, int32_t* done = nullptr) {
    // `only` = -1: both compilations, one after the other (the work item of a launch that is not split).  Split launches (Engine::launch,
    // EngineParams::done) draw the two compilations of a sub-problem as TWO work items: only = 0 compiles the restricted decision
    // diagram (or the single compile of an item that is not fused) and raises *done; only = 1 waits for *done -- every first half
    // is drawn before any second half, so the flag's writer is running or finished: no deadlock --, reads what the first half found
    // (maybe_update_best) and compiles the relaxed one.
    DD_TID_SETUP(c)
    (void)NT;
    const bool fused = (in.flags & IN_FUSED) != 0;
    int64_t lb = in.best_lb;
    bool go = true;
    if (POOLED) {
        PAR_BEGIN
        if (tid == 0) c.depth0 = in.depth;
        PAR_END
        if ((in.flags & IN_MUST_EXPLORE) && (in.flags & IN_CACHE) && c.cache_cap && c.pvr != nullptr && in.src_off == NO_POOL_SRC) {
            // the solver's pop (parallel.rs:537-549, sequential.rs:341): Cache::must_explore (cache.rs:32-39), then update_threshold(.., explored)
            PAR_BEGIN
            if (tid == 0) {
                uint64_t rs[WS];
#pragma unroll
                for (int k = 0; k < WS; ++k) rs[k] = in.state[k];
                int64_t packed = 0;
                bool explore = true;
                if (cache_get<WS>(c, rs, in.depth, &packed)) {
                    const int32_t tv = th_value(packed);
                    explore = tv != TH_INF && (in.value > tv || (in.value == tv && !th_explored(packed)));
                }
                if (explore && (in.flags & IN_MARK_EXPLORED)) cache_update<WS>(c, rs, in.depth, th_pack(in.value, true));
                c.sh->sel_above = explore ? 1 : 0;
            }
            PAR_END
            const bool explore = c.sh->sel_above != 0;
            DD_SYNC();
            if (!explore) {
                PAR_BEGIN
                if (tid == 0) {
                    res2[0].status = ST_SKIPPED;
                    res2[1].status = ST_NOT_RUN;
                }
                PAR_END
                return;
            }
        }
    }
    if (only == 1) {
        if (!fused) return;
        PAR_BEGIN
        if (tid == 0) {
#if !defined(DDO_HOST_EMULATION)
            while (__hip_atomic_load(done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(64);
#endif
            const int32_t st = LD_I32(&res2[0].status), ex = LD_I32(&res2[0].is_exact);
            c.sh->sel_above = (st == ST_OK && !ex) ? 1 : 0;
            c.sh->sel_bucket = LD_I32(&res2[0].has_best_exact) ? 1 : 0;
            c.sh->sel_digit = LD_I32(&res2[0].best_exact_value);
        }
        PAR_END
        go = c.sh->sel_above != 0;
        if (c.sh->sel_bucket && (int64_t)c.sh->sel_digit > lb) lb = c.sh->sel_digit;  // maybe_update_best
        DD_SYNC();
        if (!go) return;   // (the first half wrote ST_NOT_RUN into the second record)
    }
    for (int pass = only == 1 ? 1 : 0; pass < 2; ++pass) {
        if (pass == 1 && (only == 0 || !(fused && go))) {
            PAR_BEGIN
            if (tid == 0 && !(only == 0 && fused && go)) res2[1].status = ST_NOT_RUN;   // (a split item's second half writes its own record)
            PAR_END
            break;
        }
        run_dd2<WS, DEEP, POOLED>(c, in, fused ? (pass == 0 ? CT_RESTRICTED : CT_RELAXED) : in.comp_type, lb, &res2[pass]);
        if (pass == 0 && fused) {
            PAR_BEGIN
            if (tid == 0) {
                c.sh->sel_above = (res2[0].status == ST_OK && !res2[0].is_exact) ? 1 : 0;
                c.sh->sel_bucket = res2[0].has_best_exact ? 1 : 0;
                c.sh->sel_digit = res2[0].best_exact_value;
            }
            PAR_END
            go = c.sh->sel_above != 0;
            if (c.sh->sel_bucket && (int64_t)c.sh->sel_digit > lb) lb = c.sh->sel_digit;  // maybe_update_best
            DD_SYNC();
        }
    }
    if (only == 0) {
        PAR_BEGIN
        if (tid == 0) {
#if !defined(DDO_HOST_EMULATION)
            __threadfence();   // the records of this half are visible before the flag is
            __hip_atomic_store(done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
        }
        PAR_END
    }
}

/// The body of every kernel of this engine: workgroups draw work items from the launch's counter until it runs dry.
template <int WS, int DEEP, int POOLED>
DDO_DEV void dd2_work_loop(DD2Ctx<WS>& c, const EngineParams& P) {
#if !defined(DDO_HOST_EMULATION)
    const int total = P.done ? 2 * P.nbatch : P.nbatch;
    for (;;) {
        if (threadIdx.x == 0) c.sh->work = atomicAdd(P.work_counter, 1);
        __syncthreads();
        const int drawn = c.sh->work;
        __syncthreads();
        if (drawn >= total) break;
        const int half = P.done ? (drawn >= P.nbatch ? 1 : 0) : -1;
        const int k = half == 1 ? drawn - P.nbatch : drawn;
        const int w = P.order ? (int)P.order[k] : k;   // (longest first: Engine::launch)
        run_work_item2<WS, DEEP, POOLED>(c, P.inputs[w], P.results + 2 * (size_t)w, half, P.done ? P.done + w : (int32_t*)nullptr);
    }
#endif
}

/// LDS bytes of one in-place workgroup
/// entries of the rank-counting scratch (lex_split2: partial counts of a tie list of m <= lex_cap nodes with 2 m <= threads)
DD_HD inline int dd2_tcount_len(int nthreads, int lex_cap) {
    const int m = nthreads / 2 < lex_cap ? nthreads / 2 : lex_cap;
    return m < 64 ? 64 : m;
}

/// LDS bytes of the area the select histogram and the tie-break keys (8 bytes each) share
DD_HD inline size_t dd2_hist_bytes(int hist_bins, int lex_cap) {
    const size_t h = (size_t)(hist_bins > 0 ? hist_bins : 2048) * 4, l = (size_t)(lex_cap > 0 ? lex_cap : 0) * 8;
    return ((h > l ? h : l) + 15) & ~(size_t)15;
}

/// The context of a workgroup (DD2Ctx) lives at the start of its LDS block: every phase reads the few fields it needs right
/// there instead of the whole kernel carrying sixty pointers and sizes in registers from the first layer to the result record
/// (round 3: 397 SGPR and 183 VGPR spills at the 128-VGPR budget of the dense kernel).
constexpr size_t DD2_CTX_BYTES = 512;

inline size_t dd2_lds_bytes(int capS, int tab_cap, int npad, int nthreads, bool keys_in_lds = true, int hist_bins = 2048, int lex_cap = 1024, int ws = MAX_WS) {
    const size_t nbw = ((size_t)capS + 31) / 32;
    size_t b = DD2_CTX_BYTES;
    b += keys_in_lds ? (size_t)capS * 4 : 0;   // key32 / value_bot
    b = (b + 15) & ~(size_t)15;
    b += (size_t)tab_cap * 4;              // dedup table
    b += 4 * nbw * 4;                      // live, inex, okb, fresh
    b = (b + 15) & ~(size_t)15;
    b += (size_t)npad * 4;                 // cnt
    b += dd2_hist_bytes(hist_bins, lex_cap);               // hist / tie-break keys (squash phases only)
    b += (size_t)dd2_tcount_len(nthreads, lex_cap) * 4 * 2;   // rank-counting scratch
    b += dd2_shared_bytes(ws);
    return (b + 15) & ~(size_t)15;
}

template <int WS>
DDO_DEV void dd2_bind(DD2Ctx<WS>& c, const EngineParams& P, int slot, unsigned char* lds, int nthreads) {
    c.n = P.n;
    c.npad = P.npad;
    c.unit_weights = P.unit_weights;
    c.adj = (GLB_PTR(const uint64_t))(P.adj);
    c.weight = (GLB_PTR(const int32_t))(P.weight);
    c.capS = P.capS;
    c.capW = P.capW;
    c.max_layers = P.max_layers;
    c.nbw = (P.capS + 31) / 32;
    const size_t capS = (size_t)P.capS, capW = (size_t)P.capW, ml = (size_t)P.max_layers, s = (size_t)slot;
    c.RW = ((WS + 1 + 7) / 8) * 8;
#if defined(DDO_WORD_MAJOR)
    c.st = (GLB_PTR(uint64_t))(P.s_state + s * (size_t)WS * capS);
#endif
    c.rec = (GLB_PTR(uint64_t))(P.s_rec + s * capS * (size_t)c.RW);
    c.pt = (GLB_PTR(uint64_t))(P.s_ptree + s * (P.ev_cap / 4));
    c.tab_cap = P.tab2_cap;
    c.ev = (GLB_PTR(uint32_t))(P.s_ev + s * P.ev_cap);
    c.ev_cap = P.ev_cap;
    c.evoff = (GLB_PTR(uint32_t))(P.s_evoff + s * ml * 8);
    c.lvar = (GLB_PTR(int32_t))(P.lvar + s * ml);
    c.ldup = (GLB_PTR(int32_t))(P.ldup + s * ml * 2);
    c.lmerge = (GLB_PTR(int32_t))(P.nlayer + s * ml);          // the per-layer node counts of engine 1 are not needed here
    c.cs_slot = (GLB_PTR(uint32_t))(P.s_cs_slot + s * capW);
    c.cs_state = (GLB_PTR(uint64_t))(P.cs_state + s * (size_t)WS * (size_t)P.capN);   // capN >= capW words per row are reserved
    c.cs_pid = (GLB_PTR(uint32_t))((uint32_t*)(P.s_cs_path + s * (size_t)WS * capW));
    c.cs_value = (GLB_PTR(int32_t))(P.cs_value + s * (size_t)P.capN);
    c.cs_pop = (GLB_PTR(uint32_t))(P.cs_pop + s * (size_t)P.capN);
    unsigned char* p = lds;
    if (P.keys_global) {
    c.keyh = (GLB_PTR(uint64_t))(P.s_hash + s * capS);      // keys in HBM, packed with the hashes: the LDS footprint halves, two workgroups share a CU
        c.key32 = nullptr;
        c.h32 = nullptr;
    } else {
        c.keyh = nullptr;
    c.h32 = (GLB_PTR(uint32_t))((uint32_t*)(P.s_hash + s * capS));
        c.key32 = (LDS_PTR(uint32_t))p;
        p += ((size_t)P.capS * 4 + 15) & ~(size_t)15;
    }
    c.tab = (LDS_PTR(uint32_t))p;
    p += (size_t)P.tab2_cap * 4;
    c.live = (LDS_PTR(uint32_t))p;
    p += (size_t)c.nbw * 4;
    c.inex = (LDS_PTR(uint32_t))p;
    p += (size_t)c.nbw * 4;
    c.okb = (LDS_PTR(uint32_t))p;
    p += (size_t)c.nbw * 4;
    c.fresh = (LDS_PTR(uint32_t))p;
    p += (size_t)c.nbw * 4;
    p = (unsigned char*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    c.cnt = (LDS_PTR(int32_t))p;
    p += (size_t)P.npad * 4;
    c.hist = (LDS_PTR(uint32_t))p;
    c.lex_cap = P.lex_cap > 0 && P.lex_cap <= 1024 ? P.lex_cap : 1024;
    p += dd2_hist_bytes(P.hist_bins, c.lex_cap);
    c.wl = P.s_wl + s * 2 * capW;          // work lists live in HBM (written and read once per layer, coalesced)
    c.fl = c.wl + capW;
    c.tcount = (LDS_PTR(int32_t))p;
    p += (size_t)dd2_tcount_len(nthreads, c.lex_cap) * 4;
    c.tcount2 = (LDS_PTR(int32_t))p;
    p += (size_t)dd2_tcount_len(nthreads, c.lex_cap) * 4;
    c.sh = (LDS_PTR(DD2Shared))p;
    if (P.lists_in_lds) {   // narrow capacity tiers: the work / free lists (2 x capW x u16) behind the shared block -- one global round trip
        c.wl = (uint16_t*)(p + dd2_shared_bytes(WS));   // less between the sweep that writes them and the expand that reads them
        c.fl = c.wl + capW;
    }
    c.arena = (GLB_PTR(uint8_t))(P.arena);
    c.arena_cap = P.arena_cap;
    c.arena_head = (GLB_PTR(unsigned long long))(P.arena_head);
    c.cutoff_flag = (GLB_PTR(const int32_t))(P.cutoff_flag);
    c.pool = (GLB_PTR(uint8_t))(P.pool);
    c.pool_cap = P.pool_cap;
    c.pool_head = (GLB_PTR(unsigned long long))(P.pool_head);
    c.vbase_off = P.vbase_off;
    c.clocks = P.phase_clocks;
    c.hist_bins = P.hist_bins > 0 ? P.hist_bins : 2048;
    c.tab_limit = P.tab2_cap >= 3 * P.capW ? 0x7FFFFFFF : (int)((long)P.tab2_cap * 7 / 8);
    c.tier = P.tier;
    c.NT = nthreads;
    c.cache_tab = P.cache_tab;
    c.cache_cap = P.s_pvr ? P.cache_cap : 0;   // (only engines that keep the per-record values can serve a cache)
    c.cache_stats = P.cache_stats;
    c.cache_stride = P.cache_stride;
    c.pvr = P.s_pvr ? (GLB_PTR(uint64_t))(P.s_pvr + s * (P.ev_cap / 4)) : (GLB_PTR(uint64_t))nullptr;
    c.pst = P.s_pst ? (GLB_PTR(uint64_t))(P.s_pst + s * (size_t)P.pst_cap * (size_t)WS) : (GLB_PTR(uint64_t))nullptr;
    c.pst_cap = P.s_pst ? P.pst_cap : 0u;
    c.depth0 = 0;
}

}  // namespace ddo_hip
