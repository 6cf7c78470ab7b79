// =============================================================================
// misp_dd_core.hpp -- one workgroup compiles one decision diagram (MISP).
//
// Restates, for fixed-width bitset states, the layer loop of
// /root/reference/ddo/src/implementation/mdd/clean.rs:345-876 with the MISP model
// callbacks of /root/reference/ddo/examples/misp/main.rs:62-209 inlined:
//   next_variable  (main.rs:109-143)  -> per-vertex occurrence counters kept in LDS,
//                                        maintained by +/- deltas, argmin each layer
//   transition / transition_cost / for_each_in_domain (main.rs:77-102) -> expand phase
//   next_l dedup (clean.rs:738-775)   -> open-addressing table (LDS), tag + index entries,
//                                        full state compare on tag hit, 64-bit atomicMax of
//                                        (value, arc) == `value >= value_top` best-edge rule
//   _restrict / _relax (clean.rs:802-876) + MispRanking (main.rs:205-208)
//                                     -> exact top-K radix select on (value, popcount,
//                                        lexicographic member order), compaction, OR-merge
//   _finalize_* / _compute_local_bounds / _drain_cutset (clean.rs:407-475, 547-655)
//                                     -> backward pass over the stored arcs, LDS atomicMax
//
// Layout: candidate states are SoA, word-major (`cstate[w][c]`), so a wavefront reading
// word w of 64 consecutive candidates issues one contiguous 512-byte request.
//
// The file is written in "phase" style -- PAR_BEGIN ... PAR_END blocks separated by
// workgroup barriers, no thread-private state carried across a barrier -- so that
// the very same source also builds as a lock-step host emulation
// (-DDDO_HOST_EMULATION, used ONLY by tests/ to exercise the logic without a GPU).
// The product library never contains the emulation.
// =============================================================================
#pragma once
#include "dd_types.h"

#if defined(DDO_HOST_EMULATION)
// ---------------------------------------------------------------- host emulation (tests only)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define DDO_DEV inline
// A PAR block is a loop over the thread ids.  DDO_EMU_ORDER picks the order the "threads" of a block run in: 0 (default) ascending,
// 1 descending, 2 a fixed pseudo-random permutation.  A block whose threads only meet through atomics gives the same result in any
// order; one in which a thread reads what a lower-numbered thread wrote a moment ago -- a data race on the GPU that the ascending
// emulation cannot see -- does not (tools/diag/emu_order.sh runs the emulation suites in all three orders).
#define PAR_BEGIN for (int tid_ = 0; tid_ < NT; ++tid_) { const int tid = ::ddo_hip_emu_tid(tid_, NT);
#define PAR_END }
inline int ddo_hip_emu_order() {
    static const int mode = [] { const char* e = std::getenv("DDO_EMU_ORDER"); return e ? std::atoi(e) : 0; }();
    return mode;
}
inline int ddo_hip_emu_tid(int k, int nt) {
    const int mode = ddo_hip_emu_order();
    if (mode == 1) return nt - 1 - k;
    if (mode == 2) return (int)(((long long)k * 769 + 211) % nt);   // 769 is prime: a bijection for every workgroup size but its multiples
    return k;
}
#define DD_SYNC()
#define DDO_TID_DECL
namespace ddo_hip {
template <class T> inline T emu_atomic_add(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T emu_atomic_max(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T emu_atomic_min(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T emu_atomic_or(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> inline T emu_atomic_and(T* p, T v) { T o = *p; *p = o & v; return o; }
inline uint32_t emu_atomic_cas(uint32_t* p, uint32_t cmp, uint32_t v) { uint32_t o = *p; if (o == cmp) *p = v; return o; }
#define LDS_PTR(T) T*
#define LDS_ADD_I32(p, v) emu_atomic_add<int32_t>((p), (v))
#define LDS_ADD_U32(p, v) emu_atomic_add<uint32_t>((p), (v))
#define LDS_MIN_U32(p, v) emu_atomic_min<uint32_t>((p), (v))
#define LDS_MAX_I32(p, v) emu_atomic_max<int32_t>((p), (v))
#define LDS_OR_U64(p, v) emu_atomic_or<uint64_t>((p), (v))
#define LDS_AND_U64(p, v) emu_atomic_and<uint64_t>((p), (v))
#define LDS_MAX_U64(p, v) emu_atomic_max<uint64_t>((p), (v))
#define LDS_ADD_U64(p, v) emu_atomic_add<uint64_t>((p), (v))
#define TAB_CAS(p, c, v) emu_atomic_cas((p), (c), (v))
#define GLB_MAX_U64(p, v) emu_atomic_max<uint64_t>((p), (v))
#define GLB_OR_U32(p, v) emu_atomic_or<uint32_t>((p), (v))
#define GLB_ADD_U64(p, v) emu_atomic_add<unsigned long long>((p), (v))
#define GLB_ADD_I32(p, v) emu_atomic_add<int32_t>((p), (v))
#define LD_U64(p) (*(p))
#define LD_U32(p) (*(p))
#define LD_I32(p) (*(p))
#define FENCE_BLOCK()
#define DD_WAVE_UNIFORM(x) (x)
#define DD_SCALAR_PTR(T) const T*
template <class T> inline const T* dd_scalar_ptr(const T* p) { return p; }
inline int dd_popc(uint64_t x) { return __builtin_popcountll(x); }
inline int dd_ctz(uint64_t x) { return __builtin_ctzll(x); }
inline double dd_floor(double x) { return __builtin_floor(x); }
inline int dd_clz32(uint32_t x) { return __builtin_clz(x); }
inline uint64_t dd_brev(uint64_t x) {
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    return __builtin_bswap64(x);
}
}  // namespace ddo_hip
#else
// ---------------------------------------------------------------- gfx950 device build
#include <hip/hip_runtime.h>
#define DDO_DEV __device__ __forceinline__
#if defined(DDO_PAR_ENTRY_SYNC)   // diagnosis build (tools/diag): a barrier on ENTRY to every block as well -- separates the workgroup-uniform
#define PAR_BEGIN { __syncthreads();   // reads of shared scalars between two blocks from the writes of the second one
#else
#define PAR_BEGIN {
#endif
#define PAR_END } __syncthreads();
// explicit barrier: needed when workgroup-uniform code has read shared scalars that the very next phase rewrites
#define DD_SYNC() __syncthreads()
namespace ddo_hip {
// Pointers into LDS carry the address space: through a generic pointer every access becomes a FLAT instruction,
// which travels the vector-memory path and ties LDS traffic to vmcnt (an LDS atomic would then wait behind
// outstanding global stores).  The __hip_atomic builtins accept any address space: ds_* for LDS pointers, the usual
// flat/global atomics for generic ones (agent scope = what atomicAdd & co. use).
#define LDS_PTR(T) __attribute__((address_space(3))) T*
#define LDS_ADD_I32(p, v) __hip_atomic_fetch_add((p), (int32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LDS_ADD_U32(p, v) __hip_atomic_fetch_add((p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LDS_MIN_U32(p, v) __hip_atomic_fetch_min((p), (uint32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LDS_MAX_I32(p, v) __hip_atomic_fetch_max((p), (int32_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LDS_OR_U64(p, v) __hip_atomic_fetch_or((p), (uint64_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LDS_AND_U64(p, v) __hip_atomic_fetch_and((p), (uint64_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LDS_MAX_U64(p, v) __hip_atomic_fetch_max((p), (uint64_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LDS_ADD_U64(p, v) __hip_atomic_fetch_add((p), (uint64_t)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
template <class P>
__device__ __forceinline__ uint32_t dd_tab_cas(P p, uint32_t cmp, uint32_t v) {
    __hip_atomic_compare_exchange_strong(p, &cmp, v, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return cmp;
}
#define TAB_CAS(p, c, v) dd_tab_cas((p), (c), (v))
#define GLB_MAX_U64(p, v) atomicMax((unsigned long long*)(p), (unsigned long long)(v))
#define GLB_OR_U32(p, v) atomicOr((p), (v))
#define GLB_ADD_U64(p, v) __hip_atomic_fetch_add((p), (unsigned long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)   // (any address space)
#define GLB_ADD_I32(p, v) atomicAdd((p), (v))
// Words that other waves update with L2 atomics (ckey, cflags) or have just stored
// (dedup compare) are read with agent-scope relaxed atomic loads: they bypass the CU's
// vector L1 (MI355X_MICROARCH.md, "sc1 loads bypass L1 only").
#define LD_U64(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LD_U32(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LD_I32(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define FENCE_BLOCK() __threadfence_block()
// a value every lane of the wavefront holds: kept in a scalar register, what is computed from it runs on the scalar unit
#define DD_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// A read-only table addressed the same way by every lane (a model table indexed by the layer's variable and a loop counter): the pointer
// in scalar registers and in the constant address space, so that the look-ups are scalar loads (s_load through the scalar cache) and
// not 64-lane vector loads of one address -- a MAX2SAT child costs five such look-ups per remaining variable.
#define DD_SCALAR_PTR(T) const __attribute__((address_space(4))) T*
template <class T>
__device__ __forceinline__ DD_SCALAR_PTR(T) dd_scalar_ptr(const T* p) {
    const uint64_t a = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return (DD_SCALAR_PTR(T))(((uint64_t)hi << 32) | (uint64_t)lo);
}
__device__ __forceinline__ int dd_popc(uint64_t x) { return __popcll(x); }
__device__ __forceinline__ int dd_ctz(uint64_t x) { return __builtin_ctzll(x); }
__device__ __forceinline__ double dd_floor(double x) { return __builtin_floor(x); }
__device__ __forceinline__ int dd_clz32(uint32_t x) { return __builtin_clz(x); }
__device__ __forceinline__ uint64_t dd_brev(uint64_t x) { return __brevll(x); }
}  // namespace ddo_hip
#endif

#include "dd_thresholds.hpp"
#include "dd_tsptw.hpp"

namespace ddo_hip {

constexpr uint32_t TAB_EMPTY = 0xFFFFFFFFu;
// Bit 31 of the arc half of a candidate key says "the parent of this arc has an exact
// best path".  The 64-bit atomicMax of (value, arc) then resolves ties between equal-valued arcs in favour of such a
// parent, whatever order the threads run in: EBPO (clean.rs:643-655) and the best paths become order independent.  The
// reference leaves these ties to its hash map's iteration order; the oracle applies the same rule (Problem::canonical_ties).
constexpr uint32_t KEY_OK = 0x80000000u;
DDO_DEV uint32_t key_arc(uint64_t key) {   // candidate index of the best arc, NONE32 for the root
    const uint32_t a = (uint32_t)key;
    return a == NONE32 ? NONE32 : (a & ~KEY_OK);
}
constexpr int32_t VB_UNMARKED = INT32_MIN;  // value_bot = isize::MIN  <=> !MARKED (clean.rs:392, 464)

/// Workgroup-shared scalars (one per workgroup, lives in LDS).
#if defined(DDO_HOST_EMULATION)
DDO_DEV uint64_t dd_clock() { return 0; }
#else
DDO_DEV uint64_t dd_clock() { return (uint64_t)__builtin_readcyclecounter(); }
#endif

struct DDShared {
    int32_t work;
    uint32_t varkey;
    int32_t nU;          // unique candidates (== next_l.len())
    int32_t status;
    int32_t cutoff;
    int32_t scan_total;
    int32_t sel_digit, sel_above, sel_bucket;
    int32_t tie_n;               // select_pivot: candidates that carry the pivot's primary key
    int32_t nkept, merged_pos, recycled, dup_from, dup_to;
    int32_t ncut, ncut2;
    int32_t xbest;       // recycled merge: candidate re-added to the layer (clean.rs:868-872)
    uint32_t recycled_merges;
    int32_t maxn;
    uint64_t clk[8], clk_last;   // shader-clock ticks per phase (DDO_HIP_STATS): DD1_TICK
    uint64_t k1and, k1or;
    uint64_t pivK1;
    uint64_t pivLex[MAX_WS];
    uint64_t merged[MAX_WS];
    uint64_t mergedKey;
    uint64_t bestKey, bestExactKey;
    uint64_t nodes, arcs;
    uint64_t arena_off;
    int32_t ncache;      // candidates removed by _filter_with_cache in the layer being built
    uint32_t cache_hits;
    int32_t xcand[64];   // per-lane partial results of the recycled-merge search
    // signed-vector models (MCP): per-variable reductions of a merge (relax.rs:141-176) and the merged node's rank
    uint32_t vmin[MAX_VEC_VARS];
    uint64_t vposmask[(MAX_VEC_VARS + 63) / 64], vnegmask[(MAX_VEC_VARS + 63) / 64];
    int32_t mrank, xdelta;
};

// DD1_TICK(ph): shader-clock ticks since the previous mark are charged to phase ph (DDO_HIP_STATS; workgroup-uniform context)
#define DD1_TICK(ph)                                        \
    if (c.clocks) {                                         \
        PAR_BEGIN                                           \
        if (tid == 0) {                                     \
            const uint64_t _t = dd_clock();                 \
            sh->clk[ph] += _t - sh->clk_last;               \
            sh->clk_last = _t;                              \
        }                                                   \
        PAR_END                                             \
    }


/// Everything one workgroup needs: model, slot-local workspace and LDS carve-up.
/// two buffers: picked by a SELECT (SEL: the context can then live in registers, see DDCtx::cstate) or kept as an array member
template <class T, bool SEL>
struct Buf2 {
    T* a;
    T* b;
    DDO_DEV T* operator[](int i) const { return i ? b : a; }
    DDO_DEV void set(int i, T* p) {
        if (i) b = p;
        else a = p;
    }
};
template <class T>
struct Buf2<T, false> {
    T* v[2];
    DDO_DEV T* operator[](int i) const { return v[i]; }
    DDO_DEV void set(int i, T* p) { v[i] = p; }
};

template <int WS>
struct DDCtx {
    // model
    int n, npad, unit_weights;
    const uint64_t* adj;
    const int32_t* weight;      // MISP: vertex weights; knapsack: item profits (the cost of a TAKE arc)
    // knapsack (examples/knapsack/main.rs:53-72): kind == MODEL_KNAPSACK, the state is one word = remaining capacity
    int kind;
    int depth0;                 // depth of the residual sub-problem (static variable order: main.rs:118-125)
    const int32_t* kp_weight;   // item weights
    const int32_t* kp_order;    // items by decreasing profit/weight
    // maximum cut (examples/mcp/{model,relax}.rs): kind == MODEL_MCP
    const int32_t* vgraph;
    const int32_t* vest;
    const int32_t* vnk;
    int32_t vr;
    int32_t* lddelta;
    // MAX2SAT (examples/max2sat/{model,relax}.rs): kind == MODEL_MAX2SAT
    const int32_t *m2_wtt, *m2_wtf, *m2_wft, *m2_wff, *m2_order, *m2_rankpos;
    // capacity
    int capN, capC1, max_layers;
    TwModel tw;          // TSPTW tables (kind == MODEL_TSPTW)
    uint64_t* dkey_tab;  // TSPTW dominance: (depth, position, must_visit) -> best value
    uint64_t dkey_cap;
    unsigned long long* dkey_stats;
    int fan, dbits;      // children per node (2; TSPTW: nb_nodes) and the bits a decision index takes in arc / path words
    // slot workspace
    // the two candidate buffers (current / next layer).  In the wide instantiations (signed-vector models, WS > 16) they are picked
    // by SELECTS, not by indexing an array member: an array indexed at run time pins the whole context in scratch memory -- every
    // `c.field` a scratch load -- where it could live in scalar registers (72-word kernel: 2 224 -> 1 552 B/lane of scratch, C3 kernels
    // -13 %).  The narrow instantiations keep the array.  History: with select buffers the 16-word kernel of rounds 4 / 5 faulted on the
    // GPU for TSPTW states of four-word node sets (rbg132, n200w20.001), and round 5 blamed the DPP atomic optimizer (the fault moved or
    // vanished with unrelated code changes, no bounds check ever fired).  Round 6 (tools/diag, profiles/r06/diag):
    // it was a read of memory the compile never wrote -- DDO_HIP_ALLOC_FILL=0x55 moves the faulting address to ckey + 8 x 0x55555555 in
    // that tree -- in the one-thread-per-parent TSPTW expansion that round 5 later replaced; an entry barrier on every block does not
    // remove it (no race on shared scalars), and today's tree passes the engine-1 GPU suites under 0x55 / 0xFF fills in BOTH layouts.
    // With the fault gone the two layouts were measured again: within a few percent on every secondary workload, so the split stays.
#if defined(DDO_BUF2_SEL_ALL)   // diagnosis / A-B build: the select-based buffers in every instantiation
#define DDO_BUF2_SEL(ws) true
#else
#define DDO_BUF2_SEL(ws) ((ws) > 16)
#endif
    Buf2<uint64_t, DDO_BUF2_SEL(WS)> cstate;
    Buf2<uint64_t, DDO_BUF2_SEL(WS)> ckey;
    Buf2<uint32_t, DDO_BUF2_SEL(WS)> cpop;
    Buf2<uint32_t, DDO_BUF2_SEL(WS)> cflags;
    uint32_t* ctarget;
    uint32_t* keep;
    uint32_t* posmap;
    uint8_t* cls;
    uint32_t* ninfo;
    uint32_t* arct;
    int32_t* arcc;       // arc costs, same indexing as arct (written when the arc is created)
    int32_t* nlayer;
    int32_t* lvar;
    int32_t* ldup;
    uint64_t* cs_state;
    int32_t* cs_value;
    uint32_t* cs_pop;
    // LDS
    uint32_t* table;     // table_cap u32 (also reused as 2 x capN i32 value_bot arrays)
    int table_cap;
    int32_t* cnt;        // npad
    uint32_t* hist;      // 256
    int32_t* tcount;     // NT
    int32_t* tcount2;    // NT
    DDShared* sh;
    // frontier cut-set / thresholds / cache (dd_thresholds.hpp): every layer is kept
    int tmode, lstride;
    int cdbits;          // bits of a candidate index in a dedup table entry (tag | candidate): 20, more when fan * capN needs them
    int clocks;          // DDO_HIP_STATS: phase clocks on
    uint64_t* lstate;
    int32_t *lval, *lrub, *lvb, *lth;
    int32_t* cth;        // [capC1] theta of the candidates the cache pruned in the layer being built
    int32_t* lntot;      // [max_layers] nodes per layer including the ones the cache pruned
    // (where these sit matters: in the narrow instantiations this struct lives in scratch memory, and the offsets of the fields behind
    // them decide how the hot loops' context reads fall -- appended last they cost MCP n = 30 58 % and config C5 28 % of kernel time)
    uint64_t lpool, apool;       // per-slot pools of node / arc records (0: fixed strides), see run_dd
    uint64_t *lbase, *abase;     // [max_layers + 1] where each kept layer / its entering arcs start in the pools
    uint64_t* cache_tab;
    uint64_t cache_cap;
    int cache_stride;
    unsigned long long* cache_stats;
    uint64_t* dom_coord;
    int32_t* dom_value;
    uint32_t *dom_count, *dom_lock;
    uint32_t dom_cap;
    unsigned long long* dom_stats;
    // output
    uint8_t* arena;
    uint64_t arena_cap;
    unsigned long long* arena_head;
    const int32_t* cutoff_flag;
    int NT;
#if !defined(DDO_HOST_EMULATION)
    int tid_;
#endif
};

#if defined(DDO_HOST_EMULATION)
#define DD_TID_SETUP(c) const int NT = (c).NT;
#else
#define DD_TID_SETUP(c) const int NT = (c).NT; const int tid = (int)threadIdx.x;
#endif

DDO_DEV uint32_t bias32(int32_t v) { return (uint32_t)v ^ 0x80000000u; }
DDO_DEV int32_t unbias32(uint32_t v) { return (int32_t)(v ^ 0x80000000u); }

/// 64-bit mix of the state words (any good mix is equivalent to FxHash for dedup purposes).
template <int WS>
DDO_DEV uint64_t hash_state(const uint64_t* s) {
    uint64_t h = 0x243F6A8885A308D3ULL;
#pragma unroll
    for (int k = 0; k < WS; ++k) {
        h ^= s[k];
        h *= 0x9E3779B97F4A7C15ULL;
        h ^= h >> 29;
    }
    h *= 0xBF58476D1CE4E5B9ULL;
    h ^= h >> 32;
    return h;
}

/// cnt[i] += delta for every member i of the state (delta maintenance of the
/// next_variable counters, main.rs:130-135).
template <int WS, class CP>
DDO_DEV void add_bits(CP cnt, const uint64_t* s, int delta) {
#pragma unroll
    for (int k = 0; k < WS; ++k) {
        uint64_t x = s[k];
        while (x) {
            int b = dd_ctz(x);
            LDS_ADD_I32(&cnt[k * 64 + b], delta);
            x &= x - 1;
        }
    }
}

/// lexicographic ranking word: MISP orders states by member lists (BitSet::cmp) = brev(~word) descending;
/// knapsack ranks by remaining capacity (examples/knapsack/main.rs:187-194)
template <class Ctx>
DDO_DEV uint64_t lexkey(const Ctx& c, uint64_t w) { return c.kind != MODEL_MISP ? w : dd_brev(~w); }

/// Signed-vector states (MCP): benefit v sits in half (v & 1) of word v / 2; the word after the last pair is the depth.
template <int WS>
DDO_DEV int32_t vec_get(const uint64_t* s, int v) {
    int32_t r = 0;
#pragma unroll
    for (int k = 0; k < WS; ++k)
        if (k == (v >> 1)) r = (int32_t)(uint32_t)(s[k] >> (32 * (v & 1)));
    return r;
}
DDO_DEV int32_t iabs32(int32_t x) { return x < 0 ? -x : x; }
constexpr int TIE_PAIRS_MAX = 256;   // select_pivot: classes of ties up to this size are ranked pair by pair (<= the workgroup's threads: the list sits in a scan array)
constexpr int VEC_BATCH = 8;   // words of a signed-vector parent read ahead of the stores of its children (expand)
/// the signed-vector models (MCP, MAX2SAT) share merge, relax, ranking and state layout; transitions and bounds differ
DDO_DEV bool dd_is_vec(int kind) { return kind == MODEL_MCP || kind == MODEL_MAX2SAT; }
/// States wider than 16 words only exist for the signed-vector models (MAX2SAT / MCP beyond 30 variables): in the 32- and
/// 72-word instantiations the model kind is a compile-time fact, and the code of the other models -- which keeps whole states in
/// per-thread arrays -- is not generated at all (round 3: 7 000 spilled registers in the 72-word kernel, most of them for paths it
/// can never take).
template <int WS> DDO_DEV bool dd_is_vec_w(int kind) { return WS > 16 ? true : dd_is_vec(kind); }
template <int WS> DDO_DEV bool dd_kind_is(int kind, int what) { return WS > 16 ? false : kind == what; }
/// sum of |benefit| over the variables >= from (from = 0: McpRanking's key, model.rs:154-163)
template <int WS>
DDO_DEV int32_t vec_rank(const uint64_t* s, int n, int from) {
    int32_t r = 0;
#pragma unroll
    for (int k = 0; k < WS; ++k) {
        if (2 * k >= from && 2 * k < n) r += iabs32((int32_t)(uint32_t)s[k]);
        if (2 * k + 1 >= from && 2 * k + 1 < n) r += iabs32((int32_t)(uint32_t)(s[k] >> 32));
    }
    return r;
}

/// Relaxation::fast_upper_bound: MISP main.rs:191-193; knapsack main.rs:158-184 (fractional bound over the remaining
/// items in ratio order; the only floating point on the path: cap/weight * profit, floored)
template <int WS>
DDO_DEV int32_t rub_of(const DDCtx<WS>& c, const uint64_t* s, int pop, int depth) {
    if constexpr (tw_k_of_ws(WS) != 0)
        if (dd_kind_is<WS>(c.kind, MODEL_TSPTW)) return tw_rub<tw_k_of_ws(WS)>(c.tw, s);
    if (c.kind == MODEL_MCP)   // mcp/relax.rs:123-130
        return vec_rank<WS>(s, c.n, depth) + c.vest[depth] - c.vr + c.vnk[depth];
    if (c.kind == MODEL_MAX2SAT) {   // max2sat/model.rs:231-240; a complete assignment (depth n) has nothing left to gain
        if (depth >= c.n) return 0;
        return pop + c.vest[depth] - c.vr + c.vnk[depth];
    }
    if (dd_kind_is<WS>(c.kind, MODEL_KNAPSACK)) {
        int64_t cap = (int64_t)s[0];
        int64_t max_profit = 0;
        for (int d = depth; cap > 0 && d < c.n; ++d) {
            const int item = c.kp_order[d];
            const int64_t w = c.kp_weight[item];
            if (cap >= w) {
                max_profit += c.weight[item];
                cap -= w;
            } else {
                const double ratio = (double)cap / (double)w;
                const double pr = ratio * (double)c.weight[item];
                max_profit += (int64_t)dd_floor(pr);
                cap = 0;
            }
        }
        return (int32_t)max_profit;
    }
    if constexpr (WS > 16) return 0;   // (only the signed-vector models have states this wide: dd_is_vec_w)
    if (c.unit_weights) return pop;
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

/// a candidate that is in `curr_l`: unique (it won the dedup) and not removed by _filter_with_cache (clean.rs:710-726)
template <class Ctx>
DDO_DEV bool cand_live(const Ctx& c, int cur, int cd) {
    return c.ctarget[cd] == (uint32_t)cd && !(c.tmode && (c.cflags[cur][cd] & (NF_CACHE | NF_DOM)));
}

/// candidate numbering: the child of parent position p under decision index d lives at d * capN + p (binary models:
/// NO-children at p, YES-children at capN + p), the merged node of a relaxed layer at fan * capN.
DDO_DEV int lin2cand(int j, int nprev, int capN) { return (j / nprev) * capN + (j % nprev); }

/// 64-bit primary ranking key: (value_top, secondary rank) -- clean.rs:803-808, then the model's StateRanking where it is a
/// number: MispRanking's len() (popcount), 0 for knapsack, later the 32-bit rank of the signed-vector models.
DDO_DEV uint64_t k1_of(uint64_t key, uint32_t pop) { return (key & 0xFFFFFFFF00000000ULL) | (uint64_t)pop; }

/// Workgroup exclusive scan of a[0..NT) (Hillis-Steele, double buffered); total in *total.
template <class Ctx>
DDO_DEV void block_exclusive_scan_any(Ctx& c, int32_t* a, int32_t* tmp, int32_t* total) {
    DD_TID_SETUP(c)
    int32_t* src = a;
    int32_t* dst = tmp;
    for (int d = 1; d < NT; d <<= 1) {
        PAR_BEGIN
        dst[tid] = src[tid] + (tid >= d ? src[tid - d] : 0);
        PAR_END
        int32_t* t = src;
        src = dst;
        dst = t;
    }
    // src holds the inclusive scan; write the exclusive one into `a`
    PAR_BEGIN
    int32_t incl = src[tid];
    int32_t excl = tid ? src[tid - 1] : 0;
    if (tid == NT - 1) *total = incl;
    dst[tid] = excl;
    PAR_END
    if (dst != a) {
        PAR_BEGIN
        a[tid] = dst[tid];
        PAR_END
    }
}
template <int WS>
DDO_DEV void block_exclusive_scan(DDCtx<WS>& c, int32_t* a, int32_t* tmp) {
    block_exclusive_scan_any(c, a, tmp, &c.sh->scan_total);
}

/// Primary keys of the candidates j0, j0 + NT, ... (SEL_BATCH of them; lanes next to each other read candidates next to each other):
/// every load is issued before the first test, so that a sweep of the selection waits for memory once per batch, not once per
/// candidate (an index past the layer reads candidate 0 and is not live).
constexpr int SEL_BATCH = 4;
template <int WS>
DDO_DEV void sel_keys(const DDCtx<WS>& c, int cur, int j0, int NT, int ncl, int nprev, uint64_t* k1, bool* live, int* cds) {
    uint32_t tg[SEL_BATCH], fl[SEL_BATCH], pp[SEL_BATCH];
    uint64_t kk[SEL_BATCH];
#pragma unroll
    for (int u = 0; u < SEL_BATCH; ++u) {
        const int j = j0 + u * NT;
        cds[u] = j < ncl ? lin2cand(j, nprev, c.capN) : 0;
    }
#pragma unroll
    for (int u = 0; u < SEL_BATCH; ++u) {
        tg[u] = c.ctarget[cds[u]];
        kk[u] = LD_U64(&c.ckey[cur][cds[u]]);
        pp[u] = c.cpop[cur][cds[u]];
        fl[u] = c.tmode ? c.cflags[cur][cds[u]] : 0u;
    }
#pragma unroll
    for (int u = 0; u < SEL_BATCH; ++u) {
        live[u] = j0 + u * NT < ncl && tg[u] == (uint32_t)cds[u] && !(fl[u] & (NF_CACHE | NF_DOM));
        k1[u] = k1_of(kk[u], pp[u]);
    }
}

/// Exact K-th largest (1 <= K < nU) among the unique candidates of buffer `cur` by the total
/// order (value_top, popcount, BitSet::cmp) -- MSD radix select, 8-bit digits, skipping the
/// digits that are constant over the layer.  Result: sh->pivK1 / sh->pivLex with the
/// unresolved low digits zero-filled, so that "key >= pivot" <=> kept.
template <int WS>
DDO_DEV void select_pivot(DDCtx<WS>& c, int cur, int nprev, int K) {
    DD_TID_SETUP(c)
    DDShared* sh = c.sh;
    const int ncl = c.fan * nprev;
    const int q = (ncl + NT - 1) / NT;
    const uint64_t* key = c.ckey[cur];
    const uint32_t* pop = c.cpop[cur];
    const uint64_t* st = c.cstate[cur];
    const int capC1 = c.capC1;

    PAR_BEGIN
    if (tid == 0) {
        sh->k1and = ~0ULL;
        sh->k1or = 0;
        sh->pivK1 = 0;
        for (int k = 0; k < WS; ++k) sh->pivLex[k] = 0;
    }
    PAR_END
    int need = K;
    bool done = false;
    uint64_t pivK1 = 0;
    // ---- NARROW layers (a few hundred candidates: MCP and knapsack at width 100, the reference's test widths): the primary keys
    // are staged in LDS and every candidate counts the keys above its own.  The digit rounds below cost three barriers and a sweep
    // of global memory per byte of the key whatever the layer holds (35-40 kcycles per squash of 200 candidates, a third of an MCP
    // compile); counting is one LDS read per pair.  The LDS block is the histogram and the two scan arrays, idle during a selection.
    const int narrow_cap = 128 + NT;
    const bool narrow = ncl <= narrow_cap;
    if (narrow) {
        LDS_PTR(uint64_t) nk = (LDS_PTR(uint64_t))c.hist;   // (ds_read, not the FLAT path the engine's generic pointers take)
        const uint64_t DEAD = ~0ULL;   // no key looks like this: the low word is a popcount or a rank
        PAR_BEGIN
        for (int j = tid; j < ncl; j += NT) {
            const int cd = lin2cand(j, nprev, c.capN);
            nk[j] = cand_live(c, cur, cd) ? k1_of(LD_U64(&key[cd]), pop[cd]) : DEAD;
        }
        PAR_END
        PAR_BEGIN
        for (int j = tid; j < ncl; j += NT) {
            const uint64_t mine = nk[j];
            if (mine == DEAD) continue;
            int above = 0, equal = 0;
#pragma unroll 8
            for (int j2 = 0; j2 < ncl; ++j2) {   // (every lane reads the same word: an LDS broadcast; unrolled so that eight reads are in flight)
                const uint64_t o = nk[j2];
                above += (o != DEAD && o > mine) ? 1 : 0;
                equal += o == mine ? 1 : 0;
            }
            if (above < K && K <= above + equal) {   // the K-th best key (every candidate that carries it writes the same three words)
                sh->pivK1 = mine;
                sh->sel_above = above;
                sh->sel_bucket = equal;
            }
        }
        PAR_END
        pivK1 = sh->pivK1;
        need = K - sh->sel_above;
        const int bucket = sh->sel_bucket;
        DD_SYNC();
        
#if defined(DDO_HOST_EMULATION)
        if (getenv("DD_SELECT_TRACE")) std::fprintf(stderr, "NARROW ncl=%d K=%d need=%d bucket=%d %s\n", ncl, K, need, bucket, need == bucket ? "whole" : bucket <= 64 ? "pairs" : "radix");
#endif
        if (need == bucket) done = true;   // all the candidates with the pivot's key are kept: the pivot's state words stay zero
        else if (bucket <= 64) {           // a small class of ties: each member counts the members that rank above it
            PAR_BEGIN
            for (int j = tid; j < ncl; j += NT) {
                if (nk[j] != pivK1) continue;
                const int cd = lin2cand(j, nprev, c.capN);
                int r = 0;
                for (int j2 = 0; j2 < ncl; ++j2) {
                    if (j2 == j || nk[j2] != pivK1) continue;
                    const int c2 = lin2cand(j2, nprev, c.capN);
                    bool c2_above = false;
                    for (int k = 0; k < WS; ++k) {
                        const uint64_t la = lexkey(c, st[(size_t)k * capC1 + c2]), lb = lexkey(c, st[(size_t)k * capC1 + cd]);
                        if (la != lb) {
                            c2_above = la > lb;
                            break;
                        }
                    }
                    r += c2_above ? 1 : 0;
                }
                if (r == need - 1)
                    for (int k = 0; k < WS; ++k) sh->pivLex[k] = lexkey(c, st[(size_t)k * capC1 + cd]);
            }
            PAR_END
            done = true;
        }
    } else {
    PAR_BEGIN
    uint64_t a = ~0ULL, o = 0;
    for (int j0 = tid; j0 < ncl; j0 += SEL_BATCH * NT) {
        uint64_t k1[SEL_BATCH];
        bool live[SEL_BATCH];
        int cds[SEL_BATCH];
        sel_keys<WS>(c, cur, j0, NT, ncl, nprev, k1, live, cds);
#pragma unroll
        for (int u = 0; u < SEL_BATCH; ++u)
            if (live[u]) {
                a &= k1[u];
                o |= k1[u];
            }
    }
    if (a != ~0ULL || o != 0) {
        LDS_AND_U64(&sh->k1and, a);
        LDS_OR_U64(&sh->k1or, o);
    }
    PAR_END
    }

    const uint64_t diff = narrow ? 0 : sh->k1and ^ sh->k1or;
    // ---- primary key: 8 bytes, most significant first (bytes that are constant over the layer cost nothing)
    for (int b = 7; b >= 0 && !done && !narrow; --b) {
        const int shift = 8 * b;
        if (((diff >> shift) & 0xFF) == 0) {
            pivK1 |= sh->k1and & (0xFFULL << shift);
            continue;
        }
        PAR_BEGIN
        if (tid < 256) c.hist[tid] = 0;
        PAR_END
        PAR_BEGIN
        for (int j0 = tid; j0 < ncl; j0 += SEL_BATCH * NT) {
            uint64_t k1[SEL_BATCH];
            bool live[SEL_BATCH];
            int cds[SEL_BATCH];
            sel_keys<WS>(c, cur, j0, NT, ncl, nprev, k1, live, cds);
#pragma unroll
            for (int u = 0; u < SEL_BATCH; ++u) {
                const bool active = live[u] && ((shift + 8 >= 64) || ((k1[u] >> (shift + 8)) == (pivK1 >> (shift + 8))));
                if (active) LDS_ADD_U32(&c.hist[(k1[u] >> shift) & 0xFF], 1u);
            }
        }
        PAR_END
        PAR_BEGIN
        if (tid < 256) {
            int above = 0;
            for (int x = tid + 1; x < 256; ++x) above += (int)c.hist[x];
            int mine = (int)c.hist[tid];
            if (above < need && need <= above + mine) {
                sh->sel_digit = tid;
                sh->sel_above = above;
                sh->sel_bucket = mine;
            }
        }
        PAR_END
        pivK1 |= (uint64_t)sh->sel_digit << shift;
        need -= sh->sel_above;
        if (need == sh->sel_bucket) done = true;  // the whole bucket is kept: stop refining
    }
    // ---- tie-break: lexicographic member order; digit = byte of brev(~word), MSB first
    // The pivot words live in LDS (sh->pivLex, zeroed above); only the word in progress is a register: a local array
    // indexed by the running word number would sit in scratch memory (31- and 69-word states).
    // Which candidates are still tied with the pivot is kept as a byte per candidate (c.cls: the classification that follows
    // overwrites every entry): a sweep of a later word then costs one byte per candidate and one state word per TIED candidate.
    // (Until round 3 every sweep re-compared all the words before wj: with 31- to 72-word states and a sweep or two per word,
    // the selection of a MAX2SAT layer was a quarter of its time.)
    if (!done) {
#if defined(DDO_HOST_EMULATION)
        if (getenv("DD_SELECT_TRACE")) std::fprintf(stderr, "TIEBREAK ncl=%d K=%d need=%d bucket=%d\n", ncl, K, need, (int)sh->sel_bucket);
#endif
        PAR_BEGIN
        if (tid == 0) sh->tie_n = 0;
        PAR_END
        PAR_BEGIN
        for (int j0 = tid; j0 < ncl; j0 += SEL_BATCH * NT) {
            uint64_t k1[SEL_BATCH];
            bool live[SEL_BATCH];
            int cds[SEL_BATCH];
            sel_keys<WS>(c, cur, j0, NT, ncl, nprev, k1, live, cds);
#pragma unroll
            for (int u = 0; u < SEL_BATCH; ++u) {
                if (j0 + u * NT >= ncl) continue;
                const bool tied = live[u] && k1[u] == pivK1;
                c.cls[cds[u]] = tied ? 1 : 0;
                if (tied) {   // ... and listed, while the list has room (the two scan arrays, idle during a selection)
                    const int at = LDS_ADD_I32(&sh->tie_n, 1);
                    if (at < TIE_PAIRS_MAX) c.tcount[at] = cds[u];
                }
            }
        }
        PAR_END
        // A class of at most TIE_PAIRS_MAX ties is ranked PAIR BY PAIR by the whole workgroup: b (b - 1) / 2 comparisons of state words
        // that end at the first word that differs.  The digit rounds below narrow the class by one byte of one word per round -- three
        // barriers and a sweep over every candidate of the layer each; the signed-vector states of MAX2SAT are small integers, most of
        // their bytes 0x00 or 0xFF: config C3 took 15 rounds over 6 words per selection (classes of 130 ties on average), half a
        // million cycles per squashed layer.
        const int b = sh->tie_n;
        if (b <= TIE_PAIRS_MAX) {
            int32_t* tl = c.tcount;    // the tied candidates
            int32_t* tr = c.tcount2;   // how many tied candidates rank above each
            PAR_BEGIN
            for (int i = tid; i < b; i += NT) tr[i] = 0;
            PAR_END
            PAR_BEGIN
            for (int pi = tid; pi < b * b; pi += NT) {
                const int ia = pi / b, ib = pi % b;
                if (ia >= ib) continue;
                const int ca = tl[ia], cb = tl[ib];
                bool a_above = false;
                for (int k = 0; k < WS; ++k) {
                    const uint64_t la = lexkey(c, st[(size_t)k * capC1 + ca]), lb = lexkey(c, st[(size_t)k * capC1 + cb]);
                    if (la != lb) {
                        a_above = la > lb;
                        break;
                    }
                }
                LDS_ADD_I32(&tr[a_above ? ib : ia], 1);   // (two candidates of a layer never hold the same state)
            }
            PAR_END
            PAR_BEGIN
            for (int i = tid; i < b; i += NT)
                if (tr[i] == need - 1)
                    for (int k = 0; k < WS; ++k) sh->pivLex[k] = lexkey(c, st[(size_t)k * capC1 + tl[i]]);
            PAR_END
#if defined(DDO_HOST_EMULATION)
            if (getenv("DD_SELECT_TRACE")) std::fprintf(stderr, "TIEPAIRS b=%d need=%d\n", b, need);
#endif
            done = true;
        }
    }
    uint64_t pw = 0;
    uint64_t ldiff = 0, land = 0;   // bits of word wj in which the nodes still tied differ / agree on 1
#if defined(DDO_HOST_EMULATION)
    int dd_trace_words = 0, dd_trace_rounds = done ? -1 : 0;
#endif
    for (int qd = 0; qd < 8 * WS && !done; ++qd) {
        const int wj = qd >> 3;
        const int shift = 8 * (7 - (qd & 7));
        if ((qd & 7) == 0) {
#if defined(DDO_HOST_EMULATION)
            ++dd_trace_words;
#endif
            pw = 0;
            // one sweep per word: AND / OR of the word over the nodes still tied.  Bytes that are the same for all of them
            // decide nothing -- with 31-word signed-vector states most of the 248 digit rounds would be such bytes
            PAR_BEGIN
            if (tid == 0) {
                sh->k1and = ~0ULL;
                sh->k1or = 0;
            }
            PAR_END
            PAR_BEGIN
            uint64_t a = ~0ULL, o = 0;
            int lo = tid * q, hi = lo + q < ncl ? lo + q : ncl;
            for (int j = lo; j < hi; ++j) {
                int cd = lin2cand(j, nprev, c.capN);
                if (!c.cls[cd]) continue;
                const uint64_t lw = lexkey(c, st[(size_t)wj * capC1 + cd]);
                a &= lw;
                o |= lw;
            }
            if (a != ~0ULL || o != 0) {
                LDS_AND_U64(&sh->k1and, a);
                LDS_OR_U64(&sh->k1or, o);
            }
            PAR_END
            land = sh->k1and;
            ldiff = land ^ sh->k1or;
            DD_SYNC();   // every thread has its copy before thread 0 resets the pair for the next word
        }
        const bool same_byte = ((ldiff >> shift) & 0xFF) == 0;   // the same byte in every tied node
        if (same_byte) pw |= land & (0xFFULL << shift);
        else {
#if defined(DDO_HOST_EMULATION)
        ++dd_trace_rounds;
#endif
        PAR_BEGIN
        if (tid < 256) c.hist[tid] = 0;
        PAR_END
        PAR_BEGIN
        int lo = tid * q, hi = lo + q < ncl ? lo + q : ncl;
        for (int j = lo; j < hi; ++j) {
            int cd = lin2cand(j, nprev, c.capN);
            if (!c.cls[cd]) continue;
            uint64_t lw = lexkey(c, st[(size_t)wj * capC1 + cd]);
            if (shift + 8 < 64 && (lw >> (shift + 8)) != (pw >> (shift + 8))) continue;
            LDS_ADD_U32(&c.hist[(lw >> shift) & 0xFF], 1u);
        }
        PAR_END
        PAR_BEGIN
        if (tid < 256) {
            int above = 0;
            for (int x = tid + 1; x < 256; ++x) above += (int)c.hist[x];
            int mine = (int)c.hist[tid];
            if (above < need && need <= above + mine) {
                sh->sel_digit = tid;
                sh->sel_above = above;
                sh->sel_bucket = mine;
            }
        }
        PAR_END
        pw |= (uint64_t)sh->sel_digit << shift;
        need -= sh->sel_above;
        if (need == sh->sel_bucket) done = true;
        }
        if (done || (qd & 7) == 7) {   // the word is finished (or the selection is): publish it
            PAR_BEGIN
            if (tid == 0) sh->pivLex[wj] = pw;
            if (!done && ldiff != 0) {   // candidates whose word differs from the pivot's are decided: no longer tied
                int lo = tid * q, hi = lo + q < ncl ? lo + q : ncl;
                for (int j = lo; j < hi; ++j) {
                    int cd = lin2cand(j, nprev, c.capN);
                    if (c.cls[cd] && lexkey(c, st[(size_t)wj * capC1 + cd]) != pw) c.cls[cd] = 0;
                }
            }
            PAR_END
        }
    }
#if defined(DDO_HOST_EMULATION)
    if (getenv("DD_SELECT_TRACE") && dd_trace_rounds >= 0) std::fprintf(stderr, "TIEEND words=%d rounds=%d\n", dd_trace_words, dd_trace_rounds);
#endif
    PAR_BEGIN
    if (tid == 0) sh->pivK1 = pivK1;
    PAR_END
}

/// key(cand) >= pivot under (K1, lexkey words).
template <int WS>
DDO_DEV bool ge_pivot(const DDCtx<WS>& c, int cur, int cd, uint64_t k1) {
    const DDShared* sh = c.sh;
    if (k1 != sh->pivK1) return k1 > sh->pivK1;
    const uint64_t* st = c.cstate[cur];
    for (int k = 0; k < WS; ++k) {
        uint64_t lw = lexkey(c, st[(size_t)k * c.capC1 + cd]);
        if (lw != sh->pivLex[k]) return lw > sh->pivLex[k];
    }
    return true;
}

/// full-order "a ranks above b" for two candidates of buffer `cur`
template <int WS>
DDO_DEV bool ranks_above(const DDCtx<WS>& c, int cur, int a, int b) {
    uint64_t ka = k1_of(LD_U64(&c.ckey[cur][a]), c.cpop[cur][a]);
    uint64_t kb = k1_of(LD_U64(&c.ckey[cur][b]), c.cpop[cur][b]);
    if (ka != kb) return ka > kb;
    const uint64_t* st = c.cstate[cur];
    for (int k = 0; k < WS; ++k) {
        uint64_t la = lexkey(c, st[(size_t)k * c.capC1 + a]);
        uint64_t lb = lexkey(c, st[(size_t)k * c.capC1 + b]);
        if (la != lb) return la > lb;
    }
    return false;
}

/// Inserts candidate `cd` (state s, already stored in buffer nxt) into the dedup table.
/// Returns the winner candidate (== cd when this state is new).  clean.rs:738-775.
template <int WS>
DDO_DEV uint32_t dedup_insert(const DDCtx<WS>& c, int nxt, uint32_t cd, const uint64_t* s, int mask) {
    const uint64_t h = hash_state<WS>(s);
    const uint32_t tag = (uint32_t)(h >> (32 + c.cdbits));  // 12 bits (fewer when the candidate index needs more than 20)
    const uint32_t mine = (tag << c.cdbits) | cd;
    uint32_t slot = (uint32_t)h & (uint32_t)mask;
    const uint64_t* st = c.cstate[nxt];
    for (int probes = 0; probes <= mask; ++probes) {   // bounded: a full table is an internal error, not a hang
        uint32_t e = LD_U32(&c.table[slot]);
        if (e == TAB_EMPTY) {
            e = TAB_CAS(&c.table[slot], TAB_EMPTY, mine);
            if (e == TAB_EMPTY) return cd;
        }
        if ((e >> c.cdbits) == tag) {
            const uint32_t w = e & ((1u << c.cdbits) - 1u);
            bool eq = true;
            for (int k = 0; k < WS && eq; ++k) eq = LD_U64(&st[(size_t)k * c.capC1 + w]) == s[k];
            if (eq) return w;
        }
        slot = (slot + 1) & (uint32_t)mask;
    }
    c.sh->status = ST_ERR_INTERNAL;
    return cd;
}

/// hash_state as a stream: h = hash_begin(); h = hash_step(h, word) for every state word in order; hash_end(h)
DDO_DEV uint64_t hash_begin() { return 0x243F6A8885A308D3ULL; }
DDO_DEV uint64_t hash_step(uint64_t h, uint64_t w) {
    h ^= w;
    h *= 0x9E3779B97F4A7C15ULL;
    h ^= h >> 29;
    return h;
}
DDO_DEV uint64_t hash_end(uint64_t h) {
    h *= 0xBF58476D1CE4E5B9ULL;
    h ^= h >> 32;
    return h;
}
/// dedup_insert for a candidate whose state words ALREADY lie in cstate[nxt] (written by the caller, fenced) and whose hash the
/// caller computed while it wrote them: the wide signed-vector states (31 to 72 words) are never held in a per-thread array --
/// three such arrays per thread were 400+ registers, i.e. scratch memory, and the expansion of a MAX2SAT layer spent its time
/// there (round 3: 2 600 cycles per node).  A tag match is verified word against word out of cstate[nxt], eight words per batch.
template <int WS>
DDO_DEV uint32_t dedup_insert_stored(const DDCtx<WS>& c, int nxt, uint32_t cd, uint64_t h, int mask) {
    const uint32_t tag = (uint32_t)(h >> (32 + c.cdbits));  // 12 bits (fewer when the candidate index needs more than 20)
    const uint32_t mine = (tag << c.cdbits) | cd;
    uint32_t slot = (uint32_t)h & (uint32_t)mask;
    const uint64_t* st = c.cstate[nxt];
    for (int probes = 0; probes <= mask; ++probes) {   // bounded: a full table is an internal error, not a hang
        const uint32_t e = TAB_CAS(&c.table[slot], TAB_EMPTY, mine);
        if (e == TAB_EMPTY) return cd;
        if ((e >> c.cdbits) == tag) {
            const uint32_t w = e & ((1u << c.cdbits) - 1u);
            bool eq = true;
            for (int k0 = 0; k0 < WS && eq; k0 += 8) {
                uint64_t a[8], b[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool in = k0 + q < WS;
                    a[q] = in ? LD_U64(&st[(size_t)(k0 + q) * c.capC1 + w]) : 0;
                    b[q] = in ? LD_U64(&st[(size_t)(k0 + q) * c.capC1 + cd]) : 0;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) eq &= a[q] == b[q];
            }
            if (eq) return w;
        }
        slot = (slot + 1) & (uint32_t)mask;
    }
    c.sh->status = ST_ERR_INTERNAL;
    return cd;
}

/// read-only probe (recycled-merge detection, clean.rs:830): candidate holding state s, or NONE32
template <int WS>
DDO_DEV uint32_t dedup_find(const DDCtx<WS>& c, int buf, const uint64_t* s, int mask) {
    const uint64_t h = hash_state<WS>(s);
    const uint32_t tag = (uint32_t)(h >> (32 + c.cdbits));
    uint32_t slot = (uint32_t)h & (uint32_t)mask;
    const uint64_t* st = c.cstate[buf];
    for (int probes = 0; probes <= mask; ++probes) {
        uint32_t e = c.table[slot];
        if (e == TAB_EMPTY) return NONE32;
        if ((e >> c.cdbits) == tag) {
            const uint32_t w = e & ((1u << c.cdbits) - 1u);
            bool eq = true;
            for (int k = 0; k < WS && eq; ++k) eq = st[(size_t)k * c.capC1 + w] == s[k];
            if (eq) return w;
        }
        slot = (slot + 1) & (uint32_t)mask;
    }
    return NONE32;
}

DDO_DEV int table_size_for(int ncand_max, int cap) {
    int h = 1024;
    while (h < cap && h * 2 < ncand_max * 3) h <<= 1;  // load factor <= 2/3
    return h;
}

/// _filter_with_dominance (clean.rs:689-708) with SimpleDominanceChecker (dominance/simple.rs:67-111) for a dominance whose
/// key is the depth and whose states have one coordinate (state word 0) besides the value -- KPDominance.  The set of
/// non-dominated (coordinate, value) pairs of a depth is a Pareto front; kept sorted by coordinate (values then strictly
/// decrease), `is_dominated_or_insert` is a binary search, one walk over the dominating run (its smallest value is the
/// threshold; an entry that differs in value only contributes value - 1) or one splice that drops what the new pair
/// dominates.  The reference processes curr_l in dominance order (value, then coordinate, descending): ranks by counting,
/// then ONE thread replays that order under the depth's lock -- the data structure is shared by all compiles in flight.
template <int WS, class Ctx>
DDO_DEV void dominance_filter(Ctx& c, int cur, int nprev, int depth) {
    DD_TID_SETUP(c)
    auto* sh = c.sh;
    const int capN = c.capN;
    const int ncl = c.fan * nprev;
    uint32_t* order = c.keep;                 // free until the positions are assigned (capC1 entries with kept layers)
    PAR_BEGIN
    if (tid == 0) sh->scan_total = 0;
    PAR_END
    PAR_BEGIN   // rank of every exact node of curr_l in (value, coordinate) descending order
    for (int j = tid; j < ncl; j += NT) {
        const int cd = lin2cand(j, nprev, capN);
        if (!cand_live(c, cur, cd) || (LD_U32(&c.cflags[cur][cd]) & (NF_INEXACT | NF_RELAXED))) continue;
        const int32_t v = unbias32((uint32_t)(LD_U64(&c.ckey[cur][cd]) >> 32));
        const uint64_t co = c.cstate[cur][cd];
        int rank = 0;
        for (int j2 = 0; j2 < ncl; ++j2) {
            const int c2 = lin2cand(j2, nprev, capN);
            if (c2 == cd || !cand_live(c, cur, c2) || (LD_U32(&c.cflags[cur][c2]) & (NF_INEXACT | NF_RELAXED))) continue;
            const int32_t v2 = unbias32((uint32_t)(LD_U64(&c.ckey[cur][c2]) >> 32));
            const uint64_t co2 = c.cstate[cur][c2];
            rank += (v2 > v || (v2 == v && co2 > co)) ? 1 : 0;
        }
        order[rank] = (uint32_t)cd;
        LDS_ADD_I32(&sh->scan_total, 1);
    }
    PAR_END
    const int m = sh->scan_total;
    DD_SYNC();
    PAR_BEGIN
    if (tid == 0 && m > 0) {
        uint64_t* fc = c.dom_coord + (size_t)depth * c.dom_cap;
        int32_t* fv = c.dom_value + (size_t)depth * c.dom_cap;
#if !defined(DDO_HOST_EMULATION)
        while (atomicCAS(&c.dom_lock[depth], 0u, 1u) != 0u) {}
        __threadfence();
#endif
        int F = (int)CT_LD(&c.dom_count[depth]);
        for (int q = 0; q < m; ++q) {
            const int cd = (int)order[q];
            const int32_t v = unbias32((uint32_t)(LD_U64(&c.ckey[cur][cd]) >> 32));
            const uint64_t co = c.cstate[cur][cd];
            int lo = 0, hi = F;                       // first entry with coordinate >= co
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (CT_LD(&fc[mid]) < co) lo = mid + 1; else hi = mid;
            }
            const int idx = lo;
            bool dominated = false;
            int32_t th = TH_INF;
            if (!(idx < F && CT_LD(&fc[idx]) == co && CT_LD(&fv[idx]) == v)) {   // an equal pair is replaced, it does not dominate
                for (int k = idx; k < F; ++k) {
                    const int32_t ov = CT_LD(&fv[k]);
                    if (ov < v) break;
                    dominated = true;
                    const int32_t t = CT_LD(&fc[k]) == co ? ov - 1 : ov;   // only the value differs: threshold value - 1
                    th = t < th ? t : th;
                }
            }
            if (dominated) {
                c.cflags[cur][cd] = LD_U32(&c.cflags[cur][cd]) | NF_DOM;
                c.cth[cd] = th;
                sh->ncache += 1;
                continue;
            }
            int first = idx;                           // entries the new pair dominates or equals: a run ending at idx
            while (first > 0 && CT_LD(&fv[first - 1]) <= v) --first;
            const int last = idx + ((idx < F && CT_LD(&fc[idx]) == co) ? 1 : 0);
            const int removed = last - first;
            if (F - removed + 1 > (int)c.dom_cap) {   // front full: the pair is not recorded (sound: less pruning later)
                CT_ADD(&c.dom_stats[0], 1ULL);
                continue;
            }
            if (removed == 0) {
                for (int k = F; k > first; --k) {
                    CT_ST(&fc[k], CT_LD(&fc[k - 1]));
                    CT_ST(&fv[k], CT_LD(&fv[k - 1]));
                }
            } else if (removed > 1) {
                for (int k = last; k < F; ++k) {
                    CT_ST(&fc[k - removed + 1], CT_LD(&fc[k]));
                    CT_ST(&fv[k - removed + 1], CT_LD(&fv[k]));
                }
            }
            CT_ST(&fc[first], co);
            CT_ST(&fv[first], v);
            F = F - removed + 1;
        }
        CT_ST(&c.dom_count[depth], (uint32_t)F);
#if !defined(DDO_HOST_EMULATION)
        __threadfence();
        atomicExch(&c.dom_lock[depth], 0u);
#endif
    }
    PAR_END
}

/// One compile() -- clean.rs:345-381 -- of the sub-problem `in` with the given type.
template <int WS>
DDO_DEV void run_dd(DDCtx<WS>& c, const DDInput& in, int comp_type, int64_t best_lb, DDResult* res) {
    DD_TID_SETUP(c)
    DDShared* sh = c.sh;
    const int capN = c.capN, capC1 = c.capC1;
    const int W = in.width;
    const int MERGED = c.fan * capN;
    const bool relaxed = comp_type == CT_RELAXED;
    const bool restricted = comp_type == CT_RESTRICTED;
    const uint32_t dmask = (1u << c.dbits) - 1u;
    const int LS = c.tmode ? c.lstride : capN;                        // nodes per layer in the per-layer arrays
    // Kept layers and arc arrays either sit at fixed strides (layer X at X * LS, its arcs at X * fan * capN: a slot is sized for
    // the widest layer any DD may have in EVERY layer) or -- c.lpool != 0, round 4: TSPTW beyond 64 nodes, where that is 25 GB per
    // slot and four DDs in flight -- are carved out of two per-slot pools as the DD grows: layer X starts at lbase[X] and holds
    // lntot[X] nodes, the arcs entering it start at abase[X] and are indexed by (decision, parent position) of the nlayer[X - 1]
    // parents.  A DD whose layers outgrow the pools ends with a capacity error (loud), never with a wrong result.
    constexpr bool POOLS = tw_k_of_ws(WS) != 0;   // only the D-ary model uses the pools: the other instantiations do not even carry the offsets
    const bool dynl = POOLS && c.lpool != 0;
    // (macros, not lambdas: capturing lambdas changed the register allocation of the 72-word kernel enough to cost it 30 % on frb15-9-1)
#define LB(X) (dynl ? (size_t)c.lbase[(X)] : (size_t)(X) * (size_t)LS)
#define LSX(X, k, pos) (dynl ? (size_t)c.lbase[(X)] * WS + (size_t)(k) * (size_t)c.lntot[(X)] + (pos) : ((size_t)(X) * WS + (k)) * (size_t)LS + (pos))   /* word k of node pos of kept layer X in c.lstate */
#define AB(X) (dynl ? (size_t)c.abase[(X)] : (size_t)(X) * (size_t)c.fan * (size_t)capN)
#define AI(cd, np) (dynl ? (size_t)((cd) / (uint32_t)capN) * (size_t)(np) + (size_t)((cd) % (uint32_t)capN) : (size_t)(cd))   /* candidate cd = decision * capN + parent position, among np parents */
    uint64_t node_off = 0, arc_off = 0;   // pool heads of this DD (every thread counts along: workgroup-uniform)
    size_t lb_cur_ = 0, ab_cur_ = 0, ab_next_ = 0;   // POOLS: start of the layer being built, of the arcs entering it, of the arcs entering the next one
    // (without POOLS these are plain expressions of the layer number, as they were: the 72-word kernel is sensitive to every
    // loop-carried scalar -- carrying them cost it 27 % on frb15-9-1)
#define lb_cur (POOLS ? lb_cur_ : (size_t)L * (size_t)LS)
#define ab_cur (POOLS ? ab_cur_ : (size_t)L * (size_t)c.fan * (size_t)capN)
#define ab_next (POOLS ? ab_next_ : (size_t)(L + 1) * (size_t)c.fan * (size_t)capN)
    const bool frontier = c.tmode && (in.flags & IN_FRONTIER) != 0;   // CUTSET_TYPE == FRONTIER
    const bool use_cache = c.tmode && (in.flags & IN_CACHE) != 0 && c.cache_cap != 0;
    const bool use_dom = c.tmode && (in.flags & IN_DOMINANCE) != 0 && c.dom_cap != 0 && dd_kind_is<WS>(c.kind, MODEL_KNAPSACK);
    constexpr int TWK = tw_k_of_ws(WS) != 0 ? tw_k_of_ws(WS) : 1;       // words of a TSPTW node set at this state width (dd_tsptw.hpp)
    constexpr int DKW = 2 * TWK + 1;                                    // words of a TsptwDominance key
    const bool use_dkey = tw_k_of_ws(WS) != 0 && c.tmode && (in.flags & IN_DOMINANCE) != 0 && c.dkey_cap != 0 && dd_kind_is<WS>(c.kind, MODEL_TSPTW);

    // ---------------------------------------------------------------- _clear + _initialize
    int cur = 0;
    // The dedup table is sized PER LAYER for the candidates the layer can have -- fan x its parents -- not for the widest layer
    // the DD may ever see: a TSPTW DD asked for width 20 000 cleared a table of 2 M entries per layer for layers of a few dozen
    // nodes (round 3: 195 us per layer, 0.045 x ONE thread of the oracle on config C5).  `hmask` is the mask of the table of the
    // CURRENT unique layer: set when expand builds it, used again by the recycled-merge probe of the next layer's squash.
    int hsize = table_size_for(c.fan + 2, c.table_cap);
    int hmask = hsize - 1;
    PAR_BEGIN
    for (int i = tid; i < c.npad; i += NT) c.cnt[i] = 0;
    if (tid == 0) {
        sh->nU = 1;
        sh->status = ST_OK;
        sh->nodes = 0;
        sh->arcs = 0;
        sh->recycled_merges = 0;
        sh->maxn = 0;
        sh->cutoff = 0;
        for (int k = 0; k < 8; ++k) sh->clk[k] = 0;
        sh->clk_last = dd_clock();
        sh->ncache = 0;
        sh->cache_hits = 0;
        for (int k = 0; k < WS; ++k) c.cstate[0][(size_t)k * capC1] = in.state[k];
        int pop = 0;
        if (dd_kind_is<WS>(c.kind, MODEL_MISP))
            for (int k = 0; k < WS; ++k) pop += dd_popc(in.state[k]);
        if (dd_is_vec_w<WS>(c.kind)) pop = vec_rank<WS>(in.state, c.n, 0);
        c.ckey[0][0] = ((uint64_t)bias32(in.value) << 32) | NONE32;
        c.cpop[0][0] = (uint32_t)pop;
        c.cflags[0][0] = 0;
        c.ctarget[0] = 0;        // the root is its own "winner"
        for (int d = 1; d < c.fan; ++d) c.ctarget[(size_t)d * capN] = NONE32;
    }
    PAR_END
    PAR_BEGIN
    if (tid == 0) if (dd_kind_is<WS>(c.kind, MODEL_MISP)) add_bits<WS>(c.cnt, in.state, +1);
    // the table must describe the current unique layer (needed by recycled-merge probes)
    for (int i = tid; i < hsize; i += NT) c.table[i] = TAB_EMPTY;
    PAR_END

    int nprev = 1;   // parents that produced the current candidates
    int lel = -1;    // Option<LayerId>
    int L = 0;       // layers.len()
    int var = -1;
    bool failed = false;

    for (;;) {
        // ------------------------------------------------------------ next_variable (main.rs:109-143)
        PAR_BEGIN
        if (tid == 0) {
            sh->varkey = 0xFFFFFFFFu;
            if (c.cutoff_flag) sh->cutoff = LD_I32(c.cutoff_flag);
        }
        PAR_END
        PAR_BEGIN
        if (!dd_kind_is<WS>(c.kind, MODEL_MISP)) {   // static order (knapsack/main.rs:118-125, mcp/model.rs:88-96); an empty layer ends the DD
            if (tid == 0 && c.depth0 + L < c.n && sh->nU > 0)
                sh->varkey = (c.kind == MODEL_MCP || dd_kind_is<WS>(c.kind, MODEL_TSPTW)) ? (uint32_t)(c.depth0 + L)   // tsptw/model.rs:140-147
                             : c.kind == MODEL_MAX2SAT ? (uint32_t)c.m2_order[c.n - (c.depth0 + L) - 1]   // model.rs:330-346
                                                       : (uint32_t)c.kp_order[c.depth0 + L];
        } else {
            for (int i = tid; i < c.n; i += NT) {
                int cv = c.cnt[i];
                if (cv > 0) LDS_MIN_U32(&sh->varkey, ((uint32_t)cv << 12) | (uint32_t)i);
                else if (cv < 0) sh->status = ST_ERR_INTERNAL;
            }
        }
        PAR_END
        var = sh->varkey == 0xFFFFFFFFu ? -1 : (int)(sh->varkey & 0xFFFu);
        if (var < 0) break;
        if (sh->cutoff) {  // clean.rs:352-354
            PAR_BEGIN
            if (tid == 0) sh->status = ST_CUTOFF;
            PAR_END
            failed = true;
            break;
        }
        if (L >= c.max_layers - 1 || sh->status != ST_OK) { failed = true; break; }
        // ------------------------------------------------------------ _filter_with_cache (clean.rs:710-726)
        // not for the root layer (clean.rs:671): a unique candidate whose value does not exceed the threshold the cache
        // holds for (depth, state) leaves curr_l; it stays in the layer (its theta is propagated upwards, clean.rs:502)
        if (use_cache && L >= 1) {
            PAR_BEGIN
            if (tid == 0) sh->ncache = 0;
            PAR_END
            PAR_BEGIN
            const int nclx = c.fan * nprev;
            for (int j = tid; j < nclx; j += NT) {
                const int cd = lin2cand(j, nprev, capN);
                if (c.ctarget[cd] != (uint32_t)cd) continue;
                uint64_t s[WS];
#pragma unroll
                for (int k = 0; k < WS; ++k) s[k] = c.cstate[cur][(size_t)k * capC1 + cd];
                int64_t packed = 0;
                if (!cache_get<WS>(c, s, c.depth0 + L, &packed)) continue;
                const int32_t val = unbias32((uint32_t)(LD_U64(&c.ckey[cur][cd]) >> 32));
                const int32_t tv = th_value(packed);
                if (tv != TH_INF && val > tv) continue;            // node.value_top > threshold.value: keep
                c.cflags[cur][cd] = LD_U32(&c.cflags[cur][cd]) | NF_CACHE;
                c.cth[cd] = tv;
                if (dd_kind_is<WS>(c.kind, MODEL_MISP)) add_bits<WS>(c.cnt, s, -1);   // it leaves the layer next_variable will look at
                LDS_ADD_I32(&sh->ncache, 1);
            }
            PAR_END
        }
        int ncache = (use_cache && L >= 1) ? sh->ncache : 0;
        DD1_TICK(7)   // next_variable, cache filter
        // ------------------------------------------------------------ _filter_with_dominance (clean.rs:689-708)
        // every layer, the root's too: the exact nodes of curr_l, best (value, coordinate) first, are checked against -- and
        // added to -- the set of non-dominated states of this depth (dominance/simple.rs:67-111); a dominated node leaves
        // curr_l with the threshold the checker returns
        if (use_dom) {
            if (ncache == 0) {
                PAR_BEGIN
                if (tid == 0) sh->ncache = 0;
                PAR_END
            }
            const int before = sh->ncache;
            DD_SYNC();
            dominance_filter<WS>(c, cur, nprev, c.depth0 + L);
            ncache = sh->ncache;
            (void)before;
        }
        if (use_dkey) {
            // TsptwDominance (examples/tsptw/dominance.rs:26-60): states with the same (position, must_visit) compare on their
            // value alone.  The reference walks curr_l best value first, so a node is dominated exactly when the best value
            // recorded for its key -- by earlier compiles or by this layer -- is larger than its own: record all, then test.
            if (ncache == 0) {
                PAR_BEGIN
                if (tid == 0) sh->ncache = 0;
                PAR_END
            }
            struct { uint64_t* cache_tab; uint64_t cache_cap; int cache_stride; unsigned long long* cache_stats; } dk{c.dkey_tab, c.dkey_cap, DKW + 3, c.dkey_stats};
            const int nclx = c.fan * nprev;
            PAR_BEGIN
            for (int j = tid; j < nclx; j += NT) {
                const int cd = lin2cand(j, nprev, capN);
                if (!cand_live(c, cur, cd) || (LD_U32(&c.cflags[cur][cd]) & (NF_INEXACT | NF_RELAXED))) continue;
                uint64_t keyw[DKW];
                tw_dominance_key<TWK>(c.cstate[cur], capC1, cd, keyw);
                cache_update<DKW>(dk, keyw, c.depth0 + L, th_pack(unbias32((uint32_t)(LD_U64(&c.ckey[cur][cd]) >> 32)), false));
            }
            PAR_END
            PAR_BEGIN
            for (int j = tid; j < nclx; j += NT) {
                const int cd = lin2cand(j, nprev, capN);
                if (!cand_live(c, cur, cd) || (LD_U32(&c.cflags[cur][cd]) & (NF_INEXACT | NF_RELAXED))) continue;
                uint64_t keyw[DKW];
                tw_dominance_key<TWK>(c.cstate[cur], capC1, cd, keyw);
                int64_t packed = 0;
                if (!cache_get<DKW>(dk, keyw, c.depth0 + L, &packed)) continue;
                const int32_t best = th_value(packed), val = unbias32((uint32_t)(LD_U64(&c.ckey[cur][cd]) >> 32));
                if (best > val) {                                   // dominated: only the value differs -> threshold value - 1
                    c.cflags[cur][cd] = LD_U32(&c.cflags[cur][cd]) | NF_DOM;
                    c.cth[cd] = best - 1;
                    LDS_ADD_I32(&sh->ncache, 1);
                }
            }
            PAR_END
            ncache = sh->ncache;
        }
        const int nU = sh->nU - ncache;                               // |curr_l| after the filters

        DD1_TICK(0)   // dominance filters
        // ------------------------------------------------------------ _squash_if_needed (clean.rs:779-795)
        const bool squash = (restricted && nU > W) || (relaxed && nU > W && L > 1);
        if ((!squash && nU > capN) || (c.tmode && nU + ncache + 1 > LS)) {  // Exact DD wider than the workspace
            PAR_BEGIN
            if (tid == 0) sh->status = ST_ERR_CAPACITY;
            PAR_END
            failed = true;
            break;
        }
        const int ncl = c.fan * nprev;
        const int q = (ncl + NT - 1) / NT;
        int K = 0;
        if (squash) {
            if (lel < 0) {
                lel = L - 1;  // _maybe_save_lel
                if (relaxed) {
                    // keep a copy of the last exact layer: it becomes the cut-set (clean.rs:566-573)
                    PAR_BEGIN
                    for (int pos = tid; pos < nprev; pos += NT) {
                        uint32_t p = c.keep[pos];
                        for (int k = 0; k < WS; ++k)
                            c.cs_state[(size_t)k * capN + pos] = c.cstate[cur ^ 1][(size_t)k * capC1 + p];
                        c.cs_value[pos] = unbias32((uint32_t)(LD_U64(&c.ckey[cur ^ 1][p]) >> 32));
                        c.cs_pop[pos] = c.cpop[cur ^ 1][p];
                    }
                    PAR_END
                }
            }
            K = restricted ? W : W - 1;
            if (K > 0) select_pivot<WS>(c, cur, nprev, K);
        }
        DD1_TICK(1)   // exact top-K pivot: radix select on the key, tie-break on the state words

        // ------------------------------------------------------------ classify + count (pass 1)
        PAR_BEGIN
        if (tid == 0) {
            sh->mergedKey = 0;
            for (int k = 0; k < WS; ++k) sh->merged[k] = 0;
            if (tw_k_of_ws(WS) != 0 && dd_kind_is<WS>(c.kind, MODEL_TSPTW)) {
                for (int k = 0; k < TWK; ++k) sh->merged[TWK + k] = ~0ULL;   // intersection of the must-visit sets
                sh->vmin[0] = 0xFFFFFFFFu;                                   // earliest elapsed time
            }
            if (dd_is_vec_w<WS>(c.kind)) {
                for (int v = 0; v < MAX_VEC_VARS; ++v) sh->vmin[v] = 0xFFFFFFFFu;
                for (int q = 0; q < (MAX_VEC_VARS + 63) / 64; ++q) sh->vposmask[q] = sh->vnegmask[q] = 0;
            }
            sh->mrank = 0;
            sh->xdelta = 0;
            sh->recycled = 0;
            sh->dup_from = -1;
            sh->dup_to = -1;
            sh->xbest = -1;
        }
        PAR_END
        PAR_BEGIN
        int lo = tid * q, hi = lo + q < ncl ? lo + q : ncl;
        int kept = 0;
        uint64_t mor[WS];
#pragma unroll
        for (int k = 0; k < WS; ++k) mor[k] = 0;
        uint64_t mkey = 0;
        bool anydel = false;
        for (int j = lo; j < hi; ++j) {
            int cd = lin2cand(j, nprev, capN);
            uint8_t cl = 0;
            if (c.ctarget[cd] == (uint32_t)cd && c.tmode && (c.cflags[cur][cd] & (NF_CACHE | NF_DOM))) {
                cl = 3;   // pruned by the cache / dominated: a node of the layer, not of curr_l
            } else if (c.ctarget[cd] == (uint32_t)cd) {
                cl = 1;
                if (squash) {
                    uint64_t key = LD_U64(&c.ckey[cur][cd]);
                    bool keepit = K > 0 && ge_pivot<WS>(c, cur, cd, k1_of(key, c.cpop[cur][cd]));
                    if (!keepit) {
                        cl = 2;
                        uint64_t s[WS];
                        if (!dd_is_vec_w<WS>(c.kind)) {   // (the wide signed-vector states are streamed below, word by word: no array)
#pragma unroll
                            for (int k = 0; k < WS; ++k) s[k] = c.cstate[cur][(size_t)k * capC1 + cd];
                        }
                        if (dd_kind_is<WS>(c.kind, MODEL_MISP)) add_bits<WS>(c.cnt, s, -1);  // it leaves the layer
                        if (relaxed && dd_is_vec_w<WS>(c.kind)) {
                            // McpRelax::merge (relax.rs:141-176): per variable the signs seen and the smallest |benefit|;
                            // relax (relax.rs:115-121) adds rank(victim) - rank(merged) to every redirected arc, so the
                            // merged node's value is max(value + rank) over the victims minus its own rank
                            uint64_t pmw = 0, nmw = 0;   // sign masks of the 64 variables of the current mask word (32 state words)
#pragma unroll 4
                            for (int k = 0; k < WS; ++k) {
                                if (2 * k >= c.n) break;
                                const uint64_t sw = c.cstate[cur][(size_t)k * capC1 + cd];
                                const int32_t a0 = (int32_t)(uint32_t)sw, a1 = (int32_t)(uint32_t)(sw >> 32);
                                // (the minimum settles after a few victims: a plain read -- same address for the whole wave, a
                                // broadcast -- spares the atomic, which the hardware runs lane after lane)
                                if ((uint32_t)iabs32(a0) < sh->vmin[2 * k]) LDS_MIN_U32(&sh->vmin[2 * k], (uint32_t)iabs32(a0));
                                if (a0 > 0) pmw |= 1ULL << ((2 * k) & 63);
                                if (a0 < 0) nmw |= 1ULL << ((2 * k) & 63);
                                if (2 * k + 1 < c.n) {
                                    if ((uint32_t)iabs32(a1) < sh->vmin[2 * k + 1]) LDS_MIN_U32(&sh->vmin[2 * k + 1], (uint32_t)iabs32(a1));
                                    if (a1 > 0) pmw |= 1ULL << ((2 * k + 1) & 63);
                                    if (a1 < 0) nmw |= 1ULL << ((2 * k + 1) & 63);
                                }
                                if ((k & 31) == 31 || 2 * (k + 1) >= c.n) {   // the mask word is complete
                                    if (pmw) LDS_OR_U64(&sh->vposmask[k >> 5], pmw);
                                    if (nmw) LDS_OR_U64(&sh->vnegmask[k >> 5], nmw);
                                    pmw = nmw = 0;
                                }
                            }
                            const int32_t adj = unbias32((uint32_t)(key >> 32)) + (int32_t)c.cpop[cur][cd];
                            const uint64_t akey = ((uint64_t)bias32(adj) << 32) | (uint32_t)key;
                            if (akey > mkey) mkey = akey;
                            anydel = true;
                        } else if (relaxed && dd_kind_is<WS>(c.kind, MODEL_TSPTW)) {
                            // TsptwRelax::merge (relax.rs:65-191): union of the positions, intersection / union of the must-visit
                            // sets, union of the maybe-visit sets, earliest and latest elapsed time
                            // (sh->merged: K words each of positions | agreed must | all must | all maybe -- 4K <= WS)
                            if constexpr (tw_k_of_ws(WS) != 0) {
                                const uint64_t meta = tw_meta<TWK>(s);
#pragma unroll
                                for (int q = 0; q < TWK; ++q) {
                                    const uint64_t here = (meta & TW_VIRTUAL) ? s[q] : ((int)((meta & 0xFFFF) >> 6) == q ? 1ULL << (meta & 63) : 0ULL);
                                    if (here) LDS_OR_U64(&sh->merged[q], here);
                                    LDS_AND_U64(&sh->merged[TWK + q], s[TWK + q]);
                                    if (s[TWK + q]) LDS_OR_U64(&sh->merged[2 * TWK + q], s[TWK + q]);
                                    if ((meta & TW_MAYBE) && s[2 * TWK + q]) LDS_OR_U64(&sh->merged[3 * TWK + q], s[2 * TWK + q]);
                                }
                                LDS_MIN_U32(&sh->vmin[0], tw_earliest<TWK>(s));
                                LDS_MAX_I32(&sh->mrank, (int32_t)tw_latest<TWK>(s));
                            }
                            if (key > mkey) mkey = key;
                            anydel = true;
                        } else if (relaxed) {
                            if (dd_kind_is<WS>(c.kind, MODEL_KNAPSACK)) {   // KPRelax::merge: the largest capacity (main.rs:150-152)
#pragma unroll
                                for (int k = 0; k < WS; ++k)
                                    if (s[k] > mor[k]) mor[k] = s[k];   // word 1 is the depth, equal over the layer
                            } else {
#pragma unroll
                                for (int k = 0; k < WS; ++k) mor[k] |= s[k];
                            }
                            if (key > mkey) mkey = key;
                            anydel = true;
                        }
                    }
                }
                kept += (cl == 1);
            }
            c.cls[cd] = cl;
        }
        c.tcount[tid] = kept;
        if (anydel) {
            if (dd_is_vec_w<WS>(c.kind) || dd_kind_is<WS>(c.kind, MODEL_TSPTW)) {
                // reductions already done per victim
            } else if (dd_kind_is<WS>(c.kind, MODEL_KNAPSACK)) {
#pragma unroll
                for (int k = 0; k < WS; ++k) LDS_MAX_U64(&sh->merged[k], mor[k]);
            } else {
#pragma unroll
                for (int k = 0; k < WS; ++k)
                    if (mor[k]) LDS_OR_U64(&sh->merged[k], mor[k]);   // MispRelax::merge (main.rs:172-178)
            }
            LDS_MAX_U64(&sh->mergedKey, mkey);
        }
        PAR_END
        block_exclusive_scan<WS>(c, c.tcount, c.tcount2);
        const int nkept = sh->scan_total;
#if defined(DDO_HOST_EMULATION)
        if (getenv("DD_TRACE")) std::printf("SQ L=%d squash=%d nU=%d W=%d K=%d nkept=%d pivK1=%llx nprev=%d\n", L, (int)squash, nU, W, K, nkept, (unsigned long long)sh->pivK1, nprev);
#endif

        // ------------------------------------------------------------ assign positions (pass 2)
        PAR_BEGIN
        int lo = tid * q, hi = lo + q < ncl ? lo + q : ncl;
        int pos = c.tcount[tid];
        for (int j = lo; j < hi; ++j) {
            int cd = lin2cand(j, nprev, capN);
            if (c.cls[cd] == 1) {
                c.keep[pos] = (uint32_t)cd;
                c.posmap[cd] = (uint32_t)pos;
                ++pos;
            }
        }
        PAR_END

        int n = nkept;
        int merged_pos = -1;
        if (squash && relaxed) {
            // ------------------------------------------------------------ _relax: merged node (clean.rs:826-875)
            PAR_BEGIN
            if (tid == 0) {
                uint64_t ms[WS];
                for (int k = 0; k < WS; ++k) ms[k] = sh->merged[k];
                if (dd_kind_is<WS>(c.kind, MODEL_TSPTW)) {   // RelaxHelper::get_* (relax.rs:120-166)
                    const uint32_t e = sh->vmin[0], l = (uint32_t)sh->mrank;
                    for (int k = 0; k < WS; ++k) ms[k] = 0;
                    if constexpr (tw_k_of_ws(WS) != 0) {
                        uint64_t any_maybe = 0;
                        for (int q = 0; q < TWK; ++q) {
                            const uint64_t agree = sh->merged[TWK + q], all_must = sh->merged[2 * TWK + q], all_maybe = sh->merged[3 * TWK + q];
                            const uint64_t maybe = (all_maybe | all_must) & ~agree;
                            ms[q] = sh->merged[q];
                            ms[TWK + q] = agree;
                            ms[2 * TWK + q] = maybe;
                            any_maybe |= maybe;
                        }
                        ms[3 * TWK] = (uint64_t)e | ((uint64_t)(e != l ? l : e) << 32);
                        ms[3 * TWK + 1] = TW_VIRTUAL | (e != l ? TW_FUZZY : 0) | (any_maybe ? TW_MAYBE : 0) | ((uint64_t)(c.depth0 + L) << 32);
                    }
                    sh->mrank = 0;
                }
                if (dd_is_vec_w<WS>(c.kind)) {
                    // merged benefit: all signs agree -> the value closest to zero, else 0 (relax.rs:141-176)
                    int32_t mrank = 0;
                    for (int k = 0; k < WS; ++k) {
                        uint64_t w = 0;
                        for (int hsel = 0; hsel < 2; ++hsel) {
                            const int v = 2 * k + hsel;
                            if (v >= c.n) continue;
                            const bool posi = (sh->vposmask[v >> 6] >> (v & 63)) & 1ULL, nega = (sh->vnegmask[v >> 6] >> (v & 63)) & 1ULL;
                            int32_t b = 0;
                            if (posi && !nega) b = (int32_t)sh->vmin[v];
                            else if (nega && !posi) b = -(int32_t)sh->vmin[v];
                            mrank += iabs32(b);
                            w |= (uint64_t)(uint32_t)b << (32 * hsel);
                        }
                        ms[k] = w;
                    }
                    ms[(c.n + 1) / 2] = (uint64_t)(c.depth0 + L);   // depth word (every node of the layer has it)
                    sh->mrank = mrank;
                    const int32_t mv = unbias32((uint32_t)(sh->mergedKey >> 32)) - mrank;
                    sh->mergedKey = ((uint64_t)bias32(mv) << 32) | (uint32_t)sh->mergedKey;
                }
                uint32_t r = dedup_find<WS>(c, cur, ms, hmask);
                if (r != NONE32 && c.cls[r] == 1) {
                    sh->recycled = 1;  // clean.rs:830
                    sh->recycled_merges += 1;
                    sh->merged_pos = (int32_t)c.posmap[r];
                    uint64_t old = LD_U64(&c.ckey[cur][r]);
                    if (sh->mergedKey > old) c.ckey[cur][r] = sh->mergedKey;
                    c.cflags[cur][r] = LD_U32(&c.cflags[cur][r]) | NF_RELAXED;
                } else {
                    int pop = 0;
                    for (int k = 0; k < WS; ++k) {
                        c.cstate[cur][(size_t)k * capC1 + MERGED] = ms[k];
                        if (dd_kind_is<WS>(c.kind, MODEL_MISP)) pop += dd_popc(ms[k]);
                    }
                    if (dd_is_vec_w<WS>(c.kind)) pop = sh->mrank;
                    c.ckey[cur][MERGED] = sh->mergedKey;
                    c.cpop[cur][MERGED] = (uint32_t)pop;
                    c.cflags[cur][MERGED] = NF_RELAXED | NF_INEXACT;
                    c.keep[nkept] = (uint32_t)MERGED;
                    c.posmap[MERGED] = (uint32_t)nkept;
                    c.cls[MERGED] = 1;
                    sh->merged_pos = nkept;
                    if (dd_kind_is<WS>(c.kind, MODEL_MISP)) add_bits<WS>(c.cnt, ms, +1);
                }
            }
            PAR_END
            merged_pos = sh->merged_pos;
            if (!sh->recycled) {
                n = nkept + 1;
            } else {
                // clean.rs:868-872: the layer is truncated to W entries, i.e. the best-ranked node of
                // the merged set stays in the layer (un-deleted) next to the recycled node.
                PAR_BEGIN
                if (tid < 64) {
                    int best = -1;
                    for (int j = tid; j < ncl; j += 64) {
                        int cd = lin2cand(j, nprev, capN);
                        if (c.cls[cd] == 2 && (best < 0 || ranks_above<WS>(c, cur, cd, best))) best = cd;
                    }
                    sh->xcand[tid] = best;
                }
                PAR_END
                PAR_BEGIN
                if (tid == 0) {
                    int best = -1;
                    for (int l = 0; l < 64; ++l) {
                        int cd = sh->xcand[l];
                        if (cd >= 0 && (best < 0 || ranks_above<WS>(c, cur, cd, best))) best = cd;
                    }
                    sh->xbest = best;
                    uint64_t s[WS];
                    for (int k = 0; k < WS; ++k) s[k] = c.cstate[cur][(size_t)k * capC1 + best];
                    if (dd_kind_is<WS>(c.kind, MODEL_MISP)) add_bits<WS>(c.cnt, s, +1);
                    c.cls[best] = 1;
                    c.keep[nkept] = (uint32_t)best;
                    c.posmap[best] = (uint32_t)nkept;
                    sh->dup_from = nkept;
                    sh->dup_to = sh->merged_pos;
                    // its arcs were ALSO redirected to the recycled node, with relaxed costs (relax.rs:115-121)
                    if (dd_is_vec_w<WS>(c.kind)) sh->xdelta = (int32_t)c.cpop[cur][best] - sh->mrank;
                }
                PAR_END
                n = nkept + 1;
            }
        }

        // nodes the cache pruned: positions behind the nodes of curr_l (they are part of the layer: arcs point at them and
        // their theta travels upwards in _compute_thresholds)
        int ntot = n;
        if (ncache > 0) {
            PAR_BEGIN
            int lo = tid * q, hi = lo + q < ncl ? lo + q : ncl;
            int cnt3 = 0;
            for (int j = lo; j < hi; ++j) cnt3 += c.cls[lin2cand(j, nprev, capN)] == 3 ? 1 : 0;
            c.tcount[tid] = cnt3;
            PAR_END
            block_exclusive_scan<WS>(c, c.tcount, c.tcount2);
            PAR_BEGIN
            int lo = tid * q, hi = lo + q < ncl ? lo + q : ncl;
            int pos = n + c.tcount[tid];
            for (int j = lo; j < hi; ++j) {
                int cd = lin2cand(j, nprev, capN);
                if (c.cls[cd] == 3) {
                    c.keep[pos] = (uint32_t)cd;
                    c.posmap[cd] = (uint32_t)pos;
                    ++pos;
                }
            }
            if (tid == 0) sh->cache_hits += (uint32_t)ncache;   // (nodes removed by the cache or by dominance)
            PAR_END
            ntot = n + ncache;
        }

        DD1_TICK(3)   // classify, positions, merged node
        // ------------------------------------------------------------ layers.push (clean.rs:678-684)
        if (dynl) {   // room for this layer's nodes in the pool?
            if (node_off + (uint64_t)ntot > c.lpool) {
                PAR_BEGIN
                if (tid == 0) sh->status = ST_ERR_LPOOL;
                PAR_END
                failed = true;
                break;
            }
            lb_cur_ = (size_t)node_off;
            node_off += (uint64_t)ntot;
        } else {
            lb_cur_ = (size_t)L * (size_t)LS;
        }
        PAR_BEGIN
        if (tid == 0) {
            if (c.tmode) c.lntot[L] = ntot;
            if (dynl) c.lbase[L] = (uint64_t)lb_cur;
            c.nlayer[L] = n;
            c.lvar[L] = var;
            c.ldup[2 * L] = sh->dup_from;
            c.ldup[2 * L + 1] = sh->dup_to;
            if (c.lddelta) c.lddelta[L] = sh->xdelta;
        }
        uint32_t* ni = c.ninfo + lb_cur;
        for (int pos = tid; pos < ntot; pos += NT) {
            uint32_t cd = c.keep[pos];
            const uint64_t nkey = LD_U64(&c.ckey[cur][cd]);
            if (c.tmode) {   // the layer is kept: state, value, and the slots the backward passes fill
                const size_t li = lb_cur + pos;
#pragma unroll
                for (int k = 0; k < WS; ++k)
                    c.lstate[(dynl ? lb_cur * WS + (size_t)k * (size_t)ntot : ((size_t)L * WS + k) * (size_t)LS) + pos] = c.cstate[cur][(size_t)k * capC1 + cd];
                c.lval[li] = unbias32((uint32_t)(nkey >> 32));
                c.lrub[li] = INT32_MAX;
                c.lvb[li] = VB_UNMARKED;
                c.lth[li] = pos >= n ? c.cth[cd] : TH_NONE;
            }
            uint32_t arc = key_arc(nkey);
            uint32_t fl = LD_U32(&c.cflags[cur][cd]);
            if (arc != NONE32 && ((uint32_t)nkey & KEY_OK)) fl |= NF_OKPATH;
            uint32_t w;
            if (arc == NONE32) w = NI_NOARC;
            else {
                uint32_t d = arc / (uint32_t)capN;            // decision index of the best arc, parent position
                uint32_t pp = arc - d * (uint32_t)capN;
                w = (pp << c.dbits) | d;
            }
            if (fl & NF_INEXACT) w |= NI_INEXACT;
            if (fl & NF_RELAXED) w |= NI_RELAXED;
            if (fl & NF_OKPATH) w |= NI_OKPATH;
            if (pos >= n) w |= (fl & NF_DOM) ? NI_DOM : NI_CACHE;
            ni[pos] = w;
        }
        // arcs entering this layer, translated to node positions (needed by the backward pass)
        if (((relaxed && lel >= 0) || c.tmode) && L >= 1) {
            uint32_t* at = c.arct + ab_cur;   // (the arcs entering this layer: allocated when the layer above was expanded)
            int32_t* ac = c.arcc + ab_cur;
            for (int j = tid; j < ncl; j += NT) {
                int cd = lin2cand(j, nprev, capN);
                const size_t ai = dynl ? (size_t)j : (size_t)cd;
                uint32_t t = c.ctarget[cd];
                uint32_t out = NONE32;
                if (t != NONE32) {
                    if (c.cls[t] == 1 || c.cls[t] == 3) out = c.posmap[t];
                    else {
                        out = (uint32_t)merged_pos;
                        // Relaxation::relax of a redirected arc (mcp/relax.rs:115-121): + rank(old target) - rank(merged)
                        if (dd_is_vec_w<WS>(c.kind)) ac[ai] += (int32_t)c.cpop[cur][t] - sh->mrank;
                    }
                }
                at[ai] = out;
            }
        }
        PAR_END

        DD1_TICK(5)   // layer bookkeeping (kept layers, arcs)
        // ------------------------------------------------------------ expand (clean.rs:360-370, 728-776)
        const int nxt = cur ^ 1;
        ab_next_ = (size_t)(L + 1) * (size_t)c.fan * (size_t)capN;
        if (dynl) {   // room for the arcs entering layer L + 1: fan per expanded node of this layer
            if (arc_off + (uint64_t)c.fan * (uint64_t)n > c.apool) {
                PAR_BEGIN
                if (tid == 0) sh->status = ST_ERR_APOOL;
                PAR_END
                failed = true;
                break;
            }
            ab_next_ = (size_t)arc_off;
            arc_off += (uint64_t)c.fan * (uint64_t)n;
        }
        hsize = table_size_for(c.fan * (n + 1) + 2, c.table_cap);   // at most fan children per node of this layer
        hmask = hsize - 1;
        PAR_BEGIN
        for (int i = tid; i < hsize; i += NT) c.table[i] = TAB_EMPTY;
        if (tid == 0) sh->nU = 0;
        PAR_END
        PAR_BEGIN
        const bool kp = dd_kind_is<WS>(c.kind, MODEL_KNAPSACK);
        uint64_t adjv[WS];
#pragma unroll
        for (int k = 0; k < WS; ++k) adjv[k] = kp ? 0 : c.adj[(size_t)var * WS + k];
        const int vw = var >> 6;
        const uint64_t vbit = 1ULL << (var & 63);
        const int32_t wv = c.weight[var];
        const uint64_t kpw = kp ? (uint64_t)c.kp_weight[var] : 0;
        int32_t* ac_next = c.arcc + ab_next;   // costs of the arcs entering layer L + 1
        int myarcs = 0, myuniq = 0;
        // TSPTW: a WAVEFRONT per parent, a lane per child (decision = node to visit next, up to n of them).  What belongs to the
        // parent -- its state, rough upper bound and domain -- is the same in all 64 lanes (one scalar computation); the children's
        // transitions and table insertions run side by side.  (One thread per parent walked the n children one after the other:
        // 185-290 kcycles per layer of 20-70 nodes on config C5, nine tenths of the compile.)
        const bool twl = tw_k_of_ws(WS) != 0 && dd_kind_is<WS>(c.kind, MODEL_TSPTW);
        const int lane = twl ? (tid & 63) : 0, lanes = twl ? 64 : 1;
        for (int pos = twl ? DD_WAVE_UNIFORM(tid >> 6) : tid; pos < n; pos += twl ? (NT >> 6) : NT) {
            const uint32_t p = c.keep[pos];
            if (dd_is_vec_w<WS>(c.kind)) {
                // ---- signed-vector models, STREAMED: the parent's words are read, turned into the two children's words and written
                // out one at a time (hashes and sums accumulate alongside); no per-thread state arrays (see dedup_insert_stored)
                const uint64_t pkey = LD_U64(&c.ckey[cur][p]);
                const int32_t val = unbias32((uint32_t)(pkey >> 32));
                const int pop = (int)c.cpop[cur][p];
                const uint32_t pfl = LD_U32(&c.cflags[cur][p]);
                const uint32_t inexact = (pfl & (NF_INEXACT | NF_RELAXED)) ? NF_INEXACT : 0u;
                const uint32_t pok = (!inexact || (!(pfl & NF_RELAXED) && (uint32_t)pkey != NONE32 && ((uint32_t)pkey & KEY_OK))) ? KEY_OK : 0u;
                const int depth = c.depth0 + L;
                const uint64_t* src = c.cstate[cur] + p;
                uint64_t* dst = c.cstate[nxt];
                const int dw = (c.n + 1) / 2;   // index of the depth word
                int32_t rub;
                if (c.kind == MODEL_MAX2SAT) {   // max2sat/model.rs:231-240
                    rub = depth >= c.n ? 0 : pop + c.vest[depth] - c.vr + c.vnk[depth];
                } else {                          // mcp/relax.rs:123-130: sum of |benefit| over the vertices >= depth
                    int32_t r = 0;
#pragma unroll 4
                    for (int k = 0; k < WS; ++k) {
                        const uint64_t sw = src[(size_t)k * capC1];
                        if (2 * k >= depth && 2 * k < c.n) r += iabs32((int32_t)(uint32_t)sw);
                        if (2 * k + 1 >= depth && 2 * k + 1 < c.n) r += iabs32((int32_t)(uint32_t)(sw >> 32));
                    }
                    rub = r + c.vest[depth] - c.vr + c.vnk[depth];
                }
                if (c.tmode) c.lrub[lb_cur + pos] = rub;
                if (rub == RUB_NEG_INF || (int64_t)rub + (int64_t)val <= best_lb) {   // clean.rs:364-365
                    for (int d = 0; d < c.fan; ++d) c.ctarget[(size_t)d * capN + pos] = NONE32;
                    continue;
                }
                const uint32_t cd0 = (uint32_t)pos, cd1 = (uint32_t)(capN + pos);
                int32_t sum0 = 0, sum1 = 0, rank0 = 0, rank1 = 0, sx = 0;
                uint64_t h0 = hash_begin(), h1 = hash_begin();
                const bool two = !(c.kind == MODEL_MCP && depth == 0);   // MCP: the first vertex is fixed on side S (model.rs:60-63)
                if (c.kind == MODEL_MAX2SAT) {
                    // max2sat/model.rs:270-329 (side 0 = T, side 1 = F)
                    const int kx = DD_WAVE_UNIFORM(var), nfree = c.n - depth - 1;
                    const size_t row = (size_t)kx * DD_WAVE_UNIFORM(c.n);
                    DD_SCALAR_PTR(int32_t) m_wtt = dd_scalar_ptr(c.m2_wtt) + row;   // the variable's rows of the four weight tables
                    DD_SCALAR_PTR(int32_t) m_wtf = dd_scalar_ptr(c.m2_wtf) + row;
                    DD_SCALAR_PTR(int32_t) m_wft = dd_scalar_ptr(c.m2_wft) + row;
                    DD_SCALAR_PTR(int32_t) m_wff = dd_scalar_ptr(c.m2_wff) + row;
                    DD_SCALAR_PTR(int32_t) m_rankpos = dd_scalar_ptr(c.m2_rankpos);
                    sum0 = m_wtt[kx];
                    sum1 = m_wff[kx];   // unit clauses (k) / (-k)
                    // (the parent's words arrive eight at a time: a load behind every pair of stores -- which may alias it, as far
                    // as the compiler can tell -- left each of the 16 to 72 words waiting a full memory latency of its own)
#pragma unroll 1
                    for (int kb = 0; kb < WS; kb += VEC_BATCH) {
                    uint64_t pw8[VEC_BATCH];
#pragma unroll
                    for (int i = 0; i < VEC_BATCH; ++i) pw8[i] = kb + i < WS ? src[(size_t)(kb + i) * capC1] : 0;
#pragma unroll
                    for (int i = 0; i < VEC_BATCH; ++i) {
                        const int k = kb + i;
                        if (k >= WS) break;
                        const uint64_t sw = pw8[i];
                        uint64_t w0 = 0, w1 = 0;
#pragma unroll
                        for (int hsel = 0; hsel < 2; ++hsel) {
                            const int v = 2 * k + hsel;
                            if (v >= c.n) continue;
                            const int32_t sl = (int32_t)(uint32_t)(sw >> (32 * hsel));
                            int32_t a = sl, b = sl;                       // child benefit under T / under F
                            if (v == kx) {
                                sx = sl;
                                a = b = 0;
                            } else if (m_rankpos[v] < nfree) {
                                const int32_t wtt = m_wtt[v], wtf = m_wtf[v];
                                const int32_t wft = m_wft[v], wff = m_wff[v];
                                const int32_t ps = sl > 0 ? sl : 0, ns = sl < 0 ? -sl : 0;
                                const int32_t mt = ps + wft < ns + wff ? ps + wft : ns + wff;
                                const int32_t mf = ps + wtt < ns + wtf ? ps + wtt : ns + wtf;
                                sum0 += (wtf + wtt) + mt;
                                sum1 += (wff + wft) + mf;
                                a = sl + wft - wff;
                                b = sl + wtt - wtf;
                            }
                            rank0 += iabs32(a);
                            rank1 += iabs32(b);
                            w0 |= (uint64_t)(uint32_t)a << (32 * hsel);
                            w1 |= (uint64_t)(uint32_t)b << (32 * hsel);
                        }
                        if (k == dw) w0 = w1 = (uint64_t)(depth + 1);   // depth word
                        dst[(size_t)k * capC1 + cd0] = w0;
                        dst[(size_t)k * capC1 + cd1] = w1;
                        h0 = hash_step(h0, w0);
                        h1 = hash_step(h1, w1);
                    }
                    }
                } else {
                    // mcp/model.rs:60-130 (side 0 = S, side 1 = T)
                    const int x = DD_WAVE_UNIFORM(var);
                    DD_SCALAR_PTR(int32_t) wrow = dd_scalar_ptr(c.vgraph) + (size_t)x * DD_WAVE_UNIFORM(c.n);
#pragma unroll 1
                    for (int kb = 0; kb < WS; kb += VEC_BATCH) {
                    uint64_t pw8[VEC_BATCH];
#pragma unroll
                    for (int i = 0; i < VEC_BATCH; ++i) pw8[i] = kb + i < WS ? src[(size_t)(kb + i) * capC1] : 0;
#pragma unroll
                    for (int i = 0; i < VEC_BATCH; ++i) {
                        const int k = kb + i;
                        if (k >= WS) break;
                        const uint64_t sw = pw8[i];
                        uint64_t w0 = 0, w1 = 0;
#pragma unroll
                        for (int hsel = 0; hsel < 2; ++hsel) {
                            const int v = 2 * k + hsel;
                            if (v >= c.n) continue;
                            const int32_t skl = (int32_t)(uint32_t)(sw >> (32 * hsel));
                            if (v == x) sx = skl;
                            if (v < x) continue;
                            const int32_t wkl = wrow[v];
                            const int32_t mn = iabs32(skl) < iabs32(wkl) ? iabs32(skl) : iabs32(wkl);
                            const int64_t prod = (int64_t)skl * (int64_t)wkl;
                            if (prod <= 0) sum0 += mn;
                            if (prod >= 0) sum1 += mn;
                            const int32_t a = skl + wkl, b = skl - wkl;
                            rank0 += iabs32(a);
                            rank1 += iabs32(b);
                            w0 |= (uint64_t)(uint32_t)a << (32 * hsel);
                            w1 |= (uint64_t)(uint32_t)b << (32 * hsel);
                        }
                        if (k == dw) w0 = w1 = (uint64_t)(depth + 1);   // depth word
                        dst[(size_t)k * capC1 + cd0] = w0;
                        if (two) dst[(size_t)k * capC1 + cd1] = w1;
                        h0 = hash_step(h0, w0);
                        h1 = hash_step(h1, w1);
                    }
                    }
                }
                int32_t cost0, cost1;
                if (c.kind == MODEL_MAX2SAT) {
                    cost0 = (sx > 0 ? sx : 0) + sum0;
                    cost1 = (sx < 0 ? -sx : 0) + sum1;
                } else {
                    cost0 = depth == 0 ? 0 : (sx < 0 ? -sx : 0) + sum0;
                    cost1 = depth == 0 ? 0 : (sx > 0 ? sx : 0) + sum1;
                }
                for (int side = 0; side < 2; ++side) {
                    const uint32_t cd = side == 0 ? cd0 : cd1;
                    if (side == 1 && !two) {
                        c.ctarget[cd] = NONE32;
                        break;
                    }
                    const int32_t cost = side == 0 ? cost0 : cost1;
                    const uint64_t mykey = ((uint64_t)bias32(val + cost) << 32) | pok | cd;
                    c.ckey[nxt][cd] = mykey;
                    ac_next[AI(cd, n)] = cost;
                    c.cpop[nxt][cd] = (uint32_t)(side == 0 ? rank0 : rank1);
                    c.cflags[nxt][cd] = inexact;
                }
                FENCE_BLOCK();   // both children are visible to the workgroup before the table publishes them
                for (int side = 0; side < (two ? 2 : 1); ++side) {
                    const uint32_t cd = side == 0 ? cd0 : cd1;
                    const int32_t cost = side == 0 ? cost0 : cost1;
                    const uint64_t mykey = ((uint64_t)bias32(val + cost) << 32) | pok | cd;
                    const uint32_t w = dedup_insert_stored<WS>(c, nxt, cd, hash_end(side == 0 ? h0 : h1), hmask);
                    c.ctarget[cd] = w;
                    ++myarcs;
                    if (w == cd) ++myuniq;
                    else {
                        GLB_MAX_U64(&c.ckey[nxt][w], mykey);            // append_edge_to!: value >= value_top
                        if (inexact) GLB_OR_U32(&c.cflags[nxt][w], NF_INEXACT);
                    }
                }
                continue;
            }
            uint64_t s[WS];
#pragma unroll
            for (int k = 0; k < WS; ++k) s[k] = c.cstate[cur][(size_t)k * capC1 + p];
            const uint64_t pkey = LD_U64(&c.ckey[cur][p]);
            const int32_t val = unbias32((uint32_t)(pkey >> 32));
            const int pop = (int)c.cpop[cur][p];
            const uint32_t pfl = LD_U32(&c.cflags[cur][p]);
            const uint32_t inexact = (pfl & (NF_INEXACT | NF_RELAXED)) ? NF_INEXACT : 0u;
            // does this parent have an exact best path (see KEY_OK) ?
            const uint32_t pok = (!inexact || (!(pfl & NF_RELAXED) && (uint32_t)pkey != NONE32 && ((uint32_t)pkey & KEY_OK))) ? KEY_OK : 0u;
            const int32_t rub = rub_of<WS>(c, s, pop, c.depth0 + L);
            if (c.tmode) c.lrub[lb_cur + pos] = rub;   // node.rub (clean.rs:363), read again by _compute_thresholds
            if (rub == RUB_NEG_INF || (int64_t)rub + (int64_t)val <= best_lb) {  // clean.rs:364-365: not expanded (isize::MIN + value saturates)
                if (dd_kind_is<WS>(c.kind, MODEL_MISP)) add_bits<WS>(c.cnt, s, -1);
                for (int d = lane; d < c.fan; d += lanes) c.ctarget[(size_t)d * capN + pos] = NONE32;
                continue;
            }
            if (tw_k_of_ws(WS) != 0 && dd_kind_is<WS>(c.kind, MODEL_TSPTW)) {
                // examples/tsptw/model.rs:65-139: one child per node the salesman may visit next; decision index = node
                uint64_t dom[TWK];
                if constexpr (tw_k_of_ws(WS) != 0) tw_domain<TWK>(c.tw, s, dom);
                for (int j = lane; j < c.fan; j += 64) {
                    const uint32_t cd = (uint32_t)((size_t)j * capN + pos);
                    uint64_t dw = dom[0];
#pragma unroll
                    for (int q = 1; q < TWK; ++q) dw = (j >> 6) == q ? dom[q] : dw;
                    if (!((dw >> (j & 63)) & 1ULL)) {
                        c.ctarget[cd] = NONE32;
                        continue;
                    }
                    uint64_t y[WS];
#pragma unroll
                    for (int k = 0; k < WS; ++k) y[k] = 0;
                    int32_t cost = 0;
                    if constexpr (tw_k_of_ws(WS) != 0) tw_transition<TWK>(c.tw, s, j, y, &cost);
#pragma unroll
                    for (int k = 0; k < WS; ++k) c.cstate[nxt][(size_t)k * capC1 + cd] = y[k];
                    const uint64_t mykey = ((uint64_t)bias32(val + cost) << 32) | pok | cd;
                    c.ckey[nxt][cd] = mykey;
                    ac_next[AI(cd, n)] = cost;
                    c.cpop[nxt][cd] = 0;                       // TsptwRanking compares depths: equal within a layer
                    c.cflags[nxt][cd] = inexact;
                    FENCE_BLOCK();
                    const uint32_t w = dedup_insert<WS>(c, nxt, cd, y, hmask);
                    c.ctarget[cd] = w;
                    ++myarcs;
                    if (w == cd) ++myuniq;
                    else {
                        GLB_MAX_U64(&c.ckey[nxt][w], mykey);            // append_edge_to!: value >= value_top
                        if (inexact) GLB_OR_U32(&c.cflags[nxt][w], NF_INEXACT);
                    }
                }
                continue;
            }
            bool hasv = false;   // MISP: the vertex is in the state; knapsack: the item fits (main.rs:93-99)
            if (kp) hasv = s[0] >= kpw;
            else {
#pragma unroll
                for (int k = 0; k < WS; ++k)
                    if (k == vw) hasv = (s[k] & vbit) != 0;
            }
            // ---- decision NO (main.rs:77-85 with value == NO): state minus the variable; knapsack LEAVE_IT_OUT: same state
            if (hasv && !kp) {
#pragma unroll
                for (int k = 0; k < WS; ++k)
                    if (k == vw) s[k] &= ~vbit;
                LDS_ADD_I32(&c.cnt[var], -1);
            }
            if (kp) {   // KnapsackState = (capacity, depth) (main.rs:44-50): both children are one level deeper
#pragma unroll
                for (int k = 0; k < WS; ++k)
                    if (k == 1) s[k] += 1;
            }
            {
                const uint32_t cd = (uint32_t)pos;
#pragma unroll
                for (int k = 0; k < WS; ++k) c.cstate[nxt][(size_t)k * capC1 + cd] = s[k];
                const uint64_t mykey = ((uint64_t)bias32(val) << 32) | pok | cd;
                c.ckey[nxt][cd] = mykey;
                ac_next[AI(cd, n)] = 0;                                   // transition_cost of NO / LEAVE_IT_OUT
                c.cpop[nxt][cd] = (uint32_t)(pop - ((hasv && !kp) ? 1 : 0));
                c.cflags[nxt][cd] = inexact;
                FENCE_BLOCK();
                const uint32_t w = dedup_insert<WS>(c, nxt, cd, s, hmask);
                c.ctarget[cd] = w;
                ++myarcs;
                if (w == cd) ++myuniq;
                else {
                    GLB_MAX_U64(&c.ckey[nxt][w], mykey);            // append_edge_to!: value >= value_top
                    if (inexact) GLB_OR_U32(&c.cflags[nxt][w], NF_INEXACT);
                    if (dd_kind_is<WS>(c.kind, MODEL_MISP)) add_bits<WS>(c.cnt, s, -1);                      // duplicate: not a new member of next_l
                }
            }
            // ---- decision YES (only when the variable is in the state, main.rs:95-102)
            if (hasv) {
                uint64_t y[WS];
                int ypop = 0;
#pragma unroll
                for (int k = 0; k < WS; ++k) {
                    y[k] = s[k] & adjv[k];
                    ypop += dd_popc(y[k]);
                }
                if (kp) {   // TAKE_IT (main.rs:106-113): the capacity shrinks by the item's weight
#pragma unroll
                    for (int k = 0; k < WS; ++k) y[k] = k == 0 ? s[0] - kpw : s[k];
                    ypop = 0;
                }
                const uint32_t cd = (uint32_t)(capN + pos);
#pragma unroll
                for (int k = 0; k < WS; ++k) c.cstate[nxt][(size_t)k * capC1 + cd] = y[k];
                const uint64_t mykey = ((uint64_t)bias32(val + wv) << 32) | pok | cd;
                c.ckey[nxt][cd] = mykey;
                ac_next[AI(cd, n)] = wv;                                  // transition_cost of YES / TAKE_IT
                c.cpop[nxt][cd] = (uint32_t)ypop;
                c.cflags[nxt][cd] = inexact;
                FENCE_BLOCK();
                const uint32_t w = dedup_insert<WS>(c, nxt, cd, y, hmask);
                c.ctarget[cd] = w;
                ++myarcs;
                if (w == cd) {
                    ++myuniq;
                    if (dd_kind_is<WS>(c.kind, MODEL_MISP)) add_bits<WS>(c.cnt, y, +1);
                } else {
                    GLB_MAX_U64(&c.ckey[nxt][w], mykey);
                    if (inexact) GLB_OR_U32(&c.cflags[nxt][w], NF_INEXACT);
                }
            } else {
                c.ctarget[capN + pos] = NONE32;
            }
        }
        if (myarcs) LDS_ADD_U64(&sh->arcs, (uint64_t)myarcs);
        if (myuniq) LDS_ADD_I32(&sh->nU, myuniq);
        if (tid == 0) {
            sh->nodes += (uint64_t)n;
            if (n > sh->maxn) sh->maxn = n;
            if (dynl) c.abase[L + 1] = (uint64_t)ab_next;
        }
        PAR_END
#if defined(DDO_HOST_EMULATION)
        if (getenv("DD_TRACE")) std::printf("E1 L=%d var=%d n=%d arcs=%llu nU_next=%d squash=%d\n", L, var, n, (unsigned long long)sh->arcs, sh->nU, (int)squash);
#endif

        DD1_TICK(2)   // expand: rough upper bounds, transitions, dedup table
        cur = nxt;
        nprev = n;
        ab_cur_ = ab_next_;
        L += 1;
    }

    DD1_TICK(4)
    // ==================================================================== _finalize (clean.rs:407-414)
    int n_layers = L;
    int nT = 0;          // nodes of the terminal layer
    const int nU = sh->nU;
    const int ncl = c.fan * nprev;
    const int q = (ncl + NT - 1) / NT;
    if (!failed && nU > (c.tmode ? LS : capN)) {
        PAR_BEGIN
        if (tid == 0) sh->status = ST_ERR_CAPACITY;
        PAR_END
        failed = true;
    }
    if (!failed && nU > 0) {
        // _finalize_layers: the terminal layer is whatever is in next_l, never squashed (clean.rs:608-618)
        PAR_BEGIN
        int lo = tid * q, hi = lo + q < ncl ? lo + q : ncl;
        int kept = 0;
        for (int j = lo; j < hi; ++j) {
            int cd = lin2cand(j, nprev, capN);
            uint8_t cl = c.ctarget[cd] == (uint32_t)cd ? 1 : 0;
            c.cls[cd] = cl;
            kept += cl;
        }
        c.tcount[tid] = kept;
        if (tid == 0) {
            sh->bestKey = 0;
            sh->bestExactKey = 0;
        }
        PAR_END
        block_exclusive_scan<WS>(c, c.tcount, c.tcount2);
        nT = sh->scan_total;
        PAR_BEGIN
        int lo = tid * q, hi = lo + q < ncl ? lo + q : ncl;
        int pos = c.tcount[tid];
        for (int j = lo; j < hi; ++j) {
            int cd = lin2cand(j, nprev, capN);
            if (c.cls[cd] == 1) {
                c.keep[pos] = (uint32_t)cd;
                c.posmap[cd] = (uint32_t)pos;
                ++pos;
            }
        }
        PAR_END
        bool pool_full = false;
        if (dynl) {
            pool_full = node_off + (uint64_t)nT > c.lpool;
            lb_cur_ = (size_t)node_off;
            node_off += (uint64_t)nT;
        } else {
            lb_cur_ = (size_t)L * (size_t)LS;
        }
        if (pool_full) {
            PAR_BEGIN
            if (tid == 0) sh->status = ST_ERR_LPOOL;
            PAR_END
            nT = 0;
            failed = true;   // as the overflow checks inside the layer loop: no backward pass over arcs this layer never wrote
        }
        PAR_BEGIN
        if (tid == 0) {
            c.nlayer[L] = nT;
            if (c.tmode) c.lntot[L] = nT;
            if (dynl) c.lbase[L] = (uint64_t)lb_cur;
            c.lvar[L] = -1;
            c.ldup[2 * L] = -1;
            c.ldup[2 * L + 1] = -1;
        }
        uint32_t* ni = c.ninfo + lb_cur;
        for (int pos = tid; pos < nT; pos += NT) {
            uint32_t cd = c.keep[pos];
            uint64_t key = LD_U64(&c.ckey[cur][cd]);
            if (c.tmode) {
                const size_t li = lb_cur + pos;
#pragma unroll
                for (int k = 0; k < WS; ++k)
                    c.lstate[(dynl ? lb_cur * WS + (size_t)k * (size_t)nT : ((size_t)L * WS + k) * (size_t)LS) + pos] = c.cstate[cur][(size_t)k * capC1 + cd];
                c.lval[li] = unbias32((uint32_t)(key >> 32));
                c.lrub[li] = INT32_MAX;      // the terminal layer is never bounded (clean.rs:360 does not reach it)
                c.lvb[li] = VB_UNMARKED;
                c.lth[li] = TH_NONE;
            }
            uint32_t arc = key_arc(key);
            uint32_t fl = LD_U32(&c.cflags[cur][cd]);
            const bool okp = arc != NONE32 && ((uint32_t)key & KEY_OK) && !(fl & NF_RELAXED);
            uint32_t w;
            if (arc == NONE32) w = NI_NOARC;
            else {
                uint32_t d = arc / (uint32_t)capN;
                w = ((arc - d * (uint32_t)capN) << c.dbits) | d;
            }
            if (fl & NF_INEXACT) w |= NI_INEXACT;
            if (fl & NF_RELAXED) w |= NI_RELAXED;
            if (okp) w |= NI_OKPATH;
            ni[pos] = w;
            // _find_best_node (clean.rs:620-632): max value_top; among equal values a node with an exact best path first
            // (order-independent EBPO, see KEY_OK), then the position
            const bool nok = !(fl & (NF_INEXACT | NF_RELAXED)) || okp;
            uint64_t bk = (key & 0xFFFFFFFF00000000ULL) | (nok ? KEY_OK : 0u) | (uint32_t)pos;
            LDS_MAX_U64(&sh->bestKey, bk + 1);  // +1 so that 0 means "none"
            if (!(fl & (NF_INEXACT | NF_RELAXED))) LDS_MAX_U64(&sh->bestExactKey, bk + 1);
        }
        if (((relaxed && lel >= 0) || c.tmode) && L >= 1) {
            uint32_t* at = c.arct + ab_cur;
            for (int j = tid; j < (pool_full ? 0 : ncl); j += NT) {
                int cd = lin2cand(j, nprev, capN);
                uint32_t t = c.ctarget[cd];
                at[dynl ? (size_t)j : (size_t)cd] = t != NONE32 ? c.posmap[t] : NONE32;
            }
        }
        PAR_END
        n_layers = L + 1;
    } else if (!failed) {
        PAR_BEGIN
        if (tid == 0) {
            sh->bestKey = 0;
            sh->bestExactKey = 0;
        }
        PAR_END
    }

    DD1_TICK(6)   // _finalize: terminal layer, best node
    // ---------------------------------------------------------------- results
    const bool is_exact = lel < 0;                                     // clean.rs:635
    const bool has_best = !failed && sh->bestKey != 0;
    int best_pos = -1, best_value = 0;
    if (has_best) {
        uint64_t bk = sh->bestKey - 1;
        best_pos = (int)((uint32_t)bk & ~KEY_OK);
        best_value = unbias32((uint32_t)(bk >> 32));
    }
    bool has_best_exact = !failed && sh->bestExactKey != 0;
    int exact_pos = -1, exact_value = 0;
    if (has_best_exact) {
        uint64_t bk = sh->bestExactKey - 1;
        exact_pos = (int)((uint32_t)bk & ~KEY_OK);
        exact_value = unbias32((uint32_t)(bk >> 32));
    }
    // _has_exact_best_path (clean.rs:643-655), EBPO
    bool ebpo = false;
    if (relaxed && !failed) {
        PAR_BEGIN
        if (tid == 0) {
            int res_e = 1;
            if (has_best) {
                // _has_exact_best_path (clean.rs:643-655) walks the best arcs up to the first exact (-> true) or relaxed
                // (-> false) node; the best arc of every node already prefers a parent for which that walk succeeds
                // (KEY_OK), so the answer for the best terminal node is in its own word
                const uint32_t w = c.ninfo[LB((n_layers - 1)) + best_pos];
                if (!(w & (NI_INEXACT | NI_RELAXED))) res_e = 1;
                else if (w & NI_RELAXED) res_e = 0;
                else res_e = (w & NI_OKPATH) ? 1 : 0;
            }
            sh->sel_digit = res_e;
        }
        PAR_END
        ebpo = sh->sel_digit != 0;
        if (ebpo && has_best) {  // clean.rs:638-640
            has_best_exact = true;
            exact_pos = best_pos;
            exact_value = best_value;
        } else if (ebpo && !has_best) {
            // best_node == None => has_exact_best_path(None) == true, best_exact_node stays None
        }
    }

    // ---------------------------------------------------------------- local bounds (clean.rs:448-475)
    const bool want_cutset = relaxed && !failed && lel >= 0 && lel < n_layers && has_best;
    int32_t* vbA = (int32_t*)c.table;
    int32_t* vbB = vbA + capN;
    // ================================================================ kept layers: local bounds of EVERY layer, cut-set flags,
    // thresholds and cache updates (clean.rs:448-606); the cut-set is emitted from the kept layers further down
    if (c.tmode && !failed) {
        const int T = n_layers - 1;
        const int lel_eff = lel < 0 ? n_layers : lel;                       // clean.rs:548-550
        const bool th_on = relaxed || is_exact;                              // clean.rs:479, 551
        if (relaxed && lel >= 0 && lel < n_layers) {                          // _compute_local_bounds (clean.rs:448-475)
            PAR_BEGIN
            for (int pos = tid; pos < c.lntot[T]; pos += NT) c.lvb[LB(T) + pos] = 0;
            PAR_END
            for (int Lc = T; Lc >= 1; --Lc) {
                const int nP = c.nlayer[Lc - 1];
                PAR_BEGIN
                const uint32_t* at = c.arct + AB(Lc);
                const int32_t* ac = c.arcc + AB(Lc);
                const int dfrom = c.ldup[2 * Lc], dto = c.ldup[2 * Lc + 1];
                for (int j = tid; j < c.fan * nP; j += NT) {
                    const int d = j / nP;
                    const int pp = j - d * nP;
                    const uint32_t t = at[dynl ? j : d * capN + pp];
                    if (t == NONE32) continue;
                    const int32_t cost = ac[dynl ? j : d * capN + pp];
                    const int32_t v = LD_I32(&c.lvb[LB(Lc) + t]);
                    if (v != VB_UNMARKED) GLB_MAX_I32(&c.lvb[LB((Lc - 1)) + pp], v + cost);
                    if ((int)t == dfrom) {   // the arcs of the node a recycled merge re-added also go to the merged node
                        const int32_t v2 = LD_I32(&c.lvb[LB(Lc) + dto]);
                        if (v2 != VB_UNMARKED) GLB_MAX_I32(&c.lvb[LB((Lc - 1)) + pp], v2 + cost + (c.lddelta ? c.lddelta[Lc] : 0));
                    }
                }
                PAR_END
            }
        }
        if (th_on) {
            // ---- _finalize_cutset (clean.rs:547-606)
            PAR_BEGIN
            if (!frontier) {
                if (lel_eff < n_layers)
                    for (int pos = tid; pos < c.lntot[lel_eff]; pos += NT) c.ninfo[LB(lel_eff) + pos] |= NI_CUTSET;
            } else {
                for (int Lc = 1; Lc <= T; ++Lc) {
                    const int nP = c.nlayer[Lc - 1];
                    const uint32_t* at = c.arct + AB(Lc);
                    const int dfrom = c.ldup[2 * Lc];
                    for (int j = tid; j < c.fan * nP; j += NT) {
                        const int d = j / nP;
                        const int pp = j - d * nP;
                        const uint32_t t = at[dynl ? j : d * capN + pp];
                        if (t == NONE32) continue;
                        const bool child_inexact = (c.ninfo[LB(Lc) + t] & (NI_INEXACT | NI_RELAXED)) != 0 || (int)t == dfrom;
                        if (!child_inexact) continue;
                        const uint32_t pw = LD_U32(&c.ninfo[LB((Lc - 1)) + pp]);
                        if (!(pw & (NI_INEXACT | NI_RELAXED | NI_CUTSET))) GLB_OR_U32(&c.ninfo[LB((Lc - 1)) + pp], NI_CUTSET);
                    }
                }
            }
            PAR_END
            // (_compute_thresholds and the cache updates follow the reservation of the output arena, see below)
        }
    }
    if (want_cutset && !c.tmode) {
        const int T = n_layers - 1;
        PAR_BEGIN
        for (int pos = tid; pos < c.nlayer[T]; pos += NT) vbA[pos] = 0;
        PAR_END
        for (int Lc = T; Lc > lel; --Lc) {
            const int nP = c.nlayer[Lc - 1];
            PAR_BEGIN
            for (int pos = tid; pos < nP; pos += NT) vbB[pos] = VB_UNMARKED;
            PAR_END
            PAR_BEGIN
            const uint32_t* at = c.arct + AB(Lc);
            const int32_t* ac = c.arcc + AB(Lc);
            const int dfrom = c.ldup[2 * Lc], dto = c.ldup[2 * Lc + 1];
            for (int j = tid; j < c.fan * nP; j += NT) {
                const int d = j / nP;
                const int pp = j - d * nP;
                const uint32_t t = at[dynl ? j : d * capN + pp];
                if (t == NONE32) continue;
                const int32_t cost = ac[dynl ? j : d * capN + pp];
                int32_t v = vbA[t];
                if (v != VB_UNMARKED) LDS_MAX_I32(&vbB[pp], v + cost);
                if ((int)t == dfrom) {  // arcs of the re-added node were also redirected (clean.rs:851-866)
                    int32_t v2 = vbA[dto];
                    if (v2 != VB_UNMARKED) LDS_MAX_I32(&vbB[pp], v2 + cost + (c.lddelta ? c.lddelta[Lc] : 0));
                }
            }
            PAR_END
            int32_t* t = vbA;
            vbA = vbB;
            vbB = t;
        }
    }

    // ---------------------------------------------------------------- cut-set size (clean.rs:417-445)
    const int ncs_layer = want_cutset ? c.nlayer[lel] : 0;
    const bool filter = (in.flags & IN_FILTER_CUTSET) != 0;
    PAR_BEGIN
    if (tid == 0) {
        sh->ncut = 0;
        sh->ncut2 = 0;
    }
    PAR_END
    // kept layers: the cut-set nodes carry NI_CUTSET -- the last exact layer, or (frontier) exact nodes with an inexact child
    // anywhere in the DD (clean.rs:417-445 emits the MARKED ones)
    const int cs_first = frontier ? 0 : lel, cs_last = frontier ? n_layers - 2 : lel;
    if (want_cutset && c.tmode) {
        PAR_BEGIN
        int mine = 0;
        for (int Lc = cs_first; Lc <= cs_last; ++Lc)
            for (int pos = tid; pos < c.lntot[Lc]; pos += NT) {
                const size_t li = LB(Lc) + pos;
                if (!(LD_U32(&c.ninfo[li]) & NI_CUTSET)) continue;
                const int32_t vb = LD_I32(&c.lvb[li]);
                if (vb == VB_UNMARKED) continue;
                if (filter) {
                    const int64_t v = c.lval[li];
                    int64_t ub = c.lrub[li] == INT32_MAX ? INT64_MAX : v + c.lrub[li];
                    if (v + vb < ub) ub = v + vb;
                    if (best_value < ub) ub = best_value;
                    if (ub <= best_lb) continue;
                }
                ++mine;
            }
        if (mine) LDS_ADD_I32(&sh->ncut, mine);
        PAR_END
    }
    if (want_cutset && !c.tmode) {
        PAR_BEGIN
        int mine = 0;
        for (int pos = tid; pos < ncs_layer; pos += NT) {
            int32_t vb = vbA[pos];
            if (vb == VB_UNMARKED) continue;
            if (filter) {
                uint64_t s[WS];
#pragma unroll
                for (int k = 0; k < WS; ++k) s[k] = c.cs_state[(size_t)k * capN + pos];
                int64_t v = c.cs_value[pos];
                int64_t ub = v + rub_of<WS>(c, s, (int)c.cs_pop[pos], c.depth0 + lel);
                if (v + vb < ub) ub = v + vb;
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
    const bool same = emit_best && emit_exact && exact_pos == best_pos;
    const int path_len = n_layers > 0 ? n_layers - 1 : 0;
    const int best_len = emit_best ? path_len : 0;
    const int exact_len = (emit_exact && !same) ? path_len : 0;
    const int cs_path_len = frontier ? (n_layers > 1 ? n_layers - 1 : 0) : (lel > 0 ? lel : 0);

    // arena layout (all 8-byte aligned)
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
    const uint64_t cs_path_off = off;
    off += ((uint64_t)ncut * cs_path_len * 4 + 7) & ~7ULL;
    const uint64_t cs_depth_off = off;
    if (c.tmode) off += ((uint64_t)ncut * 4 + 7) & ~7ULL;
    const uint64_t total = off;

    PAR_BEGIN
    if (tid == 0) {
        unsigned long long a = total ? GLB_ADD_U64(c.arena_head, (unsigned long long)total) : 0ULL;
        sh->arena_off = a;
        if (a + total > c.arena_cap) sh->status = ST_ERR_ARENA;
    }
    PAR_END
    const bool arena_ok = sh->status == ST_OK || sh->status == ST_CUTOFF;
    uint8_t* base = c.arena + sh->arena_off;

    // A compile whose output does not fit the shared arena is compiled AGAIN by the host (on its own, arena enlarged): it must
    // leave no trace in the cache, or the second run would be pruned by the thresholds of the first.  So the thresholds -- and
    // with them every cache update -- are computed only once the arena space of this DD is reserved.
    if (c.tmode && !failed && (relaxed || is_exact) && arena_ok) {
        const int T = n_layers - 1;
        const int lel_eff = lel < 0 ? n_layers : lel;                       // clean.rs:548-550
        {
            // ---- _compute_thresholds (clean.rs:478-532) + _maybe_update_cache (:534-545)
            int64_t bk64 = best_lb;
            if (has_best_exact && (int64_t)exact_value > bk64) bk64 = exact_value;
            const int32_t bk = bk64 < -(1 << 30) ? -(1 << 30) : (int32_t)bk64;   // values are far above: same comparisons
            const bool bk_min = bk64 <= -((int64_t)1 << 39);                      // no bound known: best_known == isize::MIN
            if (has_best_exact && T >= 0 && nT > 0) {
                PAR_BEGIN
                for (int pos = tid; pos < nT; pos += NT) {
                    const bool nex = !(c.ninfo[LB(T) + pos] & (NI_INEXACT | NI_RELAXED));
                    if ((!frontier && is_exact) || (frontier && nex)) c.lth[LB(T) + pos] = bk;
                }
                PAR_END
            }
            for (int Lc = T; Lc >= 0; --Lc) {
                PAR_BEGIN
                for (int pos = tid; pos < c.lntot[Lc]; pos += NT) {
                    const size_t li = LB(Lc) + pos;
                    const uint32_t w = LD_U32(&c.ninfo[li]);
                    if (w & NI_CACHE) continue;                       // its theta is the cached threshold: only propagated
                    const bool nex = !(w & (NI_INEXACT | NI_RELAXED));
                    const int32_t val = c.lval[li], rub = c.lrub[li];
                    int32_t th = LD_I32(&c.lth[li]);
                    if (rub == RUB_NEG_INF) {
                        th = TH_INF;                                   // value (+sat) isize::MIN <= anything; best_known (-sat) MIN is huge
                    } else if (rub != INT32_MAX && !bk_min && (int64_t)val + rub <= (int64_t)bk) {
                        th = bk - rub;
                    } else if (w & NI_CUTSET) {
                        const int32_t vb = LD_I32(&c.lvb[li]);
                        // value_bot of an unmarked node is isize::MIN: value_top (+sat) MIN <= best_known unless there is no
                        // bound at all (best_known == isize::MIN) and value_top > 0; best_known (-sat) MIN is then 0, else huge
                        const bool locb_le = vb == VB_UNMARKED ? (!bk_min || val <= 0) : (!bk_min && (int64_t)val + vb <= (int64_t)bk);
                        if (locb_le) {
                            const int32_t cand = vb == VB_UNMARKED ? (bk_min ? 0 : TH_INF) : bk - vb;
                            const int32_t old = th == TH_NONE ? TH_INF : th;
                            th = cand < old ? cand : old;
                        } else {
                            th = val;
                        }
                    } else if (nex && th == TH_NONE) {
                        th = TH_INF;                                   // large theta for dangling nodes
                    }
                    c.lth[li] = th;
                    const bool above = frontier ? nex : Lc <= lel_eff;
                    if (use_cache && th != TH_NONE && above) {
                        uint64_t st[WS];
#pragma unroll
                        for (int k = 0; k < WS; ++k) st[k] = c.lstate[LSX(Lc, k, pos)];
                        cache_update<WS>(c, st, c.depth0 + Lc, th_pack(th, !(w & NI_CUTSET)));
                    }
                }
                PAR_END
                if (Lc == 0) break;
                const int nP = c.nlayer[Lc - 1];
                PAR_BEGIN
                const uint32_t* at = c.arct + AB(Lc);
                const int32_t* ac = c.arcc + AB(Lc);
                const int dfrom = c.ldup[2 * Lc], dto = c.ldup[2 * Lc + 1];
                for (int j = tid; j < c.fan * nP; j += NT) {
                    const int d = j / nP;
                    const int pp = j - d * nP;
                    const uint32_t t = at[dynl ? j : d * capN + pp];
                    if (t == NONE32) continue;
                    const int32_t cost = ac[dynl ? j : d * capN + pp];
                    const int32_t th = LD_I32(&c.lth[LB(Lc) + t]);
                    if (th != TH_NONE) GLB_MIN_I32(&c.lth[LB((Lc - 1)) + pp], th_sub(th, cost));
                    if ((int)t == dfrom) {
                        const int32_t th2 = LD_I32(&c.lth[LB(Lc) + dto]);
                        if (th2 != TH_NONE) GLB_MIN_I32(&c.lth[LB((Lc - 1)) + pp], th_sub(th2, cost + (c.lddelta ? c.lddelta[Lc] : 0)));
                    }
                }
                PAR_END
            }
        }
    }

    if (arena_ok && !failed) {
        PAR_BEGIN
        // best paths (clean.rs:329-343): decisions from the terminal node back to the DD root
        if (tid == 0 && best_len) {
            uint32_t* out = (uint32_t*)(base + path_off);
            int p = best_pos;
            for (int Lc = n_layers - 1, i = 0; Lc >= 1; --Lc, ++i) {
                uint32_t w = c.ninfo[LB(Lc) + p];
                uint32_t arc = w & NI_ARC_MASK;
                out[i] = ((uint32_t)c.lvar[Lc - 1] << c.dbits) | (arc & dmask);
                p = (int)(arc >> c.dbits);
            }
        }
        if (tid == 64 % NT && exact_len) {
            uint32_t* out = (uint32_t*)(base + exact_off);
            int p = exact_pos;
            for (int Lc = n_layers - 1, i = 0; Lc >= 1; --Lc, ++i) {
                uint32_t w = c.ninfo[LB(Lc) + p];
                uint32_t arc = w & NI_ARC_MASK;
                out[i] = ((uint32_t)c.lvar[Lc - 1] << c.dbits) | (arc & dmask);
                p = (int)(arc >> c.dbits);
            }
        }
        // cut-set nodes (clean.rs:421-443)
        if (want_cutset && ncut && c.tmode) {
            uint64_t* o_state = (uint64_t*)(base + cs_state_off);
            int32_t* o_value = (int32_t*)(base + cs_value_off);
            int32_t* o_ub = (int32_t*)(base + cs_ub_off);
            uint32_t* o_path = (uint32_t*)(base + cs_path_off);
            int32_t* o_depth = (int32_t*)(base + cs_depth_off);
            for (int Lc = cs_first; Lc <= cs_last; ++Lc)
                for (int pos = tid; pos < c.lntot[Lc]; pos += NT) {
                    const size_t li = LB(Lc) + pos;
                    if (!(LD_U32(&c.ninfo[li]) & NI_CUTSET)) continue;
                    const int32_t vb = LD_I32(&c.lvb[li]);
                    if (vb == VB_UNMARKED) continue;
                    const int64_t v = c.lval[li];
                    int64_t ub = c.lrub[li] == INT32_MAX ? INT64_MAX : v + c.lrub[li];
                    if (v + vb < ub) ub = v + vb;
                    if (best_value < ub) ub = best_value;
                    if (filter && ub <= best_lb) continue;
                    const int idx = LDS_ADD_I32(&sh->ncut2, 1);
#pragma unroll
                    for (int k = 0; k < WS; ++k) o_state[(size_t)idx * WS + k] = c.lstate[LSX(Lc, k, pos)];
                    o_value[idx] = (int32_t)v;
                    o_ub[idx] = (int32_t)ub;
                    o_depth[idx] = Lc;
                    int p = pos;
                    for (int Lw = Lc, i = 0; Lw >= 1; --Lw, ++i) {
                        const uint32_t arc = c.ninfo[LB(Lw) + p] & NI_ARC_MASK;
                        o_path[(size_t)idx * cs_path_len + i] = ((uint32_t)c.lvar[Lw - 1] << c.dbits) | (arc & dmask);
                        p = (int)(arc >> c.dbits);
                    }
                }
        } else if (want_cutset && ncut) {
            uint64_t* o_state = (uint64_t*)(base + cs_state_off);
            int32_t* o_value = (int32_t*)(base + cs_value_off);
            int32_t* o_ub = (int32_t*)(base + cs_ub_off);
            uint32_t* o_path = (uint32_t*)(base + cs_path_off);
            for (int pos = tid; pos < ncs_layer; pos += NT) {
                int32_t vb = vbA[pos];
                if (vb == VB_UNMARKED) continue;
                uint64_t s[WS];
#pragma unroll
                for (int k = 0; k < WS; ++k) s[k] = c.cs_state[(size_t)k * capN + pos];
                int64_t v = c.cs_value[pos];
                int64_t ub = v + rub_of<WS>(c, s, (int)c.cs_pop[pos], c.depth0 + lel);
                if (v + vb < ub) ub = v + vb;
                if (best_value < ub) ub = best_value;
                if (filter && ub <= best_lb) continue;
                int idx = LDS_ADD_I32(&sh->ncut2, 1);
#pragma unroll
                for (int k = 0; k < WS; ++k) o_state[(size_t)idx * WS + k] = s[k];
                o_value[idx] = (int32_t)v;
                o_ub[idx] = (int32_t)ub;
                int p = pos;
                for (int Lc = lel, i = 0; Lc >= 1; --Lc, ++i) {
                    uint32_t w = c.ninfo[LB(Lc) + p];
                    uint32_t arc = w & NI_ARC_MASK;
                    o_path[(size_t)idx * cs_path_len + i] = ((uint32_t)c.lvar[Lc - 1] << c.dbits) | (arc & dmask);
                    p = (int)(arc >> c.dbits);
                }
            }
        }
        PAR_END
    }

    PAR_BEGIN
    if (tid == 0) {
        DDResult r;
        r.status = sh->status;
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
        for (int k = 0; k < 32; ++k) r.phase_clk[k] = k < 8 && c.clocks ? sh->clk[k] : 0;
        r.pool_off = NO_POOL_SRC;
        r.cs_depth_off = c.tmode ? cs_depth_off : 0;
        r.cs_path_stride = cs_path_len;
        r.cs_lvar_off = 0;   // (IN_PATH_BITS is the in-place engine's)
        r.cache_hits = sh->cache_hits;
        *res = r;
    }
    PAR_END
}

#undef lb_cur
#undef ab_cur
#undef ab_next
#undef LB
#undef LSX
#undef AB
#undef AI

/// Restricted, then -- when that was not exact -- relaxed: the device half of
/// ParallelSolver::process_one_node (parallel.rs:391-437).
template <int WS>
DDO_DEV void run_work_item(DDCtx<WS>& c, const DDInput& in, DDResult* res2) {
    DD_TID_SETUP(c)
    (void)NT;
    c.depth0 = in.depth;
    if ((in.flags & IN_MUST_EXPLORE) && c.tmode && c.cache_cap) {
        // the solver's pop (parallel.rs:537-549): Cache::must_explore (cache.rs:32-39), then update_threshold(.., explored)
        PAR_BEGIN
        if (tid == 0) {
            int64_t packed = 0;
            bool explore = true;
            if (cache_get<WS>(c, in.state, in.depth, &packed)) {
                const int32_t tv = th_value(packed);
                explore = tv != TH_INF && (in.value > tv || (in.value == tv && !th_explored(packed)));
            }
            if (explore && (in.flags & IN_MARK_EXPLORED)) cache_update<WS>(c, in.state, in.depth, th_pack(in.value, true));
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
    if (in.flags & IN_FUSED) {
        run_dd<WS>(c, in, CT_RESTRICTED, in.best_lb, &res2[0]);
        // all threads read the restricted result through shared memory state written by thread 0
        PAR_BEGIN
        if (tid == 0) {
            c.sh->sel_above = (res2[0].status == ST_OK && !res2[0].is_exact) ? 1 : 0;
            c.sh->sel_bucket = res2[0].has_best_exact ? 1 : 0;
            c.sh->sel_digit = res2[0].best_exact_value;
        }
        PAR_END
        const bool go = c.sh->sel_above != 0;
        int64_t lb = in.best_lb;
        if (c.sh->sel_bucket && (int64_t)c.sh->sel_digit > lb) lb = c.sh->sel_digit;  // maybe_update_best
        if (go) run_dd<WS>(c, in, CT_RELAXED, lb, &res2[1]);
        else {
            PAR_BEGIN
            if (tid == 0) res2[1].status = ST_NOT_RUN;
            PAR_END
        }
    } else {
        run_dd<WS>(c, in, in.comp_type, in.best_lb, &res2[0]);
        PAR_BEGIN
        if (tid == 0) res2[1].status = ST_NOT_RUN;
        PAR_END
    }
}

/// LDS bytes needed by one workgroup.
inline size_t dd_lds_bytes(int table_cap_lds, int npad, int nthreads, int tw_words = 0) {
    size_t b = (size_t)table_cap_lds * 4;
    b += (size_t)npad * 4;
    b += 256 * 4;
    b += (size_t)nthreads * 4 * 2;
    b += (sizeof(DDShared) + 15) & ~(size_t)15;
    b += (size_t)tw_words * 4;
    return (b + 15) & ~(size_t)15;
}
/// words of the TSPTW tables a workgroup may keep in LDS (EngineParams::tw_lds): distances, time windows, cheapest entering
/// edges and their order.  The rough upper bound and the domain of a state walk its must-visit set one node at a time, each
/// step a dependent table lookup: from LDS that step costs a fifth of what it costs from L2.
inline int tw_lds_words(int n) { return n * (n + 4); }

/// Binds slot `slot` of the workspace and the LDS block `lds` to a context.  TLDS: the dedup
/// table lives in LDS (else in HBM: P.gtable, for widths whose table exceeds 160 KB of LDS).
template <int WS, bool TLDS>
DDO_DEV void dd_bind(DDCtx<WS>& c, const EngineParams& P, int slot, unsigned char* lds, int nthreads) {
    c.n = P.n;
    c.npad = P.npad;
    c.unit_weights = P.unit_weights;
    c.adj = P.adj;
    c.weight = P.weight;
    c.kind = P.model_kind;
    c.depth0 = 0;   // set per work item
    c.kp_weight = P.kp_weight;
    c.kp_order = P.kp_order;
    c.vgraph = P.vgraph;
    c.vest = P.vest;
    c.vnk = P.vnk;
    c.vr = P.vr;
    c.lddelta = P.lddelta ? P.lddelta + (size_t)slot * (size_t)P.max_layers : nullptr;
    c.m2_wtt = P.m2_wtt;
    c.m2_wtf = P.m2_wtf;
    c.m2_wft = P.m2_wft;
    c.m2_wff = P.m2_wff;
    c.m2_order = P.m2_order;
    c.m2_rankpos = P.m2_rankpos;
    c.capN = P.capN;
    c.capC1 = P.capC1;
    c.max_layers = P.max_layers;
    const size_t capC1 = (size_t)P.capC1, capN = (size_t)P.capN, ml = (size_t)P.max_layers, s = (size_t)slot;
    for (int b = 0; b < 2; ++b) {
        c.cstate.set(b, P.cstate + (s * 2 + b) * (size_t)WS * capC1);
        c.ckey.set(b, P.ckey + (s * 2 + b) * capC1);
        c.cpop.set(b, P.cpop + (s * 2 + b) * capC1);
        c.cflags.set(b, P.cflags + (s * 2 + b) * capC1);
    }
    c.fan = P.fan > 2 ? P.fan : 2;
    c.dbits = P.fan > 2 ? P.dbits : 1;
    c.ctarget = P.ctarget + s * (size_t)c.fan * capN;
    c.keep = P.keep + s * capN;
    c.posmap = P.posmap + s * capC1;
    c.cls = P.cls + s * capC1;
    c.ninfo = P.ninfo + s * ml * capN;
    const bool dynpool = P.tmode && P.lpool_nodes != 0;
    c.lpool = dynpool ? P.lpool_nodes : 0;
    c.apool = dynpool ? P.apool_arcs : 0;
    c.lbase = dynpool ? P.lbase + s * (ml + 1) : nullptr;
    c.abase = dynpool ? P.abase + s * (ml + 1) : nullptr;
    c.arct = P.arct + s * (dynpool ? (size_t)P.apool_arcs : ml * (size_t)c.fan * capN);
    c.arcc = P.arcc + s * (dynpool ? (size_t)P.apool_arcs : ml * (size_t)c.fan * capN);
    c.nlayer = P.nlayer + s * ml;
    c.lvar = P.lvar + s * ml;
    c.ldup = P.ldup + s * ml * 2;
    c.cs_state = P.cs_state + s * (size_t)WS * capN;
    c.cs_value = P.cs_value + s * capN;
    c.cs_pop = P.cs_pop + s * capN;
    c.table_cap = P.table_cap;
    c.tmode = P.tmode;
    c.clocks = P.phase_clocks;
    c.lstride = P.tmode ? P.lstride : P.capN;
    c.cdbits = 20;   // an entry is never TAB_EMPTY: candidate indices stay below 2^cdbits - 1
    while ((1u << c.cdbits) <= (uint32_t)P.capC1) ++c.cdbits;
    c.lstate = nullptr;
    c.lval = c.lrub = c.lvb = c.lth = c.cth = nullptr;
    if (P.tmode) {
        const size_t lsz = dynpool ? (size_t)P.lpool_nodes : ml * (size_t)P.lstride;
        c.ninfo = P.ninfo + s * lsz;
        c.lstate = P.lstate + s * lsz * (size_t)WS;
        c.lval = P.lval + s * lsz;
        c.lrub = P.lrub + s * lsz;
        c.lvb = P.lvb + s * lsz;
        c.lth = P.lth + s * lsz;
        c.cth = (int32_t*)(P.lth + (size_t)P.nslots * lsz) + s * capC1;   // behind the theta arrays: [slot][capC1]
        c.keep = P.keep + s * capC1;
    }
    c.lntot = P.lntot + s * ml;
    c.cache_tab = P.cache_tab;
    c.cache_cap = P.tmode ? P.cache_cap : 0;
    c.cache_stride = P.cache_stride;
    c.cache_stats = P.cache_stats;
    c.dom_coord = P.dom_coord;
    c.dom_value = P.dom_value;
    c.dom_count = P.dom_count;
    c.dom_lock = P.dom_lock;
    c.dom_cap = P.tmode ? P.dom_cap : 0;
    c.dom_stats = P.dom_stats;
    c.tw = TwModel{P.n, P.tw_dist, P.tw_early, P.tw_late, P.tw_cheap, P.tw_order};
    c.dkey_tab = P.dkey_tab;
    c.dkey_cap = P.tmode ? P.dkey_cap : 0;
    c.dkey_stats = P.dkey_stats;
    unsigned char* p = lds;
    if (TLDS) {
        c.table = (uint32_t*)p;
        p += (size_t)P.table_cap * 4;
    } else {
        c.table = P.gtable + s * (size_t)P.table_cap;
    }
    c.cnt = (int32_t*)p;
    p += (size_t)P.npad * 4;
    c.hist = (uint32_t*)p;
    p += 256 * 4;
    c.tcount = (int32_t*)p;
    p += (size_t)nthreads * 4;
    c.tcount2 = (int32_t*)p;
    p += (size_t)nthreads * 4;
    c.sh = (DDShared*)p;
    p += (sizeof(DDShared) + 15) & ~(size_t)15;
    if (P.tw_lds) {   // filled by dd_stage_tables
        const int32_t* t = (const int32_t*)p;
        const size_t n = (size_t)P.n;
        c.tw = TwModel{P.n, t, t + n * n, t + n * n + n, t + n * n + 2 * n, t + n * n + 3 * n};
    }
    c.arena = P.arena;
    c.arena_cap = P.arena_cap;
    c.arena_head = P.arena_head;
    c.cutoff_flag = P.cutoff_flag;
    c.NT = nthreads;
}

/// Once per workgroup, after dd_bind: the model tables that live in LDS (TSPTW, EngineParams::tw_lds).
template <int WS>
DDO_DEV void dd_stage_tables(DDCtx<WS>& c, const EngineParams& P) {
    if (!P.tw_lds) return;
    DD_TID_SETUP(c)
    PAR_BEGIN
    int32_t* t = const_cast<int32_t*>(c.tw.dist);
    const int n = P.n, nn = n * n;
    for (int i = tid; i < nn; i += NT) t[i] = P.tw_dist[i];
    for (int i = tid; i < n; i += NT) {
        t[nn + i] = P.tw_early[i];
        t[nn + n + i] = P.tw_late[i];
        t[nn + 2 * n + i] = P.tw_cheap[i];
        t[nn + 3 * n + i] = P.tw_order[i];
    }
    PAR_END
}

}  // namespace ddo_hip
