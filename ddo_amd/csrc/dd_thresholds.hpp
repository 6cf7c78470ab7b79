// =============================================================================
// dd_thresholds.hpp -- the cache side of the layer-rebuilding engine (misp_dd_core.hpp):
//   SimpleCache (cache/simple.rs:36-73) as one open-addressing table in HBM, shared by every decision diagram a solver
//   compiles, keyed by (depth, state) and holding Threshold { value, explored } (common.rs) packed into one word so that
//   `update_threshold` (keep the larger threshold; at equal values `explored` wins) is a single atomicMax.
// The table is never cleared by layer: ParallelSolver::get_workload clears the layers above the first active one
// (parallel.rs:506-511) only to free memory -- no sub-problem of such a depth can appear again, so those entries are
// never queried -- and a full table refuses insertions (counted in cache_stats[1]): the search stays correct, it only
// prunes less.
// =============================================================================
#pragma once
#include "dd_types.h"

namespace ddo_hip {

constexpr int32_t TH_NONE = INT32_MAX;        // theta: None
constexpr int32_t TH_INF = INT32_MAX - 1;     // theta: isize::MAX and everything derived from it by saturating arithmetic
constexpr int64_t TH_INF64 = (int64_t)1 << 40;
// An entry lives at most this many slots from its home slot (update and get agree).  Round 6: 256 (was 4 096) -- a search that fills the
// table (Pooled DDs cache every exact node they expand: hundreds of millions of (state, depth) pairs on brock200_4) made every look-up
// and every refused insertion walk the whole budget, a dependent global load per step: seconds per decision diagram.  At the load
// factors a table is sized for a chain of 256 does not occur; in a full one the look-up is bounded and the insertion refused (counted).
constexpr uint64_t CACHE_MAX_PROBES = 256;
constexpr uint64_t CT_EMPTY = 0ULL, CT_LOCKED = 2ULL;   // tag word: 0 empty, 2 being written, (hash | 1) ready

DD_HD inline int64_t th_pack(int32_t theta, bool explored) {
    const int64_t v = theta >= TH_INF ? TH_INF64 : (int64_t)theta;
    return v * 2 + (explored ? 1 : 0);
}
DD_HD inline int32_t th_value(int64_t packed) {
    const int64_t v = packed >> 1;
    return v >= (TH_INF64 >> 1) ? TH_INF : (int32_t)v;
}
DD_HD inline bool th_explored(int64_t packed) { return (packed & 1) != 0; }
/// theta - cost with the reference's saturating arithmetic collapsed to one +infinity
DD_HD inline int32_t th_sub(int32_t theta, int32_t cost) { return theta >= TH_INF ? TH_INF : theta - cost; }

#if defined(DDO_HOST_EMULATION)
#define CT_LD(p) (*(p))
#define CT_CAS(p, c, v) emu_cas64((p), (c), (v))
#define CT_ST(p, v) (*(p) = (v))
#define CT_MAX_I64(p, v) emu_atomic_max<long long>((long long*)(p), (long long)(v))
#define CT_ADD(p, v) emu_atomic_add<unsigned long long>((p), (v))
#define CT_FENCE()
#define GLB_MIN_I32(p, v) emu_atomic_min<int32_t>((p), (v))
#define GLB_MAX_I32(p, v) emu_atomic_max<int32_t>((p), (v))
inline uint64_t emu_cas64(uint64_t* p, uint64_t c, uint64_t v) { uint64_t o = *p; if (o == c) *p = v; return o; }
#else
#define CT_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define CT_CAS(p, c, v) atomicCAS((unsigned long long*)(p), (unsigned long long)(c), (unsigned long long)(v))
#define CT_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define CT_MAX_I64(p, v) atomicMax((long long*)(p), (long long)(v))
#define CT_ADD(p, v) atomicAdd((p), (v))
#define CT_FENCE() __threadfence()
#define GLB_MIN_I32(p, v) atomicMin((p), (v))
#define GLB_MAX_I32(p, v) atomicMax((p), (v))
#endif

template <int WS>
DDO_DEV uint64_t cache_hash(const uint64_t* s, int depth) {
    uint64_t h = 0x9E3779B97F4A7C15ULL * (uint64_t)(depth + 1);
#pragma unroll
    for (int k = 0; k < WS; ++k) {
        h ^= s[k];
        h *= 0xFF51AFD7ED558CCDULL;
        h ^= h >> 29;
    }
    h *= 0xC4CEB9FE1A85EC53ULL;
    h ^= h >> 32;
    return h;
}

/// Cache::get_threshold (simple.rs:60-62): true when (depth, state) has a threshold; *packed receives it
template <int WS, class Ctx>
DDO_DEV bool cache_get(const Ctx& c, const uint64_t* s, int depth, int64_t* packed) {
    if (!c.cache_cap) return false;
    const uint64_t h = cache_hash<WS>(s, depth);
    const uint64_t tag = h | 1ULL;
    const uint64_t mask = c.cache_cap - 1;
    uint64_t slot = (h >> 1) & mask;
    // cache_update never stores an entry more than CACHE_MAX_PROBES slots from its home slot, so a look-up that has gone
    // that far without a hit is a miss: a full or heavily clustered table costs a bounded scan per candidate
    for (uint64_t probes = 0; probes <= mask && probes < CACHE_MAX_PROBES; ++probes) {
        uint64_t* e = c.cache_tab + slot * (uint64_t)c.cache_stride;
        uint64_t t = CT_LD(&e[0]);
        while (t == CT_LOCKED) t = CT_LD(&e[0]);   // another compile is writing this entry right now
        if (t == CT_EMPTY) return false;
        CT_FENCE();   // acquire: the writer stores payload, fence, tag -- the payload loads below must not move above the tag load
        if (t == tag && CT_LD(&e[2]) == (uint64_t)depth) {
            bool eq = true;
#pragma unroll
            for (int k = 0; k < WS; ++k) eq &= CT_LD(&e[3 + k]) == s[k];
            if (eq) {
                const int64_t p = (int64_t)CT_LD(&e[1]);
                if (p == INT64_MIN) return false;   // claimed but no threshold yet
                *packed = p;
                return true;
            }
        }
        slot = (slot + 1) & mask;
    }
    return false;
}

/// Cache::update_threshold (simple.rs:64-68): the entry keeps max(old, new) under Threshold's order (value, explored)
template <int WS, class Ctx>
DDO_DEV void cache_update(const Ctx& c, const uint64_t* s, int depth, int64_t packed) {
    if (!c.cache_cap) return;
    const uint64_t h = cache_hash<WS>(s, depth);
    const uint64_t tag = h | 1ULL;
    const uint64_t mask = c.cache_cap - 1;
    uint64_t slot = (h >> 1) & mask;
    // a table that is HALF full takes no new entries (thresholds of states it already holds still rise): sound -- less pruning -- and the
    // look-ups keep their short chains (linear probing: 2.5 probes per miss at one half, 32 at seven eighths -- each a dependent global
    // load; brock200_4 under ParCachingSolverPooled, 4 M entries: 18.6 s of kernels with a 7/8 limit, 0.93 s with a table that never fills)
    const bool full = CT_LD(&c.cache_stats[0]) >= (unsigned long long)(c.cache_cap >> 1);
    for (uint64_t probes = 0; probes <= mask && probes < CACHE_MAX_PROBES; ++probes) {
        uint64_t* e = c.cache_tab + slot * (uint64_t)c.cache_stride;
        bool done = false, next = false;
        while (!done && !next) {   // the winner of the claim finishes inside one iteration: lanes of a wave cannot starve each other
            uint64_t t = CT_LD(&e[0]);
            if (t == CT_EMPTY) {
                if (full) {
                    CT_ADD(&c.cache_stats[1], 1ULL);
                    return;
                }
                if (CT_CAS(&e[0], CT_EMPTY, CT_LOCKED) == CT_EMPTY) {
                    CT_ST(&e[1], (uint64_t)INT64_MIN);
                    CT_ST(&e[2], (uint64_t)depth);
#pragma unroll
                    for (int k = 0; k < WS; ++k) CT_ST(&e[3 + k], s[k]);
                    CT_FENCE();
                    CT_ST(&e[0], tag);
                    CT_ADD(&c.cache_stats[0], 1ULL);
                    CT_MAX_I64(&e[1], packed);
                    done = true;
                }
            } else if (t == CT_LOCKED) {
                // spin: re-read
            } else if (t == tag && CT_LD(&e[2]) == (uint64_t)depth) {
                bool eq = true;
#pragma unroll
                for (int k = 0; k < WS; ++k) eq &= CT_LD(&e[3 + k]) == s[k];
                if (eq) {
                    CT_MAX_I64(&e[1], packed);
                    done = true;
                } else next = true;
            } else next = true;
        }
        if (done) return;
        slot = (slot + 1) & mask;
    }
    CT_ADD(&c.cache_stats[1], 1ULL);   // table (or this neighbourhood) full: the threshold is dropped
}

}  // namespace ddo_hip
