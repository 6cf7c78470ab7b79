// =============================================================================
// dd_tsptw.hpp -- the TSPTW model of the reference's example (examples/tsptw/{state,model,relax,heuristics,dominance}.rs)
// for the layer-rebuilding engine (misp_dd_core.hpp): variable fan-out (one child per node that may be visited next),
// "fuzzy" relaxed states (a set of possible positions, an interval of elapsed times, must / maybe visit sets).
//
// State on the wire and in HBM -- 3K + 2 words (include/ddo_hip.h, ddo_model_create_tsptw), K = words of one of the
// reference's Set256 fields (state.rs:34-69): K = 1 up to 64 nodes, 2 up to 128, 4 up to 256:
//   w[0 .. K)     Position::Virtual(set): bit i <=> the salesman may be at node i        (0 for Position::Node)
//   w[K .. 2K)    must_visit
//   w[2K .. 3K)   maybe_visit                                                            (0 when None)
//   w[3K]         elapsed: earliest (bits 0..31) | latest (bits 32..63); ElapsedTime::FixedAt(d): both = d
//   w[3K + 1]     Position::Node index (bits 0..15) | flags (bit 16 virtual position, 17 fuzzy elapsed time, 18 maybe_visit
//                 is Some) | depth (bits 32..47)
// Every logical state has exactly one encoding, so state equality (dedup, cache keys) is word equality.  K is a template
// constant tied to the kernel's state width (tw_k_of_ws: 7 -> 1, 8 -> 2, 16 -> 4) so that the state stays in registers.
// =============================================================================
#pragma once
#include "dd_types.h"

namespace ddo_hip {

constexpr uint64_t TW_VIRTUAL = 1ULL << 16, TW_FUZZY = 1ULL << 17, TW_MAYBE = 1ULL << 18;
constexpr int32_t RUB_NEG_INF = INT32_MIN + 1;   // fast_upper_bound == isize::MIN: the state cannot be completed (relax.rs:207-262)

/// words of one node set for nb_nodes nodes, the state words that go with it and the kernel width that carries them
constexpr int tw_set_words(int n) { return n <= 64 ? 1 : (n <= 128 ? 2 : 4); }
constexpr int tw_state_words(int n) { return 3 * tw_set_words(n) + 2; }
constexpr int tw_k_of_ws(int ws_template) { return ws_template == 7 ? 1 : (ws_template == 8 ? 2 : (ws_template == 16 ? 4 : 0)); }

struct TwModel {
    int n;
    const int32_t* dist;      // [n][n]
    const int32_t* early;     // [n] time windows (instance.rs:27-35)
    const int32_t* late;      // [n]
    const int32_t* cheap;     // [n] cheapest edge entering each node (relax.rs:50-63)
    const int32_t* order;     // [n] the nodes by increasing `cheap` (fast_upper_bound takes the cheapest of the maybe set)
};

// (included by misp_dd_core.hpp after its DDO_DEV / dd_ctz definitions)
template <int K> DDO_DEV uint64_t tw_meta(const uint64_t* s) { return s[3 * K + 1]; }
template <int K> DDO_DEV uint32_t tw_earliest(const uint64_t* s) { return (uint32_t)s[3 * K]; }
template <int K> DDO_DEV uint32_t tw_latest(const uint64_t* s) { return (tw_meta<K>(s) & TW_FUZZY) ? (uint32_t)(s[3 * K] >> 32) : (uint32_t)s[3 * K]; }
template <int K> DDO_DEV int tw_depth(const uint64_t* s) { return (int)((tw_meta<K>(s) >> 32) & 0xFFFF); }

/// model.rs:194-215: distance from the (possibly virtual) position to node j: smallest / largest over the position set
template <int K>
DDO_DEV int64_t tw_min_dist(const TwModel& m, const uint64_t* s, int j) {
    if (!(tw_meta<K>(s) & TW_VIRTUAL)) return m.dist[(size_t)(tw_meta<K>(s) & 0xFFFF) * m.n + j];
    int64_t best = INT64_MAX;
#pragma unroll
    for (int q = 0; q < K; ++q)
        for (uint64_t x = s[q]; x; x &= x - 1) {
            const int64_t d = m.dist[(size_t)(64 * q + dd_ctz(x)) * m.n + j];
            best = d < best ? d : best;
        }
    return best;
}
template <int K>
DDO_DEV int64_t tw_max_dist(const TwModel& m, const uint64_t* s, int j) {
    if (!(tw_meta<K>(s) & TW_VIRTUAL)) return m.dist[(size_t)(tw_meta<K>(s) & 0xFFFF) * m.n + j];
    int64_t best = 0;
#pragma unroll
    for (int q = 0; q < K; ++q)
        for (uint64_t x = s[q]; x; x &= x - 1) {
            const int64_t d = m.dist[(size_t)(64 * q + dd_ctz(x)) * m.n + j];
            best = d > best ? d : best;
        }
    return best;
}
/// model.rs:150-157
template <int K>
DDO_DEV bool tw_can_move_to(const TwModel& m, const uint64_t* s, int j) {
    const int64_t md = tw_min_dist<K>(m, s, j);
    return md != INT64_MAX && (int64_t)tw_earliest<K>(s) + md <= (int64_t)m.late[j];
}
/// for_each_in_domain (model.rs:65-94): bit j of dom[] set <=> the salesman may go to node j next
template <int K>
DDO_DEV void tw_domain(const TwModel& m, const uint64_t* s, uint64_t* dom) {
#pragma unroll
    for (int q = 0; q < K; ++q) dom[q] = 0;
    if (tw_depth<K>(s) == m.n - 1) {
        dom[0] = tw_can_move_to<K>(m, s, 0) ? 1ULL : 0ULL;
        return;
    }
#pragma unroll
    for (int q = 0; q < K; ++q)
        for (uint64_t x = s[K + q]; x; x &= x - 1)
            if (!tw_can_move_to<K>(m, s, 64 * q + dd_ctz(x))) return;          // a node that must be visited is out of reach
#pragma unroll
    for (int q = 0; q < K; ++q) {
        uint64_t d = s[K + q];
        if (tw_meta<K>(s) & TW_MAYBE)
            for (uint64_t x = s[2 * K + q]; x; x &= x - 1)
                if (tw_can_move_to<K>(m, s, 64 * q + dd_ctz(x))) d |= 1ULL << dd_ctz(x);
        dom[q] = d;
    }
}
/// transition (model.rs:95-115, arrival time :158-193) and transition_cost (:116-139: minus travel and waiting time)
template <int K>
DDO_DEV void tw_transition(const TwModel& m, const uint64_t* s, int j, uint64_t* r, int32_t* cost) {
    const int64_t mind = tw_min_dist<K>(m, s, j), maxd = tw_max_dist<K>(m, s, j);
    const int64_t mn = (int64_t)tw_earliest<K>(s) + mind;
    const int64_t mx = (int64_t)tw_latest<K>(s) + maxd;
    const int64_t twe = m.early[j], twl = m.late[j];
    int64_t e, l;
    bool fuzzy;
    if (mn == mx) {
        e = l = mn > twe ? mn : twe;
        fuzzy = false;
    } else {
        e = mn > twe ? mn : twe;
        l = mx < twl ? mx : twl;
        fuzzy = e != l;
        if (!fuzzy) l = e;
    }
    const uint64_t meta = tw_meta<K>(s);
#pragma unroll
    for (int q = 0; q < K; ++q) {
        const uint64_t bit = (j >> 6) == q ? 1ULL << (j & 63) : 0ULL;
        r[q] = 0;
        r[K + q] = s[K + q] & ~bit;
        r[2 * K + q] = (meta & TW_MAYBE) ? (s[2 * K + q] & ~bit) : 0;
    }
    r[3 * K] = (uint64_t)(uint32_t)e | ((uint64_t)(uint32_t)(fuzzy ? l : e) << 32);
    r[3 * K + 1] = (uint64_t)j | (fuzzy ? TW_FUZZY : 0) | (meta & TW_MAYBE) | ((uint64_t)(tw_depth<K>(s) + 1) << 32);
    const int64_t arrive = mn;
    const int64_t waiting = arrive < twe ? twe - arrive : 0;
    *cost = -(int32_t)(mind + waiting);
}
/// TsptwRelax::fast_upper_bound (relax.rs:196-264)
template <int K>
DDO_DEV int32_t tw_rub(const TwModel& m, const uint64_t* s) {
    int complete_tour = m.n - tw_depth<K>(s);
    int64_t mandatory = 0, back = INT64_MAX;
    const int64_t now = tw_earliest<K>(s);
#pragma unroll
    for (int q = 0; q < K; ++q)
        for (uint64_t x = s[K + q]; x; x &= x - 1) {
            const int i = 64 * q + dd_ctz(x);
            complete_tour -= 1;
            mandatory += m.cheap[i];
            const int64_t d0 = m.dist[(size_t)i * m.n];
            back = d0 < back ? d0 : back;
            if (now + m.cheap[i] > (int64_t)m.late[i]) return RUB_NEG_INF;
        }
    if (tw_meta<K>(s) & TW_MAYBE) {
        // the reference sorts the entering edges of the maybe set and adds the `complete_tour` cheapest: the nodes are walked
        // in the order of their cheapest entering edge instead (the sum of the smallest ones does not depend on how ties fall)
        int cnt = 0, violations = 0, taken = 0;
        for (int idx = 0; idx < m.n; ++idx) {
            const int i = m.order[idx];
            uint64_t w = s[2 * K];
#pragma unroll
            for (int q = 1; q < K; ++q) w = (i >> 6) == q ? s[2 * K + q] : w;
            if (!((w >> (i & 63)) & 1ULL)) continue;
            ++cnt;
            const int32_t v = m.cheap[i];
            if (taken < complete_tour) {
                mandatory += v;
                ++taken;
            }
            const int64_t d0 = m.dist[(size_t)i * m.n];
            back = d0 < back ? d0 : back;
            if (now + v > (int64_t)m.late[i]) violations += 1;
        }
        if (cnt - violations < complete_tour) return RUB_NEG_INF;
    }
    if (mandatory == 0) {
        const int64_t here = tw_min_dist<K>(m, s, 0);
        back = here < back ? here : back;
    }
    const int64_t total = mandatory + back;
    if (now + total > (int64_t)m.late[0]) return RUB_NEG_INF;
    return -(int32_t)total;
}

/// TsptwDominance's key (dominance.rs:26-42): (position, must_visit) of candidate cd, read from the word-major candidate
/// states: 2K + 1 words -- the position set (or the node in word 0), whether the position is virtual, the must-visit set
template <int K>
DDO_DEV void tw_dominance_key(const uint64_t* cst, size_t cap, int cd, uint64_t* key) {
    const uint64_t meta = cst[(size_t)(3 * K + 1) * cap + cd];
    const bool virt = (meta & TW_VIRTUAL) != 0;
#pragma unroll
    for (int q = 0; q < K; ++q) {
        key[q] = virt ? cst[(size_t)q * cap + cd] : (q == 0 ? (meta & 0xFFFF) : 0ULL);
        key[K + 1 + q] = cst[(size_t)(K + q) * cap + cd];
    }
    key[K] = virt ? 1ULL : 0ULL;
}

}  // namespace ddo_hip
