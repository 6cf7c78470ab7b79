// =============================================================================
// dd_tsptw.hpp -- the TSPTW model of the reference's example (examples/tsptw/{state,model,relax,heuristics,dominance}.rs)
// for the layer-rebuilding engine (misp_dd_core.hpp): variable fan-out (one child per node that may be visited next),
// "fuzzy" relaxed states (a set of possible positions, an interval of elapsed times, must / maybe visit sets).
//
// State on the wire and in HBM -- 5 words (include/ddo_hip.h, ddo_model_create_tsptw; at most 64 nodes so that each of
// the reference's Set256 fields is one word):
//   w0  Position::Virtual(set): bit i <=> the salesman may be at node i        (0 for Position::Node)
//   w1  must_visit
//   w2  maybe_visit                                                            (0 when None)
//   w3  elapsed: earliest (bits 0..31) | latest (bits 32..63); ElapsedTime::FixedAt(d): both = d
//   w4  Position::Node index (bits 0..15) | flags (bit 16 virtual position, 17 fuzzy elapsed time, 18 maybe_visit is
//       Some) | depth (bits 32..47)
// Every logical state has exactly one encoding, so state equality (dedup, cache keys) is word equality.
// =============================================================================
#pragma once
#include "dd_types.h"

namespace ddo_hip {

constexpr uint64_t TW_VIRTUAL = 1ULL << 16, TW_FUZZY = 1ULL << 17, TW_MAYBE = 1ULL << 18;
constexpr int32_t RUB_NEG_INF = INT32_MIN + 1;   // fast_upper_bound == isize::MIN: the state cannot be completed (relax.rs:207-262)

struct TwModel {
    int n;
    const int32_t* dist;      // [n][n]
    const int32_t* early;     // [n] time windows (instance.rs:27-35)
    const int32_t* late;      // [n]
    const int32_t* cheap;     // [n] cheapest edge entering each node (relax.rs:50-63)
};

// (included by misp_dd_core.hpp after its DDO_DEV / dd_ctz definitions)
DDO_DEV uint32_t tw_earliest(const uint64_t* s) { return (uint32_t)s[3]; }
DDO_DEV uint32_t tw_latest(const uint64_t* s) { return (s[4] & TW_FUZZY) ? (uint32_t)(s[3] >> 32) : (uint32_t)s[3]; }
DDO_DEV int tw_depth(const uint64_t* s) { return (int)((s[4] >> 32) & 0xFFFF); }

/// model.rs:194-215: distance from the (possibly virtual) position to node j: smallest / largest over the position set
DDO_DEV int64_t tw_min_dist(const TwModel& m, const uint64_t* s, int j) {
    if (!(s[4] & TW_VIRTUAL)) return m.dist[(size_t)(s[4] & 0xFFFF) * m.n + j];
    int64_t best = INT64_MAX;
    for (uint64_t x = s[0]; x; x &= x - 1) {
        const int64_t d = m.dist[(size_t)dd_ctz(x) * m.n + j];
        best = d < best ? d : best;
    }
    return best;
}
DDO_DEV int64_t tw_max_dist(const TwModel& m, const uint64_t* s, int j) {
    if (!(s[4] & TW_VIRTUAL)) return m.dist[(size_t)(s[4] & 0xFFFF) * m.n + j];
    int64_t best = 0;
    for (uint64_t x = s[0]; x; x &= x - 1) {
        const int64_t d = m.dist[(size_t)dd_ctz(x) * m.n + j];
        best = d > best ? d : best;
    }
    return best;
}
/// model.rs:150-157
DDO_DEV bool tw_can_move_to(const TwModel& m, const uint64_t* s, int j) {
    const int64_t md = tw_min_dist(m, s, j);
    return md != INT64_MAX && (int64_t)tw_earliest(s) + md <= (int64_t)m.late[j];
}
/// for_each_in_domain (model.rs:65-94): bit j set <=> the salesman may go to node j next
DDO_DEV uint64_t tw_domain(const TwModel& m, const uint64_t* s) {
    if (tw_depth(s) == m.n - 1) return tw_can_move_to(m, s, 0) ? 1ULL : 0ULL;
    for (uint64_t x = s[1]; x; x &= x - 1)
        if (!tw_can_move_to(m, s, dd_ctz(x))) return 0ULL;          // a node that must be visited is out of reach
    uint64_t dom = s[1];
    if (s[4] & TW_MAYBE)
        for (uint64_t x = s[2]; x; x &= x - 1)
            if (tw_can_move_to(m, s, dd_ctz(x))) dom |= 1ULL << dd_ctz(x);
    return dom;
}
/// transition (model.rs:95-115, arrival time :158-193) and transition_cost (:116-139: minus travel and waiting time)
DDO_DEV void tw_transition(const TwModel& m, const uint64_t* s, int j, uint64_t* r, int32_t* cost) {
    const uint64_t bit = 1ULL << j;
    const int64_t mind = tw_min_dist(m, s, j), maxd = tw_max_dist(m, s, j);
    const int64_t mn = (int64_t)tw_earliest(s) + mind;
    const int64_t mx = (int64_t)tw_latest(s) + maxd;
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
    r[0] = 0;
    r[1] = s[1] & ~bit;
    r[2] = (s[4] & TW_MAYBE) ? (s[2] & ~bit) : 0;
    r[3] = (uint64_t)(uint32_t)e | ((uint64_t)(uint32_t)(fuzzy ? l : e) << 32);
    r[4] = (uint64_t)j | (fuzzy ? TW_FUZZY : 0) | (s[4] & TW_MAYBE) | ((uint64_t)(tw_depth(s) + 1) << 32);
    const int64_t arrive = mn;
    const int64_t waiting = arrive < twe ? twe - arrive : 0;
    *cost = -(int32_t)(mind + waiting);
}
/// TsptwRelax::fast_upper_bound (relax.rs:196-264)
DDO_DEV int32_t tw_rub(const TwModel& m, const uint64_t* s) {
    int complete_tour = m.n - tw_depth(s);
    int64_t mandatory = 0, back = INT64_MAX;
    const int64_t now = tw_earliest(s);
    for (uint64_t x = s[1]; x; x &= x - 1) {
        const int i = dd_ctz(x);
        complete_tour -= 1;
        mandatory += m.cheap[i];
        const int64_t d0 = m.dist[(size_t)i * m.n];
        back = d0 < back ? d0 : back;
        if (now + m.cheap[i] > (int64_t)m.late[i]) return RUB_NEG_INF;
    }
    if (s[4] & TW_MAYBE) {
        int32_t tmp[64];
        int cnt = 0, violations = 0;
        for (uint64_t x = s[2]; x; x &= x - 1) {
            const int i = dd_ctz(x);
            int k = cnt++;                         // insertion sort: the cheapest `complete_tour` of them are needed
            const int32_t v = m.cheap[i];
            while (k > 0 && tmp[k - 1] > v) {
                tmp[k] = tmp[k - 1];
                --k;
            }
            tmp[k] = v;
            const int64_t d0 = m.dist[(size_t)i * m.n];
            back = d0 < back ? d0 : back;
            if (now + v > (int64_t)m.late[i]) violations += 1;
        }
        if (cnt - violations < complete_tour) return RUB_NEG_INF;
        for (int k = 0; k < complete_tour && k < cnt; ++k) mandatory += tmp[k];
    }
    if (mandatory == 0) {
        const int64_t here = tw_min_dist(m, s, 0);
        back = here < back ? here : back;
    }
    const int64_t total = mandatory + back;
    if (now + total > (int64_t)m.late[0]) return RUB_NEG_INF;
    return -(int32_t)total;
}

}  // namespace ddo_hip
