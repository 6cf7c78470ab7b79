#!/bin/bash
# ThreadSanitizer over the CPU oracle (test infrastructure): the known-answer tests and the ParallelSolver with 8 and 32 threads on
# brock200_2 / W = 100.  Round 3: no report.  (The device host solver's own helper threads -- LazyFringe::prepare_block over
# disjoint blocks -- need a GPU under the sanitizer and were not run.)
set -e
cd "$(dirname "$0")/../oracle"
D=${TMPDIR:-/tmp}/ddo_tsan; mkdir -p $D
g++ -std=c++17 -O1 -g -fsanitize=thread -pthread -o $D/kat kat_main.cpp
(cd $D && ./kat | tail -1)
cat > $D/drv.cpp <<'EOC'
#include "oracle_capi.cpp"
#include <cstdio>
int main(int argc, char** argv) {
    void* h = oracle_misp_load(argv[1]);
    if (!h) return 2;
    oracle_solve_out out;
    for (int threads : {8, 32}) {
        int rc = oracle_misp_solve(h, (uint64_t)atoi(argv[2]), threads, 0.0, &out, nullptr);
        std::printf("threads %d rc %d best %lld explored %llu\n", threads, rc, (long long)out.best_value, (unsigned long long)out.explored);
    }
    return 0;
}
EOC
g++ -std=c++17 -O1 -g -fsanitize=thread -pthread -I. -o $D/drv $D/drv.cpp
cd .. && $D/drv data/misp/brock200_2.clq 100
