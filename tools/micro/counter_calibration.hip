// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on the access patterns of the in-place engine, and the throughput the
// memory system sustains for each of them ("Next round" item 2).  Every kernel moves a KNOWN number of bytes:
//   stream16   lane i reads / writes 16 consecutive bytes (the guide's reference pattern: FETCH_SIZE reports 1/2)
//   line64     every lane reads / writes its own random 64-byte line as 4 x 16 B (thread-per-node record I/O)
//   word8      every lane reads / writes 8 bytes of its own random 64-byte line (the work-list sweep; key|hash stores)
//   group8     8 lanes x 8 B cover one random 64-byte line (8-lanes-per-node record I/O)
// Random lines are drawn either over the whole buffer (>= 2 GB: beyond the 256 MB Infinity Cache) or within a per-workgroup
// region of `region` lines (a DD slot: 14 352 lines = 918 KB per workgroup).
// Output: one JSON line per kernel launch with the bytes the kernel needed, the sectors it touched and its duration; the
// launch order is fixed, so tools/micro/calibrate_counters.py joins it with the rocprofv3 counter_collection.csv by dispatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct alignas(16) U64x2 { uint64_t a, b; };

__device__ inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

enum { P_STREAM16 = 0, P_LINE64 = 1, P_WORD8 = 2, P_GROUP8 = 3 };

// nlines: lines of the whole buffer; region: 0 = whole buffer, else lines per workgroup region (region * gridDim <= nlines)
template <int PAT, int U>
__global__ void __launch_bounds__(512) k_read(const uint64_t* __restrict__ buf, uint64_t nlines, uint32_t region, int iters, uint64_t* sink) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    uint64_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint64_t v[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = (uint32_t)(it * U + u);
            if (PAT == P_STREAM16) {
                const uint64_t idx = ((uint64_t)q * nthreads + gid) * 2;   // in u64 words
                const U64x2 x = *(const U64x2*)(buf + (idx % (nlines * 8)));
                v[u][0] = x.a; v[u][1] = x.b;
            } else {
                const uint32_t who = PAT == P_GROUP8 ? (gid >> 3) : gid;
                const uint32_t r = mix32(who * 2654435761u + q * 40503u + 12345u);
                const uint64_t line = region ? (uint64_t)blockIdx.x * region + (r % region) : ((uint64_t)r * 2654435761ull >> 7) % nlines;
                const uint64_t* p = buf + line * 8;
                if (PAT == P_LINE64) {
                    const U64x2* p2 = (const U64x2*)p;
                    U64x2 a = p2[0], b = p2[1], c = p2[2], d = p2[3];
                    v[u][0] = a.a ^ b.b ^ c.a; v[u][1] = a.b ^ b.a ^ c.b ^ d.a ^ d.b;
                } else if (PAT == P_WORD8) {
                    v[u][0] = p[q % 7]; v[u][1] = 0;
                } else {
                    v[u][0] = p[gid & 7]; v[u][1] = 0;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u][0] * 3 + v[u][1];
    }
    if (acc == 0x123456789ULL) sink[gid] = acc;
}

template <int PAT, int U>
__global__ void __launch_bounds__(512) k_write(uint64_t* __restrict__ buf, uint64_t nlines, uint32_t region, int iters) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t q = (uint32_t)(it * U + u);
            if (PAT == P_STREAM16) {
                const uint64_t idx = ((uint64_t)q * nthreads + gid) * 2;
                *(U64x2*)(buf + (idx % (nlines * 8))) = U64x2{idx, (uint64_t)q};
            } else {
                const uint32_t who = PAT == P_GROUP8 ? (gid >> 3) : gid;
                const uint32_t r = mix32(who * 2654435761u + q * 40503u + 12345u);
                const uint64_t line = region ? (uint64_t)blockIdx.x * region + (r % region) : ((uint64_t)r * 2654435761ull >> 7) % nlines;
                uint64_t* p = buf + line * 8;
                if (PAT == P_LINE64) {
                    U64x2* p2 = (U64x2*)p;
                    p2[0] = U64x2{line, 1}; p2[1] = U64x2{line, 2}; p2[2] = U64x2{line, 3}; p2[3] = U64x2{line, 4};
                } else if (PAT == P_WORD8) {
                    p[q % 7] = line;
                } else {
                    p[gid & 7] = line;
                }
            }
        }
    }
}

struct Case { const char* name; int pat; bool write; uint32_t region; };

int main(int argc, char** argv) {
    const int nblocks = argc > 1 ? atoi(argv[1]) : 512;
    const int nth = argc > 2 ? atoi(argv[2]) : 512;
    const int iters = argc > 3 ? atoi(argv[3]) : 64;
    const uint64_t nlines = (uint64_t)(argc > 4 ? atoll(argv[4]) : 2048) * 1024 * 1024 / 64;   // MB -> lines
    uint64_t* buf; uint64_t* sink;
    CK(hipMalloc(&buf, nlines * 64)); CK(hipMemset(buf, 1, nlines * 64));
    CK(hipMalloc(&sink, (size_t)nblocks * nth * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    constexpr int U = 4;
    const Case cases[] = {
        {"read_stream16", P_STREAM16, false, 0}, {"read_line64_far", P_LINE64, false, 0}, {"read_word8_far", P_WORD8, false, 0}, {"read_group8_far", P_GROUP8, false, 0},
        {"read_line64_slot", P_LINE64, false, 14352}, {"read_word8_slot", P_WORD8, false, 14352}, {"read_group8_slot", P_GROUP8, false, 14352},
        {"write_stream16", P_STREAM16, true, 0}, {"write_line64_far", P_LINE64, true, 0}, {"write_word8_far", P_WORD8, true, 0}, {"write_group8_far", P_GROUP8, true, 0},
        {"write_line64_slot", P_LINE64, true, 14352}, {"write_word8_slot", P_WORD8, true, 14352}, {"write_group8_slot", P_GROUP8, true, 14352},
    };
    for (const Case& c : cases) {
        if (c.region && (uint64_t)c.region * nblocks > nlines) { printf("{\"case\": \"%s\", \"skipped\": \"buffer too small\"}\n", c.name); continue; }
        float best = 1e30f;
        for (int rep = 0; rep < 2; ++rep) {   // (two launches per case: the second one is the calibrated one -- warm TLBs)
            CK(hipEventRecord(e0));
#define LR(P) hipLaunchKernelGGL((k_read<P, U>), dim3(nblocks), dim3(nth), 0, 0, buf, nlines, c.region, iters, sink)
#define LW(P) hipLaunchKernelGGL((k_write<P, U>), dim3(nblocks), dim3(nth), 0, 0, buf, nlines, c.region, iters)
            if (!c.write) { if (c.pat == 0) LR(0); else if (c.pat == 1) LR(1); else if (c.pat == 2) LR(2); else LR(3); }
            else { if (c.pat == 0) LW(0); else if (c.pat == 1) LW(1); else if (c.pat == 2) LW(2); else LW(3); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double accesses = (double)nblocks * nth * iters * U;
        const double useful = accesses * (c.pat == P_STREAM16 ? 16 : c.pat == P_LINE64 ? 64 : 8);
        const double sectors = c.pat == P_STREAM16 ? useful / 64 : (c.pat == P_GROUP8 ? accesses / 8 : accesses);   // 64-byte lines touched (with repeats)
        printf("{\"case\": \"%s\", \"write\": %d, \"region_lines\": %u, \"useful_bytes\": %.0f, \"line_bytes\": %.0f, \"ms\": %.4f, \"useful_GBps\": %.1f, \"line_GBps\": %.1f, \"Mlines_per_s\": %.1f}\n",
               c.name, c.write ? 1 : 0, c.region, useful, sectors * 64, best, useful / best / 1e6, sectors * 64 / best / 1e6, sectors / best / 1e3);
    }
    return 0;
}
