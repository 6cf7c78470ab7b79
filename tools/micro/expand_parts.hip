// Which memory operations of the expand loop cost what?  One branching node per thread and round (dense tier occupancy:
// 512 threads, 2 workgroups per CU), random parent / child slots, components switched by a bit mask:
//   1 parent record read   2 parent path read   4 NO-child record word (8 B)   8 NO-child key|hash (8 B)
//   16 YES-child record    32 YES-child path    64 YES-child key|hash (8 B)    128 event record (16 B, consecutive)
//   256 records and paths by 4 lanes x 16 B (quad) instead of one lane x 4 x 16 B     512 key|hash read of the parent (8 B)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct alignas(16) U64x2 { uint64_t a, b; };
struct alignas(16) U32x4 { uint32_t x, y, z, w; };
constexpr int CAPS = 20008;
constexpr int ROUNDS = 48;
constexpr int NTH = 512;
__device__ inline uint32_t rng(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

__global__ void __launch_bounds__(NTH) k_parts(uint8_t* base, int mask, unsigned long long* cyc) {
    const size_t per = (size_t)CAPS * (64 + 64 + 8) + (size_t)ROUNDS * NTH * 16;
    uint8_t* p0 = base + (size_t)blockIdx.x * per;
    uint64_t* rec = (uint64_t*)p0;
    uint64_t* path = (uint64_t*)(p0 + (size_t)CAPS * 64);
    uint64_t* keyh = (uint64_t*)(p0 + (size_t)CAPS * 128);
    U32x4* ev = (U32x4*)(p0 + (size_t)CAPS * 136);
    uint32_t seed = blockIdx.x * 7919u + threadIdx.x * 31u + 1;
    const bool quad = mask & 256;
    const int lane = threadIdx.x & 63, qb = lane & ~3, pr = lane & 3;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < ROUNDS; ++r) {
        const int par = rng(seed) % CAPS, ny = rng(seed) % CAPS, vw = r % 7;
        uint64_t st[8] = {1, 2, 3, 4, 5, 6, 7, 8}, pa[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t kh = 0;
        if (mask & 512) kh = __hip_atomic_load(&keyh[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!quad) {
            if (mask & 1) {
                const U64x2* p = (const U64x2*)(rec + (size_t)par * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) { U64x2 v = p[k]; st[2 * k] = v.a; st[2 * k + 1] = v.b; }
            }
            if (mask & 2) {
                const U64x2* q = (const U64x2*)(path + (size_t)par * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) { U64x2 v = q[k]; pa[2 * k] = v.a; pa[2 * k + 1] = v.b; }
            }
        } else {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int slot = __shfl(par, qb + c4, 64);
                if (mask & 1) { U64x2 v = ((const U64x2*)(rec + (size_t)slot * 8))[pr]; st[2 * c4] = v.a; st[2 * c4 + 1] = v.b; }
                if (mask & 2) { U64x2 w = ((const U64x2*)(path + (size_t)slot * 8))[pr]; pa[2 * c4] = w.a; pa[2 * c4 + 1] = w.b; }
            }
        }
        uint64_t y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = (st[k] & (0x9E3779B97F4A7C15ULL * (k + 1 + r))) ^ pa[k] ^ kh;
        const uint64_t h = y[0] * 31 + y[3];
        if (mask & 4) rec[(size_t)par * 8 + vw] = y[vw];
        if (mask & 8) __hip_atomic_store(&keyh[par], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!quad) {
            if (mask & 32) {
                U64x2* q = (U64x2*)(path + (size_t)ny * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = U64x2{pa[2 * k], pa[2 * k + 1] | 1};
            }
            if (mask & 16) {
                U64x2* p = (U64x2*)(rec + (size_t)ny * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) p[k] = U64x2{y[2 * k], y[2 * k + 1]};
            }
        } else {
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int slot = __shfl(ny, qb + c4, 64);
                if (mask & 32) ((U64x2*)(path + (size_t)slot * 8))[pr] = U64x2{pa[2 * c4], pa[2 * c4 + 1] | 1};
                if (mask & 16) ((U64x2*)(rec + (size_t)slot * 8))[pr] = U64x2{y[2 * c4], y[2 * c4 + 1]};
            }
        }
        if (mask & 64) __hip_atomic_store(&keyh[ny], h + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mask & 128) ev[(size_t)r * NTH + threadIdx.x] = U32x4{(uint32_t)par, (uint32_t)ny, (uint32_t)h, (uint32_t)y[1]};
        __threadfence_block();
        __syncthreads();
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    int nblocks = argc > 1 ? atoi(argv[1]) : 512;
    uint8_t* base; unsigned long long* cyc;
    const size_t per = (size_t)CAPS * (64 + 64 + 8) + (size_t)ROUNDS * NTH * 16;
    CK(hipMalloc(&base, per * nblocks)); CK(hipMemset(base, 1, per * nblocks));
    CK(hipMalloc(&cyc, nblocks * 8));
    std::vector<unsigned long long> h(nblocks);
    const int masks[] = {1023 - 256, 1023, 1023 - 256 - 2 - 32, 1023 - 2 - 32, 1, 2, 512, 4, 8, 16, 32, 64, 128, 4 + 8, 16 + 64, 16 + 256, 32 + 256, 1 + 256, 1 + 2 + 512,
                         1 + 2 + 512 + 256, 4 + 8 + 16 + 32 + 64 + 128, 4 + 8 + 16 + 32 + 64 + 128 + 256, 4 + 8 + 16 + 64 + 128 + 256};
    for (int rep = 0; rep < 2; ++rep)
        for (int m : masks) {
            hipLaunchKernelGGL(k_parts, dim3(nblocks), dim3(NTH), 0, 0, base, m, cyc);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h.data(), cyc, nblocks * 8, hipMemcpyDeviceToHost));
            double s = 0; for (auto x : h) s += x;
            printf("mask %4d [%s%s%s%s%s%s%s%s%s%s]: %9.0f cycles per round of %d nodes\n", m, m & 1 ? "recR " : "", m & 2 ? "pathR " : "", m & 512 ? "khR " : "", m & 4 ? "noW " : "",
                   m & 8 ? "noKH " : "", m & 16 ? "yesRec " : "", m & 32 ? "yesPath " : "", m & 64 ? "yesKH " : "", m & 128 ? "ev " : "", m & 256 ? "QUAD" : "", s / nblocks / ROUNDS, NTH);
        }
    return 0;
}
