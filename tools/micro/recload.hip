// Microbenchmark: how long does one "round" of random 64-byte record reads take for a 1024-thread workgroup?
// (a) every thread reads its own record with 8 x 8-byte loads, (b) 4 x 16-byte loads, (c) 4 lanes share a record,
// one 16-byte load each (16 records per wave instruction), (d) 8 lanes x 8 bytes.  Also record stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct alignas(16) U64x2 { uint64_t a, b; };
constexpr int CAPS = 20008;
constexpr int ROUNDS = 64;

__device__ inline uint32_t rng(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>
__global__ void __launch_bounds__(1024) k_read(const uint64_t* __restrict__ rec, uint64_t* out, unsigned long long* cyc) {
    const uint64_t* my = rec + (size_t)blockIdx.x * CAPS * 8;
    uint32_t seed = blockIdx.x * 7919u + threadIdx.x * 31u + 1;
    uint64_t acc = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < ROUNDS; ++r) {
        if (MODE == 0) {
            int s = rng(seed) % CAPS;
            const uint64_t* p = my + (size_t)s * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= __builtin_nontemporal_load(p + k) * (k + 1);
        } else if (MODE == 1) {
            int s = rng(seed) % CAPS;
            const U64x2* p = (const U64x2*)(my + (size_t)s * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) { U64x2 v = p[k]; acc ^= v.a * (2 * k + 1) ^ v.b * (2 * k + 2); }
        } else if (MODE == 2) {   // 4 lanes per record, 4 rounds of records per "item round" to fetch the same number of records
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t sd = blockIdx.x * 7919u + (threadIdx.x >> 2) * 131u + r * 4 + q;
                int s = rng(sd) % CAPS;
                const U64x2* p = (const U64x2*)(my + (size_t)s * 8) + (threadIdx.x & 3);
                U64x2 v = *p;
                acc ^= v.a ^ (v.b * 3);
            }
        } else if (MODE == 3) {   // 8 lanes per record, 8 bytes each; 8 sub-rounds
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                uint32_t sd = blockIdx.x * 7919u + (threadIdx.x >> 3) * 131u + r * 8 + q;
                int s = rng(sd) % CAPS;
                acc ^= my[(size_t)s * 8 + (threadIdx.x & 7)] * (q + 1);
            }
        }
        __syncthreads();
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
__global__ void __launch_bounds__(1024) k_write(uint64_t* __restrict__ rec, unsigned long long* cyc) {
    uint64_t* my = rec + (size_t)blockIdx.x * CAPS * 8;
    uint32_t seed = blockIdx.x * 7919u + threadIdx.x * 31u + 1;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < ROUNDS; ++r) {
        if (MODE == 0) {
            int s = rng(seed) % CAPS;
            uint64_t* p = my + (size_t)s * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) p[k] = seed + k;
        } else if (MODE == 1) {
            int s = rng(seed) % CAPS;
            U64x2* p = (U64x2*)(my + (size_t)s * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] = U64x2{seed + k, seed - k};
        } else if (MODE == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t sd = blockIdx.x * 7919u + (threadIdx.x >> 2) * 131u + r * 4 + q;
                int s = rng(sd) % CAPS;
                U64x2* p = (U64x2*)(my + (size_t)s * 8) + (threadIdx.x & 3);
                *p = U64x2{sd, sd + 1};
            }
        } else if (MODE == 3) {  // word-major SoA scatter: 7 words to 7 different arrays (like st[k][slot])
            int s = rng(seed) % CAPS;
#pragma unroll
            for (int k = 0; k < 7; ++k) my[(size_t)k * CAPS + s] = seed + k;
        }
        __syncthreads();
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    int nblocks = argc > 1 ? atoi(argv[1]) : 256;
    uint64_t *rec, *out; unsigned long long* cyc;
    size_t bytes = (size_t)nblocks * CAPS * 64;
    CK(hipMalloc(&rec, bytes)); CK(hipMemset(rec, 1, bytes));
    CK(hipMalloc(&out, (size_t)nblocks * 1024 * 8));
    CK(hipMalloc(&cyc, nblocks * 8));
    std::vector<unsigned long long> h(nblocks);
    auto report = [&](const char* name) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), cyc, nblocks * 8, hipMemcpyDeviceToHost));
        double s = 0; for (auto x : h) s += x;
        printf("%-34s blocks %d: %.0f cycles per round (1024 records)\n", name, nblocks, s / nblocks / ROUNDS);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_read<0>, dim3(nblocks), dim3(1024), 0, 0, rec, out, cyc); report("read  8x8B per thread");
        hipLaunchKernelGGL(k_read<1>, dim3(nblocks), dim3(1024), 0, 0, rec, out, cyc); report("read  4x16B per thread");
        hipLaunchKernelGGL(k_read<2>, dim3(nblocks), dim3(1024), 0, 0, rec, out, cyc); report("read  4 lanes x 16B per record");
        hipLaunchKernelGGL(k_read<3>, dim3(nblocks), dim3(1024), 0, 0, rec, out, cyc); report("read  8 lanes x 8B per record");
        hipLaunchKernelGGL(k_write<0>, dim3(nblocks), dim3(1024), 0, 0, rec, cyc); report("write 8x8B per thread");
        hipLaunchKernelGGL(k_write<1>, dim3(nblocks), dim3(1024), 0, 0, rec, cyc); report("write 4x16B per thread");
        hipLaunchKernelGGL(k_write<2>, dim3(nblocks), dim3(1024), 0, 0, rec, cyc); report("write 4 lanes x 16B per record");
        hipLaunchKernelGGL(k_write<3>, dim3(nblocks), dim3(1024), 0, 0, rec, cyc); report("write 7 words SoA scatter");
    }
    return 0;
}
