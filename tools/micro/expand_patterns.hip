// Microbenchmark for the store / load patterns of the in-place engine's expand and work-list phases, at the occupancy of
// the dense tier (2 workgroups of 512 threads per CU).  Every workgroup owns a slot-like region (records, paths,
// word-major rows, hashes, keys) of CAPS nodes; one "round" handles one branching node per thread at random slots,
// then fences and synchronises -- like one pass of the expand loop.
//   A  today:       YES-child: record 4 x 16 B by one lane, 7-word word-major scatter, hash 8 B, key 4 B, path 4 x 16 B;
//                   NO-child: record word 8 B + record hash 8 B + hash 8 B + word-major 8 B + key 4 B
//   B  E1+E2':      YES: record and path written by 4 lanes x 16 B (quad transpose), 7-word scatter, key|hash 8 B;
//                   NO: record word 8 B + key|hash 8 B + word-major 8 B
//   C  B w/o word-major copy
// and the work-list sweep over N slots: (s1) two coalesced 8-byte streams, (s2) one 16-byte piece of every 64-byte record.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct alignas(16) U64x2 { uint64_t a, b; };
constexpr int CAPS = 20008;
constexpr int ROUNDS = 48;
constexpr int NTH = 512;

__device__ inline uint32_t rng(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

struct Region { uint64_t *rec, *path, *wm, *hsh, *keyh; uint32_t* key; };
__device__ inline Region region(uint8_t* base, int b) {
    Region r;
    const size_t per = (size_t)CAPS * (64 + 64 + 56 + 8 + 8 + 8);
    uint8_t* p = base + (size_t)b * per;
    r.rec = (uint64_t*)p; p += (size_t)CAPS * 64;
    r.path = (uint64_t*)p; p += (size_t)CAPS * 64;
    r.wm = (uint64_t*)p; p += (size_t)CAPS * 56;
    r.hsh = (uint64_t*)p; p += (size_t)CAPS * 8;
    r.keyh = (uint64_t*)p; p += (size_t)CAPS * 8;
    r.key = (uint32_t*)p;
    return r;
}

template <int MODE>
__global__ void __launch_bounds__(NTH) k_expand(uint8_t* base, unsigned long long* cyc) {
    Region R = region(base, blockIdx.x);
    uint32_t seed = blockIdx.x * 7919u + threadIdx.x * 31u + 1;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < ROUNDS; ++r) {
        const int par = rng(seed) % CAPS;   // parent slot (NO-child in place)
        const int ny = rng(seed) % CAPS;    // YES-child slot
        const int vw = r % 7;
        // parent record + path read
        uint64_t st[8], pa[8];
        if (MODE == 0) {
            const U64x2* p = (const U64x2*)(R.rec + (size_t)par * 8);
            const U64x2* q = (const U64x2*)(R.path + (size_t)par * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) { U64x2 v = p[k]; st[2 * k] = v.a; st[2 * k + 1] = v.b; }
#pragma unroll
            for (int k = 0; k < 4; ++k) { U64x2 v = q[k]; pa[2 * k] = v.a; pa[2 * k + 1] = v.b; }
        } else {
            // quad-cooperative: lane (4q + r) reads piece r of the records of the quad's 4 nodes (4 loads), no transpose
            // needed for timing purposes: the data volume and request shape are what count
            const int lane = threadIdx.x & 63, qb = lane & ~3, pr = lane & 3;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int slot = __shfl(par, qb + c4, 64);
                U64x2 v = ((const U64x2*)(R.rec + (size_t)slot * 8))[pr];
                U64x2 w = ((const U64x2*)(R.path + (size_t)slot * 8))[pr];
                st[2 * c4] = v.a; st[2 * c4 + 1] = v.b; pa[2 * c4] = w.a; pa[2 * c4 + 1] = w.b;
            }
        }
        uint64_t y[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) y[k] = st[k] & (0x9E3779B97F4A7C15ULL * (k + 1 + r)) ^ pa[k];
        const uint64_t h = y[0] * 31 + y[3];
        if (MODE == 0) {
            // NO-child
            R.rec[(size_t)par * 8 + vw] = y[vw];
            R.rec[(size_t)par * 8 + 7] = h;
            R.hsh[par] = h;
            R.wm[(size_t)vw * CAPS + par] = y[vw];
            R.key[par] = (uint32_t)h;
            // YES-child
            U64x2* p = (U64x2*)(R.rec + (size_t)ny * 8);
            U64x2* q = (U64x2*)(R.path + (size_t)ny * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = U64x2{pa[2 * k], pa[2 * k + 1] | 1};
#pragma unroll
            for (int k = 0; k < 4; ++k) p[k] = U64x2{y[2 * k], y[2 * k + 1]};
#pragma unroll
            for (int k = 0; k < 7; ++k) R.wm[(size_t)k * CAPS + ny] = y[k];
            R.hsh[ny] = h + 1;
            R.key[ny] = (uint32_t)h + 1;
        } else {
            R.rec[(size_t)par * 8 + vw] = y[vw];
            R.keyh[par] = h;
            if (MODE == 1) R.wm[(size_t)vw * CAPS + par] = y[vw];
            const int lane = threadIdx.x & 63, qb = lane & ~3, pr = lane & 3;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int slot = __shfl(ny, qb + c4, 64);
                ((U64x2*)(R.path + (size_t)slot * 8))[pr] = U64x2{pa[2 * c4], pa[2 * c4 + 1] | 1};
                ((U64x2*)(R.rec + (size_t)slot * 8))[pr] = U64x2{y[2 * c4], y[2 * c4 + 1]};
            }
            if (MODE == 1) {
#pragma unroll
                for (int k = 0; k < 7; ++k) R.wm[(size_t)k * CAPS + ny] = y[k];
            }
            R.keyh[ny] = h + 1;
        }
        __threadfence_block();
        __syncthreads();
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// sweep over N slots: MODE 0 = row word + hash streams (coalesced); MODE 1 = 16-byte piece of each 64-byte record + key|hash stream;
// MODE 2 = 16-byte piece only
template <int MODE>
__global__ void __launch_bounds__(NTH) k_sweep(uint8_t* base, int N, uint64_t* out, unsigned long long* cyc) {
    Region R = region(base, blockIdx.x);
    uint64_t acc = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < ROUNDS; ++r) {
        const int vw = r % 7;
        for (int b0 = 0; b0 < N; b0 += NTH * 4) {
            uint64_t a[4], h[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = b0 + q * NTH + threadIdx.x;
                if (MODE == 0) {
                    a[q] = s < N ? R.wm[(size_t)vw * CAPS + s] : 0;
                    h[q] = s < N ? R.hsh[s] : 0;
                } else {
                    U64x2 v = s < N ? ((const U64x2*)(R.rec + (size_t)s * 8))[vw >> 1] : U64x2{0, 0};
                    a[q] = (vw & 1) ? v.b : v.a;
                    h[q] = (MODE == 1 && s < N) ? R.keyh[s] : 0;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc += a[q] * 3 + h[q];
        }
        __syncthreads();
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[(size_t)blockIdx.x * NTH + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main(int argc, char** argv) {
    int nblocks = argc > 1 ? atoi(argv[1]) : 512;
    int N = argc > 2 ? atoi(argv[2]) : 10000;
    uint8_t* base; uint64_t* out; unsigned long long* cyc;
    const size_t per = (size_t)CAPS * (64 + 64 + 56 + 8 + 8 + 8);
    CK(hipMalloc(&base, per * nblocks)); CK(hipMemset(base, 1, per * nblocks));
    CK(hipMalloc(&out, (size_t)nblocks * NTH * 8));
    CK(hipMalloc(&cyc, nblocks * 8));
    std::vector<unsigned long long> h(nblocks);
    auto report = [&](const char* name, double per_round_items) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), cyc, nblocks * 8, hipMemcpyDeviceToHost));
        double s = 0; for (auto x : h) s += x;
        printf("%-58s blocks %d: %9.0f cycles per round (%.0f items)\n", name, nblocks, s / nblocks / ROUNDS, per_round_items);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_expand<0>, dim3(nblocks), dim3(NTH), 0, 0, base, cyc); report("expand A (today: one lane per node, hsh, word-major)", NTH);
        hipLaunchKernelGGL(k_expand<1>, dim3(nblocks), dim3(NTH), 0, 0, base, cyc); report("expand B (quad records/paths, key|hash, word-major)", NTH);
        hipLaunchKernelGGL(k_expand<2>, dim3(nblocks), dim3(NTH), 0, 0, base, cyc); report("expand C (B without the word-major copy)", NTH);
        hipLaunchKernelGGL(k_sweep<0>, dim3(nblocks), dim3(NTH), 0, 0, base, N, out, cyc); report("sweep s1 (row word + hash streams)", N);
        hipLaunchKernelGGL(k_sweep<1>, dim3(nblocks), dim3(NTH), 0, 0, base, N, out, cyc); report("sweep s2 (16 B of every record + key|hash stream)", N);
        hipLaunchKernelGGL(k_sweep<2>, dim3(nblocks), dim3(NTH), 0, 0, base, N, out, cyc); report("sweep s3 (16 B of every record only)", N);
    }
    return 0;
}
