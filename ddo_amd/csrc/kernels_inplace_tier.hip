// gfx950 kernels of the in-place engine for the capacity tiers (engine.hpp: Engine::create_tier): workgroups of 64 to 256
// threads, many of them per CU.  Narrow decision diagrams are all latency (a layer is a chain of dependent memory round
// trips whatever its size), so the lever is how many DDs a CU overlaps: 3 waves per SIMD leave 168 VGPRs per lane -- no
// spills -- and let 12 one-wave workgroups (or 3 of 256 threads) share a CU.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "misp_dd_inplace.hpp"

namespace ddo_hip {

template <int WS>
#if !defined(DDO_TIER_WAVES)
#define DDO_TIER_WAVES 3
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DDO_TIER_WAVES, DDO_TIER_WAVES))) misp_compile_kernel2_tier(EngineParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    static_assert(sizeof(DD2Ctx<WS>) <= DD2_CTX_BYTES, "DD2_CTX_BYTES");
    DD2Ctx<WS>& c = *(DD2Ctx<WS>*)lds;   // the context lives in LDS (misp_dd_inplace.hpp: DD2_CTX_BYTES)
    if (threadIdx.x == 0) dd2_bind<WS>(c, P, (int)blockIdx.x, lds + DD2_CTX_BYTES, (int)blockDim.x);
    __syncthreads();
    dd2_work_loop<WS, 0, 0>(c, P);
}

// Two decision diagrams per CU at full width: 512 threads each, 4 waves per SIMD (128 VGPRs like the 1024-thread kernel), so
// that two workgroups -- 16 waves -- share a CU and overlap each other's memory latency (Engine: "dense" configuration).
template <int WS>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) misp_compile_kernel2_dense(EngineParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    static_assert(sizeof(DD2Ctx<WS>) <= DD2_CTX_BYTES, "DD2_CTX_BYTES");
    DD2Ctx<WS>& c = *(DD2Ctx<WS>*)lds;   // the context lives in LDS (misp_dd_inplace.hpp: DD2_CTX_BYTES)
    if (threadIdx.x == 0) dd2_bind<WS>(c, P, (int)blockIdx.x, lds + DD2_CTX_BYTES, (int)blockDim.x);
    __syncthreads();
    dd2_work_loop<WS, 0, 0>(c, P);
}

kernel_fn pick_kernel2_dense(int wsT) {
    switch (wsT) {
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 1
        case 1: return misp_compile_kernel2_dense<1>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 2
        case 2: return misp_compile_kernel2_dense<2>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 4
        case 4: return misp_compile_kernel2_dense<4>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 7
        case 7: return misp_compile_kernel2_dense<7>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 8
        case 8: return misp_compile_kernel2_dense<8>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 16
        case 16: return misp_compile_kernel2_dense<16>;
#endif
        default: return nullptr;
    }
}

int tier_waves_per_simd() { return DDO_TIER_WAVES; }

kernel_fn pick_kernel2_tier(int wsT) {
    switch (wsT) {
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 1
        case 1: return misp_compile_kernel2_tier<1>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 2
        case 2: return misp_compile_kernel2_tier<2>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 4
        case 4: return misp_compile_kernel2_tier<4>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 7
        case 7: return misp_compile_kernel2_tier<7>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 8
        case 8: return misp_compile_kernel2_tier<8>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 16
        case 16: return misp_compile_kernel2_tier<16>;
#endif
        default: return nullptr;
    }
}

}  // namespace ddo_hip
