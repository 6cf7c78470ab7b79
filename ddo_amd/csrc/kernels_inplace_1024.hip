// gfx950 kernels of the in-place engine, 1024-thread workgroups (see kernels.hpp)
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "misp_dd_inplace.hpp"

namespace ddo_hip {

// MAXT = 512 lets the register allocator use 256 VGPRs (the 1024-thread variant is capped at 128 and spills)
template <int WS, int MAXT>
__global__ void __launch_bounds__(MAXT) misp_compile_kernel2(EngineParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    static_assert(sizeof(DD2Ctx<WS>) <= DD2_CTX_BYTES, "DD2_CTX_BYTES");
    DD2Ctx<WS>& c = *(DD2Ctx<WS>*)lds;   // the context lives in LDS (misp_dd_inplace.hpp: DD2_CTX_BYTES)
    if (threadIdx.x == 0) dd2_bind<WS>(c, P, (int)blockIdx.x, lds + DD2_CTX_BYTES, (int)blockDim.x);
    __syncthreads();
    dd2_work_loop<WS, 0, 0>(c, P);
}

kernel_fn pick_kernel2_1024(int wsT) {
    switch (wsT) {
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 1
        case 1: return misp_compile_kernel2<1, 1024>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 2
        case 2: return misp_compile_kernel2<2, 1024>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 4
        case 4: return misp_compile_kernel2<4, 1024>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 7
        case 7: return misp_compile_kernel2<7, 1024>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 8
        case 8: return misp_compile_kernel2<8, 1024>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 16
        case 16: return misp_compile_kernel2<16, 1024>;
#endif
        default: return nullptr;
    }
}

kernel_fn pick_kernel2(int wsT, int threads) { return threads <= 512 ? pick_kernel2_512(wsT) : pick_kernel2_1024(wsT); }

}  // namespace ddo_hip
