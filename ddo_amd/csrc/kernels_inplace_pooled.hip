// gfx950 kernel of the in-place engine compiling POOLED decision diagrams (mdd/pooled.rs:117-823; misp_dd_inplace.hpp:
// run_dd2<WS, DEEP, POOLED = 1>): one 512-thread workgroup per decision diagram (256 VGPRs per lane: no spills), one per CU --
// a pool is sized for as many nodes as the LDS dedup table admits, whatever the width its layers are squashed to.
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "misp_dd_inplace.hpp"

namespace ddo_hip {

template <int WS>
__global__ void __launch_bounds__(512) misp_compile_kernel2_pooled(EngineParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    static_assert(sizeof(DD2Ctx<WS>) <= DD2_CTX_BYTES, "DD2_CTX_BYTES");
    DD2Ctx<WS>& c = *(DD2Ctx<WS>*)lds;   // the context lives in LDS (misp_dd_inplace.hpp: DD2_CTX_BYTES)
    if (threadIdx.x == 0) dd2_bind<WS>(c, P, (int)blockIdx.x, lds + DD2_CTX_BYTES, (int)blockDim.x);
    __syncthreads();
    dd2_work_loop<WS, 1, 1>(c, P);
}

kernel_fn pick_kernel2_pooled(int wsT) {
    switch (wsT) {
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 1
        case 1: return misp_compile_kernel2_pooled<1>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 2
        case 2: return misp_compile_kernel2_pooled<2>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 4
        case 4: return misp_compile_kernel2_pooled<4>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 7
        case 7: return misp_compile_kernel2_pooled<7>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 8
        case 8: return misp_compile_kernel2_pooled<8>;
#endif
#if !defined(DDO_WS_ONLY) || DDO_WS_ONLY == 16
        case 16: return misp_compile_kernel2_pooled<16>;
#endif
        default: return nullptr;
    }
}

}  // namespace ddo_hip
