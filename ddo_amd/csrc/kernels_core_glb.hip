// gfx950 kernels of the layer-rebuilding engine, dedup table in HBM (see kernels.hpp)
#include <hip/hip_runtime.h>

#include "kernels.hpp"
#include "misp_dd_core.hpp"

namespace ddo_hip {

template <int WS, bool TLDS>
__global__ void __launch_bounds__(1024) misp_compile_kernel(EngineParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    DDCtx<WS> c;
    dd_bind<WS, TLDS>(c, P, (int)blockIdx.x, lds, (int)blockDim.x);
    c.tid_ = (int)threadIdx.x;
    dd_stage_tables<WS>(c, P);
    for (;;) {
        if (threadIdx.x == 0) c.sh->work = atomicAdd(P.work_counter, 1);
        __syncthreads();
        const int w = c.sh->work;
        __syncthreads();
        if (w >= P.nbatch) break;
        run_work_item<WS>(c, P.inputs[w], P.results + 2 * (size_t)w);
    }
}

kernel_fn pick_kernel_glb(int wsT) {
    switch (wsT) {
        case 1: return misp_compile_kernel<1, false>;
        case 2: return misp_compile_kernel<2, false>;
        case 4: return misp_compile_kernel<4, false>;
        case 7: return misp_compile_kernel<7, false>;
        case 8: return misp_compile_kernel<8, false>;
        case 16: return misp_compile_kernel<16, false>;
        case 32: return misp_compile_kernel<32, false>;   // signed-vector models only (MAX2SAT n <= 62)
        case 72: return misp_compile_kernel<72, false>;   // signed-vector models up to n = 142 (frb15-9-x: n = 135)
        default: return nullptr;
    }
}

}  // namespace ddo_hip
