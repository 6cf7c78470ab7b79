// =============================================================================
// kernels.hpp -- entry points of the gfx950 kernels, one translation unit per
// group so that the build compiles them in parallel.
// =============================================================================
#pragma once
#include "dd_types.h"

namespace ddo_hip {

typedef void (*kernel_fn)(EngineParams);

/// in-place engine (misp_dd_inplace.hpp): wsT in {1, 2, 4, 7, 8, 16}; threads <= 512 picks the 256-VGPR variant
kernel_fn pick_kernel2(int wsT, int threads);
/// layer-rebuilding engine (misp_dd_core.hpp): wsT in {1, 2, 4, 7, 8, 16, 32}; dedup table in LDS or in HBM
kernel_fn pick_kernel(int wsT, bool table_in_lds);

// the per-unit halves
kernel_fn pick_kernel2_1024(int wsT);
kernel_fn pick_kernel2_512(int wsT);
/// capacity tiers: workgroups of 64..256 threads, 3 waves per SIMD (168 VGPRs: no spills, 12 waves per CU)
kernel_fn pick_kernel2_tier(int wsT);
/// waves per SIMD the capacity-tier kernel is compiled for (its register budget): 4 x that many waves share a CU
int tier_waves_per_simd();
/// Pooled decision diagrams (mdd/pooled.rs) out of the in-place engine's node slots: 512 threads, one decision diagram per CU
kernel_fn pick_kernel2_pooled(int wsT);
/// two full-width DDs per CU: 512 threads, 4 waves per SIMD
kernel_fn pick_kernel2_dense(int wsT);
kernel_fn pick_kernel_lds(int wsT);
kernel_fn pick_kernel_glb(int wsT);

}  // namespace ddo_hip
