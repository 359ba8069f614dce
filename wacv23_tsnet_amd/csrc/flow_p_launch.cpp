// flow_p_launch.cpp -- instantiation and launcher of the large-map flow kernel (flow_persist.hpp).  Its own translation unit because it is
// compiled WITHOUT the SLP vectorizer (build.py UNIT_FLAGS): see the note at the top of flow_persist.hpp.
#include <stdexcept>

#include "flow_persist.hpp"
#include "kernels.hpp"

namespace tsnet {

void launch_flow_p(const FlowArgs& a, int variant, hipStream_t s) {
    const size_t lds = flowp_lds_bytes(a.h, a.w, a.C);
    const dim3 grid(a.B * (a.P / 64) * a.G), block(64 * kFlowWaves);
#ifdef TSNET_TOOLS
    if (variant == 2) {                          // ablation: no exp / accumulate pass
        ensure_dynamic_lds(reinterpret_cast<const void*>(flow_kernel_p<2>), lds);
        hipLaunchKernelGGL(flow_kernel_p<2>, grid, block, lds, s, a);
        return;
    }
#endif
    if (variant) throw std::invalid_argument("flow: experiment variants exist in the tools build only");
    ensure_dynamic_lds(reinterpret_cast<const void*>(flow_kernel_p<0>), lds);
    hipLaunchKernelGGL(flow_kernel_p<0>, grid, block, lds, s, a);
}

}  // namespace tsnet
