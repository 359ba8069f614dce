// flow_p_launch.cpp -- instantiations and launchers of the two flow kernels (flow_persist.hpp: large maps; flow_sweep.hpp: the others).
// Their own translation unit because it is compiled WITHOUT the SLP vectorizer (build.py UNIT_FLAGS): see the note at the top of
// flow_persist.hpp.
#include <stdexcept>

#include "flow_persist.hpp"
#include "flow_sweep.hpp"
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

// flow_kernel<NT> (flow_sweep.hpp): NT = 2 (64 targets per workgroup) while their planes fit the LDS, else 1; grid = workgroups
void launch_flow(const FlowArgs& a, int NT, size_t lds, unsigned grid, hipStream_t s) {
    if (NT == 2) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(flow_kernel<2>), lds);
        hipLaunchKernelGGL(flow_kernel<2>, dim3(grid), dim3(64 * kFlowWaves), lds, s, a);
    } else if (NT == 1) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(flow_kernel<1>), lds);
        hipLaunchKernelGGL(flow_kernel<1>, dim3(grid), dim3(64 * kFlowWaves), lds, s, a);
    } else {
        throw std::invalid_argument("flow: 32 or 64 targets per workgroup");
    }
}

}  // namespace tsnet
