// kernels.hpp -- launchers of the convolution kernels and the large-map flow kernel.  Each kernel family lives in its own translation unit
// (conv_h2_launch.cpp, conv_h2r_launch.cpp, conv_g64_launch.cpp, conv_w1_launch.cpp, flow_p_launch.cpp) so that the library builds in parallel and the MFMA
// kernels build without the SLP vectorizer; engine.cpp holds the host logic and the small kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "conv_common.hpp"
#include "flow_args.hpp"

namespace tsnet {

// conv_h2.hpp -- patch kernels.  a.tiles_m / tiles_n / tpi are set by the caller; throws std::invalid_argument for a combination that
// is not instantiated.  nprod: 1 (bf16 operands), 3, or 4 (h2 only); affine = a.in_alpha != null.
//   h2  (3x3 / stride 1): pr x bn in {4x32, 4x64, 4x128, 2x128}; abl / opt: tools build only (ablation / experiment masks)
//   h2s (7x7 stem, 8 input channels): 4 x 64
//   h2d (3x3 / stride 2): bn = 64 (four waves) or 128 (eight waves)
void launch_conv_h2(const ConvArgs& a, int pr, int bn, int nprod, int abl, int opt, hipStream_t s);
void launch_conv_h2s(const ConvArgs& a, int nprod, hipStream_t s);
void launch_conv_h2d(const ConvArgs& a, int pr, int bn, int nprod, bool deep, hipStream_t s);
// conv_w1.hpp -- 3x3 / stride 1 / pad 1 as Winograd F(2,3) along x: 4 x 32 pixels x 64 channels per tile (eight waves); the layer's
// weights are the TRANSFORMED filters (12 "taps": tap row ky x position p, pack_weights_kernel with kh = 3, kw = 4)
void launch_conv_w1(const ConvArgs& a, int nprod, int abl, hipStream_t s);      // abl: tools build only
// conv_h2r.hpp -- general implicit GEMM: ks in {1, 3, 7}, bn = 64 (any) or 128 (ks = 3, Cin >= 16); Cin = 8 or a power of two >= 16
void launch_conv_h2r(const ConvArgs& a, int ks, int bn, int nprod, hipStream_t s);
// conv_g64.hpp -- the same GEMM in 64-deep K steps (Cin, and the concat split, multiples of 64; Npad a multiple of 128): ks in {1, 3},
// bm = 64 (four waves) or 128 (eight waves) rows x 128 channels; the same bits as conv_h2r
void launch_conv_g64(const ConvArgs& a, int ks, int bm, int nprod, hipStream_t s);
// conv_h2s32.hpp -- 7 x 7 stems at 32 raw input channels (the pose model) as a patch kernel: 4 x 32 pixels x 64 channels; the same bits as conv_h2r
void launch_conv_h2s32(const ConvArgs& a, int nprod, hipStream_t s);

// flow_persist.hpp -- flow_kernel_p: a.K, a.G (flowp_plan), a.part, a.cnt set by the caller; variant: tools build only
void launch_flow_p(const FlowArgs& a, int variant, hipStream_t s);
// flow_sweep.hpp -- flow_kernel<NT>: NT = 1 or 2 target blocks per workgroup, `grid` workgroups, `lds` = flow_lds_bytes(NT, h, w, C)
void launch_flow(const FlowArgs& a, int NT, size_t lds, unsigned grid, hipStream_t s);

// dynamic LDS above the default limit needs hipFuncAttributeMaxDynamicSharedMemorySize: set once per (kernel, device), not per launch
void ensure_dynamic_lds(const void* kernel, size_t bytes);

}  // namespace tsnet
