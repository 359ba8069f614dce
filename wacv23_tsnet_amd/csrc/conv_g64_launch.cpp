// conv_g64_launch.cpp -- instantiations and launchers of the general implicit GEMM with 64-deep K steps (conv_g64.hpp) and of the 32-channel
// stem patch kernel (conv_h2s32.hpp): the two kernels that took over conv_h2r's hot layers in round 6.
#include <stdexcept>

#include "conv_g64.hpp"
#include "conv_h2s32.hpp"
#include "kernels.hpp"

namespace tsnet {
namespace {

template <int KS, int BM, int NPROD>
void go(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)g64_lds_bytes(BM, NPROD == 1 ? 1 : 2, a.Cin);
    if (a.in_alpha) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(conv_g64_kernel<KS, BM, NPROD, true>), lds);
        hipLaunchKernelGGL((conv_g64_kernel<KS, BM, NPROD, true>), dim3(a.tiles_m * a.tiles_n), dim3(BM * 4), lds, s, a);
    } else {
        ensure_dynamic_lds(reinterpret_cast<const void*>(conv_g64_kernel<KS, BM, NPROD, false>), lds);
        hipLaunchKernelGGL((conv_g64_kernel<KS, BM, NPROD, false>), dim3(a.tiles_m * a.tiles_n), dim3(BM * 4), lds, s, a);
    }
}

template <int NPROD>
void go_np(const ConvArgs& a, int ks, int bm, hipStream_t s) {
    if (bm != 64 && bm != 128) throw std::invalid_argument("conv(g64): tiles have 64 or 128 rows");
    if (ks == 1) { if (bm == 64) go<1, 64, NPROD>(a, s); else go<1, 128, NPROD>(a, s); }
    else if (ks == 3) { if (bm == 64) go<3, 64, NPROD>(a, s); else go<3, 128, NPROD>(a, s); }
    else throw std::invalid_argument("conv(g64): kernel size must be 1 or 3");
}

}  // namespace

void launch_conv_g64(const ConvArgs& a, int ks, int bm, int nprod, hipStream_t s) {
    if ((a.Cin & 63) || (a.x2 && (a.Csplit & 63)) || (a.Npad & 127)) throw std::invalid_argument("conv(g64): input channels (and the concat split) must be multiples of 64, the padded width of 128");
    if (nprod == 3) go_np<3>(a, ks, bm, s);
    else if (nprod == 1) go_np<1>(a, ks, bm, s);
    else throw std::invalid_argument("conv(g64): 1 (bf16 operands) or 3 products");
}

void launch_conv_h2s32(const ConvArgs& a, int nprod, hipStream_t s) {
    if (a.Cin != 32 || a.taps != 49 || a.stride != 1 || a.pad != 3 || !a.reflect || a.in_alpha) throw std::invalid_argument("conv(h2s32): a 7 x 7 / reflection-pad-3 stem on 32 raw input channels");
    if (nprod == 3) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(conv_h2s32_kernel<3>), (size_t)h2s32_lds_bytes(2));
        hipLaunchKernelGGL((conv_h2s32_kernel<3>), dim3(a.tiles_m * a.tiles_n), dim3(256), (size_t)h2s32_lds_bytes(2), s, a);
    } else if (nprod == 1) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(conv_h2s32_kernel<1>), (size_t)h2s32_lds_bytes(1));
        hipLaunchKernelGGL((conv_h2s32_kernel<1>), dim3(a.tiles_m * a.tiles_n), dim3(256), (size_t)h2s32_lds_bytes(1), s, a);
    } else {
        throw std::invalid_argument("conv(h2s32): 1 (bf16 operands) or 3 products");
    }
}

}  // namespace tsnet
