// conv_w1_launch.cpp -- instantiations and launcher of the Winograd-along-x convolution (conv_w1.hpp).
#include <stdexcept>

#include "conv_w1.hpp"
#include "conv_w1_one.hpp"
#include "kernels.hpp"

namespace tsnet {

namespace {

// a launch of one tile per workgroup runs the one-tile kernel (conv_w1_one.hpp: round 4's, shorter prologue); chunks run conv_w1.hpp's
template <int NPROD, bool AFFINE, int OPT>
void go_w1_one(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)w1_lds_bytes(a.Cin, NPROD == 1 ? 1 : 2, 1);
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv_w1_one_kernel<NPROD, AFFINE, OPT>), lds);
    hipLaunchKernelGGL((conv_w1_one_kernel<NPROD, AFFINE, OPT>), dim3(a.tiles_m * a.tiles_n), dim3(64 * kW1Waves), lds, s, a);
}

template <int NPROD, bool AFFINE, int OPT>
void go_w1_k(const ConvArgs& a, size_t lds, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv_w1_kernel<NPROD, AFFINE, OPT>), lds);
    const int c = a.w1_chunk > 1 ? a.w1_chunk : 1;
    if (c > 1 && (a.tiles_m % ((a.xcd_gn > 0 ? 8 / a.xcd_gn : 8) * c) || (a.xcd_gn > 0 && a.tiles_n % a.xcd_gn)))
        throw std::invalid_argument("conv(w1): the chunk size must divide every XCD's rows of the tile matrix");
    hipLaunchKernelGGL((conv_w1_kernel<NPROD, AFFINE, OPT>), dim3(a.tiles_m * a.tiles_n / c), dim3(64 * kW1Waves), lds, s, a);
}

template <int NPROD>
void go_w1(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)w1_lds_bytes(a.Cin, NPROD == 1 ? 1 : 2, a.w1_tab2 ? 2 : 1);
    if (NPROD == 1 && a.w1_chunk > 1) throw std::invalid_argument("conv(w1): chunks of several tiles need the two-plane stages");
    if (a.w1_chunk <= 1) {
        if (!a.in_alpha && !a.in_relu) go_w1_one<NPROD, false, 2>(a, s);
        else if (!a.in_alpha) go_w1_one<NPROD, false, 0>(a, s);
        else if (a.reflect) go_w1_one<NPROD, true, 0>(a, s);
        else go_w1_one<NPROD, true, 1>(a, s);
        return;
    }
    if constexpr (NPROD == 3) {                       // (one plane: chunks of one tile only -- the chunk kernel is not instantiated for it)
        if (!a.in_alpha && !a.in_relu) go_w1_k<NPROD, false, 2>(a, lds, s);
        else if (!a.in_alpha) go_w1_k<NPROD, false, 0>(a, lds, s);
        else if (a.reflect) go_w1_k<NPROD, true, 0>(a, lds, s);
        else go_w1_k<NPROD, true, 1>(a, lds, s);      // zero padding under a fused InstanceNorm: padded pixels re-zeroed after the transform
    }
}

}  // namespace

void launch_conv_w1(const ConvArgs& a, int nprod, int abl, hipStream_t s) {
    if (a.Cin > 2 * 64 * kW1Waves) throw std::invalid_argument("conv(w1): the prologue's table request covers two channels per thread");      // (the LDS budget stops at ~1200)
    if (abl) {
#ifdef TSNET_TOOLS
        const size_t lds = (size_t)w1_lds_bytes(a.Cin, 2, a.w1_tab2 ? 2 : 1);
        if (abl == 31 && nprod == 3 && a.in_alpha) { go_w1_k<3, true, 512>(a, lds, s); return; }   // time stamps, IN + ReLU input (reflect)
        if (nprod != 3 || a.in_alpha) throw std::invalid_argument("conv(w1): ablations are built for three products on a raw input");
#define TSNET_W1_ABL(A_) if (abl == A_) { go_w1_k<3, false, ((A_) << 4) | 2>(a, lds, s); return; }
        if (abl == 30) { go_w1_k<3, false, 512 | 2>(a, lds, s); return; }            // time stamps (tsnet_w1_prof_read), raw input
        TSNET_W1_ABL(1) TSNET_W1_ABL(2) TSNET_W1_ABL(4) TSNET_W1_ABL(3) TSNET_W1_ABL(7) TSNET_W1_ABL(8) TSNET_W1_ABL(15) TSNET_W1_ABL(16) TSNET_W1_ABL(17)
#undef TSNET_W1_ABL
        throw std::invalid_argument("conv(w1): this ablation is not instantiated");
#else
        throw std::invalid_argument("conv(w1): ablation variants are only built into the tools library");
#endif
    }
    if (nprod == 3) go_w1<3>(a, s);
    else if (nprod == 1) go_w1<1>(a, s);
    else throw std::invalid_argument("conv(w1): 1 (bf16 operands) or 3 products");
}

}  // namespace tsnet

#ifdef TSNET_TOOLS
// tools build: the time stamps of the last stamped launch (abl 30 / 31): [tile][wave][16] of s_memrealtime ticks (100 MHz); slot 8 = XCC_ID << 32 | HW_ID
extern "C" int tsnet_w1_prof_read(unsigned long long* out, int ntiles) {
    using namespace tsnet;
    if (!out || ntiles < 1 || ntiles > kW1ProfTiles) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_w1_prof), (size_t)ntiles * kW1Waves * kW1ProfSlots * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
#endif
