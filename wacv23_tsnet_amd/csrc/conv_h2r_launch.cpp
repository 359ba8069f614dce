// conv_h2r_launch.cpp -- instantiations and launcher of the general implicit-GEMM convolution (conv_h2r.hpp).
#include <stdexcept>

#include "conv_h2r.hpp"
#include "kernels.hpp"

namespace tsnet {
namespace {

template <int KS, int BN, int NPROD, bool SMALL>
void go(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)h2r_lds_bytes(a.Cin);
    if (a.in_alpha) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(conv_h2r_kernel<KS, BN, 2, 2, NPROD, true, SMALL>), lds);
        hipLaunchKernelGGL((conv_h2r_kernel<KS, BN, 2, 2, NPROD, true, SMALL>), dim3(a.tiles_m * a.tiles_n), dim3(256), lds, s, a);
    } else {
        ensure_dynamic_lds(reinterpret_cast<const void*>(conv_h2r_kernel<KS, BN, 2, 2, NPROD, false, SMALL>), lds);
        hipLaunchKernelGGL((conv_h2r_kernel<KS, BN, 2, 2, NPROD, false, SMALL>), dim3(a.tiles_m * a.tiles_n), dim3(256), lds, s, a);
    }
}

template <int NPROD>
void go_np(const ConvArgs& a, int ks, int bn, hipStream_t s) {
    const bool small = a.Cin == 8;
    if (bn == 128) {
        if (ks != 3 || small) throw std::invalid_argument("conv(h2r): 128-wide tiles are built for 3x3 layers with at least 16 input channels");
        go<3, 128, NPROD, false>(a, s);
        return;
    }
    if (bn != 64) throw std::invalid_argument("conv(h2r): tile width must be 64 or 128");
    switch (ks) {
        case 1: if (small) throw std::invalid_argument("conv(h2r): 1x1 layers need at least 16 input channels"); go<1, 64, NPROD, false>(a, s); break;
        case 3: if (small) go<3, 64, NPROD, true>(a, s); else go<3, 64, NPROD, false>(a, s); break;
        case 7: if (small) go<7, 64, NPROD, true>(a, s); else go<7, 64, NPROD, false>(a, s); break;
        default: throw std::invalid_argument("conv: kernel size must be 1, 3 or 7");
    }
}

}  // namespace

void launch_conv_h2r(const ConvArgs& a, int ks, int bn, int nprod, hipStream_t s) {
    if (nprod == 3) go_np<3>(a, ks, bn, s);
    else if (nprod == 1) go_np<1>(a, ks, bn, s);
    else throw std::invalid_argument("conv(h2r): 1 (bf16 operands) or 3 products");
}

}  // namespace tsnet
