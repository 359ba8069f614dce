// conv_g64_launch.cpp -- instantiations and launcher of the general implicit GEMM with 64-deep K steps (conv_g64.hpp).
#include <stdexcept>

#include "conv_g64.hpp"
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

}  // namespace tsnet
