// conv_h2_launch.cpp -- instantiations and launchers of the patch convolution kernels (conv_h2.hpp).
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>

#include "conv_h2.hpp"
#include "kernels.hpp"

namespace tsnet {

void ensure_dynamic_lds(const void* kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return;
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, size_t> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    size_t& have = done[std::make_pair(kernel, dev)];
    if (have >= bytes) return;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) throw std::runtime_error(std::string("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: ") + hipGetErrorString(e));
    have = bytes;
}

namespace {

template <int PR, int BN, int WM, int WN, int NPROD, bool AFFINE, int HABL, int OPT>
void go_h2_k(const ConvArgs& a, size_t lds, int threads, hipStream_t s) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(conv_h2_kernel<PR, BN, WM, WN, NPROD, AFFINE, HABL, OPT>), lds);
    hipLaunchKernelGGL((conv_h2_kernel<PR, BN, WM, WN, NPROD, AFFINE, HABL, OPT>), dim3(a.tiles_m * a.tiles_n), dim3(threads), lds, s, a);
}

// OPT bit 0 (re-zero the padded pixels after the affine transform) is the launcher's: set for a zero-padding layer with a fused InstanceNorm
template <int PR, int BN, int WM, int WN, int NPROD, int HABL = 0, int OPT = 0>
void go_h2(const ConvArgs& a, hipStream_t s) {
    constexpr int KG = (OPT & 16) ? 2 : 1;
    static_assert((OPT & 1) == 0, "bit 0 is chosen here");
    if ((a.Cin >> 4) % KG) throw std::invalid_argument("conv(h2): two K groups need an even number of 16-channel slabs");
    const size_t lds = (size_t)h2_lds_bytes(PR, a.Cin, KG);
    if (!a.in_alpha) go_h2_k<PR, BN, WM, WN, NPROD, false, HABL, OPT>(a, lds, 256 * KG, s);
    else if (a.reflect) go_h2_k<PR, BN, WM, WN, NPROD, true, HABL, OPT>(a, lds, 256 * KG, s);
    else go_h2_k<PR, BN, WM, WN, NPROD, true, HABL, OPT | 1>(a, lds, 256 * KG, s);
}

template <int NPROD>
void go_h2_shape(const ConvArgs& a, int pr, int bn, hipStream_t s) {
    if (pr == 4 && bn == 32) go_h2<4, 32, 4, 1, NPROD>(a, s);
    else if (pr == 4 && bn == 64) go_h2<4, 64, 2, 2, NPROD>(a, s);
    else if (pr == 4 && bn == 128) go_h2<4, 128, 2, 2, NPROD>(a, s);
    else if (pr == 2 && bn == 128) go_h2<2, 128, 1, 4, NPROD>(a, s);
    else if (pr == 5 && bn == 128) {                 // 4 rows x 128 channels with the four waves side by side (1 x 4, wave tile 128 x 32): every weight fragment is loaded once per workgroup
        if constexpr (NPROD == 1) go_h2<4, 128, 1, 4, 1>(a, s);
        else throw std::invalid_argument("conv(h2): the 1 x 4 wave grid of the 4 x 128 tile is built for bf16 operands");
    }
    else throw std::invalid_argument("conv(h2): tile must be 4x32, 4x64, 4x128 or 2x128");
}

template <int BN, int NWV, int NPROD, int PR = kPatchRows, bool DEEP = false>
void go_h2d(const ConvArgs& a, hipStream_t s) {
    const size_t lds = (size_t)h2d_lds_bytes(PR) + (size_t)2 * a.Cin * 4;
    if (a.in_alpha) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(conv_h2d_kernel<BN, NWV, NPROD, true, PR, DEEP>), lds);
        hipLaunchKernelGGL((conv_h2d_kernel<BN, NWV, NPROD, true, PR, DEEP>), dim3(a.tiles_m * a.tiles_n), dim3(64 * NWV), lds, s, a);
    } else {
        ensure_dynamic_lds(reinterpret_cast<const void*>(conv_h2d_kernel<BN, NWV, NPROD, false, PR, DEEP>), lds);
        hipLaunchKernelGGL((conv_h2d_kernel<BN, NWV, NPROD, false, PR, DEEP>), dim3(a.tiles_m * a.tiles_n), dim3(64 * NWV), lds, s, a);
    }
}

}  // namespace

void launch_conv_h2(const ConvArgs& a, int pr, int bn, int nprod, int abl, int opt, hipStream_t s) {
    if (!abl && opt == 24) {           // single-frame launches: deep prefetch + two K groups, eight waves (conv_h2.hpp)
        if (nprod != 1 && nprod != 3) throw std::invalid_argument("conv(h2): the two-group tiles run 1 or 3 products");
        if (pr == 4 && bn == 32) { if (nprod == 3) go_h2<4, 32, 4, 1, 3, 0, 24>(a, s); else go_h2<4, 32, 4, 1, 1, 0, 24>(a, s); }
        else if (pr == 4 && bn == 64) { if (nprod == 3) go_h2<4, 64, 2, 2, 3, 0, 24>(a, s); else go_h2<4, 64, 2, 2, 1, 0, 24>(a, s); }
        else throw std::invalid_argument("conv(h2): the two-group tiles are 4x32 and 4x64");
        return;
    }
    if (abl || opt) {
#ifdef TSNET_TOOLS
        // experiment / ablation instantiations (tools/h2_variants.py): 3 products, raw or transformed input
        if (nprod == 1) {               // bf16 operands: the ablations of the 4 x 128 tile (what binds the bf16 modes' 3 x 3 kernel)
#define TSNET_H2_VAR1(A_) if (pr == 4 && bn == 128 && abl == A_ && opt == 0) { go_h2<4, 128, 2, 2, 1, A_, 0>(a, s); return; }
            TSNET_H2_VAR1(1) TSNET_H2_VAR1(2) TSNET_H2_VAR1(4) TSNET_H2_VAR1(3) TSNET_H2_VAR1(7) TSNET_H2_VAR1(16)
#undef TSNET_H2_VAR1
#define TSNET_H2_VAR1S(A_) if (pr == 5 && bn == 128 && abl == A_ && opt == 0) { go_h2<4, 128, 1, 4, 1, A_, 0>(a, s); return; }       // ... and of its side-by-side form
            TSNET_H2_VAR1S(1) TSNET_H2_VAR1S(2) TSNET_H2_VAR1S(4) TSNET_H2_VAR1S(3) TSNET_H2_VAR1S(7) TSNET_H2_VAR1S(16) TSNET_H2_VAR1S(8)
#undef TSNET_H2_VAR1S
            throw std::invalid_argument("conv(h2): this bf16 experiment variant is not instantiated");
        }
        if (nprod != 3) throw std::invalid_argument("conv(h2): experiment variants are built for one or three products");
#define TSNET_H2_VAR(PR_, BN_, WM_, WN_, A_, O_) if (pr == PR_ && bn == BN_ && abl == A_ && opt == O_) { go_h2<PR_, BN_, WM_, WN_, 3, A_, O_>(a, s); return; }
        TSNET_H2_VAR(4, 64, 2, 2, 0, 2) TSNET_H2_VAR(4, 64, 2, 2, 0, 4) TSNET_H2_VAR(4, 64, 2, 2, 0, 8) TSNET_H2_VAR(4, 64, 2, 2, 0, 16)
        TSNET_H2_VAR(4, 128, 2, 2, 0, 2) TSNET_H2_VAR(4, 128, 2, 2, 0, 4) TSNET_H2_VAR(4, 128, 2, 2, 0, 16) TSNET_H2_VAR(4, 32, 4, 1, 0, 8)
        TSNET_H2_VAR(4, 64, 2, 2, 0, 32) TSNET_H2_VAR(4, 64, 2, 2, 0, 64) TSNET_H2_VAR(4, 64, 2, 2, 0, 36) TSNET_H2_VAR(2, 128, 1, 4, 0, 32)
        TSNET_H2_VAR(4, 64, 2, 2, 1, 0) TSNET_H2_VAR(4, 64, 2, 2, 2, 0) TSNET_H2_VAR(4, 64, 2, 2, 4, 0) TSNET_H2_VAR(4, 64, 2, 2, 7, 0)
        TSNET_H2_VAR(4, 64, 2, 2, 8, 0) TSNET_H2_VAR(4, 64, 2, 2, 16, 0) TSNET_H2_VAR(4, 64, 2, 2, 15, 0) TSNET_H2_VAR(4, 64, 2, 2, 31, 0)
        TSNET_H2_VAR(2, 128, 1, 4, 1, 0) TSNET_H2_VAR(2, 128, 1, 4, 2, 0) TSNET_H2_VAR(2, 128, 1, 4, 7, 0)
#undef TSNET_H2_VAR
        throw std::invalid_argument("conv(h2): this experiment variant is not instantiated");
#else
        throw std::invalid_argument("conv(h2): experiment / ablation variants are only built into the tools library");
#endif
    }
    if (nprod == 3) go_h2_shape<3>(a, pr, bn, s);
    else if (nprod == 1) go_h2_shape<1>(a, pr, bn, s);
    else if (nprod == 4) {
        if (pr == 4 && bn == 64) go_h2<4, 64, 2, 2, 4>(a, s);
        else if (pr == 4 && bn == 128) go_h2<4, 128, 2, 2, 4>(a, s);
        else throw std::invalid_argument("conv(h2): four products on 4x64 or 4x128 tiles");
    } else throw std::invalid_argument("conv(h2): 1 (bf16 operands), 3 or 4 products");
}

void launch_conv_h2s(const ConvArgs& a, int nprod, hipStream_t s) {
    if (nprod == 3) hipLaunchKernelGGL((conv_h2s_kernel<3>), dim3(a.tiles_m * a.tiles_n), dim3(256), kH2sLds, s, a);
    else if (nprod == 1) hipLaunchKernelGGL((conv_h2s_kernel<1>), dim3(a.tiles_m * a.tiles_n), dim3(256), kH2sLds, s, a);
    else throw std::invalid_argument("conv(h2s): 1 (bf16 operands) or 3 products");
}

void launch_conv_h2d(const ConvArgs& a, int pr, int bn, int nprod, bool deep, hipStream_t s) {
    if (nprod != 1 && nprod != 3) throw std::invalid_argument("conv(h2d): 1 (bf16 operands) or 3 products");
    if (deep) {                        // launches that cannot fill the chip: weights eight steps ahead, the three staging rounds in flight together (conv_h2.hpp)
        if (pr != 2 || bn != 128 || nprod != 3) throw std::invalid_argument("conv(h2d): the deep schedule is the 2 x 128 tile's, fp16 x 2 operands");
        go_h2d<128, 4, 3, 2, true>(a, s);
        return;
    }
    if (pr == 4 && bn == 64) { if (nprod == 3) go_h2d<64, 4, 3>(a, s); else go_h2d<64, 4, 1>(a, s); }
    else if (pr == 4 && bn == 128) { if (nprod == 3) go_h2d<128, 8, 3>(a, s); else go_h2d<128, 8, 1>(a, s); }
    else if (pr == 2 && bn == 128) { if (nprod == 3) go_h2d<128, 4, 3, 2>(a, s); else go_h2d<128, 4, 1, 2>(a, s); }
    else throw std::invalid_argument("conv(h2d): tile must be 4x64 (four waves), 4x128 (eight waves) or 2x128 (four waves)");
}

}  // namespace tsnet
