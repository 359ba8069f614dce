// pack_weights.hpp -- weights of a layer -> the operand planes the convolution kernels read (once, at tsnet_finalize).
#pragma once
#include "conv_common.hpp"

namespace tsnet {

// ---------------------------------------------------------------------------------------------------------------
// OIHW fp32 (kernel kh x kw) -> operand planes in MFMA fragment order:
//   out[p][((kc*Npad + n)*2 + o)*8 + e] = part_p( scale * W[k = kc*16 + (o ^ ((n>>3)&1))*8 + e][n] ),  k = tap*cin_pad + c
// A wave's fragment of 32 columns x 16 k is 1 KiB contiguous; the octet swizzle by bit 3 of the column is the one the A tile uses.
// planes = 2: fp16 (hi, lo) of w * scale;  planes = 1: one bf16 plane of w (bf16-operand mode, scale ignored)
__global__ void pack_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, float scale, int planes,
                                    int cout, int cin_real, int cin_pad, int kh, int kw, int kpad, int npad, int cin_total, int cin_off) {
    const size_t plane = (size_t)kpad * npad;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < plane; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 7;
        const int o = (idx >> 3) & 1;
        const size_t rest = idx >> 4;
        const int n = (int)(rest % npad);
        const int kc = (int)(rest / npad);
        const int k = kc * 16 + (o ^ ((n >> 3) & 1)) * 8 + e;
        const int tap = k / cin_pad, c = k - tap * cin_pad;
        float v = 0.f;
        if (tap < kh * kw && c < cin_real && n < cout) {
            const int ky = tap / kw, kx = tap - ky * kw;
            v = w[(((size_t)n * cin_total + cin_off + c) * kh + ky) * kw + kx];
        }
        if (planes == 1) {
            out[idx] = bf16_rne(v);
        } else {
            unsigned hi, lo;
            split_h2(v * scale, hi, lo);
            out[idx] = (unsigned short)hi; out[plane + idx] = (unsigned short)lo;
        }
    }
}

}  // namespace tsnet
