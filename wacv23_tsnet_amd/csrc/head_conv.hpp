// head_conv.hpp -- the decoder's RGB head: ReflectionPad2d(3) + Conv2d(64 -> 3, 7x7) + bias + Tanh
// (model/TSNet.py:151-152), with the producer's InstanceNorm+ReLU applied on load and, for the pose
// model, the fixed-background composite (model/TSNet_pose.py:416-417) in the epilogue.
//
// Why not the MFMA kernel: with 3 output channels the GEMM's N pads to 32, so 91 % of the matrix work
// is wasted (0.58 ms for 4.9 GFLOP).  Here every thread owns one output pixel and its 3 channels:
// the (16+6)x(16+6) input patch is staged through LDS 16 channels at a time in a [channel-quad][pixel]
// image (adjacent pixels = adjacent 16-byte slots: conflict-free ds_read_b128), and the weights are
// wave-uniform, so they arrive through the scalar cache as SGPR operands of the FMAs (12 FMA per LDS
// read).  VALU-bound: 2.47 GFMA per forward = ~35 us at the fp32 vector peak.
#pragma once
#include <hip/hip_runtime.h>

namespace tsnet {

struct HeadArgs {
    const float* x;       // (N,H,W,C) raw output of the last up-conv, NHWC
    const float* alpha;   // (N*C) InstanceNorm scale / shift of x (null = x is already an activation)
    const float* beta;
    const float* w;       // (7*7, C, 4) packed: [tap][cin][cout padded to 4]
    const float* bias;    // (3)
    float* y;             // (N,3,H,W) NCHW
    int N, H, W, C;
    int composite, fore_x0, fore_x1;
    float bg[3];
};

constexpr int kHeadT = 16, kHeadP = kHeadT + 6, kHeadCh = 16;

__global__ __launch_bounds__(256) void head_conv_kernel(HeadArgs a) {
    __shared__ float4 tile[(kHeadCh / 4) * kHeadP * kHeadP];     // [channel quad][22*22 pixels]
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int tiles_x = (a.W + kHeadT - 1) / kHeadT;
    const int n = blockIdx.y;
    const int bx = (blockIdx.x % tiles_x) * kHeadT, by = (blockIdx.x / tiles_x) * kHeadT;
    float tot0 = 0.f, tot1 = 0.f, tot2 = 0.f;
    for (int c0 = 0; c0 < a.C; c0 += kHeadCh) {
        // ---- stage the patch (reflection in the address, IN+ReLU on the value)
        for (int i = tid; i < (kHeadCh / 4) * kHeadP * kHeadP; i += 256) {
            const int q = i / (kHeadP * kHeadP), p = i - q * (kHeadP * kHeadP);
            const int py = p / kHeadP, px = p - py * kHeadP;
            int iy = by + py - 3, ix = bx + px - 3;
            iy = iy < 0 ? -iy : iy; iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
            ix = ix < 0 ? -ix : ix; ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
            iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);       // tiles hanging over the edge: any valid address
            ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
            const int c = c0 + q * 4;
            float4 v = *reinterpret_cast<const float4*>(a.x + (((size_t)n * a.H + iy) * a.W + ix) * a.C + c);
            if (a.alpha) {
                const float4 al = *reinterpret_cast<const float4*>(a.alpha + (size_t)n * a.C + c);
                const float4 be = *reinterpret_cast<const float4*>(a.beta + (size_t)n * a.C + c);
                v.x = __builtin_fmaf(v.x, al.x, be.x); v.y = __builtin_fmaf(v.y, al.y, be.y);
                v.z = __builtin_fmaf(v.z, al.z, be.z); v.w = __builtin_fmaf(v.w, al.w, be.w);
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            }
            tile[i] = v;
        }
        __syncthreads();
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;      // one fmaf chain per 16-channel slab (784 products), then folded
        for (int ky = 0; ky < 7; ++ky) {
            for (int kx = 0; kx < 7; ++kx) {
                const float* wt = a.w + ((size_t)(ky * 7 + kx) * a.C + c0) * 4;      // wave-uniform: scalar loads
                const int p = (ty + ky) * kHeadP + tx + kx;
#pragma unroll
                for (int q = 0; q < kHeadCh / 4; ++q) {
                    const float4 v = tile[q * (kHeadP * kHeadP) + p];
                    // whole 16-float rows (padding lane included): one wide scalar load per channel quad instead of
                    // twelve dword / dwordx2 loads, each of which forced an lgkmcnt(0) shared with the LDS reads
                    // (315 -> 275 us).  Four pixels per thread (4x fewer weight loads) was slower: 256 workgroups
                    // leave one wave per SIMD and the scalar-cache latency shows (298 us).
                    const float4* w4 = reinterpret_cast<const float4*>(wt + q * 16);
                    const float4 wx = w4[0], wy = w4[1], wz = w4[2], ww = w4[3];
                    a0 = __builtin_fmaf(v.x, wx.x, a0); a1 = __builtin_fmaf(v.x, wx.y, a1); a2 = __builtin_fmaf(v.x, wx.z, a2);
                    a0 = __builtin_fmaf(v.y, wy.x, a0); a1 = __builtin_fmaf(v.y, wy.y, a1); a2 = __builtin_fmaf(v.y, wy.z, a2);
                    a0 = __builtin_fmaf(v.z, wz.x, a0); a1 = __builtin_fmaf(v.z, wz.y, a1); a2 = __builtin_fmaf(v.z, wz.z, a2);
                    a0 = __builtin_fmaf(v.w, ww.x, a0); a1 = __builtin_fmaf(v.w, ww.y, a1); a2 = __builtin_fmaf(v.w, ww.z, a2);
                }
            }
        }
        tot0 += a0; tot1 += a1; tot2 += a2;
        __syncthreads();
    }
    const int ox = bx + tx, oy = by + ty;
    if (ox < a.W && oy < a.H) {
        float o[3] = {tanhf(tot0 + a.bias[0]), tanhf(tot1 + a.bias[1]), tanhf(tot2 + a.bias[2])};
        if (a.composite && (ox < a.fore_x0 || ox >= a.fore_x1)) { o[0] = a.bg[0]; o[1] = a.bg[1]; o[2] = a.bg[2]; }
        const size_t hw = (size_t)a.H * a.W;
#pragma unroll
        for (int c = 0; c < 3; ++c) a.y[((size_t)n * 3 + c) * hw + (size_t)oy * a.W + ox] = o[c];
    }
}

// Second generation (round 2).  PMC of the kernel above (profiles/round1_pmc_summary.txt): 0.28 ms for 35 us of FMAs -- every
// (tap, channel quad) iteration waits for a 64-byte scalar load AND an LDS read on the one lgkm counter (scalar loads return out of
// order, so the compiler drains it to zero each time), and the 22-pixel row pitch costs 12.8 M LDS bank conflicts per launch.  Here:
//   * the slab's weights (49 taps x 16 channels x (3+1) floats = 9.4 KB) are staged in LDS next to the patch and read as wave-uniform
//     broadcasts -- LDS returns in order, the compiler counts lgkmcnt and keeps several reads in flight;
//   * every thread owns TWO output pixels 16 columns apart (24 FMAs per 6 LDS reads instead of 12 per 2 + a scalar load);
//   * the patch row pitch is 48 pixels (768 B = 0 mod 256 B): the two rows a 16-lane read group touches never share a bank.
// Same fmaf chains per output (tap-major, then channel) as the first kernel: the two produce identical bits.
constexpr int kHead2W = 32, kHead2H = 16, kHead2Pitch = 48, kHead2Rows = kHead2H + 6;

__global__ __launch_bounds__(256) void head_conv2_kernel(HeadArgs a) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    float4* tile = reinterpret_cast<float4*>(smem_raw);                                   // [channel quad][22 rows][48 pixels]
    float4* wts = tile + (kHeadCh / 4) * kHead2Rows * kHead2Pitch;                        // [49 taps][16 channels]: (w_r, w_g, w_b, 0)
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int tiles_x = (a.W + kHead2W - 1) / kHead2W;
    const int n = blockIdx.y;
    const int bx = (blockIdx.x % tiles_x) * kHead2W, by = (blockIdx.x / tiles_x) * kHead2H;
    float tot[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    constexpr int PW = kHead2W + 6;                                                       // 38 patch columns in use
    for (int c0 = 0; c0 < a.C; c0 += kHeadCh) {
        for (int i = tid; i < (kHeadCh / 4) * kHead2Rows * PW; i += 256) {
            const int q = i / (kHead2Rows * PW), p = i - q * (kHead2Rows * PW);
            const int py = p / PW, px = p - py * PW;
            int iy = by + py - 3, ix = bx + px - 3;
            iy = iy < 0 ? -iy : iy; iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
            ix = ix < 0 ? -ix : ix; ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
            iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);       // tiles hanging over the edge: any valid address
            ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
            const int c = c0 + q * 4;
            float4 v = *reinterpret_cast<const float4*>(a.x + (((size_t)n * a.H + iy) * a.W + ix) * a.C + c);
            if (a.alpha) {
                const float4 al = *reinterpret_cast<const float4*>(a.alpha + (size_t)n * a.C + c);
                const float4 be = *reinterpret_cast<const float4*>(a.beta + (size_t)n * a.C + c);
                v.x = __builtin_fmaf(v.x, al.x, be.x); v.y = __builtin_fmaf(v.y, al.y, be.y);
                v.z = __builtin_fmaf(v.z, al.z, be.z); v.w = __builtin_fmaf(v.w, al.w, be.w);
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            }
            tile[(q * kHead2Rows + py) * kHead2Pitch + px] = v;
        }
        for (int i = tid; i < 49 * kHeadCh; i += 256) {
            const int tap = i / kHeadCh, c = i - tap * kHeadCh;
            wts[i] = *reinterpret_cast<const float4*>(a.w + ((size_t)tap * a.C + c0 + c) * 4);
        }
        __syncthreads();
        float acc[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};   // one fmaf chain per 16-channel slab (784 products), then folded
        for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const float4* wt = wts + (ky * 7 + kx) * kHeadCh;
#pragma unroll
                for (int q = 0; q < kHeadCh / 4; ++q) {
                    const float4* row = tile + (q * kHead2Rows + ty + ky) * kHead2Pitch + tx + kx;
                    const float4 v0 = row[0], v1 = row[16];
                    const float4 wx = wt[q * 4], wy = wt[q * 4 + 1], wz = wt[q * 4 + 2], ww = wt[q * 4 + 3];
                    const float pv[2][4] = {{v0.x, v0.y, v0.z, v0.w}, {v1.x, v1.y, v1.z, v1.w}};
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        acc[k][0] = __builtin_fmaf(pv[k][0], wx.x, acc[k][0]); acc[k][1] = __builtin_fmaf(pv[k][0], wx.y, acc[k][1]); acc[k][2] = __builtin_fmaf(pv[k][0], wx.z, acc[k][2]);
                        acc[k][0] = __builtin_fmaf(pv[k][1], wy.x, acc[k][0]); acc[k][1] = __builtin_fmaf(pv[k][1], wy.y, acc[k][1]); acc[k][2] = __builtin_fmaf(pv[k][1], wy.z, acc[k][2]);
                        acc[k][0] = __builtin_fmaf(pv[k][2], wz.x, acc[k][0]); acc[k][1] = __builtin_fmaf(pv[k][2], wz.y, acc[k][1]); acc[k][2] = __builtin_fmaf(pv[k][2], wz.z, acc[k][2]);
                        acc[k][0] = __builtin_fmaf(pv[k][3], ww.x, acc[k][0]); acc[k][1] = __builtin_fmaf(pv[k][3], ww.y, acc[k][1]); acc[k][2] = __builtin_fmaf(pv[k][3], ww.z, acc[k][2]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) { tot[k][0] += acc[k][0]; tot[k][1] += acc[k][1]; tot[k][2] += acc[k][2]; }
        __syncthreads();
    }
    const size_t hw = (size_t)a.H * a.W;
    const int oy = by + ty;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int ox = bx + tx + 16 * k;
        if (ox < a.W && oy < a.H) {
            float o[3] = {tanhf(tot[k][0] + a.bias[0]), tanhf(tot[k][1] + a.bias[1]), tanhf(tot[k][2] + a.bias[2])};
            if (a.composite && (ox < a.fore_x0 || ox >= a.fore_x1)) { o[0] = a.bg[0]; o[1] = a.bg[1]; o[2] = a.bg[2]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) a.y[((size_t)n * 3 + c) * hw + (size_t)oy * a.W + ox] = o[c];
        }
    }
}

// OIHW (3, C, 7, 7) -> [tap][cin][4]
__global__ void pack_head_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int C) {
    const int total = 49 * C * 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int o = i & 3, c = (i >> 2) % C, tap = (i >> 2) / C;
        out[i] = o < 3 ? w[((size_t)o * C + c) * 49 + tap] : 0.f;
    }
}

}  // namespace tsnet
