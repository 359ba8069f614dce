// norm_elementwise.hpp -- HBM-bound kernels around the convolutions (NHWC fp32, 16 B per lane).
//
// Replaces on the reference hot path:
//   * nn.InstanceNorm2d statistics (affine=False, eps=1e-5, biased variance; TSNet.py:53 default
//     norm_layer)                                                        -> in_stats_partial + in_finalize
//   * the ResnetBlock tail  x + IN(conv(..))   (TSNet.py:47-49)          -> norm_act_kernel(resid)
//   * Encoder tail  ReLU(IN(.)) materialised once (TSNet.py:70-71)       -> norm_act_kernel(relu)
//   * FuseNet residual + the mean over sources (TSNet.py:195-200, :400)  -> fuse_resid_mean_kernel
//   * nn.Upsample(x2, bilinear, align_corners=False) (TSNet.py:145)      -> upsample2x_kernel
//   * set_test_input's /255 + torch.cat + coord_conv (TSNet.py:286,312,107-125) -> pack_input_kernel
// All reductions use a fixed order (no float atomics): results are run-to-run deterministic.
#pragma once
#include <hip/hip_runtime.h>

#include "conv_common.hpp"

namespace tsnet {

// ---------------------------------------------------------------------------------------------
// InstanceNorm statistics, stage 1: per (image n, split s) partial sum / sum of squares per channel,
// accumulated in fp64.  x: (N, HW, C).  part: (N, S, C, 2) doubles.
// Thread layout: cq = float4 column within the row, rg = row group; block covers all C/4 columns
// (or 256 of them; grid.z walks the rest).
struct StatsArgs {
    const float* x;
    double* part;
    int HW, C, S, rows_per_split;
};

__global__ __launch_bounds__(256) void in_stats_partial_kernel(StatsArgs a) {
    __shared__ double red[256 * 8];
    const int cq_total = a.C >> 2;
    const int cols = cq_total < 256 ? cq_total : 256;   // float4 columns handled per block
    const int R = 256 / cols;                            // row groups
    const int tid = threadIdx.x;
    const int cq = tid % cols + blockIdx.z * 256;
    const int rg = tid / cols;
    const int n = blockIdx.y, s = blockIdx.x;
    const int r0 = s * a.rows_per_split;
    int r1 = r0 + a.rows_per_split;
    if (r1 > a.HW) r1 = a.HW;
    double sm[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
    if (rg < R && cq < cq_total) {
        const float* base = a.x + ((size_t)n * a.HW) * a.C + (size_t)cq * 4;
        for (int r = r0 + rg; r < r1; r += R) {
            const float4 v = *reinterpret_cast<const float4*>(base + (size_t)r * a.C);
            sm[0] += v.x; sq[0] += (double)v.x * v.x;
            sm[1] += v.y; sq[1] += (double)v.y * v.y;
            sm[2] += v.z; sq[2] += (double)v.z * v.z;
            sm[3] += v.w; sq[3] += (double)v.w * v.w;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[tid * 8 + e] = sm[e]; red[tid * 8 + 4 + e] = sq[e]; }
    __syncthreads();
    if (rg == 0 && cq < cq_total) {
        double tsm[4] = {0, 0, 0, 0}, tsq[4] = {0, 0, 0, 0};
        for (int g = 0; g < R; ++g) {
            const int t = g * cols + tid;
#pragma unroll
            for (int e = 0; e < 4; ++e) { tsm[e] += red[t * 8 + e]; tsq[e] += red[t * 8 + 4 + e]; }
        }
        double* o = a.part + (((size_t)n * a.S + s) * a.C + (size_t)cq * 4) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e * 2] = tsm[e]; o[e * 2 + 1] = tsq[e]; }
    }
}

// The same reduction fused with an addition: y[n] = x[n] + add[n % add_nmod], statistics of y.  FuseNet's first convolution is split at
// the channel concat (TSNet.py:195-197: cat(src_fea, tar_fea) -> conv): the per-source half is computed once per source set
// (tsnet_set_sources, cached in clip mode), the shared target half once per driving frame; this kernel joins them and produces the
// InstanceNorm statistics of the sum.  One-shot forward and clip mode run the SAME kernels in the same order: bit-identical results.
struct AddStatsArgs {
    const float* x;      // (N, HW, C)
    const float* add;    // (add_nmod, HW, C)
    float* y;            // (N, HW, C)
    double* part;        // (N, S, C, 2)
    int HW, C, S, rows_per_split, add_nmod;
};

__global__ __launch_bounds__(256) void add_stats_partial_kernel(AddStatsArgs a) {
    __shared__ double red[256 * 8];
    const int cq_total = a.C >> 2;
    const int cols = cq_total < 256 ? cq_total : 256;
    const int R = 256 / cols;
    const int tid = threadIdx.x;
    const int cq = tid % cols + blockIdx.z * 256;
    const int rg = tid / cols;
    const int n = blockIdx.y, s = blockIdx.x;
    const int r0 = s * a.rows_per_split;
    int r1 = r0 + a.rows_per_split;
    if (r1 > a.HW) r1 = a.HW;
    double sm[4] = {0, 0, 0, 0}, sq[4] = {0, 0, 0, 0};
    if (rg < R && cq < cq_total) {
        const size_t cb = (size_t)cq * 4;
        const float* bx = a.x + ((size_t)n * a.HW) * a.C + cb;
        const float* ba = a.add + ((size_t)(n % a.add_nmod) * a.HW) * a.C + cb;
        float* by = a.y + ((size_t)n * a.HW) * a.C + cb;
        for (int r = r0 + rg; r < r1; r += R) {
            const float4 u = *reinterpret_cast<const float4*>(bx + (size_t)r * a.C);
            const float4 w = *reinterpret_cast<const float4*>(ba + (size_t)r * a.C);
            float4 v;
            v.x = u.x + w.x; v.y = u.y + w.y; v.z = u.z + w.z; v.w = u.w + w.w;
            *reinterpret_cast<float4*>(by + (size_t)r * a.C) = v;
            sm[0] += v.x; sq[0] += (double)v.x * v.x;
            sm[1] += v.y; sq[1] += (double)v.y * v.y;
            sm[2] += v.z; sq[2] += (double)v.z * v.z;
            sm[3] += v.w; sq[3] += (double)v.w * v.w;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[tid * 8 + e] = sm[e]; red[tid * 8 + 4 + e] = sq[e]; }
    __syncthreads();
    if (rg == 0 && cq < cq_total) {
        double tsm[4] = {0, 0, 0, 0}, tsq[4] = {0, 0, 0, 0};
        for (int g = 0; g < R; ++g) {
            const int t = g * cols + tid;
#pragma unroll
            for (int e = 0; e < 4; ++e) { tsm[e] += red[t * 8 + e]; tsq[e] += red[t * 8 + 4 + e]; }
        }
        double* o = a.part + (((size_t)n * a.S + s) * a.C + (size_t)cq * 4) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e * 2] = tsm[e]; o[e * 2 + 1] = tsq[e]; }
    }
}

// stage 2: alpha = 1/sqrt(var+eps), beta = -mean*alpha  (the x*alpha+beta form ATen's CPU
// batch-norm transform uses), one thread per (n, c).
__global__ void in_finalize_kernel(const double* __restrict__ part, float* __restrict__ alpha, float* __restrict__ beta,
                                   int NC, int C, int S, int HW, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC) return;
    const int n = i / C, c = i - n * C;
    double sm = 0, sq = 0;
    for (int s = 0; s < S; ++s) {
        const double* p = part + (((size_t)n * S + s) * C + c) * 2;
        sm += p[0];
        sq += p[1];
    }
    const double mean = sm / HW;
    double var = sq / HW - mean * mean;
    if (var < 0) var = 0;
    const float al = 1.0f / sqrtf((float)var + eps);
    alpha[i] = al;
    beta[i] = -((float)mean) * al;
}

// stage 2 for many partials (conv-epilogue statistics: S = tiles per image, up to 1024):
// block = (image n, 16 channels); 16 thread groups split the S partials (group g takes s = g, g + 16, ...), fixed-order LDS combine.
// (Four groups of 64 channels left 12 workgroups walking 128 dependent additions each on the 7 x 7 stem: 16 us of latency per launch.)
constexpr int kFin2Ch = 16, kFin2Groups = 16;
__global__ __launch_bounds__(256) void in_finalize2_kernel(const double* __restrict__ part, float* __restrict__ alpha,
                                                            float* __restrict__ beta, int C, int S, int HW, float eps) {
    __shared__ double red[256 * 2];
    const int n = blockIdx.y, cl = threadIdx.x & (kFin2Ch - 1), c = blockIdx.x * kFin2Ch + cl, g = threadIdx.x / kFin2Ch;
    double sm = 0, sq = 0;
    if (c < C) {
        // the loads of 8 partials are issued before the first addition (explicitly: an `unroll 8` of the one-pair loop still compiled to a
        // memory round trip per partial -- 77 us for the 512 partials of the stem); the additions stay in order, entries past S contribute +0.0
        for (int s0 = g; s0 < S; s0 += 8 * kFin2Groups) {
            double v[8], w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int s = s0 + u * kFin2Groups;
                const double* p = part + (((size_t)n * S + (s < S ? s : g)) * C + c) * 2;
                v[u] = p[0]; w[u] = p[1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool in = s0 + u * kFin2Groups < S;
                sm += in ? v[u] : 0.0;
                sq += in ? w[u] : 0.0;
            }
        }
    }
    red[threadIdx.x * 2] = sm; red[threadIdx.x * 2 + 1] = sq;
    __syncthreads();
    if (g == 0 && c < C) {
        for (int k = 1; k < kFin2Groups; ++k) { sm += red[(k * kFin2Ch + cl) * 2]; sq += red[(k * kFin2Ch + cl) * 2 + 1]; }
        const double mean = sm / HW;
        double var = sq / HW - mean * mean;
        if (var < 0) var = 0;
        const float al = 1.0f / sqrtf((float)var + eps);
        alpha[(size_t)n * C + c] = al;
        beta[(size_t)n * C + c] = -((float)mean) * al;
    }
}

// ---------------------------------------------------------------------------------------------
// y = alpha*x + beta (optionally ReLU) (+ resid).  alpha==null -> y = x (+resid).  In-place safe.
struct NormActArgs {
    const float* x;
    const float* alpha;
    const float* beta;
    const float* resid;
    float* y;
    int HW, C, relu;
    size_t total4;   // N*HW*C/4
};

// grid = (chunks, N): the image comes from blockIdx.y and the channel quad is advanced incrementally -- no per-element
// division (a 64-bit i / per_img, i % c4n per float4 made this kernel VALU-bound at 3.2 TB/s).  When the grid stride
// is a multiple of C/4 (always, for the power-of-two widths of the model) a thread stays on one channel quad and the
// scale/shift pair is loaded once.
__global__ __launch_bounds__(256) void norm_act_kernel(NormActArgs a) {
    const unsigned c4n = (unsigned)a.C >> 2;
    const unsigned per_img = (unsigned)a.HW * c4n;
    const unsigned n = blockIdx.y;
    const unsigned stride = gridDim.x * 256u;
    const unsigned sc = stride % c4n;
    unsigned j = blockIdx.x * 256u + threadIdx.x;
    unsigned cq = j % c4n;
    const bool fixed_c = sc == 0 && a.alpha != nullptr;
    float4 al = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fixed_c) {
        al = *reinterpret_cast<const float4*>(a.alpha + (size_t)n * a.C + cq * 4);
        be = *reinterpret_cast<const float4*>(a.beta + (size_t)n * a.C + cq * 4);
    }
    for (; j < per_img; j += stride) {
        const size_t i = (size_t)n * per_img + j;
        float4 v = reinterpret_cast<const float4*>(a.x)[i];
        if (a.alpha) {
            if (!fixed_c) {
                al = *reinterpret_cast<const float4*>(a.alpha + (size_t)n * a.C + cq * 4);
                be = *reinterpret_cast<const float4*>(a.beta + (size_t)n * a.C + cq * 4);
                cq += sc; if (cq >= c4n) cq -= c4n;
            }
            v.x = __builtin_fmaf(v.x, al.x, be.x);
            v.y = __builtin_fmaf(v.y, al.y, be.y);
            v.z = __builtin_fmaf(v.z, al.z, be.z);
            v.w = __builtin_fmaf(v.w, al.w, be.w);
        }
        if (a.relu) {
            v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
            v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
        }
        if (a.resid) {
            const float4 r = reinterpret_cast<const float4*>(a.resid)[i];
            v.x = r.x + v.x; v.y = r.y + v.y; v.z = r.z + v.z; v.w = r.w + v.w;
        }
        reinterpret_cast<float4*>(a.y)[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// FuseNet tail: zbar[b,p,:] = (1/K) * sum_i ( cat(src_fea[i*B+b], tar_fea[b])[p,:] + IN(y2[i*B+b])[p,:] )
// The 1x1 `fuse_net.conv` that follows is linear, so it is applied once to the mean
// (mean_i conv(z_i)+bias == conv(mean_i z_i)+bias).
struct FuseTailArgs {
    const float* src_fea;   // (K*B, P, C1)
    const float* tar_fea;   // (B, P, C1)
    const float* y2;        // (K*B, P, 2*C1)
    const float* alpha;     // (K*B * 2*C1)
    const float* beta;
    float* zbar;            // (B, P, 2*C1)
    int B, K, P, C1;
};

__global__ __launch_bounds__(256) void fuse_resid_mean_kernel(FuseTailArgs a) {
    const int C = 2 * a.C1;
    const int c4n = C >> 2;
    const size_t total = (size_t)a.B * a.P * c4n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const size_t bp = i / c4n;                  // b*P + p
        const int b = (int)(bp / a.P);
        const int p = (int)(bp - (size_t)b * a.P);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < a.K; ++s) {
            const int n = s * a.B + b;
            float4 xr;
            if (c < a.C1) xr = *reinterpret_cast<const float4*>(a.src_fea + ((size_t)n * a.P + p) * a.C1 + c);
            else          xr = *reinterpret_cast<const float4*>(a.tar_fea + ((size_t)b * a.P + p) * a.C1 + (c - a.C1));
            const float4 y = *reinterpret_cast<const float4*>(a.y2 + ((size_t)n * a.P + p) * C + c);
            const float4 al = *reinterpret_cast<const float4*>(a.alpha + (size_t)n * C + c);
            const float4 be = *reinterpret_cast<const float4*>(a.beta + (size_t)n * C + c);
            acc.x += xr.x + __builtin_fmaf(y.x, al.x, be.x);
            acc.y += xr.y + __builtin_fmaf(y.y, al.y, be.y);
            acc.z += xr.z + __builtin_fmaf(y.z, al.z, be.z);
            acc.w += xr.w + __builtin_fmaf(y.w, al.w, be.w);
        }
        const float kf = (float)a.K;
        acc.x /= kf; acc.y /= kf; acc.z /= kf; acc.w /= kf;
        *reinterpret_cast<float4*>(a.zbar + ((size_t)b * a.P + p) * C + c) = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Bilinear x2 upsample, align_corners=False: src = (dst+0.5)/2-0.5 clamped at 0, neighbour index
// clamped at in-1; horizontal lerp first, then vertical (ATen's separable order).  Optional
// producer InstanceNorm+ReLU applied to the four taps on load.
struct UpsampleArgs {
    const float* x;       // (N,H,W,C)
    const float* alpha;   // (N*C) or null
    const float* beta;
    float* y;             // (N,2H,2W,C)
    int N, H, W, C, relu;
    int x_bf16, y_bf16;   // bf16 storage mode: x / y hold bf16 (ConvArgs)
};

// four consecutive channels of an fp32 tensor, or of the same tensor stored as bf16 (exact widening)
__device__ __forceinline__ float4 ld4_f32_or_bf16(const float* base, size_t elem, bool b16) {
    if (!b16) return *reinterpret_cast<const float4*>(base + elem);
    const uint2 p = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem);
    return make_float4(__builtin_bit_cast(float, p.x << 16), __builtin_bit_cast(float, p.x & 0xFFFF0000u),
                       __builtin_bit_cast(float, p.y << 16), __builtin_bit_cast(float, p.y & 0xFFFF0000u));
}

__device__ __forceinline__ float4 na4(float4 v, const float4& al, const float4& be, bool norm, bool relu) {
    if (norm) {
        v.x = __builtin_fmaf(v.x, al.x, be.x); v.y = __builtin_fmaf(v.y, al.y, be.y);
        v.z = __builtin_fmaf(v.z, al.z, be.z); v.w = __builtin_fmaf(v.w, al.w, be.w);
    }
    if (relu) {
        v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
        v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
    }
    return v;
}

// grid = (chunks of one output row, 2H, N): row and image come from the block index, one 32-bit division per element
__global__ __launch_bounds__(256) void upsample2x_kernel(UpsampleArgs a) {
    const unsigned c4n = (unsigned)a.C >> 2;
    const int Ho = 2 * a.H, Wo = 2 * a.W;
    const int oy = blockIdx.y, n = blockIdx.z;
    const unsigned row4 = (unsigned)Wo * c4n;
    for (unsigned j = blockIdx.x * 256u + threadIdx.x; j < row4; j += gridDim.x * 256u) {
        const int ox = (int)(j / c4n);
        const int c = (int)(j - (unsigned)ox * c4n) * 4;
        const size_t i = ((size_t)n * Ho + oy) * row4 + j;
        float sy = (oy + 0.5f) * 0.5f - 0.5f; if (sy < 0.f) sy = 0.f;
        float sx = (ox + 0.5f) * 0.5f - 0.5f; if (sx < 0.f) sx = 0.f;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < a.H - 1 ? 1 : 0), x1 = x0 + (x0 < a.W - 1 ? 1 : 0);
        const float wy1 = sy - y0, wy0 = 1.f - wy1, wx1 = sx - x0, wx0 = 1.f - wx1;
        const bool norm = a.alpha != nullptr;
        float4 al = make_float4(1.f, 1.f, 1.f, 1.f), be = make_float4(0.f, 0.f, 0.f, 0.f);
        if (norm) {
            al = *reinterpret_cast<const float4*>(a.alpha + (size_t)n * a.C + c);
            be = *reinterpret_cast<const float4*>(a.beta + (size_t)n * a.C + c);
        }
        const size_t base = ((size_t)n * a.H * a.W) * a.C + c;
        const bool xb = a.x_bf16 != 0;
        const float4 v00 = na4(ld4_f32_or_bf16(a.x, base + ((size_t)y0 * a.W + x0) * a.C, xb), al, be, norm, a.relu);
        const float4 v01 = na4(ld4_f32_or_bf16(a.x, base + ((size_t)y0 * a.W + x1) * a.C, xb), al, be, norm, a.relu);
        const float4 v10 = na4(ld4_f32_or_bf16(a.x, base + ((size_t)y1 * a.W + x0) * a.C, xb), al, be, norm, a.relu);
        const float4 v11 = na4(ld4_f32_or_bf16(a.x, base + ((size_t)y1 * a.W + x1) * a.C, xb), al, be, norm, a.relu);
        float4 o;
        o.x = wy0 * (wx0 * v00.x + wx1 * v01.x) + wy1 * (wx0 * v10.x + wx1 * v11.x);
        o.y = wy0 * (wx0 * v00.y + wx1 * v01.y) + wy1 * (wx0 * v10.y + wx1 * v11.y);
        o.z = wy0 * (wx0 * v00.z + wx1 * v01.z) + wy1 * (wx0 * v10.z + wx1 * v11.z);
        o.w = wy0 * (wx0 * v00.w + wx1 * v01.w) + wy1 * (wx0 * v10.w + wx1 * v11.w);
        if (a.y_bf16) {
            uint2 q;
            q.x = TSNET_CVT_PK_BF16(o.x, o.y);
            q.y = TSNET_CVT_PK_BF16(o.z, o.w);
            reinterpret_cast<uint2*>(a.y)[i] = q;
        } else {
            reinterpret_cast<float4*>(a.y)[i] = o;
        }
    }
}

// bf16 -> fp32 (debug accessor of the bf16-storage mode)
__global__ void bf16_widen_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = __builtin_bit_cast(float, (unsigned)x[i] << 16);
}

// ---------------------------------------------------------------------------------------------
// Stem input assembly: NCHW planes -> one NHWC tensor with Cp channels
//   [ img/255 (nimg) | lbl (L) | xx yy rr (if coords) | 0 ... ]   image index n = s*B + b.
struct PackArgs {
    const float* img[8];   // per source: (B,3,H,W) or null (label encoder input)
    const float* lbl[8];   // per source: (B,L,H,W)
    const float* coords;   // (H,W,3) table or null
    float* out;            // (S*B, H, W, Cp)
    int S, B, H, W, L, nimg, Cp;
    float img_div[8];      // per source: 255 (set_test_input's /255, TSNet.py:286) or 1 (use_prev: a frame already in [0,1], TSNet.py:269-276)
    unsigned* amax_out;    // null, or amax_out[image] <- max |packed value| of that image (float bits; operand scale of the fp16 x 2 stem)
};

// grid = (blocks per image, S * B images).  Eight channels at a time: their eight loads are issued TOGETHER -- each from a wave-uniform plane
// pointer picked by selects, a dummy address for a padding channel -- and the /255 follows in a second pass.  (Written as one branch per
// channel kind and element, the round-5 kernel compiled to eight dependent load -> branch -> load links per pixel: 39 us for the 12 source
// images of the headline forward, 1 TB/s.)
__global__ __launch_bounds__(256) void pack_input_kernel(PackArgs a) {
    const size_t HW = (size_t)a.H * a.W;
    const int creal = a.nimg + a.L + (a.coords ? 3 : 0);
    const int n = blockIdx.y;
    const int s = n / a.B, b = n - s * a.B;
    const float* const img_b = a.nimg ? a.img[s] + (size_t)b * a.nimg * HW : a.lbl[s];
    const float* const lbl_b = a.lbl[s] + (size_t)b * a.L * HW;
    const float* const dummy = a.lbl[s];                         // any readable address: the value is discarded
    const float div = a.img_div[s];
    float vmax = 0.f;
    for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < HW; pix += (size_t)gridDim.x * blockDim.x) {
        float* o = a.out + ((size_t)n * HW + pix) * a.Cp;
        for (int c0 = 0; c0 < a.Cp; c0 += 8) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {                        // every pointer / stride below is wave-uniform
                const int c = c0 + e;
                const bool is_img = c < a.nimg, is_lbl = !is_img && c < a.nimg + a.L, is_crd = !is_img && !is_lbl && c < creal;
                const float* p = is_img ? img_b + (size_t)c * HW : (is_lbl ? lbl_b + (size_t)(c - a.nimg) * HW : (is_crd ? a.coords + (c - a.nimg - a.L) : dummy));
                const size_t idx = is_crd ? pix * 3 : ((is_img || is_lbl) ? pix : 0);
                const float t = p[idx];
                v[e] = (is_img || is_lbl || is_crd) ? t : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (c0 + e < a.nimg) v[e] = v[e] / div;          // set_test_input's /255 (TSNet.py:286): IEEE division, as torch divides
                vmax = __builtin_fmaxf(vmax, __builtin_fabsf(v[e]));
            }
            *reinterpret_cast<float4*>(o + c0) = make_float4(v[0], v[1], v[2], v[3]);          // Cp = 8 or a multiple of 16 (conv_cin_pad)
            *reinterpret_cast<float4*>(o + c0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
    if (a.amax_out) tsnet_publish_amax(a.amax_out + n, vmax);      // per image; block-uniform branch: every thread of the workgroup arrives
}

}  // namespace tsnet
