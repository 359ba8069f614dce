// flow_warp.hpp -- the transformation branch: mask-aware correlation -> softmax(100*) ->
// soft-argmax flow -> bilinear warp + mean over sources.
//
// Replaces (model/TSNet.py): F.normalize (:319,339), the two masked torch.bmm (:350-358), F.softmax(100*., dim=2) (:359),
// get_grid + torch.matmul (:299-307, :362-365), F.grid_sample (:366) and the stack().mean() over sources (:392).  The reference
// materialises three B x P x P tensors per source; here the P x P matrix never exists.
//
//   l2norm_split_kernel   one wave per position: x / max(|x|, 1e-12) as before, then the row is scaled by 2^14 and written as two fp16
//                         planes (hi = rne(v), lo = rne(v - hi): conv_common.hpp) in MFMA FRAGMENT ORDER
//                         [image][32-position block][16-channel step][plane][lane = half * 32 + position][8 channels]:
//                         a wave's operand fragment is one contiguous, coalesced KiB.  Positions are padded to 64 and channels to 32 with zeros.
//   flow_kernel<NT>       a workgroup (eight waves) owns NT * 32 target positions of one (source, batch) image pair.  Their planes
//                         (64 x 512 channels x 2 planes = 128 KiB) are copied to LDS once; each wave then sweeps its own pairs of
//                         32-source blocks: source fragments come straight from global memory into registers (two 16-channel steps
//                         ahead), target fragments from LDS, 3 exact fp16 products per (source block, target block, step) on
//                         v_mfma_f32_32x32x16_f16 (lo*hi, hi*lo, hi*hi -- the convolution kernels' fp16 x 2 arithmetic: the dropped
//                         lo*lo term is <= 2^-22 |s||t|).  A = sources, B = targets: a lane owns target COLUMNS, so the softmax reduction
//                         over sources is in-lane; it keeps an online-softmax state per column whose "value" is the 2-vector grid
//                         coordinate, and the 16 partial states of a column (8 waves x 2 half-waves) are merged in a fixed order.
//
//   flow_kernel_p         (flow_persist.hpp, its own translation unit) the same sweep for LARGE maps (P >= 2048: BASELINE.json configs[4] is
//                         64 x 64 positions, five sources): a workgroup keeps its 64 target positions for all K sources of its batch element
//                         and sweeps 1 / G of every source image; tiles x G workgroups = one per CU.
//
// Masks: corr = (T.S) * (mt*ms + (1-mt)*(1-ms)), which equals the reference's sum of two masked
// products exactly for 0/1 masks and to rounding for soft masks; masked pairs stay at logit 0
// (NOT -inf), as in the reference.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_common.hpp"
#include "flow_args.hpp"

namespace tsnet {

// F.normalize(p=2, dim=channel, eps=1e-12) on NHWC rows -> fp16 (hi, lo) planes in fragment order.  grid = N * Ppad / 4 blocks of 4 waves.
__global__ __launch_bounds__(256) void l2norm_split_kernel(const float* __restrict__ x, unsigned short* __restrict__ q, int N, int P, int Ppad, int C, int KC) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int n = row / Ppad, p = row - n * Ppad;
    if (n >= N) return;
    const bool valid = p < P;
    const float* px = x + ((size_t)n * P + (valid ? p : 0)) * C;
    float ss = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(px + c);
        ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
    float nrm = sqrtf(ss);
    if (nrm < 1e-12f) nrm = 1e-12f;
    unsigned short* dst = q + ((size_t)(n * (Ppad >> 5) + (p >> 5)) * KC) * 1024 + (p & 31) * 8;
    for (int o = lane; o < KC * 2; o += 64) {                // channel octets, zero past C (and for the padded positions)
        F4 t0, t1;
#pragma unroll
        for (int e = 0; e < 4; ++e) { t0.v[e] = 0.f; t1.v[e] = 0.f; }
        if (valid && o * 8 < C) {
            t0 = *reinterpret_cast<const F4*>(px + o * 8);
            t1 = *reinterpret_cast<const F4*>(px + o * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { t0.v[e] = t0.v[e] / nrm * kFlowScale; t1.v[e] = t1.v[e] / nrm * kFlowScale; }
        }
        F4 Hh, Ll;
        split_h2_octet(t0, t1, Hh, Ll);
        unsigned short* d = dst + (o >> 1) * 1024 + (o & 1) * 256;
        *reinterpret_cast<F4*>(d) = Hh;
        *reinterpret_cast<F4*>(d + 512) = Ll;
    }
}


// grid_sample(bilinear, zeros, align_corners=False) of the UN-normalised source features at the
// flow, fused with the mean over sources.  NHWC makes every neighbour a contiguous C-float row:
// the "gather" is four fully coalesced row reads per (target, source).
struct WarpArgs {
    const float* src;    // (K*B, h, w, C), n = s*B+b
    const float* flow;   // (K*B, P, 2)
    float* out;          // (B, P, C)
    int B, K, h, w, C;
};

__global__ __launch_bounds__(256) void warp_mean_kernel(WarpArgs a) {
    const int P = a.h * a.w;
    const int c4n = a.C >> 2;
    const size_t total = (size_t)a.B * P * c4n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % c4n) * 4;
        const size_t bp = i / c4n;
        const int b = (int)(bp / P), p = (int)(bp - (size_t)b * P);
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < a.K; ++s) {
            const int n = s * a.B + b;
            const float gx = a.flow[((size_t)n * P + p) * 2 + 0];
            const float gy = a.flow[((size_t)n * P + p) * 2 + 1];
            const float ix = ((gx + 1.f) * a.w - 1.f) / 2.f;    // grid_sampler_unnormalize, align_corners=False
            const float iy = ((gy + 1.f) * a.h - 1.f) / 2.f;
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
            const float wnw = (x1 - ix) * (y1 - iy), wne = (ix - x0) * (y1 - iy);
            const float wsw = (x1 - ix) * (iy - y0), wse = (ix - x0) * (iy - y0);
            const float* base = a.src + ((size_t)n * P) * a.C + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool xin0 = x0 >= 0 && x0 < a.w, xin1 = x1 >= 0 && x1 < a.w;
            const bool yin0 = y0 >= 0 && y0 < a.h, yin1 = y1 >= 0 && y1 < a.h;
            if (yin0 && xin0) { const float4 q = *reinterpret_cast<const float4*>(base + ((size_t)y0 * a.w + x0) * a.C);
                v.x = q.x * wnw; v.y = q.y * wnw; v.z = q.z * wnw; v.w = q.w * wnw; }
            if (yin0 && xin1) { const float4 q = *reinterpret_cast<const float4*>(base + ((size_t)y0 * a.w + x1) * a.C);
                v.x = __builtin_fmaf(q.x, wne, v.x); v.y = __builtin_fmaf(q.y, wne, v.y); v.z = __builtin_fmaf(q.z, wne, v.z); v.w = __builtin_fmaf(q.w, wne, v.w); }
            if (yin1 && xin0) { const float4 q = *reinterpret_cast<const float4*>(base + ((size_t)y1 * a.w + x0) * a.C);
                v.x = __builtin_fmaf(q.x, wsw, v.x); v.y = __builtin_fmaf(q.y, wsw, v.y); v.z = __builtin_fmaf(q.z, wsw, v.z); v.w = __builtin_fmaf(q.w, wsw, v.w); }
            if (yin1 && xin1) { const float4 q = *reinterpret_cast<const float4*>(base + ((size_t)y1 * a.w + x1) * a.C);
                v.x = __builtin_fmaf(q.x, wse, v.x); v.y = __builtin_fmaf(q.y, wse, v.y); v.z = __builtin_fmaf(q.z, wse, v.z); v.w = __builtin_fmaf(q.w, wse, v.w); }
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        const float kf = (float)a.K;
        sum.x /= kf; sum.y /= kf; sum.z /= kf; sum.w /= kf;
        *reinterpret_cast<float4*>(a.out + ((size_t)b * P + p) * a.C + c) = sum;
    }
}

}  // namespace tsnet
