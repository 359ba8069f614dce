// train_extras.hpp -- the training-mode extras of the forward.
//
// Replaces (model/TSNet.py, is_train branches of forward()):
//   * :373-379  F.unfold(src_img, down) -> F.grid_sample(.., warp_grid2d) -> F.fold(.., down): every down x down patch
//               of the source image is moved as a unit by the flow of its feature position.  Here one gather: the
//               output pixel (Y, X) blends the SAME in-patch offset (Y % down, X % down) of the four neighbouring
//               source patches -- the 192-channel unfolded tensor (3 x 64 channels at 32 x 32) never exists.
//   * :329-330, :381-384  re-normalisation of the warped image to the target image's per-channel mean / unbiased std
//               (frame_stats_kernel of postproc.hpp + the elementwise pass below)
//   * :386,:390 loss_warp = sum_i 10 * L1(warp_src_img_i, tar_img)
//   * :403-405  loss_align = 1 - mean_p cos(pg[:, p], sg[:, p])
// Pose model (model/TSNet_pose.py:343-346, 386-404): same warp and re-normalisation, then the fixed-background composite of the
// warped image (:399-400) before the L1 (:402); no alignment loss.
// Reductions are fp64 in a fixed order (per-block partials, then one block): deterministic.
#pragma once
#include <hip/hip_runtime.h>

namespace tsnet {

struct PatchWarpArgs {
    const float* src[8];   // per source: (B, 3, H, W) raw images (the /255 of set_train_input is applied on load)
    const float* flow;     // (K*B, P, 2), n = s*B + b
    float* out;            // (K, B, 3, H, W)
    int K, B, H, W, h, w, down;
    float div[8];          // per source: 255, or 1 for a source that is already in [0,1] (use_prev, TSNet.py:269-276)
};

// one thread per output pixel (s, b, Y, X), three channels
__global__ __launch_bounds__(256) void patch_warp_kernel(PatchWarpArgs a) {
    const size_t HW = (size_t)a.H * a.W, total = (size_t)a.K * a.B * HW;
    const int P = a.h * a.w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / HW);                         // s*B + b
        const int pix = (int)(i - (size_t)n * HW);
        const int Y = pix / a.W, X = pix - Y * a.W;
        const int s = n / a.B, b = n - s * a.B;
        const int ty = Y / a.down, tx = X / a.down, dy = Y - ty * a.down, dx = X - tx * a.down;
        const int p = ty * a.w + tx;
        const float gx = a.flow[((size_t)n * P + p) * 2 + 0];
        const float gy = a.flow[((size_t)n * P + p) * 2 + 1];
        const float ix = ((gx + 1.f) * a.w - 1.f) / 2.f;    // grid_sampler_unnormalize, align_corners=False
        const float iy = ((gy + 1.f) * a.h - 1.f) / 2.f;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
        const float wnw = (x1 - ix) * (y1 - iy), wne = (ix - x0) * (y1 - iy);
        const float wsw = (x1 - ix) * (iy - y0), wse = (ix - x0) * (iy - y0);
        const bool xin0 = x0 >= 0 && x0 < a.w, xin1 = x1 >= 0 && x1 < a.w;
        const bool yin0 = y0 >= 0 && y0 < a.h, yin1 = y1 >= 0 && y1 < a.h;
        const float* img = a.src[s] + (size_t)b * 3 * HW;
        const float dv = a.div[s];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* pl = img + (size_t)c * HW;
            float v = 0.f;
            if (yin0 && xin0) v = (pl[(size_t)(y0 * a.down + dy) * a.W + x0 * a.down + dx] / dv) * wnw;
            if (yin0 && xin1) v = __builtin_fmaf(pl[(size_t)(y0 * a.down + dy) * a.W + x1 * a.down + dx] / dv, wne, v);
            if (yin1 && xin0) v = __builtin_fmaf(pl[(size_t)(y1 * a.down + dy) * a.W + x0 * a.down + dx] / dv, wsw, v);
            if (yin1 && xin1) v = __builtin_fmaf(pl[(size_t)(y1 * a.down + dy) * a.W + x1 * a.down + dx] / dv, wse, v);
            a.out[((size_t)n * 3 + c) * HW + pix] = v;
        }
    }
}

// x <- ((x - gen_mean) / gen_std) * ref_std + ref_mean per (frame, channel), and the per-block partial of
// sum |x - tar / 255| (fp64).  x: (N, 3, HW) with N = K*B frames; the target frame of x[n] is tar[n % B].
// grid = (chunks, 3, N); part: (N*3*chunks) doubles.
__global__ __launch_bounds__(256) void renorm_l1_kernel(float* __restrict__ x, const float* __restrict__ tar, int B, int HW,
                                                        const float* __restrict__ gen_mean, const float* __restrict__ gen_std,
                                                        const float* __restrict__ ref_mean, const float* __restrict__ ref_std,
                                                        double* __restrict__ part, int W, int fore_x0, int fore_x1, float bg0, float bg1,
                                                        float bg2) {
    __shared__ double red[256];
    const int c = blockIdx.y, n = blockIdx.z, b = n % B, tid = threadIdx.x;
    const float gm = gen_mean[n * 3 + c], gs = gen_std[n * 3 + c], rm = ref_mean[b * 3 + c], rs = ref_std[b * 3 + c];
    float* px = x + ((size_t)n * 3 + c) * HW;
    const float* pt = tar + ((size_t)b * 3 + c) * HW;
    double acc = 0.0;
    for (int i = blockIdx.x * 256 + tid; i < HW; i += gridDim.x * 256) {
        float v = px[i];
        v = v - gm; v = v / gs; v = v * rs; v = v + rm;
        if (fore_x1 > fore_x0) {                              // pose composite: x * fore + bg * (1 - fore), fore in {0, 1}
            const int X = i % W;
            if (X < fore_x0 || X >= fore_x1) v = c == 0 ? bg0 : c == 1 ? bg1 : bg2;
        }
        px[i] = v;
        const float d = v - pt[i] / 255.0f;
        acc += (double)(d < 0.f ? -d : d);
    }
    red[tid] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (tid < k) red[tid] += red[tid + k]; __syncthreads(); }
    if (tid == 0) part[((size_t)n * 3 + c) * gridDim.x + blockIdx.x] = red[0];
}

// per-position cosine similarity of two NHWC feature maps, per-block partial sums (one wave per position)
__global__ __launch_bounds__(256) void cosine_partial_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int rows, int C,
                                                             double* __restrict__ part) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double acc = 0.0;
    for (int row = blockIdx.x * 4 + wv; row < rows; row += gridDim.x * 4) {
        const float* p = x1 + (size_t)row * C;
        const float* q = x2 + (size_t)row * C;
        float dot = 0.f, n1 = 0.f, n2 = 0.f;
        for (int c = lane * 4; c < C; c += 256) {
            const float4 u = *reinterpret_cast<const float4*>(p + c), v = *reinterpret_cast<const float4*>(q + c);
            dot += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
            n1 += u.x * u.x + u.y * u.y + u.z * u.z + u.w * u.w;
            n2 += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { dot += __shfl_xor(dot, off); n1 += __shfl_xor(n1, off); n2 += __shfl_xor(n2, off); }
        float a1 = sqrtf(n1), a2 = sqrtf(n2);                // F.cosine_similarity: each norm clamped at eps = 1e-8
        a1 = a1 < 1e-8f ? 1e-8f : a1; a2 = a2 < 1e-8f ? 1e-8f : a2;
        if (lane == 0) acc += (double)(dot / (a1 * a2));
    }
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = loss_warp = sum_n 10 * (sum of frame n's partials) / (3*HW);  out[1] = loss_align = 1 - (sum of cos partials) / rows
__global__ void train_losses_kernel(const double* __restrict__ l1_part, int frames, int per_frame, double l1_den,
                                    const double* __restrict__ cos_part, int ncos, double rows, float* __restrict__ out) {
    // ncos == 0: the pose model, which has no alignment loss -> out[1] = 0
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lw = 0.f;
    for (int n = 0; n < frames; ++n) {                       // frames are grouped by source: K terms of B frames each are
        double s = 0.0;                                      // summed by the caller's layout (per_frame covers one SOURCE)
        for (int i = 0; i < per_frame; ++i) s += l1_part[(size_t)n * per_frame + i];
        lw += 10.0f * (float)(s / l1_den);                   // 10 * F.l1_loss (fp32 scalar), summed over sources in fp32
    }
    double cs = 0.0;
    for (int i = 0; i < ncos; ++i) cs += cos_part[i];
    out[0] = lw;
    out[1] = ncos > 0 ? 1.0f - (float)(cs / rows) : 0.0f;
}

}  // namespace tsnet
