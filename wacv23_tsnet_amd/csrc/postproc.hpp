// postproc.hpp -- the demo scripts' per-frame post-processing on the device.
//
// Replaces (demo/demo_face.py, identical in demo/demo_pose.py):
//   * :180-182  ref_mean / ref_std  = per-channel mean and UNBIASED std of the first source image / 255
//   * :195-198  gen_mean / gen_std of the generated frame, (rec - gen_mean) / gen_std * ref_std + ref_mean
//   * :96-105   sample_img: CHW -> HWC, + IMG_MEAN/255, clip to [0,1], * 255, BGR -> RGB, then .astype('uint8') (:222)
// The reference does this on the host per frame (a D2H copy of the fp32 frame, numpy, cv2).  Here the frame never
// leaves the GPU as fp32: one reduction launch for the statistics, one elementwise launch that writes the packed
// uint8 RGB frame (3 bytes per pixel instead of 12).  Every fp32 operation is performed in the reference's order
// (no contraction: -ffp-contract=off; correctly rounded division), so a byte differs from the host result only
// where a statistic differs in its last bit.
#pragma once
#include <hip/hip_runtime.h>

namespace tsnet {

// mean and unbiased std (torch.std default) of x / div over HW elements, per (frame b, channel c); x is (B, C, HW).
// grid = (C, B), block = 256.  fp64 sums, fixed-order reduction: deterministic.
__global__ __launch_bounds__(256) void frame_stats_kernel(const float* __restrict__ x, int C, int HW, float div,
                                                          float* __restrict__ mean, float* __restrict__ stdv) {
    __shared__ double red[2 * 256];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* p = x + ((size_t)b * C + c) * HW;
    double s = 0.0, q = 0.0;
    for (int i = tid; i < HW; i += 256) {
        const float v = p[i] / div;                          // the reference divides in fp32 first (:180)
        s += (double)v; q += (double)v * (double)v;
    }
    red[tid] = s; red[256 + tid] = q;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) { red[tid] += red[tid + k]; red[256 + tid] += red[256 + tid + k]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double m = red[0] / HW;
        double var = HW > 1 ? (red[256] - red[0] * m) / (HW - 1) : 0.0;
        if (var < 0) var = 0;
        mean[b * C + c] = (float)m;
        stdv[b * C + c] = (float)sqrt(var);
    }
}

struct DemoPostArgs {
    const float* rec;        // (B, 3, H, W) generator output, channels in the reference's BGR order
    const float* gen_mean;   // (B, 3)
    const float* gen_std;    // (B, 3)
    const float* ref_mean;   // (3)
    const float* ref_std;    // (3)
    float img_mean[3];       // IMG_MEAN / 255 (demo_face.py:27,98)
    unsigned char* out;      // (B, H, W, 3) RGB bytes
    int B, HW;
};

__global__ __launch_bounds__(256) void demo_post_kernel(DemoPostArgs a) {
    const size_t total = (size_t)a.B * a.HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / a.HW);
        const size_t p = i - (size_t)b * a.HW;
        unsigned char px[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = a.rec[((size_t)b * 3 + c) * a.HW + p];
            v = v - a.gen_mean[b * 3 + c];                   // (rec - gen_mean) / gen_std          (:197)
            v = v / a.gen_std[b * 3 + c];
            v = v * a.ref_std[c];                            // * ref_std + ref_mean                (:198)
            v = v + a.ref_mean[c];
            v = v + a.img_mean[c];                           // sample_img                          (:100)
            v = v < 0.f ? 0.f : v;
            v = v > 1.f ? 1.f : v;
            v = v * 255.f;
            px[2 - c] = (unsigned char)v;                    // BGR -> RGB (:104), astype('uint8') truncates (:222)
        }
        unsigned char* o = a.out + i * 3;
        o[0] = px[0]; o[1] = px[1]; o[2] = px[2];
    }
}

}  // namespace tsnet
