// flow_warp.hpp -- the transformation branch: mask-aware correlation -> softmax(100*) ->
// soft-argmax flow -> bilinear warp + mean over sources.
//
// Replaces (model/TSNet.py): the two masked torch.bmm (:350-358), F.softmax(100*., dim=2) (:359),
// get_grid + torch.matmul (:299-307, :362-365), F.grid_sample (:366) and the stack().mean() over
// sources (:392).  The reference materialises three B x P x P tensors per source; here the
// P x P matrix never exists: each workgroup owns 32 target positions, its 4 waves sweep disjoint
// 32-source tiles with the exact-fp32 MFMA (A = normalised source rows, B = normalised target
// rows, so a lane owns ONE target column and the softmax reduction over sources is in-lane),
// keep an online softmax state whose "value" is the 2-vector grid coordinate, and merge at the end.
//
// Masks: corr = (T.S) * (mt*ms + (1-mt)*(1-ms)), which equals the reference's sum of two masked
// products exactly for 0/1 masks and to rounding for soft masks; masked pairs stay at logit 0
// (NOT -inf), as in the reference.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_common.hpp"

namespace tsnet {

struct FlowArgs {
    const float* that;        // (B, P, C)   L2-normalised target features
    const float* shat;        // (NB, P, C)  L2-normalised source features, n = s*B + b
    const float* tar_bbox;    // (B, H, W)
    const float* src_bbox[8]; // per source (B, H, W)
    const float* gx;          // (w) linspace(-1,1,w)
    const float* gy;          // (h)
    float* flow;              // (NB, P, 2)
    int B, P, C, h, w, H, W, sy, sx;
};

// grid = (ceil(P/32), NB), block = 64 * kFlowWaves.  dyn LDS: T tile [32][C+4] floats + ms[P] + merge[2*waves][32][4].
// Eight waves per workgroup: the grid is only 384 workgroups at cfg0 (12 images x 32 target tiles), and every wave
// streams its own source rows from global memory -- with four waves that was 1.5 waves per SIMD to hide the loads.
constexpr int kFlowWaves = 8;
__global__ __launch_bounds__(64 * kFlowWaves) void flow_kernel(FlowArgs a) {
    constexpr int NT = 64 * kFlowWaves;
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int LDT = a.C + 4;
    float* sT = reinterpret_cast<float*>(smem_raw);     // [32][LDT]
    float* sMs = sT + 32 * LDT;                          // [P]
    float* sRed = sMs + ((a.P + 3) & ~3);                // [2 * kFlowWaves][32][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int n = blockIdx.y;
    const int s_idx = n / a.B, b = n - s_idx * a.B;
    const int t0 = blockIdx.x * 32;

    // stage the target tile (rows beyond P are zero) and the source mask row
    const int c4n = a.C >> 2;
    for (int i = tid; i < 32 * c4n; i += NT) {
        const int r = i / c4n, c = (i - r * c4n) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t0 + r < a.P) v = *reinterpret_cast<const float4*>(a.that + ((size_t)b * a.P + t0 + r) * a.C + c);
        *reinterpret_cast<float4*>(sT + r * LDT + c) = v;
    }
    const float* sb = a.src_bbox[s_idx] + (size_t)b * a.H * a.W;
    for (int p = tid; p < a.P; p += NT) {
        const int py = p / a.w, px = p - py * a.w;
        sMs[p] = sb[(size_t)(py * a.sy) * a.W + px * a.sx];   // F.interpolate(nearest): src = dst*scale
    }
    __syncthreads();

    const int t = t0 + li;                                  // this lane's target column
    float mt = 0.f;
    if (t < a.P) {
        const int ty = t / a.w, tx = t - ty * a.w;
        mt = a.tar_bbox[(size_t)b * a.H * a.W + (size_t)(ty * a.sy) * a.W + tx * a.sx];
    }

    float m_run = -3.0e38f, l_run = 0.f, ax = 0.f, ay = 0.f;
    const int ntile = (a.P + 31) >> 5;
    const float* srow_base = a.shat + (size_t)n * a.P * a.C;
    for (int st = wave; st < ntile; st += kFlowWaves) {
        const int s0 = st * 32;
        const int srow = s0 + li < a.P ? s0 + li : a.P - 1;     // clamp (masked below)
        const float* ap = srow_base + (size_t)srow * a.C + lh * 4;
        const float* bp = sT + li * LDT + lh * 4;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k = 0; k < a.C; k += 8) {
            const F4 av = *reinterpret_cast<const F4*>(ap + k);
            const F4 bv = *reinterpret_cast<const F4*>(bp + k);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.v[e], bv.v[e], acc, 0, 0, 0);
        }
        // D[row = source][col = target]: lane owns target li, source rows (r&3)+8*(r>>2)+4*lh
        float logit[16];
        float mx = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s = s0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float lg = -3.0e38f;
            if (s < a.P) {
                const float ms = sMs[s];
                const float mf = mt * ms + (1.0f - mt) * (1.0f - ms);
                lg = 100.0f * (acc[r] * mf);
            }
            logit[r] = lg;
            mx = lg > mx ? lg : mx;
        }
        const float m_new = mx > m_run ? mx : m_run;
        const float sc = expf(m_run - m_new);
        l_run *= sc; ax *= sc; ay *= sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int s = s0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (s < a.P) {
                const float e = expf(logit[r] - m_new);
                const int py = s / a.w, px = s - py * a.w;
                l_run += e;
                ax = __builtin_fmaf(e, a.gx[px], ax);
                ay = __builtin_fmaf(e, a.gy[py], ay);
            }
        }
        m_run = m_new;
    }
    // merge the partial states (waves x 2 half-waves) of each target column
    float* o = sRed + ((wave * 2 + lh) * 32 + li) * 4;
    o[0] = m_run; o[1] = l_run; o[2] = ax; o[3] = ay;
    __syncthreads();
    if (tid < 32 && t0 + tid < a.P) {
        float M = -3.0e38f;
        for (int q = 0; q < 2 * kFlowWaves; ++q) { const float v = sRed[(q * 32 + tid) * 4]; M = v > M ? v : M; }
        float L = 0.f, X = 0.f, Y = 0.f;
        for (int q = 0; q < 2 * kFlowWaves; ++q) {
            const float* p = sRed + (q * 32 + tid) * 4;
            if (p[1] > 0.f) {
                const float sc = expf(p[0] - M);
                L += p[1] * sc; X += p[2] * sc; Y += p[3] * sc;
            }
        }
        float* f = a.flow + ((size_t)n * a.P + t0 + tid) * 2;
        f[0] = X / L;
        f[1] = Y / L;
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
