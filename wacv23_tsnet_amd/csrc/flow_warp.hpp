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


// grid = Ppad / (32 NT) * NB (1-D), block = 512.  dyn LDS: target planes NT * KC * 2 KiB, then ms[Ppad], gx[w], gy[h]; the merge buffer
// [16][NT * 32][4] floats aliases the target planes after the sweep.
template <int NT>
__global__ __launch_bounds__(64 * kFlowWaves) void flow_kernel(FlowArgs a) {
    constexpr int NTH = 64 * kFlowWaves;
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int Ppad = (a.P + 63) / 64 * 64, KC = (a.C + 31) / 32 * 2;
    const int TBYTES = NT * KC * 2048, RBYTES = 2 * kFlowWaves * NT * 32 * 16;
    float* sMs = reinterpret_cast<float*>(smem_raw + (TBYTES > RBYTES ? TBYTES : RBYTES));   // [Ppad]
    float* sGx = sMs + Ppad;                                             // [w] linspace(-1, 1, w)
    float* sGy = sGx + ((a.w + 3) & ~3);                                 // [h]
    float* sRed = reinterpret_cast<float*>(smem_raw);                    // [2 * kFlowWaves][NT * 32][4] (after the sweep)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    // 1-D grid, XCD-aware: consecutive (image, target tile) items run on one XCD, whose L2 then fetches an image's source planes once
    const int tiles = Ppad / (32 * NT);
    const int item = xcd_item(blockIdx.x, (int)gridDim.x);
    const int n = item / tiles;
    const int s_idx = n / a.B, b = n - s_idx * a.B;
    const int tb0 = (item - n * tiles) * NT;

    // the workgroup's target fragments: one contiguous region of the plane buffer
    {
        const F4* g = reinterpret_cast<const F4*>(a.tq + ((size_t)(b * (Ppad >> 5) + tb0) * KC) * 1024);
        F4* d = reinterpret_cast<F4*>(smem_raw);
        const int cnt = TBYTES / 16;                                     // a multiple of 256 (KC is even)
        for (int i0 = 0; i0 < cnt; i0 += 8 * NTH) {                      // eight loads in flight per thread
            F4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * NTH + tid; v[u] = g[i < cnt ? i : tid]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * NTH + tid; if (i < cnt) d[i] = v[u]; }
        }
    }
    const float* sb = a.src_bbox[s_idx] + (size_t)b * a.H * a.W;
    for (int p = tid; p < Ppad; p += NTH) {
        float v = 0.f;
        if (p < a.P) {
            const int py = p / a.w, px = p - py * a.w;
            v = sb[(size_t)(py * a.sy) * a.W + px * a.sx];               // F.interpolate(nearest): src = dst*scale
        }
        sMs[p] = v;
    }
    for (int p = tid; p < a.w; p += NTH) sGx[p] = a.gx[p];
    for (int p = tid; p < a.h; p += NTH) sGy[p] = a.gy[p];
    __syncthreads();

    float mt[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int t = (tb0 + j) * 32 + li;                                // this lane's target column of block j
        mt[j] = 0.f;
        if (t < a.P) {
            const int ty = t / a.w, tx = t - ty * a.w;
            mt[j] = a.tar_bbox[(size_t)b * a.H * a.W + (size_t)(ty * a.sy) * a.W + tx * a.sx];
        }
    }
    const float inv_w = 1.0f / (float)a.w;

    float m_run[NT], l_run[NT], ax[NT], ay[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { m_run[j] = -3.0e38f; l_run[j] = 0.f; ax[j] = 0.f; ay[j] = 0.f; }
    const unsigned char* sbase = reinterpret_cast<const unsigned char*>(a.sq) + ((size_t)n * (Ppad >> 5) * KC) * 2048 + lane * 16;
    const unsigned char* tbase = smem_raw + lane * 16;
    const int npair = Ppad >> 6;
    for (int sp = wave; sp < npair; sp += kFlowWaves) {
        const unsigned char* ap = sbase + (size_t)(sp * 2) * KC * 2048;  // source blocks 2 sp, 2 sp + 1: KC * 2 KiB each
        f32x16 acc[2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        F4 af[3][2][2][2];                                               // [set][step of the group][source block][plane]
        F4 bf[2][NT][2];                                                 // [set][target block][plane]
        auto load_a = [&](int set, int g) __attribute__((always_inline)) {       // the two steps of group g
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
                        af[set][u][i][pl] = *reinterpret_cast<const F4*>(ap + ((size_t)(i * KC + g * 2 + u) * 2 + pl) * 1024);
        };
        auto load_b = [&](int set, int kc) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) bf[set][j][pl] = *reinterpret_cast<const F4*>(tbase + ((j * KC + kc) * 2 + pl) * 1024);
        };
        auto mfmas = [&](int sa, int u, int sbt) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[i][j] = TSNET_MFMA_F16(af[sa][u][i][1], bf[sbt][j][0], acc[i][j]);      // lo * hi
                    acc[i][j] = TSNET_MFMA_F16(af[sa][u][i][0], bf[sbt][j][1], acc[i][j]);      // hi * lo
                    acc[i][j] = TSNET_MFMA_F16(af[sa][u][i][0], bf[sbt][j][0], acc[i][j]);      // hi * hi
                }
        };
        // groups of two steps; source fragments two groups ahead (three register sets: a load has ~48 MFMAs = 1500 cycles to land)
        const int ngrp = KC >> 1;
        load_a(0, 0);
        load_a(1, ngrp > 1 ? 1 : 0);
        load_b(0, 0);
        auto group = [&](int g, int S) __attribute__((always_inline)) {          // S = register set of group g = g % 3
            load_a((S + 2) % 3, g + 2 < ngrp ? g + 2 : g);               // past the end: re-reads a valid group (unused)
            load_b(1, 2 * g + 1);
            mfmas(S, 0, 0);
            load_b(0, 2 * g + 2 < KC ? 2 * g + 2 : 0);
            mfmas(S, 1, 1);
        };
        int g = 0;
        for (; g + 3 <= ngrp; g += 3) { group(g, 0); group(g + 1, 1); group(g + 2, 2); }
        if (g < ngrp) group(g, 0);
        if (g + 1 < ngrp) group(g + 1, 1);

        // D[row = source][col = target]: lane owns target li of each block, source rows i*32 + (r&3) + 8*(r>>2) + 4*lh.  Branch-free:
        // a source past P gets logit -3e38 and weight 0 (select); every table it indexes is padded.
        const int s0 = sp * 64 + 4 * lh;
        float mx[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) mx[j] = -3.0e38f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int s = s0 + i * 32 + (r & 3) + 8 * (r >> 2);
                const float ms = sMs[s];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float mf = mt[j] * ms + (1.0f - mt[j]) * (1.0f - ms);
                    float lg = 100.0f * ((acc[i][j][r] * kFlowUnscale) * mf);
                    lg = s < a.P ? lg : -3.0e38f;
                    acc[i][j][r] = lg;
                    mx[j] = lg > mx[j] ? lg : mx[j];
                }
            }
        float m_new[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            m_new[j] = mx[j] > m_run[j] ? mx[j] : m_run[j];
            const float sc = expf(m_run[j] - m_new[j]);
            l_run[j] *= sc; ax[j] *= sc; ay[j] *= sc;
            m_run[j] = m_new[j];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int s = s0 + i * 32 + (r & 3) + 8 * (r >> 2);
                const int sc = s < a.P ? s : a.P - 1;
                const int py = (int)(((float)sc + 0.5f) * inv_w), px = sc - py * a.w;          // exact for s < 2^20
                const float gxv = sGx[px], gyv = sGy[py];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float e = TSNET_FAST_EXP(acc[i][j][r] - m_new[j]);      // v_exp_f32((lg - m) * log2 e): relative error <= 6e-8 * |lg - m| * 1.44 (+ 1 ulp) -- weights that matter have small |lg - m|
                    e = s < a.P ? e : 0.f;
                    l_run[j] += e;
                    ax[j] = __builtin_fmaf(e, gxv, ax[j]);
                    ay[j] = __builtin_fmaf(e, gyv, ay[j]);
                }
            }
    }
    // merge the partial states (waves x 2 half-waves) of each target column, in a fixed order
    __syncthreads();                                                     // every wave is done with the target planes
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        float* o = sRed + (((wave * 2 + lh) * NT + j) * 32 + li) * 4;
        o[0] = m_run[j]; o[1] = l_run[j]; o[2] = ax[j]; o[3] = ay[j];
    }
    __syncthreads();
    if (tid < NT * 32 && tb0 * 32 + tid < a.P) {
        float M = -3.0e38f;
        for (int qd = 0; qd < 2 * kFlowWaves; ++qd) { const float v = sRed[(qd * NT * 32 + tid) * 4]; M = v > M ? v : M; }
        float L = 0.f, X = 0.f, Y = 0.f;
        for (int qd = 0; qd < 2 * kFlowWaves; ++qd) {
            const float* pr = sRed + (qd * NT * 32 + tid) * 4;
            if (pr[1] > 0.f) {
                const float sc = expf(pr[0] - M);
                L += pr[1] * sc; X += pr[2] * sc; Y += pr[3] * sc;
            }
        }
        float* f = a.flow + ((size_t)n * a.P + tb0 * 32 + tid) * 2;
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
