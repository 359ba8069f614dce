// flow_sweep.hpp -- flow_kernel<NT>: the correlation -> softmax(100*) -> soft-argmax sweep of the transformation branch for maps that fill
// the chip with one workgroup per (source, batch element, NT * 32 targets) (model/TSNet.py:336-365; the algorithm: flow_warp.hpp's header).
// Its own header since round 5: it is compiled in flow_p_launch.cpp, the translation unit built WITHOUT the SLP vectorizer -- flow_kernel_p
// built with it returned wrong flows on the MI355X (flow_persist.hpp), and this kernel carried the same kind of packed fp32 arithmetic
// (189 v_pk_mul_f32 + 85 v_pk_add_f32 beside its MFMAs) while it was compiled in engine.cpp.  tests/test_isa.py pins the absence of packed
// fp32 arithmetic in that unit; tests/test_gpu_ops.py runs this kernel twice per case and demands equal bits.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_common.hpp"
#include "flow_args.hpp"

namespace tsnet {

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

}  // namespace tsnet
