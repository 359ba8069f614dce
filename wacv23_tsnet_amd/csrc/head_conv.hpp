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
    int x_bf16;           // bf16 storage mode: x holds bf16
};

// four consecutive channels of an fp32 tensor, or of the same tensor stored as bf16 (bf16 storage mode; exact widening)
__device__ __forceinline__ float4 head_ld4(const float* base, size_t elem, bool b16) {
    if (!b16) return *reinterpret_cast<const float4*>(base + elem);
    const uint2 p = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem);
    return make_float4(__builtin_bit_cast(float, p.x << 16), __builtin_bit_cast(float, p.x & 0xFFFF0000u),
                       __builtin_bit_cast(float, p.y << 16), __builtin_bit_cast(float, p.y & 0xFFFF0000u));
}

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
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < a.C) v = head_ld4(a.x, (((size_t)n * a.H + iy) * a.W + ix) * a.C + c, a.x_bf16 != 0);
            if (a.alpha && c < a.C) {
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

// The kernel of the forward (ngf a multiple of 16).  A two-pixel-per-thread form was bound by LDS bandwidth, not by the VALU: every (pixel, tap) re-reads its 64 channels from LDS
// (262144 px x 4 images x 49 taps x 256 B = 13.2 GB per forward = 168 us at 128 B/clk/CU; measured 177 us).  Here
//   * every thread owns FOUR horizontally adjacent output pixels: per (tap row, channel quad) it reads 4 + 6 pixel quads once and
//     slides the seven horizontal taps over them in registers -- 10 LDS reads where 28 were needed -- and the (wave-uniform,
//     broadcast) weight reads are shared by four pixels instead of two; weights are stored as (r x 4 ch, g x 4 ch, b x 4 ch): three
//     16-byte reads per (tap, quad) instead of four;
//   * the patch columns are de-interleaved modulo 4 ([row][column phase][slot]) so the eight threads of a row read consecutive
//     slots, and the row pitch (40 slots = 640 B = 128 mod 256) puts the two rows of a 16-lane group on disjoint banks;
//   * 32 x 32 output tile, 8 channels per stage and half (48.6 KB patch + 4.7 KB weights each), 1.41x halo over-fetch.
// Two fmaf chains per output and 8-channel stage (even / odd channels, 196 products each: tap row, quad, tap column), added and folded
// into the running total.
// TR = rows of the tile (32, 16 or 8; 32 columns always; TR x 8 threads per half).  The kernel is bound by the CU's LDS pipe (every wave reads
// 434 x 1 KiB per stage: eight waves keep the pipe busy 3 x as long as their own FMAs take), and a head of few frames is few waves: at B = 1
// the 512 waves sat eight to a CU on 64 of the 256 CUs.  Shorter tiles spread the same waves over more CUs -- four (TR = 16) or two (TR = 8)
// per CU, where the LDS pipe no longer binds -- for a larger halo (1.63x / 2.07x staged bytes).  A pixel's arithmetic does not depend on the
// tile it sits in (same stages, same chains, same half-0 + half-1 sum): the same bits for every TR (tested), so the launcher may choose by
// the batch (engine.cpp launch_head).
#ifdef TSNET_UNIFORM
#define HEAD_UNIFORM(x) TSNET_UNIFORM(x)
#else
#define HEAD_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif
typedef float head_f2 __attribute__((ext_vector_type(2)));
typedef float head_f4 __attribute__((ext_vector_type(4)));
constexpr int kHead3T = 32, kHead3P = kHead3T + 6, kHead3Slots = 10, kHead3Ch = 8;
constexpr int head3_patch_f4(int TR) { return (kHead3Ch / 4) * (TR + 6) * 4 * kHead3Slots; }       // float4 entries of the patch of a TR-row tile
constexpr int kHead3WtsF4 = 49 * (kHead3Ch / 4) * 3;

template <int TR>
__global__ __launch_bounds__(TR * 16, 1) void head_conv3_kernel(HeadArgs a) {
    static_assert(TR == 32 || TR == 16 || TR == 8, "tile rows");
    constexpr int TH = TR * 8;                                          // threads of a half: one per four horizontally adjacent pixels
    constexpr int PRW = TR + 6, PPX = PRW * kHead3P;                    // patch rows, patch pixels (38 columns)
    constexpr int kHead3PatchF4 = head3_patch_f4(TR);
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    // 512 threads = two halves of four waves.  Both halves own the same 32 x 32 pixels; half h takes the channel stages h, h + 2, ... through
    // its own patch / weight region and the two partial sums are added at the end (half 0 + half 1, a fixed order).  At B = 4 the head is
    // only 256 tiles: splitting the channels is what puts two waves on every SIMD, and it halves the number of barrier-separated stages.
    const int half = HEAD_UNIFORM((int)(threadIdx.x / TH));           // wave-uniform: region bases stay scalar
    float4* tile = reinterpret_cast<float4*>(smem_raw) + half * (kHead3PatchF4 + kHead3WtsF4);      // [quad][38 rows][4 column phases][10 slots]
    float4* wts = tile + kHead3PatchF4;                                 // [49 taps][quad][r, g, b] x 4 channels
    float4* abt = reinterpret_cast<float4*>(smem_raw) + 2 * (kHead3PatchF4 + kHead3WtsF4);          // (alpha, beta) of the image: [C/4] alpha quads, [C/4] beta quads
    const int tid = threadIdx.x & (TH - 1);
    const int tx = tid & 7, ty = tid >> 3;
    const int tiles_x = (a.W + kHead3T - 1) / kHead3T;
    const int n = blockIdx.y;
    const int bx = (blockIdx.x % tiles_x) * kHead3T, by = (blockIdx.x / tiles_x) * TR;
    if (a.alpha) {                                                      // once per workgroup: a global load per stage would sit on the critical path
        for (int i = threadIdx.x; i < a.C / 2; i += 2 * TH)
            abt[i] = *reinterpret_cast<const float4*>((i < a.C / 4 ? a.alpha : a.beta - a.C) + (size_t)n * a.C + i * 4);
        __syncthreads();
    }
    float tot[4][3];
#pragma unroll
    for (int p = 0; p < 4; ++p) { tot[p][0] = 0.f; tot[p][1] = 0.f; tot[p][2] = 0.f; }
    // Staging geometry, fixed over the channel stages: this thread's (up to 12) patch entries and (up to 2) weight entries.  The loads of
    // stage s + 1 are issued before the FMAs of stage s and held in registers (one wave per SIMD: nothing else would cover their latency).
    constexpr int NPE = ((kHead3Ch / 4) * PPX + TH - 1) / TH;                   // 12 (TR = 32), 14, 17
    constexpr int NWE = (kHead3WtsF4 + TH - 1) / TH;                             // 2, 3, 5
    unsigned src_off[NPE];
#pragma unroll
    for (int e = 0; e < NPE; ++e) {
        const int i = tid + e * TH;
        const bool ok = i < (kHead3Ch / 4) * PPX;
        const int q = ok ? i / PPX : 0, p = ok ? i - q * PPX : 0;
        const int py = p / kHead3P, px = p - py * kHead3P;
        int iy = by + py - 3, ix = bx + px - 3;
        iy = iy < 0 ? -iy : iy; iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
        ix = ix < 0 ? -ix : ix; ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
        iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);           // tiles hanging over the edge: any valid address
        ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
        src_off[e] = (unsigned)((((size_t)n * a.H + iy) * a.W + ix) * a.C + q * 4);
    }
    unsigned wsrc[NWE];
#pragma unroll
    for (int e = 0; e < NWE; ++e) {
        const int i = tid + e * TH;
        const int ii = i < kHead3WtsF4 ? i : 0;
        const int col = ii % 3, tq = ii / 3, q = tq % (kHead3Ch / 4), tap = tq / (kHead3Ch / 4);
        wsrc[e] = (unsigned)((tap * a.C + q * 4) * 4 + col);    // [tap][cin][4]: component `col` of four channels (+ c0 * 4 per stage)
    }
    float4 sreg[NPE], wreg[NWE];
    auto stage_load = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < NPE; ++e) sreg[e] = head_ld4(a.x, (size_t)src_off[e] + c0, a.x_bf16 != 0);
#pragma unroll
        for (int e = 0; e < NWE; ++e) {
            const float* src = a.w + wsrc[e] + c0 * 4;
            wreg[e] = make_float4(src[0], src[4], src[8], src[12]);
        }
    };
    auto stage_store = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < NPE; ++e) {
            float4 v = sreg[e];
            if (a.alpha) {
                const int c = c0 + (e * TH + tid >= PPX ? 4 : 0);                        // entries are quad-major: quad 1 starts at PPX
                const float4 al = abt[c >> 2], be = abt[(a.C + c) >> 2];
                v.x = __builtin_fmaf(v.x, al.x, be.x); v.y = __builtin_fmaf(v.y, al.y, be.y);
                v.z = __builtin_fmaf(v.z, al.z, be.z); v.w = __builtin_fmaf(v.w, al.w, be.w);
                v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f;
                v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f;
            }
            const int i = tid + e * TH;                             // recomputed, not kept: registers are the scarce resource here
            if (i < (kHead3Ch / 4) * PPX) {
                const int q = i / PPX, p = i - q * PPX;
                const int py = p / kHead3P, px = p - py * kHead3P;
                tile[((q * PRW + py) * 4 + (px & 3)) * kHead3Slots + (px >> 2)] = v;
            }
        }
#pragma unroll
        for (int e = 0; e < NWE; ++e)
            if (tid + e * TH < kHead3WtsF4) wts[tid + e * TH] = wreg[e];
    };
    static_assert(kHead3Ch / 4 == 2, "stage_store: two channel quads per stage");
    const int nst = a.C / kHead3Ch;                                     // stages: an even number (C % 16 == 0, checked by the launcher)
    stage_load(half * kHead3Ch);
    for (int stg = half; stg < nst; stg += 2) {
        const int c0 = stg * kHead3Ch;
        stage_store(c0);
        __syncthreads();
        if (stg + 2 < nst) stage_load(c0 + 2 * kHead3Ch);
        // Packed fp32 FMAs (v_pk_fma_f32: two lanes per instruction) want register PAIRS as operands.  The natural pairs here are two
        // adjacent channels: (x.c, x.c+1) of a pixel quad and (w.c, w.c+1) of one output's weight quad are adjacent registers of their
        // 16-byte loads, so each output keeps an (even-channel, odd-channel) pair of partial sums and no operand needs a move.
        head_f2 acc[4][3];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int o = 0; o < 3; ++o) acc[p][o] = head_f2{0.f, 0.f};
        // 98 steps (tap row, quad, tap column) of 24 packed FMAs.  At B = 4 the whole head is 1024 waves -- ONE per SIMD -- so nothing but
        // this wave's own instruction stream hides the LDS latency: the pixel quads of the next (tap row, quad) and the weights two steps
        // ahead are requested before the current step's FMAs (register double / triple buffers, scheduling fences keep the order).
        head_f4 pxb[2][10], wb[3][3];
        auto load_px = [&](int buf, int it) __attribute__((always_inline)) {
            const head_f4* row = reinterpret_cast<const head_f4*>(tile + (((it & 1) * PRW + ty + (it >> 1)) * 4) * kHead3Slots + tx);
#pragma unroll
            for (int j = 0; j < 10; ++j) pxb[buf][j] = row[(j & 3) * kHead3Slots + (j >> 2)];      // patch column 4 tx + j
        };
        auto load_w = [&](int buf, int st) __attribute__((always_inline)) {
            const int it = st / 7, kx = st - it * 7;
            const head_f4* wt = reinterpret_cast<const head_f4*>(wts + (((it >> 1) * 7 + kx) * (kHead3Ch / 4) + (it & 1)) * 3);
#pragma unroll
            for (int o = 0; o < 3; ++o) wb[buf][o] = wt[o];
        };
        static_assert(kHead3Ch == 8, "step index = (tap row * 2 + quad) * 7 + tap column");
        load_px(0, 0);
        load_w(0, 0);
        load_w(1, 1);
#pragma unroll
        for (int st = 0; st < 98; ++st) {
            const int it = st / 7, kx = st - it * 7;
            if (st + 2 < 98) load_w((st + 2) % 3, st + 2);
            if (kx == 0 && it + 1 < 14) load_px((it + 1) & 1, it + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int o = 0; o < 3; ++o) {
                const head_f4 w = wb[st % 3][o];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const head_f4 v = pxb[it & 1][p + kx];
                    acc[p][o] = __builtin_elementwise_fma(v.xy, w.xy, acc[p][o]);
                    acc[p][o] = __builtin_elementwise_fma(v.zw, w.zw, acc[p][o]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) { tot[p][0] += acc[p][0].x + acc[p][0].y; tot[p][1] += acc[p][1].x + acc[p][1].y; tot[p][2] += acc[p][2].x + acc[p][2].y; }
        __syncthreads();
    }
    // half 1 hands its sums over through LDS (its own patch region is free after the last barrier)
    float* xch = reinterpret_cast<float*>(reinterpret_cast<float4*>(smem_raw) + (kHead3PatchF4 + kHead3WtsF4));
    if (half == 1) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int o = 0; o < 3; ++o) xch[(p * 3 + o) * TH + tid] = tot[p][o];
    }
    __syncthreads();
    if (half == 1) return;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int o = 0; o < 3; ++o) tot[p][o] += xch[(p * 3 + o) * TH + tid];
    const size_t hw = (size_t)a.H * a.W;
    const int oy = by + ty;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int ox = bx + 4 * tx + p;
        if (ox < a.W && oy < a.H) {
            float o[3] = {tanhf(tot[p][0] + a.bias[0]), tanhf(tot[p][1] + a.bias[1]), tanhf(tot[p][2] + a.bias[2])};
            if (a.composite && (ox < a.fore_x0 || ox >= a.fore_x1)) { o[0] = a.bg[0]; o[1] = a.bg[1]; o[2] = a.bg[2]; }
#pragma unroll
            for (int c = 0; c < 3; ++c) a.y[((size_t)n * 3 + c) * hw + (size_t)oy * a.W + ox] = o[c];
        }
    }
}

__global__ void pack_head_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int C) {
    const int total = 49 * C * 4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int o = i & 3, c = (i >> 2) % C, tap = (i >> 2) / C;
        out[i] = o < 3 ? w[((size_t)o * C + c) * 49 + tap] : 0.f;
    }
}

}  // namespace tsnet
