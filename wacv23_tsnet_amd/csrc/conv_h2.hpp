// conv_h2.hpp -- 3x3 / stride-1 convolution on the fp16 MFMA from fp32 input, fp32-class accuracy, with the producer's
// InstanceNorm + ReLU applied while the input patch is staged.
//
// Numerics (tools/probes/split_probe.hip, profiles/round2_notes.md).  An fp32 value x scaled by a power of two s is
// stored as two fp16 numbers  hi = rne(x*s),  lo = rne(x*s - hi):  the residual of a round-to-nearest hi has at most 12
// significant bits left and lo rounds away at most the last one, so hi + lo = x*s up to 2^-24 |x*s| (or 2^-25 absolute
// where lo is subnormal) -- the rounding class of fp32 itself.  A product a*w is the three exact fp16 products
//   lo*hi + hi*lo + hi*hi      (dropped: lo*lo <= 2^-24 |a||w|, random sign)
// on v_mfma_f32_32x32x16_f16, fp32 accumulate, two accumulation levels exactly like conv_x3p.hpp's x3q tile: 3 MFMAs per
// 16-deep k-group where the bf16x3 scheme needs 6 and the fp32 MFMA the time of 16.  The power-of-two scales (exact)
// keep |x*s| < 65504: activations are bounded by construction -- InstanceNorm output is <= sqrt(HW-1) in magnitude,
// the residual stream <= (blocks+1) sqrt(HW) -- and the host derives s from that bound; weights are scaled per layer
// from their maximum.  The epilogue multiplies by 2^-(sa+sw) (exact) before bias, statistics and store.
//
// Data flow of a tile (128 output positions = a 4 x 32 pixel rectangle of one image, BN output channels):
//   * per 16-channel slab the (4+2) x (32+2) input patch is fetched ONCE as fp32 (two 16-byte buffer loads per thread
//     and round, reflection / zero padding resolved in the lane's offset), transformed in registers
//     (x*alpha+beta, ReLU, *s, split) and written to LDS as two fp16 planes in the swizzled image of conv_x3p.hpp; the
//     nine taps are nine shifted views (immediate offsets of the ds_read).  norm_act_kernel's separate pass over the
//     tensor (4 B read + 6 B written per element) disappears, and the conv reads 4 B per element instead of 6;
//   * weight fragments (packed in MFMA fragment order by pack_weights_h2_kernel) go straight into registers, two steps
//     ahead (three register sets); A fragments one step ahead (two sets) -- with three products per step a load has
//     only half the cover it had in x3q, so the re-use-after-last-use trick is replaced by explicit double buffering;
//   * one __syncthreads per slab; no inline asm: every load is compiler-visible.
// K order is slab-major (slab, tap), chains = taps 0..3 and 4..8 of a slab, folded into the running total.
#pragma once
#include "conv_x3p.hpp"

namespace tsnet {

struct H2Args {
    const float* x;            // (N,H,W,Cin) fp32 NHWC
    const float* in_alpha;     // null, or (N*Cin): x*alpha+beta on load (the producer's InstanceNorm statistics)
    const float* in_beta;
    int in_relu;
    float in_scale;            // 2^sa
    const unsigned short* w;   // fp16 planes [2][K/16][Npad][2 swizzled octets][8] of w * 2^sw
    float in_unscale;          // 2^-sa
    const float* w_unscale;    // device scalar 2^-sw (lives in the packed weight buffer, so replicas receive it with the broadcast)
    const float* bias;
    float* y;
    unsigned short* y3;        // always null here (member of the shared epilogue's contract)
    double* stat_part;
    const float* addend; int add_nmod;
    int N, H, W, Cin, Ho, Wo, Cout, Npad, reflect, nchunks, M;
    int tiles_m, tiles_n;
    float* fin_alpha; float* fin_beta; int* fin_counter; int fin_S; float fin_eps;
    unsigned* amax_out;        // see X3Args
    // operand scale from the data instead of an a-priori bound: in_amax = bit pattern of max |x| published by x's producer
    // (X3Args::amax_out), bound = that + in_bound_add (what later stages may add, e.g. InstanceNorm outputs of a residual stream);
    // null = the host's in_scale / in_unscale
    const unsigned* in_amax; float in_bound_add;          // in_amax[image]: one slot per image, so a sample's result never depends on its batch
};

// power-of-two operand scale for |x| <= bound: |x * 2^sa| <= 2^15 (the host's h2_scale_log2, engine.cpp)
__device__ __forceinline__ void h2_device_scale(const unsigned* amax, float add, float& scale, float& unscale) {
    const float bound = __builtin_bit_cast(float, __hip_atomic_load(amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + add;
    int e = 0;
    (void)frexpf(bound, &e);
    int sa = 15 - e;
    sa = sa > 24 ? 24 : (sa < -24 ? -24 : sa);
    scale = ldexpf(1.0f, sa);
    unscale = ldexpf(1.0f, -sa);
}

#ifndef TSNET_MFMA_F16
typedef _Float16 tsnet_f16x8 __attribute__((ext_vector_type(8)));
#define TSNET_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(tsnet_f16x8, a), __builtin_bit_cast(tsnet_f16x8, b), c, 0, 0, 0)
#endif

// x (already scaled) -> (hi, lo) fp16 bit patterns
__device__ __forceinline__ void split_h2(float v, unsigned& hi, unsigned& lo) {
    const _Float16 h = (_Float16)v;                              // round to nearest even
    const _Float16 l = (_Float16)(v - (float)h);                 // exact residual, then RNE
    hi = (unsigned)__builtin_bit_cast(unsigned short, h);
    lo = (unsigned)__builtin_bit_cast(unsigned short, l);
}

// eight consecutive channels (two float4) -> one 16-byte octet per plane
__device__ __forceinline__ void split_h2_octet(const F4& x0, const F4& x1, F4& H, F4& L) {
    unsigned h[8], l[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { split_h2(x0.v[e], h[e], l[e]); split_h2(x1.v[e], h[4 + e], l[4 + e]); }
    unsigned hw[4], lw[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { hw[e] = h[2 * e] | (h[2 * e + 1] << 16); lw[e] = l[2 * e] | (l[2 * e + 1] << 16); }
#pragma unroll
    for (int e = 0; e < 4; ++e) { H.v[e] = __builtin_bit_cast(float, hw[e]); L.v[e] = __builtin_bit_cast(float, lw[e]); }
}

// eight consecutive channels -> one octet of bf16 (round to nearest even): the bf16-operand mode's single plane
__device__ __forceinline__ void bf16_octet(const F4& x0, const F4& x1, F4& H) {
    unsigned hw[4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        hw[e] = (unsigned)bf16_rne(x0.v[2 * e]) | ((unsigned)bf16_rne(x0.v[2 * e + 1]) << 16);
        hw[2 + e] = (unsigned)bf16_rne(x1.v[2 * e]) | ((unsigned)bf16_rne(x1.v[2 * e + 1]) << 16);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) H.v[e] = __builtin_bit_cast(float, hw[e]);
}

// NPROD = 3: lo*hi, hi*lo, hi*hi;  NPROD = 4: lo*lo first (kept for the accuracy comparison in the op tests);
// NPROD = 1: bf16-operand mode (BASELINE.json configs[2] / [4]): the transformed input is rounded to ONE bf16 plane while it is
// staged, the weights are the bf16 hi plane of the bf16x3 packing (conv_x3.hpp), one v_mfma_f32_32x32x16_bf16 per k-group; no scales.
// HABL (tools build only, tools/x3_ablate.py h2; non-zero computes garbage): bit0 no patch staging in the loop, bit1 weight fragments
// loaded once, bit2 A fragments read once, bit3 no fold, bit4 no slab barrier
template <int BN, int WARPS_M, int WARPS_N, int NPROD, bool AFFINE, int HABL = 0>
__device__ __forceinline__ void h2_tile(const H2Args& a, unsigned char* smem_raw, const int tile_m, const int n0) {
    constexpr int BM = kPatchRows * kPatchCols;
    constexpr int NW = WARPS_M * WARPS_N;
    static_assert(NW == 4, "four waves: patch blocks are dealt w, w+4");
    static_assert(NPROD == 1 || NPROD == 3 || NPROD == 4, "one (bf16 operands), three or four products");
    constexpr int NPL = NPROD == 1 ? 1 : 2;                          // operand planes
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 32, NTL = WN / 32;
    constexpr int PC = kPatchCols + 2, PP = (kPatchRows + 2) * PC;   // 34, 204 patch pixels
    constexpr int PBLK = (PP + 31) / 32;                             // 7 blocks of 32 pixel slots
    constexpr int REGION = PBLK * 512;                               // one octet region: 224 slots x 16 B
    constexpr int PLANE_P = 2 * REGION, PATCH_BYTES = NPL * PLANE_P; // planes x 7 KiB per stage
    constexpr int OFF_SCRATCH = 2 * 2 * PLANE_P;                     // 2 KiB sink for the wave whose second block does not exist (fixed offsets in both modes)
    constexpr int OFF_TAB = OFF_SCRATCH + 2048;                      // (alpha*s, beta*s) table of the image: 2 x Cin floats

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int wrow = wave / WARPS_N;
    const int wn0 = (wave % WARPS_N) * WN;
    const int li = lane & 31, lh = lane >> 5;

    const int tcols = a.Wo / kPatchCols, tper = (a.Ho / kPatchRows) * tcols;
    const int img = tile_m / tper, tin = tile_m - img * tper;
    const int oy0 = (tin / tcols) * kPatchRows, ox0 = (tin % tcols) * kPatchCols;
    const int ncc = a.Cin >> 4;
    float in_scale = a.in_scale, in_unscale = a.in_unscale;
    if (NPROD != 1 && a.in_amax) h2_device_scale(a.in_amax + img, a.in_bound_add, in_scale, in_unscale);

    const size_t planew = (size_t)((a.nchunks + 1) / 2 * 2) * a.Npad * 16;
    const tsnet_brsrc_t rsx = tsnet_make_brsrc(a.x, (unsigned)((size_t)a.N * a.H * a.W * a.Cin * 4));
    tsnet_brsrc_t rsw[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) rsw[p] = tsnet_make_brsrc(a.w + p * planew, (unsigned)(planew * 2));

    // ---- LDS image of a patch stage: per plane two octet REGIONS (channels 0..7 / 8..15 of the slab), each one 16-byte entry per
    //      pixel slot.  A fragment read of lane (li, lh) is region lh, slot p0 + tap shift: 16 consecutive lanes read 256
    //      consecutive bytes (conflict-free without a swizzle) and the tap shift is an IMMEDIATE of the ds_read -- one
    //      address register for all 18 (row, tap) combinations.
    // ---- staging geometry: this wave owns pixel blocks wave and wave + 4; lane -> (slot b*32 + (lane & 31), octet lane >> 5):
    //      the 8 lanes of a ds_write_b128 group write 128 consecutive bytes.
    const int oct = lane >> 5;
    unsigned vP[2];
    float vM[2];                                                     // 1, or 0 for a zero-padded / unused slot
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int pp = (wave + 4 * r) * 32 + (lane & 31);
        const int pr = pp / PC, pc = pp - pr * PC;
        int iy = oy0 - 1 + pr, ix = ox0 - 1 + pc;
        bool ok = pp < PP;
        if (a.reflect) {
            iy = iy < 0 ? -iy : iy;
            iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
            ix = ix < 0 ? -ix : ix;
            ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
        } else {
            ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        }
        const int pix = iy * a.W + ix;
        vP[r] = ok ? (unsigned)(((img * a.H * a.W + pix) * a.Cin + oct * 8) * 4) : kOOB;
        vM[r] = ok ? 1.f : 0.f;
    }
    // per-(image, channel) transform table in LDS, pre-multiplied by the operand scale (a power of two: exact, and
    // fl(x*(al*s) + be*s) == s * fl(x*al + be)); without a producer InstanceNorm: (s, 0)
    float* tab = reinterpret_cast<float*>(smem_raw + OFF_TAB);       // [Cin] alpha*s, then [Cin] beta*s
    if (AFFINE) {
        for (int c = tid; c < a.Cin; c += 256) {
            tab[c] = a.in_alpha[(size_t)img * a.Cin + c] * in_scale;
            tab[a.Cin + c] = a.in_beta[(size_t)img * a.Cin + c] * in_scale;
        }
        __syncthreads();
    }
    const float relu_floor = a.in_relu ? 0.f : -__builtin_inff();    // branch-free ReLU switch
    F4 sx[2];                                                        // staging registers: one round of x
    auto stage_load_x = [&](int cn, int r) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) sx[q] = TSNET_BUF_LOAD16(rsx, vP[r], (unsigned)(cn * 64 + q * 16));
    };
    auto stage_store = [&](int cn, int r) __attribute__((always_inline)) {
        const int b = wave + 4 * r;
        F4 t[2];
        if (AFFINE) {
            const float* ta = tab + (cn < ncc ? cn * 16 : 0) + oct * 8;     // past the last slab: any valid entry (result unused)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const F4 al = *reinterpret_cast<const F4*>(ta + q * 4), be = *reinterpret_cast<const F4*>(ta + a.Cin + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = __builtin_fmaf(sx[q].v[e], al.v[e], be.v[e]);
                    v = v > relu_floor ? v : relu_floor;
                    t[q].v[e] = v * vM[r];
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = sx[q].v[e] * in_scale;                    // a padded slot loaded zeros
                    t[q].v[e] = v > relu_floor ? v : relu_floor;
                }
        }
        // block 7 does not exist: its wave writes into the sink (wave-uniform select, no branch)
        unsigned char* dst = smem_raw + (b < PBLK ? (cn & 1) * PATCH_BYTES + oct * REGION + b * 512 : OFF_SCRATCH + oct * 512) + (lane & 31) * 16;
        F4 Hh, Ll;
        if (NPROD == 1) {
            bf16_octet(t[0], t[1], Hh);
            *reinterpret_cast<F4*>(dst) = Hh;
        } else {
            split_h2_octet(t[0], t[1], Hh, Ll);
            *reinterpret_cast<F4*>(dst) = Hh;
            *reinterpret_cast<F4*>(dst + (b < PBLK ? PLANE_P : 1024)) = Ll;
        }
    };

    // ---- fragments.  Weights: lane (li, lh) takes the 16 bytes of column wn0 + j*32 + li, logical octet lh
    const unsigned vB = (unsigned)((wn0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16);
    F4 af[2][NPL][MT], bf[3][NPL][NTL];                              // [register set][plane][tile]
    auto load_b = [&](int set, int cc, int t) __attribute__((always_inline)) {     // past the end of K the descriptor returns zeros
        const int kc = t * ncc + cc;
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int j = 0; j < NTL; ++j) bf[set][p][j] = TSNET_BUF_LOAD16(rsw[p], vB, (unsigned)((kc * a.Npad + n0 + j * 32) * 32));
    };
    const unsigned char* abase = smem_raw + lh * REGION + (wrow * MT * PC + li) * 16;
    auto load_a = [&](int set, int cc, int t) __attribute__((always_inline)) {
        const int ky = t / 3, kx = t - ky * 3;
        const unsigned char* pbase = abase + (cc & 1) * PATCH_BYTES;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int p = 0; p < NPL; ++p) af[set][p][i] = *reinterpret_cast<const F4*>(pbase + p * PLANE_P + ((i + ky) * PC + kx) * 16);
    };

    f32x16 acc[MT][NTL], tot[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

    auto product = [&](int sa, int sb, int pa, int pb, bool fresh) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                f32x16 c = acc[i][j];
                if (fresh) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = 0.f;
                }
                if (NPROD == 1) acc[i][j] = TSNET_MFMA_BF16(af[sa][pa][i], bf[sb][pb][j], c);
                else acc[i][j] = TSNET_MFMA_F16(af[sa][pa][i], bf[sb][pb][j], c);
            }
    };
    // step (cc, t): A(cc,t) in set SA, B(cc,t) in set t%3; issues A(cc,t+1) and B of two steps ahead first
    auto step = [&](int cc, int t, int SA) __attribute__((always_inline)) {
        const bool fresh = t == 0 || t == 4;                         // chains: taps 0..3 and 4..8 of the slab
        const int t2 = (t + 2) % 9;
        if (!(HABL & 2)) load_b(t2 % 3, cc + (t + 2 >= 9 ? 1 : 0), t2);
        if (t < 8 && !(HABL & 4)) load_a(SA ^ 1, cc, t + 1);
        // staging of slab cc+1: round 0 fetched at tap 0 and written at tap 2, round 1 fetched at tap 3 and written at tap 5
        // (past the last slab the loads run into the next pixel's channels or return zeros: written to the idle stage, never read)
        if (t == 0 && !(HABL & 1)) stage_load_x(cc + 1, 0);
        if (t == 3 && !(HABL & 1)) stage_load_x(cc + 1, 1);
        const int SB = t % 3;
        if (NPROD == 1) {
            product(SA, SB, 0, 0, fresh);                            // bf16 * bf16
        } else {
            if (NPROD == 4) product(SA, SB, NPL - 1, NPL - 1, fresh);        // lo * lo
            product(SA, SB, NPL - 1, 0, fresh && NPROD == 3);        // lo * hi
            product(SA, SB, 0, NPL - 1, false);                      // hi * lo
            product(SA, SB, 0, 0, false);                            // hi * hi
        }
        if (t == 2 && !(HABL & 1)) stage_store(cc + 1, 0);
        if (t == 5 && !(HABL & 1)) stage_store(cc + 1, 1);
        if ((t == 3 || t == 8) && !(HABL & 8)) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) tot[i][j] += acc[i][j];
        }
    };
    auto slab = [&](int cc, int S0) __attribute__((always_inline)) {  // S0 = A register set of tap 0 = cc & 1 (9 taps: parity flips per slab)
        if (!(HABL & 16)) __syncthreads();                           // patch(cc) complete and visible; slab cc-1 fully read
        if (!(HABL & 4) || cc == 0) load_a(S0, cc, 0);
        step(cc, 0, S0); step(cc, 1, S0 ^ 1); step(cc, 2, S0);
        step(cc, 3, S0 ^ 1); step(cc, 4, S0); step(cc, 5, S0 ^ 1);
        step(cc, 6, S0); step(cc, 7, S0 ^ 1); step(cc, 8, S0);
    };

    // prologue: patch of slab 0, weight fragments of steps (0,0) and (0,1)
    stage_load_x(0, 0); stage_store(0, 0);
    stage_load_x(0, 1); stage_store(0, 1);
    load_b(0, 0, 0);
    load_b(1, 0, 1);
    if (HABL & 2) load_b(2, 0, 2);
    if (HABL & 4) { __syncthreads(); load_a(0, 0, 0); load_a(1, 0, 1); }
    int cc = 0;
    for (; cc + 2 <= ncc; cc += 2) { slab(cc, 0); slab(cc + 1, 1); }
    if (cc < ncc) slab(cc, 0);

    const float unscale = a.w_unscale ? in_unscale * a.w_unscale[0] : in_unscale;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[i][j][r] *= unscale;    // exact: power of two
    const int m_img = img * a.Ho * a.Wo;
    x3_epilogue<BN, WARPS_M, WARPS_N, MT, NTL>(a, tot, smem_raw, tid, wave, n0, (size_t)img * tper + tin,
                                               [&](int l) { return m_img + (oy0 + (l >> 5)) * a.Wo + ox0 + (l & 31); });
}

template <int BN, int WARPS_M, int WARPS_N, int NPROD, bool AFFINE, int HABL = 0>
__global__ __launch_bounds__(256, BN <= 64 ? 3 : 2)   // three workgroups per CU with 64-wide tiles (768 tiles = 3 per CU on the ResnetBlock layers)
void conv_h2_kernel(H2Args a) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int bid = x3p_item(blockIdx.x, a.tiles_m * a.tiles_n);
    const int tile_m = bid / a.tiles_n;
    h2_tile<BN, WARPS_M, WARPS_N, NPROD, AFFINE, HABL>(a, smem_raw, tile_m, (bid - tile_m * a.tiles_n) * BN);
}

// ---------------------------------------------------------------------------------------------------------------
// h2s: the 7 x 7 stems at 8 input channels (TSNet.py:66 with label_nc = 2: image 3 + label 2 + coordinates 3, or label 2 + coordinates 3
// padded to 8) as a PATCH kernel.  As an implicit GEMM (conv_h2r, SMALL_CIN) the stem gathers 128 rows x 16 k of fp32 from L1 for every
// k-step -- 49 taps re-read every input pixel 49 times, and with only 64 output channels there is little MFMA work per gathered byte
// (217 us for 39.5 GFLOP).  Here the (4+6) x (32+6) x 8-channel patch of a 4 x 32 output rectangle is fetched ONCE (12 KB of fp32),
// scaled, split into two fp16 planes of one 16-byte octet per pixel, and the 25 k-steps (two taps per 16-deep k-group, the 50th tap has
// zero weights) read it through shifted views: no barrier and no global A traffic inside the loop.
// A fragment of step s: lane (li, lh) supplies output pixel li of a row and k-half lh = tap 2s + lh, i.e. patch slot
// (row + ky) * 38 + li + kx of THAT tap.  Tap 2s + 1 is one slot right of tap 2s, or -- when tap 2s is the last of its row -- 32 slots
// on: one per-lane base, a wave-uniform tap offset and lh x delta per step.
template <int NPROD>
__device__ __forceinline__ void h2s_tile(const H2Args& a, unsigned char* smem_raw, const int tile_m, const int n0) {
    constexpr int BN = 64, WARPS_M = 2, WARPS_N = 2, MT = 2, NTL = 1;
    constexpr int NPL = NPROD == 1 ? 1 : 2;
    constexpr int PC = kPatchCols + 6, PR = kPatchRows + 6, PP = PR * PC;       // 38 x 10 = 380 patch pixels
    constexpr int PLANE_S = 384 * 16;                                             // one plane: a 16-byte octet per pixel slot
    static_assert(NPROD == 1 || NPROD == 3, "one (bf16 operands) or three products");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int wrow = wave / WARPS_N;
    const int wn0 = (wave % WARPS_N) * 32;
    const int li = lane & 31, lh = lane >> 5;

    const int tcols = a.Wo / kPatchCols, tper = (a.Ho / kPatchRows) * tcols;
    const int img = tile_m / tper, tin = tile_m - img * tper;
    const int oy0 = (tin / tcols) * kPatchRows, ox0 = (tin % tcols) * kPatchCols;
    float in_scale = a.in_scale, in_unscale = a.in_unscale;
    if (NPROD != 1 && a.in_amax) h2_device_scale(a.in_amax + img, a.in_bound_add, in_scale, in_unscale);

    const size_t planew = (size_t)((a.nchunks + 1) / 2 * 2) * a.Npad * 16;
    const tsnet_brsrc_t rsx = tsnet_make_brsrc(a.x, (unsigned)((size_t)a.N * a.H * a.W * 8 * 4));
    tsnet_brsrc_t rsw[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) rsw[p] = tsnet_make_brsrc(a.w + p * planew, (unsigned)(planew * 2));

    // ---- weight fragments first (their latency hides behind the patch staging), two steps ahead afterwards
    const unsigned vB = (unsigned)((wn0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16);
    F4 af[2][NPL][MT], bf[3][NPL];
    auto load_b = [&](int set, int kc) __attribute__((always_inline)) {           // past the end of K the descriptor returns zeros
#pragma unroll
        for (int p = 0; p < NPL; ++p) bf[set][p] = TSNET_BUF_LOAD16(rsw[p], vB, (unsigned)((kc * a.Npad + n0) * 32));
    };
    load_b(0, 0);
    load_b(1, 1);

    // ---- patch staging: thread t takes pixel slots t and t + 256 (380 in all); reflection padding resolved in the address
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int pp = tid + r * 256;
        const int pr = pp / PC, pc = pp - pr * PC;
        int iy = oy0 - 3 + pr, ix = ox0 - 3 + pc;
        iy = iy < 0 ? -iy : iy;
        iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
        ix = ix < 0 ? -ix : ix;
        ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
        const unsigned v = pp < PP ? (unsigned)((((img * a.H + iy) * a.W) + ix) * 32) : kOOB;
        F4 x0 = TSNET_BUF_LOAD16(rsx, v, 0u), x1 = TSNET_BUF_LOAD16(rsx, v, 16u);
#pragma unroll
        for (int e = 0; e < 4; ++e) { x0.v[e] *= in_scale; x1.v[e] *= in_scale; }
        if (pp < 384) {                                               // slots 380..383 hold zeros (never read, kept finite)
            F4 Hh, Ll;
            if (NPROD == 1) {
                bf16_octet(x0, x1, Hh);
                *reinterpret_cast<F4*>(smem_raw + pp * 16) = Hh;
            } else {
                split_h2_octet(x0, x1, Hh, Ll);
                *reinterpret_cast<F4*>(smem_raw + pp * 16) = Hh;
                *reinterpret_cast<F4*>(smem_raw + PLANE_S + pp * 16) = Ll;
            }
        }
    }
    __syncthreads();

    // per-lane address of step st: slot of tap 2 st for lh = 0; for lh = 1 the next tap = one slot right, or the first slot of the next patch
    // row when tap 2 st ends its row, or (last step: the 50th tap does not exist, its weights are zero) the same slot again
    const int base = (wrow * MT * PC + li) * 16;
    auto load_a = [&](int set, int st) __attribute__((always_inline)) {
        const int t0 = 2 * st, ky = t0 / 7, kx = t0 - ky * 7;        // wave-uniform
        const int delta = st >= 24 ? 0 : (kx == 6 ? (PC - 6) * 16 : 16);
        const unsigned char* b = smem_raw + base + (ky * PC + kx) * 16 + lh * delta;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int p = 0; p < NPL; ++p) af[set][p][i] = *reinterpret_cast<const F4*>(b + p * PLANE_S + i * PC * 16);
    };

    f32x16 acc[MT], tot[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; tot[i][0][r] = 0.f; }
    auto product = [&](int sa, int sb, int pa, int pb, bool fresh) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            f32x16 c = acc[i];
            if (fresh) {
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = 0.f;
            }
            if (NPROD == 1) acc[i] = TSNET_MFMA_BF16(af[sa][pa][i], bf[sb][pb], c);
            else acc[i] = TSNET_MFMA_F16(af[sa][pa][i], bf[sb][pb], c);
        }
    };
    // step st = 6 c + j: A(st) in set j & 1, B(st) in set j % 3 (six steps per chain keep both rotations static); issues A(st + 1), B(st + 2) first
    auto step = [&](int st, int j) __attribute__((always_inline)) {
        load_b((j + 2) % 3, st + 2);
        load_a((j + 1) & 1, st + 1 < 25 ? st + 1 : 24);
        const int SA = j & 1, SB = j % 3;
        if (NPROD == 1) {
            product(SA, SB, 0, 0, j == 0);
        } else {
            product(SA, SB, 1, 0, j == 0);                            // lo * hi
            product(SA, SB, 0, 1, false);                             // hi * lo
            product(SA, SB, 0, 0, false);                             // hi * hi
        }
    };
    auto fold = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i) tot[i][0] += acc[i];
    };
    load_a(0, 0);
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {                                     // four chains of six k-groups, then the 25th
        const int s0 = 6 * c;
        step(s0, 0); step(s0 + 1, 1); step(s0 + 2, 2); step(s0 + 3, 3); step(s0 + 4, 4); step(s0 + 5, 5);
        fold();
    }
    step(24, 0);
    fold();

    const float unscale = a.w_unscale ? in_unscale * a.w_unscale[0] : in_unscale;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][0][r] *= unscale;                     // exact: power of two
    const int m_img = img * a.Ho * a.Wo;
    __syncthreads();                                                  // the epilogue reuses the patch region for its reduction
    x3_epilogue<BN, WARPS_M, WARPS_N, MT, NTL>(a, tot, smem_raw, tid, wave, n0, (size_t)img * tper + tin,
                                               [&](int l) { return m_img + (oy0 + (l >> 5)) * a.Wo + ox0 + (l & 31); });
}

template <int NPROD>
__global__ __launch_bounds__(256, 3)
void conv_h2s_kernel(H2Args a) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int bid = x3p_item(blockIdx.x, a.tiles_m * a.tiles_n);
    const int tile_m = bid / a.tiles_n;
    h2s_tile<NPROD>(a, smem_raw, tile_m, (bid - tile_m * a.tiles_n) * 64);
}

// ---------------------------------------------------------------------------------------------------------------
// h2d: the encoders' 3 x 3 / stride-2 / zero-pad-1 downsampling convolutions (TSNet.py:70) as a patch kernel -- h2_tile with the patch
// geometry of stride 2.  A 4 x 32 output rectangle reads the (2*4+1) x (2*32+1) = 9 x 65 input patch; per 16-channel slab it is fetched once
// (585 pixels x 64 B), transformed and split like h2_tile's, and written to LDS with its columns DE-INTERLEAVED by parity: row pitch 66
// slots = 33 even columns, then 32 odd ones.  Output column x under tap column kx reads input column 2x + kx: kx = 0 -> even slot x,
// kx = 1 -> odd slot x, kx = 2 -> even slot x + 1, so the 32 lanes of a fragment read 32 consecutive 16-byte slots (conflict-free) and
// row / tap shifts are immediates, exactly as in the stride-1 kernel.  Against the implicit GEMM (conv_h2r) on these layers: half the
// staged elements (the im2col tile holds every input element 2.25 times), no global A traffic per k-step, one barrier per slab instead of
// one per k-step.  Five staging rounds per slab (19 blocks of 32 pixel slots over four waves), spread over the nine taps on two register
// sets.  K order, chains and fold points are h2_tile's.
template <int BN, int WARPS_M, int WARPS_N, int NPROD, bool AFFINE>
__device__ __forceinline__ void h2d_tile(const H2Args& a, unsigned char* smem_raw, const int tile_m, const int n0) {
    constexpr int BM = kPatchRows * kPatchCols;
    constexpr int NW = WARPS_M * WARPS_N;
    static_assert(NW == 4 && WARPS_M == 2, "four waves, two output rows per wave");
    static_assert(NPROD == 1 || NPROD == 3, "one (bf16 operands) or three products");
    constexpr int NPL = NPROD == 1 ? 1 : 2;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 32, NTL = WN / 32;
    constexpr int PCI = 2 * kPatchCols + 1, PRI = 2 * kPatchRows + 1, PP = PRI * PCI;     // 65 x 9 = 585 patch pixels
    constexpr int RP = 66;                                           // row pitch in slots: 33 even columns, 32 odd, 1 spare
    constexpr int REGION = PRI * RP * 16;                            // one octet region: 594 slots x 16 B
    constexpr int PLANE_P = 2 * REGION, PATCH_BYTES = NPL * PLANE_P;
    constexpr int OFF_SCRATCH = 2 * 2 * PLANE_P;                     // sink of the pixel block that does not exist (fixed offsets in both modes)
    constexpr int OFF_TAB = OFF_SCRATCH + 2048;
    constexpr int NR = 5;                                            // staging rounds: blocks wave, wave + 4, .., wave + 16 of 32 pixels

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int wrow = wave / WARPS_N;
    const int wn0 = (wave % WARPS_N) * WN;
    const int li = lane & 31, lh = lane >> 5;

    const int tcols = a.Wo / kPatchCols, tper = (a.Ho / kPatchRows) * tcols;
    const int img = tile_m / tper, tin = tile_m - img * tper;
    const int oy0 = (tin / tcols) * kPatchRows, ox0 = (tin % tcols) * kPatchCols;
    const int ncc = a.Cin >> 4;
    float in_scale = a.in_scale, in_unscale = a.in_unscale;
    if (NPROD != 1 && a.in_amax) h2_device_scale(a.in_amax + img, a.in_bound_add, in_scale, in_unscale);

    const size_t planew = (size_t)((a.nchunks + 1) / 2 * 2) * a.Npad * 16;
    const tsnet_brsrc_t rsx = tsnet_make_brsrc(a.x, (unsigned)((size_t)a.N * a.H * a.W * a.Cin * 4));
    tsnet_brsrc_t rsw[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) rsw[p] = tsnet_make_brsrc(a.w + p * planew, (unsigned)(planew * 2));

    // ---- staging geometry: round r of this wave = pixel block wave + 4 r; lane -> (pixel b*32 + (lane & 31), octet lane >> 5)
    const int oct = lane >> 5;
    unsigned vP[NR];
    auto slot_of = [&](int r) __attribute__((always_inline)) {       // LDS byte offset of the lane's slot inside an octet region, or -1 (sink); recomputed, not kept
        const int pp = (wave + 4 * r) * 32 + (lane & 31);
        const int pr = pp / PCI, pc = pp - pr * PCI;
        return pp < PP ? (pr * RP + ((pc & 1) ? 33 + (pc >> 1) : (pc >> 1))) * 16 : -1;
    };
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int pp = (wave + 4 * r) * 32 + (lane & 31);
        const int pr = pp / PCI, pc = pp - pr * PCI;
        const int iy = 2 * oy0 - 1 + pr, ix = 2 * ox0 - 1 + pc;
        const bool inside = pp < PP;
        const bool ok = inside && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;       // zero padding
        vP[r] = ok ? (unsigned)(((img * a.H * a.W + iy * a.W + ix) * a.Cin + oct * 8) * 4) : kOOB;
    }
    float* tab = reinterpret_cast<float*>(smem_raw + OFF_TAB);       // [Cin] alpha*s, then [Cin] beta*s
    if (AFFINE) {
        for (int c = tid; c < a.Cin; c += 256) {
            tab[c] = a.in_alpha[(size_t)img * a.Cin + c] * in_scale;
            tab[a.Cin + c] = a.in_beta[(size_t)img * a.Cin + c] * in_scale;
        }
        __syncthreads();
    }
    const float relu_floor = a.in_relu ? 0.f : -__builtin_inff();
    F4 sx[1][2];                                                     // staging registers: one round of x in flight
    auto stage_load_x = [&](int cn, int r, int set) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) sx[set][q] = TSNET_BUF_LOAD16(rsx, vP[r], (unsigned)(cn * 64 + q * 16));
    };
    auto stage_store = [&](int cn, int r, int set) __attribute__((always_inline)) {
        F4 t[2];
        const float keep = vP[r] == kOOB ? 0.f : 1.f;               // a padded pixel is zero AFTER the transform
        if (AFFINE) {
            const float* ta = tab + (cn < ncc ? cn * 16 : 0) + oct * 8;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const F4 al = *reinterpret_cast<const F4*>(ta + q * 4), be = *reinterpret_cast<const F4*>(ta + a.Cin + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = __builtin_fmaf(sx[set][q].v[e], al.v[e], be.v[e]);
                    v = v > relu_floor ? v : relu_floor;
                    t[q].v[e] = v * keep;
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = sx[set][q].v[e] * in_scale;     // a padded slot loaded zeros
                    t[q].v[e] = v > relu_floor ? v : relu_floor;
                }
        }
        const int so = slot_of(r);
        const bool real = so >= 0;
        unsigned char* dst = smem_raw + (real ? (cn & 1) * PATCH_BYTES + oct * REGION + so : OFF_SCRATCH + (lane & 63) * 16);
        F4 Hh, Ll;
        if (NPROD == 1) {
            bf16_octet(t[0], t[1], Hh);
            *reinterpret_cast<F4*>(dst) = Hh;
        } else {
            split_h2_octet(t[0], t[1], Hh, Ll);
            *reinterpret_cast<F4*>(dst) = Hh;
            *reinterpret_cast<F4*>(dst + (real ? PLANE_P : 1024)) = Ll;
        }
    };

    // ---- fragments
    const unsigned vB = (unsigned)((wn0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16);
    F4 af[2][NPL][MT], bf[3][NPL][NTL];
    auto load_b = [&](int set, int cc, int t) __attribute__((always_inline)) {
        const int kc = t * ncc + cc;
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int j = 0; j < NTL; ++j) bf[set][p][j] = TSNET_BUF_LOAD16(rsw[p], vB, (unsigned)((kc * a.Npad + n0 + j * 32) * 32));
    };
    const unsigned char* abase = smem_raw + lh * REGION + (2 * wrow * MT * RP + li) * 16;
    auto load_a = [&](int set, int cc, int t) __attribute__((always_inline)) {
        const int ky = t / 3, kx = t - ky * 3;
        const unsigned char* pbase = abase + (cc & 1) * PATCH_BYTES;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                af[set][p][i] = *reinterpret_cast<const F4*>(pbase + p * PLANE_P + ((2 * i + ky) * RP + (kx == 1 ? 33 : (kx >> 1))) * 16);
    };

    f32x16 acc[MT][NTL], tot[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
    auto product = [&](int sa, int sb, int pa, int pb, bool fresh) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                f32x16 c = acc[i][j];
                if (fresh) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = 0.f;
                }
                if (NPROD == 1) acc[i][j] = TSNET_MFMA_BF16(af[sa][pa][i], bf[sb][pb][j], c);
                else acc[i][j] = TSNET_MFMA_F16(af[sa][pa][i], bf[sb][pb][j], c);
            }
    };
    // staging of slab cc + 1 over the nine taps of slab cc, one round in flight at a time (registers are the scarce resource at 128-wide tiles):
    //   round 0: load at tap 0, store at tap 1;  1: 1 -> 3;  2: 3 -> 4;  3: 4 -> 6;  4: 6 -> 8   (a store precedes the next load of its tap)
    auto step = [&](int cc, int t, int SA) __attribute__((always_inline)) {
        const bool fresh = t == 0 || t == 4;
        const int t2 = (t + 2) % 9;
        load_b(t2 % 3, cc + (t + 2 >= 9 ? 1 : 0), t2);
        if (t < 8) load_a(SA ^ 1, cc, t + 1);
        if (t == 1) stage_store(cc + 1, 0, 0);
        if (t == 3) stage_store(cc + 1, 1, 0);
        if (t == 4) stage_store(cc + 1, 2, 0);
        if (t == 6) stage_store(cc + 1, 3, 0);
        if (t == 0) stage_load_x(cc + 1, 0, 0);
        if (t == 1) stage_load_x(cc + 1, 1, 0);
        if (t == 3) stage_load_x(cc + 1, 2, 0);
        if (t == 4) stage_load_x(cc + 1, 3, 0);
        if (t == 6) stage_load_x(cc + 1, 4, 0);
        const int SB = t % 3;
        if (NPROD == 1) {
            product(SA, SB, 0, 0, fresh);
        } else {
            product(SA, SB, NPL - 1, 0, fresh);                      // lo * hi
            product(SA, SB, 0, NPL - 1, false);                      // hi * lo
            product(SA, SB, 0, 0, false);                            // hi * hi
        }
        if (t == 8) stage_store(cc + 1, 4, 0);
        if (t == 3 || t == 8) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) tot[i][j] += acc[i][j];
        }
    };
    auto slab = [&](int cc, int S0) __attribute__((always_inline)) {
        __syncthreads();                                             // patch(cc) complete and visible; slab cc-1 fully read
        load_a(S0, cc, 0);
        step(cc, 0, S0); step(cc, 1, S0 ^ 1); step(cc, 2, S0);
        step(cc, 3, S0 ^ 1); step(cc, 4, S0); step(cc, 5, S0 ^ 1);
        step(cc, 6, S0); step(cc, 7, S0 ^ 1); step(cc, 8, S0);
    };

    // prologue: patch of slab 0, weight fragments of steps (0,0) and (0,1)
#pragma unroll
    for (int r = 0; r < NR; ++r) { stage_load_x(0, r, 0); stage_store(0, r, 0); }
    load_b(0, 0, 0);
    load_b(1, 0, 1);
    int cc = 0;
    for (; cc + 2 <= ncc; cc += 2) { slab(cc, 0); slab(cc + 1, 1); }
    if (cc < ncc) slab(cc, 0);

    const float unscale = a.w_unscale ? in_unscale * a.w_unscale[0] : in_unscale;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[i][j][r] *= unscale;
    const int m_img = img * a.Ho * a.Wo;
    __syncthreads();                                                 // the epilogue reuses the patch region
    x3_epilogue<BN, WARPS_M, WARPS_N, MT, NTL>(a, tot, smem_raw, tid, wave, n0, (size_t)img * tper + tin,
                                               [&](int l) { return m_img + (oy0 + (l >> 5)) * a.Wo + ox0 + (l & 31); });
}

constexpr int kH2dLds = 2 * 2 * 2 * (2 * kPatchRows + 1) * 66 * 16 + 2048;      // two stages x two planes x two octet regions + sink (+ 2 Cin floats x 2 of the table)

template <int BN, int NPROD, bool AFFINE>
__global__ __launch_bounds__(256, 2)
void conv_h2d_kernel(H2Args a) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int bid = x3p_item(blockIdx.x, a.tiles_m * a.tiles_n);
    const int tile_m = bid / a.tiles_n;
    h2d_tile<BN, 2, 2, NPROD, AFFINE>(a, smem_raw, tile_m, (bid - tile_m * a.tiles_n) * BN);
}

// OIHW fp32 -> two fp16 planes of w * scale in the fragment order of pack_weights_x3_kernel:
//   out[p][((kc*Npad + n)*2 + o)*8 + e] = part_p( scale * W[k = kc*16 + (o ^ ((n>>3)&1))*8 + e][n] ),  k = tap*cin_pad + c
__global__ void pack_weights_h2_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, float scale,
                                       int cout, int cin_real, int cin_pad, int ks, int kpad, int npad, int cin_total, int cin_off) {
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
        if (tap < ks * ks && c < cin_real && n < cout) {
            const int ky = tap / ks, kx = tap - ky * ks;
            v = w[(((size_t)n * cin_total + cin_off + c) * ks + ky) * ks + kx];
        }
        unsigned hi, lo;
        split_h2(v * scale, hi, lo);
        out[idx] = (unsigned short)hi; out[plane + idx] = (unsigned short)lo;
    }
}



// ---------------------------------------------------------------------------------------------------------------
// h2r: the same arithmetic (fp32 input, producer InstanceNorm + ReLU applied on load, fp16 x 2 operands, three products, or one bf16
// product) as an implicit GEMM for the layers the patch tile does not take: the 3 x 3 / stride-2 / zero-pad downsampling convolutions
// of the encoders (TSNet.py:70).  Structure of conv_x3r.hpp: a thread stages one (row, 8-channel octet) slot of the im2col A tile per
// 16-deep step -- two 16-byte fp32 loads two steps ahead, transformed and split in registers, two ds_writes -- weight fragments go
// straight into registers one step ahead (two register sets: the loop is unrolled by four, so parities are static), one barrier per
// step, chains of four k-groups folded into the running total.  K order is tap-major (k = tap * Cin + c), like conv_x3r.
// What it removes next to conv_x3r on these layers: the norm_act pass that materialised relu(IN(x)) as three bf16 planes (10 bytes per
// element of the largest activations of the network), half of the MFMA products and a third of the gathered bytes.
struct H2rArgs : H2Args {
    int stride, pad, taps, cin_log2;
};

// SMALL_CIN: Cin = 8 (the 7 x 7 stems at label_nc = 2: 3 + 2 + 3 coordinate channels = 8, or 2 + 3 padded to 8): the two octets of a
// 16-deep k-group are two different taps of the same pixel row, as in conv_x3r.hpp
template <int KS, int BN, int WARPS_M, int WARPS_N, int NPROD, bool AFFINE, bool SMALL_CIN = false>
__global__ __launch_bounds__(256, 2)
void conv_h2r_kernel(H2rArgs a) {
    constexpr int BM = 128;
    constexpr int NW = WARPS_M * WARPS_N;
    static_assert(NW == 4, "256 threads: one 16-byte slot of the A tile per thread and plane");
    static_assert(NPROD == 1 || NPROD == 3, "one (bf16 operands) or three products");
    constexpr int NPL = NPROD == 1 ? 1 : 2;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 32, NTL = WN / 32;
    constexpr int PLANE_A = BM * 32, STAGE = 2 * PLANE_A;           // 8 KiB per stage (two planes)
    constexpr int OFF_TAB = 2 * STAGE;                               // (alpha*s, beta*s) of the tile's image: 2 x Cin floats

    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int wm0 = (wave / WARPS_N) * WM, wn0 = (wave % WARPS_N) * WN;
    const int li = lane & 31, lh = lane >> 5;

    const int bid = x3p_item(blockIdx.x, a.tiles_m * a.tiles_n);
    const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int hw = a.Ho * a.Wo;
    const int img = m0 / hw;                                         // a tile lies inside one image (hw % 128 == 0, checked on the host)
    float in_scale = a.in_scale, in_unscale = a.in_unscale;
    if (NPROD != 1 && a.in_amax) h2_device_scale(a.in_amax + img, a.in_bound_add, in_scale, in_unscale);

    const size_t planew = (size_t)((a.nchunks + 1) / 2 * 2) * a.Npad * 16;
    const tsnet_brsrc_t rsx = tsnet_make_brsrc(a.x, (unsigned)((size_t)a.N * a.H * a.W * a.Cin * 4));
    tsnet_brsrc_t rsw[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) rsw[p] = tsnet_make_brsrc(a.w + p * planew, (unsigned)(planew * 2));

    float* tab = reinterpret_cast<float*>(smem_raw + OFF_TAB);
    if (AFFINE) {
        for (int c = tid; c < a.Cin; c += 256) {
            tab[c] = a.in_alpha[(size_t)img * a.Cin + c] * in_scale;
            tab[a.Cin + c] = a.in_beta[(size_t)img * a.Cin + c] * in_scale;
        }
        __syncthreads();
    }
    const float relu_floor = a.in_relu ? 0.f : -__builtin_inff();

    // ---- A staging: thread t owns row t/2, physical octet t&1 (LDS slot t*16 inside a plane), logical octet swizzled by bit 3 of the row
    const int srow = tid >> 1;
    const int oct_log = (tid & 1) ^ ((srow >> 3) & 1);
    int s_pix, s_oy, s_ox;
    {
        const int rem = (m0 + srow) - img * hw;
        const int oy = rem / a.Wo;
        s_pix = img * a.H * a.W;
        s_oy = oy * a.stride - a.pad;
        s_ox = (rem - oy * a.Wo) * a.stride - a.pad;
    }
    const int cpt_log2 = SMALL_CIN ? 0 : a.cin_log2 - 4;
    F4 ar[2][2];                                                     // register stage: [set][half of the octet]
    float am[2];                                                     // 1, or 0 where the tap lies in the zero padding / past the last tap
    int ac0[2];                                                      // first channel of the staged octet (for the transform table)
    auto load_a = [&](int kc, int set) __attribute__((always_inline)) {
        int tap, c0;
        if (SMALL_CIN) { tap = kc * 2 + oct_log; c0 = 0; }
        else { tap = kc >> cpt_log2; c0 = ((kc << 4) & (a.Cin - 1)) + oct_log * 8; }      // tap: wave-uniform
        const int ky = tap / KS, kx = tap - ky * KS;
        int iy = s_oy + ky, ix = s_ox + kx;
        bool ok = tap < a.taps;
        if (a.reflect) {
            iy = iy < 0 ? -iy : iy;
            iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
            ix = ix < 0 ? -ix : ix;
            ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
        } else {
            ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        }
        const unsigned v = ok ? (unsigned)(((s_pix + iy * a.W + ix) * a.Cin + c0) * 4) : kOOB;
        ar[set][0] = TSNET_BUF_LOAD16(rsx, v, 0u);
        ar[set][1] = TSNET_BUF_LOAD16(rsx, v, 16u);
        am[set] = ok ? 1.f : 0.f;
        ac0[set] = ok ? c0 : 0;
    };
    auto store_a = [&](int set, int stage) __attribute__((always_inline)) {
        F4 t[2];
        if (AFFINE) {
            const float* ta = tab + ac0[set];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const F4 al = *reinterpret_cast<const F4*>(ta + q * 4), be = *reinterpret_cast<const F4*>(ta + a.Cin + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = __builtin_fmaf(ar[set][q].v[e], al.v[e], be.v[e]);
                    v = v > relu_floor ? v : relu_floor;
                    t[q].v[e] = v * am[set];
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = ar[set][q].v[e] * in_scale;
                    t[q].v[e] = v > relu_floor ? v : relu_floor;
                }
        }
        unsigned char* dst = smem_raw + stage * STAGE + tid * 16;
        F4 Hh, Ll;
        if (NPROD == 1) {
            bf16_octet(t[0], t[1], Hh);
            *reinterpret_cast<F4*>(dst) = Hh;
        } else {
            split_h2_octet(t[0], t[1], Hh, Ll);
            *reinterpret_cast<F4*>(dst) = Hh;
            *reinterpret_cast<F4*>(dst + PLANE_A) = Ll;
        }
    };

    // ---- fragments
    const int a_off = (wm0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16;
    const unsigned vB = (unsigned)((wn0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16);
    F4 af[NPL][MT], bf[2][NPL][NTL];
    auto frag_a = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int i = 0; i < MT; ++i) af[p][i] = *reinterpret_cast<const F4*>(smem_raw + stage * STAGE + p * PLANE_A + i * 1024 + a_off);
    };
    auto load_b = [&](int set, int kc) __attribute__((always_inline)) {       // past the end of K the descriptor returns zeros
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int j = 0; j < NTL; ++j) bf[set][p][j] = TSNET_BUF_LOAD16(rsw[p], vB, (unsigned)((kc * a.Npad + n0 + j * 32) * 32));
    };

    f32x16 acc[MT][NTL], tot[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
    auto product = [&](int sb, int pa, int pb, bool fresh) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                f32x16 c = acc[i][j];
                if (fresh) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = 0.f;
                }
                if (NPROD == 1) acc[i][j] = TSNET_MFMA_BF16(af[pa][i], bf[sb][pb][j], c);
                else acc[i][j] = TSNET_MFMA_F16(af[pa][i], bf[sb][pb][j], c);
            }
    };

    // step kc (u = kc mod 4, static): A(kc) in LDS stage u&1, A(kc+1) in register set (u+1)&1, B(kc) in bf[u&1]
    auto step = [&](int kc, int u) __attribute__((always_inline)) {
        __syncthreads();                                             // stage u&1 complete; stage (u+1)&1 no longer read
        frag_a(u & 1);
        load_b((u + 1) & 1, kc + 1);
        store_a((u + 1) & 1, (u + 1) & 1);                           // A(kc+1): loaded during step kc-1
        load_a(kc + 2, u & 1);                                       // register set u&1 held A(kc), already in LDS
        if (NPROD == 1) {
            product(u & 1, 0, 0, u == 0);
        } else {
            product(u & 1, 1, 0, u == 0);                            // lo * hi; chains of 4 k-groups counted from k = 0
            product(u & 1, 0, 1, false);                             // hi * lo
            product(u & 1, 0, 0, false);                             // hi * hi
        }
        if (u == 3) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) tot[i][j] += acc[i][j];
        }
    };

    load_a(0, 0);
    load_a(1, 1);
    store_a(0, 0);
    load_b(0, 0);
    const int nst = a.nchunks;
    int kc = 0;
    for (; kc + 4 <= nst; kc += 4) { step(kc, 0); step(kc + 1, 1); step(kc + 2, 2); step(kc + 3, 3); }
    if (kc < nst) {                                                  // 1..3 trailing k-groups: a last, partial chain
        step(kc, 0);
        if (kc + 1 < nst) step(kc + 1, 1);
        if (kc + 2 < nst) step(kc + 2, 2);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) tot[i][j] += acc[i][j];
    }

    const float unscale = a.w_unscale ? in_unscale * a.w_unscale[0] : in_unscale;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[i][j][r] *= unscale;
    x3_epilogue<BN, WARPS_M, WARPS_N, MT, NTL>(a, tot, smem_raw, tid, wave, n0, (size_t)img * (hw / BM) + (m0 - img * hw) / BM,
                                               [&](int l) { const int m = m0 + l; return m < a.M ? m : -1; });
}

}  // namespace tsnet
