// conv_h2s32.hpp -- the 7 x 7 stems at 32 input channels as a PATCH kernel (round 6): the pose model's encoders (TSNet_pose: label_nc = 25 ->
// image 3 + label 25 + coordinates 3 = 31, and label 25 + coordinates 3 = 28, both padded to 32; TSNet.py:66, demo_pose.py:120-124).  Until
// round 5 these two layers ran on the general implicit GEMM (conv_h2r<7>): every 16-deep step gathers 128 rows x 16 k of fp32 again -- 49 taps
// re-read and re-split every input element 49 times -- 653 + 231 us of a 6.3 ms configs[3] forward (profiles/round6_kernel_trace_cfg3.txt).
// Here, as in the 8-channel stem (conv_h2.hpp h2s_tile):
//   * the (4+6) x (32+6) x 32-channel patch of a 4 x 32 output rectangle is fetched ONCE (48.6 KB of fp32; the 8 lanes of a pixel read its 32
//     channels as one 128-byte line), scaled, split and written as 8-byte halves of the fragment octets into an octet-planar LDS image
//     ([plane][16-channel group][octet][pixel slot] x 16 B: the 32 lanes of a fragment read 32 consecutive slots, conflict-free; the regions
//     are padded by 32 B so that a 16-lane write group -- two pixels x four regions x two halves -- covers all banks);
//   * the 98 k-steps (49 taps x two 16-channel groups) read it through shifted views: no barrier and no global A traffic inside the loop,
//     weight fragments straight into registers three steps ahead (four sets), A fragments one step ahead;
//   * K order and chains are conv_h2r's (k = tap * 32 + c; chains of four k-groups from k = 0, a last partial chain of two): THE SAME BITS as
//     the general kernel on the same layer (tests: torch.equal), so the layer's kernel could change without touching a tolerance.
// A tile = 4 x 32 output pixels x 64 channels, four waves 2 x 2, wave tile 64 x 32; 49 KiB of LDS: three workgroups per CU.
#pragma once
#include "conv_common.hpp"

namespace tsnet {

constexpr int kS32Region = 384 * 16 + 32;                        // one (group, octet) region of a plane: 384 pixel slots x 16 B + the bank pad
constexpr int kS32Plane = 4 * kS32Region;
constexpr int h2s32_lds_bytes(int npl) { return npl * kS32Plane < 8192 + 64 ? 8192 + 64 : npl * kS32Plane; }    // (the epilogue's flag word sits at 8192)

template <int NPROD>
__global__ __launch_bounds__(256, 3)
void conv_h2s32_kernel(ConvArgs a) {
    constexpr int BN = 64, WARPS_M = 2, WARPS_N = 2, MT = 2, NTL = 1;
    constexpr int NPL = NPROD == 1 ? 1 : 2;
    constexpr int PC = kPatchCols + 6, PRW = kPatchRows + 6, PP = PRW * PC;     // 38 x 10 = 380 patch pixels
    constexpr int REGION = kS32Region, PLANE = kS32Plane;
    constexpr int NSTEPS = 98;                                                  // 49 taps x two 16-channel groups
    static_assert(NPROD == 1 || NPROD == 3, "one (bf16 operands) or three products");

    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int wrow = wave / WARPS_N;
    const int wn0 = (wave % WARPS_N) * 32;
    const int li = lane & 31, lh = lane >> 5;

    const int bid = xcd_item(blockIdx.x, a.tiles_m * a.tiles_n);
    const int tile_m = bid / a.tiles_n;
    const int n0 = (bid - tile_m * a.tiles_n) * BN;
    const int tcols = a.Wo / kPatchCols, tper = (a.Ho / kPatchRows) * tcols;
    const int img = tile_m / tper, tin = tile_m - img * tper;
    const int oy0 = (tin / tcols) * kPatchRows, ox0 = (tin % tcols) * kPatchCols;
    float in_scale = a.in_scale, in_unscale = a.in_unscale;
    if (NPROD != 1 && a.in_amax) h2_device_scale(a.in_amax + img, a.in_bound_add, in_scale, in_unscale);

    const size_t planew = (size_t)((a.nchunks + 1) / 2 * 2) * a.Npad * 16;
    const tsnet_brsrc_t rsx = tsnet_make_brsrc(a.x, (unsigned)((size_t)a.N * a.H * a.W * 32 * 4));
    tsnet_brsrc_t rsw[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) rsw[p] = tsnet_make_brsrc(a.w + p * planew, (unsigned)(planew * 2));

    // ---- weight fragments first (their latency hides behind the patch staging), three steps ahead afterwards
    const unsigned vB = (unsigned)((wn0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16);
    F4 af[2][NPL][MT], bf[4][NPL];
    auto load_b = [&](int set, int kc) __attribute__((always_inline)) {           // past the end of K the descriptor returns zeros
#pragma unroll
        for (int p = 0; p < NPL; ++p) bf[set][p] = TSNET_BUF_LOAD16(rsw[p], vB, (unsigned)((kc * a.Npad + n0) * 32));
    };
    load_b(0, 0); load_b(1, 1); load_b(2, 2);

    // ---- patch staging: 384 pixel slots x 8 channel quads = 3072 lane slots, twelve per thread (slots 380..383 hold zeros: never read, kept
    //      finite); lane -> (pixel, quad): the eight lanes of a pixel read one 128-byte line; reflection padding resolved in the address
    {
        constexpr int ROUNDS = 12;
        F4 x[ROUNDS];
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int idx = tid + r * 256, pp = idx >> 3, q8 = idx & 7;
            const int pr = pp / PC, pc = pp - pr * PC;
            int iy = oy0 - 3 + pr, ix = ox0 - 3 + pc;
            iy = iy < 0 ? -iy : iy;
            iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
            ix = ix < 0 ? -ix : ix;
            ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
            const unsigned v = pp < PP ? (unsigned)((((img * a.H + iy) * a.W) + ix) * 128 + q8 * 16) : kOOB;
            x[r] = TSNET_BUF_LOAD16(rsx, v, 0u);
        }
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int idx = tid + r * 256, pp = idx >> 3, q8 = idx & 7;
            // group q8 >> 2, octet (q8 >> 1) & 1, half octet q8 & 1
            unsigned char* dst = smem_raw + ((q8 >> 2) * 2 + ((q8 >> 1) & 1)) * REGION + pp * 16 + (q8 & 1) * 8;
            float t[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] = x[r].v[e] * in_scale;
            if (NPROD == 1) {
                uint2 w;
                w.x = TSNET_CVT_PK_BF16(t[0], t[1]); w.y = TSNET_CVT_PK_BF16(t[2], t[3]);
                *reinterpret_cast<uint2*>(dst) = w;
            } else {
                unsigned h0, l0, h1, l1;
                TSNET_SPLIT_2PAIRS(t[0], t[1], t[2], t[3], h0, l0, h1, l1);
                uint2 hw2, lw2;
                hw2.x = h0; hw2.y = h1; lw2.x = l0; lw2.y = l1;
                *reinterpret_cast<uint2*>(dst) = hw2;
                *reinterpret_cast<uint2*>(dst + PLANE) = lw2;
            }
        }
    }
    __syncthreads();

    // slot offset of every tap (ky * 38 + kx) x 16 B: a table in constant memory, read with scalar loads (the step index is wave-uniform)
    struct TapTab { int off[50]; };
    static constexpr TapTab kTap = [] {
        TapTab t{};
        for (int tp = 0; tp < 50; ++tp) { const int tq = tp < 49 ? tp : 48; t.off[tp] = ((tq / 7) * PC + tq % 7) * 16; }
        return t;
    }();
    const unsigned char* abase = smem_raw + lh * REGION + (wrow * MT * PC + li) * 16;
    auto load_a = [&](int set, int tap, int g) __attribute__((always_inline)) {   // tap: wave-uniform
        const unsigned char* b = abase + kTap.off[tap] + g * 2 * REGION;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int p = 0; p < NPL; ++p) af[set][p][i] = *reinterpret_cast<const F4*>(b + p * PLANE + i * PC * 16);
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
    // step s = 4 c + j (tap 2 c + (j >> 1), group j & 1): A(s) in set j & 1, B(s) in set j; issues A(s + 1) and B(s + 3) first
    auto step = [&](int c, int j) __attribute__((always_inline)) {
        const int s = 4 * c + j;
        load_b((j + 3) & 3, s + 3);
        const int jn = j + 1;                                                     // the next step: tap 2 c + (jn >> 1) (jn = 4: the next chain's first)
        load_a(jn & 1, 2 * c + (jn >> 1), jn & 1);
        if (NPROD == 1) {
            product(j & 1, j, 0, 0, j == 0);
        } else {
            product(j & 1, j, 1, 0, j == 0);                                      // lo * hi; chains of four k-groups counted from k = 0 (conv_h2r's association)
            product(j & 1, j, 0, 1, false);                                       // hi * lo
            product(j & 1, j, 0, 0, false);                                       // hi * hi
        }
    };
    auto fold = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i) tot[i][0] += acc[i];
    };
    static_assert(NSTEPS == 24 * 4 + 2, "24 whole chains, then the two groups of the last tap");
    load_a(0, 0, 0);
#pragma unroll 1
    for (int c = 0; c < 24; ++c) {
        step(c, 0); step(c, 1); step(c, 2); step(c, 3);
        fold();
    }
    step(24, 0); step(24, 1);                                                     // tap 48 (their look-ahead reads tap 48 again and zeros past K)
    fold();

    const float unscale = a.w_unscale ? in_unscale * a.w_unscale[0] : in_unscale;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][0][r] *= unscale;                     // exact: power of two
    const int m_img = img * a.Ho * a.Wo;
    __syncthreads();                                                              // the epilogue reuses the patch region for its reduction
    conv_epilogue<BN, WARPS_M, WARPS_N, MT, NTL>(a, tot, smem_raw, tid, wave, n0, (size_t)img * tper + tin,
                                                 [&](int l) { return m_img + (oy0 + (l >> 5)) * a.Wo + ox0 + (l & 31); });
}

}  // namespace tsnet
