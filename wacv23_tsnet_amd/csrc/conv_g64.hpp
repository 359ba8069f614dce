// conv_g64.hpp -- the general implicit GEMM with 64-deep K steps (round 6).  Same arithmetic as conv_h2r.hpp -- fp32 input, the producer's
// InstanceNorm + ReLU applied on load, fp16 x 2 operands / three products or one bf16 product, K order tap-major (k = tap * Cin + c), chains of
// four 16-deep k-groups folded into a running fp32 total, fp64 statistics in the shared epilogue -- so a layer whose input channels are a
// multiple of 64 gets THE SAME BITS from either kernel (tests: torch.equal against conv_h2r).  What differs is the schedule.  conv_h2r stages one
// (row, 8-channel octet) slot per thread and 16-deep step: a barrier, an address computation and two half-used cache lines per 32 bytes --
// 17.9 (fp16 x 2) to 36.9 (bf16) non-MFMA instructions per MFMA on the layers it carries (profiles/round5_pmc_summary*.txt): the two 1 x 1
// convolutions (fuse_net.conv, TSNet.py:193; dec.map_conv on cat(pg, sg), :139, :163), the 64 -> 128 stride-2 layer (:70) and, with bf16
// operands, every stride-2 layer.  Here a step is one (tap, 64-channel block):
//   * the eight lanes of a row read its 64 channels as two whole 128-byte lines (lane = channel quad; conv_w1's producer layout), transform
//     and split them in registers and write 8-byte halves of the fragment octets (ds_write_b64, conflict-free with the padded group pitch);
//   * ONE barrier and one address computation per 64-deep step (a quarter of conv_h2r's), two LDS stages;
//   * every wave's tile is 64 rows x 32 channels: two A fragments per plane and group from LDS, weight fragments straight into registers a
//     whole step ahead (four sets: the group index is the set index, static).
// Tiles: BM = 64 rows (four waves side by side over 128 channels: the 1 x 1 layers' 4096 rows make 256 workgroups) or 128 rows (eight waves,
// 2 x 4); BN = 128.  An M tile lies inside one image (ragged last tile), as in conv_h2r.
#pragma once
#include "conv_common.hpp"

namespace tsnet {

constexpr int g64_group_pitch(int BM) { return BM * 32 + 64; }       // bytes of one 16-deep k-group of a plane (+ 64: odd groups shift by half a bank row, the b64 writes of a 16-lane group then cover all 32 banks)
constexpr int g64_plane_bytes(int BM) { return 4 * g64_group_pitch(BM); }
constexpr int g64_lds_bytes(int BM, int NPL, int Cin) { return 2 * NPL * g64_plane_bytes(BM) + 64 + 2 * Cin * 4; }

template <int KS, int BM, int NPROD, bool AFFINE>
__global__ __launch_bounds__(BM * 4, 1)
void conv_g64_kernel(ConvArgs a) {
    constexpr int BN = 128, WARPS_N = 4, WARPS_M = BM / 64, NT = BM * 4;
    constexpr int MT = 2, NTL = 1;                                   // wave tile 64 rows x 32 channels
    constexpr int RS = 2;                                            // row slots per thread: rows r and r + BM / 2, eight lanes per row
    static_assert(BM == 64 || BM == 128, "64 or 128 rows");
    static_assert(NPROD == 1 || NPROD == 3, "one (bf16 operands) or three products");
    constexpr int NPL = NPROD == 1 ? 1 : 2;
    constexpr int GP = g64_group_pitch(BM), PLANE = g64_plane_bytes(BM), STAGE = NPL * PLANE;
    constexpr int OFF_TAB = 2 * STAGE + 64;                          // (alpha*s, beta*s) of the tile's image behind the two stages

    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int wm0 = (wave / WARPS_N) * 64, wn0 = (wave % WARPS_N) * 32;
    const int li = lane & 31, lh = lane >> 5;

    const int bid = xcd_item(blockIdx.x, a.tiles_m * a.tiles_n);
    const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
    const int n0 = tile_n * BN;
    const int hw = a.Ho * a.Wo;
    const int img = tile_m / a.tpi;
    const int r0 = (tile_m - img * a.tpi) * BM;                      // first position of the tile inside its image
    float in_scale = a.in_scale, in_unscale = a.in_unscale;
    if (NPROD != 1 && a.in_amax) h2_device_scale(a.in_amax + img, a.in_bound_add, in_scale, in_unscale);

    const int C2 = a.Cin - a.Csplit;
    const size_t planew = (size_t)((a.nchunks + 1) / 2 * 2) * a.Npad * 16;
    const bool xb16 = NPROD == 1 && a.x_bf16 && !a.x2;               // bf16 storage: the (single) input tensor holds bf16
    const tsnet_brsrc_t rsx = tsnet_make_brsrc(a.x, (unsigned)((size_t)a.N * a.H * a.W * a.Csplit * (xb16 ? 2 : 4)));
    const tsnet_brsrc_t rsx2 = tsnet_make_brsrc(a.x2 ? a.x2 : a.x, a.x2 ? (unsigned)((size_t)a.x2_nmod * a.H * a.W * C2 * 4) : 0u);
    tsnet_brsrc_t rsw[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) rsw[p] = tsnet_make_brsrc(a.w + p * planew, (unsigned)(planew * 2));

    float* tab = reinterpret_cast<float*>(smem_raw + OFF_TAB);
    if (AFFINE) {
        for (int c = tid; c < a.Cin; c += NT) {
            tab[c] = a.in_alpha[(size_t)img * a.Cin + c] * in_scale;
            tab[a.Cin + c] = a.in_beta[(size_t)img * a.Cin + c] * in_scale;
        }
    }

    // ---- A staging: thread t owns rows t/8 and t/8 + BM/2, channel quad q8 = t & 7 of each 32-channel half of the step's 64 channels
    const int q8 = tid & 7;
    const int srow0 = tid >> 3;
    bool s_valid[RS];
    int s_oy[RS], s_ox[RS];
    unsigned lds_w[RS];                                              // byte offset of the slot's first write inside a plane
#pragma unroll
    for (int s = 0; s < RS; ++s) {
        const int row = srow0 + s * (BM / 2);
        s_valid[s] = r0 + row < hw;
        const int rem = s_valid[s] ? r0 + row : 0;
        const int oy = rem / a.Wo;
        s_oy[s] = oy * a.stride - a.pad;
        s_ox[s] = (rem - oy * a.Wo) * a.stride - a.pad;
        // group (q8 >> 2) of the half, logical octet (q8 >> 1) & 1 swizzled by bit 3 of the row, half octet q8 & 1
        lds_w[s] = (unsigned)((q8 >> 2) * GP + row * 32 + ((((q8 >> 1) & 1) ^ ((row >> 3) & 1)) * 16) + (q8 & 1) * 8);
    }
    const int pix0 = img * a.H * a.W, pix2 = (img % a.x2_nmod) * a.H * a.W;
    const float relu_floor = a.in_relu ? 0.f : -__builtin_inff();    // branch-free ReLU switch
    F4 ar[RS][2];                                                    // register stage: [row slot][32-channel half]
    float am[RS];                                                    // 1, or 0 where the tap lies in the zero padding
    int nx_tap = 0, nx_c = 0, cur_c = 0;                             // (tap, first channel) of the next step to load; of the step held in `ar`
    const int nsteps = a.nchunks / 4;                                // Cin % 64 == 0: whole steps
    auto load_a = [&]() __attribute__((always_inline)) {             // called for steps 0, 1, 2, ... in order; past the end: harmless zeros
        const int tap = nx_tap, cb = nx_c;
        cur_c = cb;
        nx_c += 64;
        if (nx_c == a.Cin) { nx_c = 0; ++nx_tap; }
        const int ky = tap / KS, kx = tap - ky * KS;
        const bool second = cb >= a.Csplit;                          // torch.cat on the channel axis, formed on load (Csplit % 64 == 0: step-uniform)
#pragma unroll
        for (int s = 0; s < RS; ++s) {
            int iy = s_oy[s] + ky, ix = s_ox[s] + kx;
            bool ok = s_valid[s] && tap < a.taps;
            if (a.reflect) {
                iy = iy < 0 ? -iy : iy;
                iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
                ix = ix < 0 ? -ix : ix;
                ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
            } else {
                ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            }
            am[s] = ok ? 1.f : 0.f;
            if (second) {
                const unsigned v = ok ? (unsigned)(((pix2 + iy * a.W + ix) * C2 + (cb - a.Csplit) + q8 * 4) * 4) : kOOB;
                ar[s][0] = TSNET_BUF_LOAD16(rsx2, v, 0u); ar[s][1] = TSNET_BUF_LOAD16(rsx2, v, 128u);
            } else if (xb16) {                                       // bf16 storage: the quad is 8 bytes, the two 32-channel halves 64 bytes apart; widened exactly
                const unsigned v = ok ? (unsigned)(((pix0 + iy * a.W + ix) * a.Csplit + cb + q8 * 4) * 2) : kOOB;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const F2 p = TSNET_BUF_LOAD8(rsx, v, (unsigned)(h * 64));
                    const unsigned w0 = __builtin_bit_cast(unsigned, p.v[0]), w1 = __builtin_bit_cast(unsigned, p.v[1]);
                    ar[s][h].v[0] = __builtin_bit_cast(float, w0 << 16); ar[s][h].v[1] = __builtin_bit_cast(float, w0 & 0xFFFF0000u);
                    ar[s][h].v[2] = __builtin_bit_cast(float, w1 << 16); ar[s][h].v[3] = __builtin_bit_cast(float, w1 & 0xFFFF0000u);
                }
            } else {
                const unsigned v = ok ? (unsigned)(((pix0 + iy * a.W + ix) * a.Csplit + cb + q8 * 4) * 4) : kOOB;
                ar[s][0] = TSNET_BUF_LOAD16(rsx, v, 0u); ar[s][1] = TSNET_BUF_LOAD16(rsx, v, 128u);
            }
        }
    };
    // transform + split + store of the step held in `ar` (its first channel: cur_c, captured before the next load_a overwrites it)
    auto store_a = [&](int stage, int cb) __attribute__((always_inline)) {
        unsigned char* base = smem_raw + stage * STAGE;
#pragma unroll
        for (int s = 0; s < RS; ++s) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float t[4];
                if (AFFINE) {
                    const F4 al = *reinterpret_cast<const F4*>(tab + cb + h * 32 + q8 * 4), be = *reinterpret_cast<const F4*>(tab + a.Cin + cb + h * 32 + q8 * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaxf(__builtin_fmaf(ar[s][h].v[e], al.v[e], be.v[e]), relu_floor) * am[s];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) t[e] = __builtin_fmaxf(ar[s][h].v[e] * in_scale, relu_floor);
                }
                unsigned char* dst = base + lds_w[s] + h * 2 * GP;
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
    };

    // ---- fragments
    const int a_off = (wm0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16;
    const unsigned vB = (unsigned)((wn0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16);
    F4 af[NPL][MT], bf[4][NPL];
    auto frag_a = [&](int stage, int g) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int i = 0; i < MT; ++i) af[p][i] = *reinterpret_cast<const F4*>(smem_raw + stage * STAGE + p * PLANE + g * GP + i * 1024 + a_off);
    };
    auto load_b = [&](int g, int kc) __attribute__((always_inline)) {         // past the end of K the descriptor returns zeros
#pragma unroll
        for (int p = 0; p < NPL; ++p) bf[g][p] = TSNET_BUF_LOAD16(rsw[p], vB, (unsigned)((kc * a.Npad + n0) * 32));
    };

    f32x16 acc[MT], tot[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; tot[i][r] = 0.f; }
    auto product = [&](int g, int pa, int pb, bool fresh) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            f32x16 c = acc[i];
            if (fresh) {
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = 0.f;
            }
            if (NPROD == 1) acc[i] = TSNET_MFMA_BF16(af[pa][i], bf[g][pb], c);
            else acc[i] = TSNET_MFMA_F16(af[pa][i], bf[g][pb], c);
        }
    };

    // ---- prologue: A(0) into stage 0, A(1) in registers, B(0) in the four sets
    load_a();
    int cb_held = cur_c;
    if (AFFINE) __syncthreads();                                     // the transform table
    store_a(0, cb_held);
    load_a();
    cb_held = cur_c;
#pragma unroll
    for (int g = 0; g < 4; ++g) load_b(g, g);

    for (int st = 0; st < nsteps; ++st) {
        __syncthreads();                                             // stage st & 1 complete; stage (st + 1) & 1 no longer read
        const int stage = st & 1;
        store_a(stage ^ 1, cb_held);                                 // A(st + 1): loaded during step st - 1
        load_a();                                                    // A(st + 2)
        cb_held = cur_c;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            frag_a(stage, g);
            if (NPROD == 1) {
                product(g, 0, 0, g == 0);
            } else {
                product(g, 1, 0, g == 0);                            // lo * hi; chains of 4 k-groups counted from k = 0 (conv_h2r's association)
                product(g, 0, 1, false);                             // hi * lo
                product(g, 0, 0, false);                             // hi * hi
            }
            load_b(g, (st + 1) * 4 + g);                             // the same group of the next step into the set just consumed
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) tot[i] += acc[i];              // (a vector add: v_pk_add_f32, as in conv_h2r -- the element-wise form is 15 - 60 instructions longer per step here and spills in the bf16 form)
    }

    const float unscale = a.w_unscale ? in_unscale * a.w_unscale[0] : in_unscale;
    f32x16 out[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[i][0][r] = tot[i][r] * unscale;
    const int m_img = img * hw;
    __syncthreads();                                                 // the stages are free: the epilogue reduces through them
    conv_epilogue<BN, WARPS_M, WARPS_N, MT, NTL>(a, out, smem_raw, tid, wave, n0, (size_t)tile_m,
                                                 [&](int l) { return r0 + l < hw ? m_img + r0 + l : -1; });
}

}  // namespace tsnet
