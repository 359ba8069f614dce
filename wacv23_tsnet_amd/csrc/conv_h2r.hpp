// conv_h2r.hpp -- the GENERAL convolution of the forward: an implicit GEMM on the same arithmetic as the patch kernels (fp32 input, the
// producer's InstanceNorm + ReLU applied on load, fp16 x 2 operands / three products or one bf16 product, two-level fp32 accumulation,
// fp64 statistics in the epilogue).  It takes every layer the patch kernels of conv_h2.hpp do not:
//   * the 1 x 1 convolutions (fuse_net.conv on the mean over sources, TSNet.py:193; dec.map_conv on cat(pg, sg), :139 -- the concat is
//     formed on load from two tensors),
//   * the 7 x 7 stems with more than 8 input channels (the pose model's 31 / 28 + padding = 32) and the 8-channel stems of frames that do
//     not split into 4 x 32 rectangles,
//   * 3 x 3 layers on feature maps that are not multiples of 4 x 32 (small frames), stride 1 or 2, reflection or zero padding.
// Structure: a thread stages one (row, 8-channel octet) slot of the im2col A tile per 16-deep step -- two 16-byte fp32 loads two steps
// ahead, transformed and split in registers, two ds_writes -- weight fragments go straight into registers one step ahead (two register
// sets: the loop is unrolled by four, so parities are static), one barrier per step, chains of four k-groups folded into the running
// total.  K order is tap-major (k = tap * Cin + c).  An M tile is 128 output positions of ONE image (images whose size is not a multiple
// of 128 get a ragged last tile): the transform table and the data-derived operand scale are per image.
#pragma once
#include "conv_common.hpp"

namespace tsnet {

// SMALL_CIN: Cin = 8: the two octets of a 16-deep k-group are two different taps of the same channels
template <int KS, int BN, int WARPS_M, int WARPS_N, int NPROD, bool AFFINE, bool SMALL_CIN = false>
__global__ __launch_bounds__(256, 2)
void conv_h2r_kernel(ConvArgs a) {
    constexpr int BM = 128;
    constexpr int NW = WARPS_M * WARPS_N;
    static_assert(NW == 4, "256 threads: one 16-byte slot of the A tile per thread and plane");
    static_assert(NPROD == 1 || NPROD == 3, "one (bf16 operands) or three products");
    constexpr int NPL = NPROD == 1 ? 1 : 2;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 32, NTL = WN / 32;
    constexpr int PLANE_A = BM * 32, STAGE = 2 * PLANE_A;           // 8 KiB per stage (two planes)
    constexpr int OFF_TAB = 2 * STAGE + 64;                          // (alpha*s, beta*s) of the tile's image: 2 x Cin floats (after the epilogue's flag word)

    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int wm0 = (wave / WARPS_N) * WM, wn0 = (wave % WARPS_N) * WN;
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
        for (int c = tid; c < a.Cin; c += 256) {
            tab[c] = a.in_alpha[(size_t)img * a.Cin + c] * in_scale;
            tab[a.Cin + c] = a.in_beta[(size_t)img * a.Cin + c] * in_scale;
        }
        __syncthreads();
    }

    // ---- A staging: thread t owns row t/2, physical octet t&1 (LDS slot t*16 inside a plane), logical octet swizzled by bit 3 of the row
    const int srow = tid >> 1;
    const int oct_log = (tid & 1) ^ ((srow >> 3) & 1);
    const bool s_valid = r0 + srow < hw;                             // a ragged last tile: rows past the image read zeros
    int s_pix, s_pix2, s_oy, s_ox;
    {
        const int rem = s_valid ? r0 + srow : 0;
        const int oy = rem / a.Wo;
        s_pix = img * a.H * a.W;
        s_pix2 = (img % a.x2_nmod) * a.H * a.W;
        s_oy = oy * a.stride - a.pad;
        s_ox = (rem - oy * a.Wo) * a.stride - a.pad;
    }
    const float relu_floor = a.in_relu ? 0.f : -__builtin_inff();    // branch-free ReLU switch
    F4 ar[2][2];                                                     // register stage: [set][half of the octet]
    float am[2];                                                     // 1, or 0 where the tap lies in the zero padding / past the last tap
    int ac0[2];                                                      // first channel of the staged octet (for the transform table)
    int nx_tap = 0, nx_c = 0;                                        // (tap, first channel) of the next k-group: load_a runs for kc = 0, 1, 2, ... in order
    auto load_a = [&](int kc, int set) __attribute__((always_inline)) {
        int tap, c0;
        if (SMALL_CIN) { tap = kc * 2 + oct_log; c0 = 0; }
        else {                                                       // tap, slab: wave-uniform, advanced incrementally (Cin is any multiple of 16)
            tap = nx_tap; c0 = nx_c + oct_log * 8;
            nx_c += 16;
            if (nx_c == a.Cin) { nx_c = 0; ++nx_tap; }
        }
        const int ky = tap / KS, kx = tap - ky * KS;
        int iy = s_oy + ky, ix = s_ox + kx;
        bool ok = s_valid && tap < a.taps;
        if (a.reflect) {
            iy = iy < 0 ? -iy : iy;
            iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
            ix = ix < 0 ? -ix : ix;
            ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
        } else {
            ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        }
        const bool second = !SMALL_CIN && c0 >= a.Csplit;             // torch.cat on the channel axis, formed on load
        const unsigned v1 = ok ? (unsigned)(((s_pix + iy * a.W + ix) * a.Csplit + c0) * 4) : kOOB;
        const unsigned v2 = ok ? (unsigned)(((s_pix2 + iy * a.W + ix) * C2 + (c0 - a.Csplit)) * 4) : kOOB;
        const unsigned v = second ? v2 : v1;
        if (second) { ar[set][0] = TSNET_BUF_LOAD16(rsx2, v, 0u); ar[set][1] = TSNET_BUF_LOAD16(rsx2, v, 16u); }
        else if (NPROD == 1) load_x_octet(rsx, xb16, v, 0u, ar[set]);
        else { ar[set][0] = TSNET_BUF_LOAD16(rsx, v, 0u); ar[set][1] = TSNET_BUF_LOAD16(rsx, v, 16u); }
        am[set] = ok ? 1.f : 0.f;
        ac0[set] = ok ? c0 : 0;
    };
    auto store_a = [&](int set, int stage) __attribute__((always_inline)) {
        F4 t[2];
        transform_octet<AFFINE>(ar[set], tab + ac0[set], a.Cin, in_scale, relu_floor, am[set], t);
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
    const int m_img = img * hw;
    conv_epilogue<BN, WARPS_M, WARPS_N, MT, NTL>(a, tot, smem_raw, tid, wave, n0, (size_t)tile_m,
                                                 [&](int l) { return r0 + l < hw ? m_img + r0 + l : -1; });
}

constexpr int h2r_lds_bytes(int Cin) { return 2 * 2 * 128 * 32 + 64 + 2 * Cin * 4; }

}  // namespace tsnet
