// conv_x3r.hpp -- general bf16x3 implicit-GEMM convolution with both operands fetched by ordinary buffer loads.
//
// For the layers the patch kernels (conv_x3p.hpp) do not take -- 7x7 stems, stride-2 downsampling, 1x1 -- the
// implicit GEMM of conv_x3.hpp is bound by the LDS-DMA path (profiles/round1_notes.md: 7.5 TB/s of 32-byte gathers
// against a 6.4-6.8 TB/s fill ceiling; MFMA pipe 0.2-0.35 busy).  This kernel keeps conv_x3's arithmetic -- same K
// order, same product order, same fold points: results are bit-identical, the launcher may pick either -- and moves
// the data the way x3q_tile does:
//   * the im2col A tile (BM rows x 16 k x 3 planes) goes global -> VGPR -> ds_write into a two-stage LDS image; a
//     thread stages one 16-byte slot per plane per step, loaded two steps ahead (register double buffer);
//   * weight fragments go straight into the MFMA operand registers, each plane re-loaded for the next step right
//     after its last use in this one;
//   * every load is a compiler-visible buffer load (descriptor range check = zero padding), so hipcc counts vmcnt
//     itself; one __syncthreads per 16-deep step; static fold points (the loop is unrolled by four).
#pragma once
#include "conv_x3p.hpp"

namespace tsnet {

// NP = 1: bf16-operand mode -- only the hi plane is staged and multiplied (see conv_x3p.hpp)
template <int KS, int BN, int WARPS_M, int WARPS_N, bool SMALL_CIN, int NP = 3>
__global__ __launch_bounds__(256)
void conv_x3r_kernel(X3Args a) {
    constexpr int BM = 128;
    constexpr int NW = WARPS_M * WARPS_N;
    static_assert(NW == 4, "256 threads: one 16-byte slot of the A tile per thread and plane");
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 32, NTL = WN / 32;
    constexpr int PLANE_A = BM * 32, STAGE = 3 * PLANE_A;           // 12 KiB per stage

    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int wm0 = (wave / WARPS_N) * WM, wn0 = (wave % WARPS_N) * WN;
    const int li = lane & 31, lh = lane >> 5;

    const int bid = x3p_item(blockIdx.x, a.tiles_m * a.tiles_n);
    const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int C2 = a.Cin - a.Csplit;
    const size_t plane1 = (size_t)a.N * a.H * a.W * a.Csplit, plane2 = (size_t)a.x2_nmod * a.H * a.W * C2;
    const size_t planew = (size_t)((a.nchunks + 1) / 2 * 2) * a.Npad * 16;
    tsnet_brsrc_t rs1[3], rs2[3], rsw[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        rs1[p] = tsnet_make_brsrc(a.x + p * plane1, (unsigned)(plane1 * 2));
        rs2[p] = tsnet_make_brsrc(a.x2 ? a.x2 + p * plane2 : a.x, a.x2 ? (unsigned)(plane2 * 2) : 0u);
        rsw[p] = tsnet_make_brsrc(a.w + p * planew, (unsigned)(planew * 2));
    }

    // ---- A staging: thread t owns row t/2, physical octet t&1 of the tile (LDS slot = t*16 inside a plane);
    //      the logical octet is swizzled by bit 3 of the row so that a ds_read_b128 phase spreads over all banks
    const int srow = tid >> 1;
    const int oct_log = (tid & 1) ^ ((srow >> 3) & 1);
    const int sm = m0 + srow;
    const bool s_ok = sm < a.M;
    int s_pix1, s_pix2, s_oy, s_ox;
    {
        const int mm = s_ok ? sm : 0;
        const int hw = a.Ho * a.Wo;
        const int img = mm / hw;
        const int rem = mm - img * hw;
        const int oy = rem / a.Wo;
        s_pix1 = img * a.H * a.W;
        s_pix2 = (img % a.x2_nmod) * a.H * a.W;
        s_oy = oy * a.stride - a.pad;
        s_ox = (rem - oy * a.Wo) * a.stride - a.pad;
    }
    const int cpt_log2 = SMALL_CIN ? 0 : a.cin_log2 - 4;
    F4 ar[2][3];                                             // register stage of the A tile: [set][plane]
    // fetch this thread's slot of k-group kc (three planes) into register set `set`
    auto load_a = [&](int kc, int set) __attribute__((always_inline)) {
        int tap, c_lane;
        unsigned so;
        bool second = false;
        if (SMALL_CIN) {                                     // Cin = 8: the two octets of a row are two different taps
            const int k = kc * 16 + oct_log * 8;
            tap = k >> a.cin_log2; c_lane = k & (a.Cin - 1); so = 0;
        } else {
            tap = kc >> cpt_log2;                            // wave-uniform
            const int c0 = (kc << 4) & (a.Cin - 1);
            second = c0 >= a.Csplit;
            so = (unsigned)((second ? c0 - a.Csplit : c0) * 2);
            c_lane = oct_log * 8;
        }
        const int ky = tap / KS, kx = tap - ky * KS;
        int iy = s_oy + ky, ix = s_ox + kx;
        bool ok = s_ok && tap < a.taps;
        if (a.reflect) {
            iy = iy < 0 ? -iy : iy;
            iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
            ix = ix < 0 ? -ix : ix;
            ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
        } else {
            ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        }
        const int pix = iy * a.W + ix;
        const unsigned v1 = ok ? (unsigned)(((s_pix1 + pix) * a.Csplit + c_lane) * 2) : kOOB;
        const unsigned v2 = ok ? (unsigned)(((s_pix2 + pix) * C2 + c_lane) * 2) : kOOB;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const tsnet_brsrc_t rs = second ? rs2[p] : rs1[p];
            ar[set][p] = TSNET_BUF_LOAD16(rs, second ? v2 : v1, so);
        }
    };
    auto store_a = [&](int set, int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NP; ++p) *reinterpret_cast<F4*>(smem_raw + stage * STAGE + p * PLANE_A + tid * 16) = ar[set][p];
    };

    // ---- fragments
    const int a_off = (wm0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16;
    const unsigned vB = (unsigned)((wn0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16);
    F4 af[3][MT], bf[3][NTL];
    auto frag_a = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int i = 0; i < MT; ++i) af[p][i] = *reinterpret_cast<const F4*>(smem_raw + stage * STAGE + p * PLANE_A + i * 1024 + a_off);
    };
    auto load_b = [&](int p, int kc) __attribute__((always_inline)) {       // past the end of K the descriptor returns zeros
#pragma unroll
        for (int j = 0; j < NTL; ++j) bf[p][j] = TSNET_BUF_LOAD16(rsw[p], vB, (unsigned)((kc * a.Npad + n0 + j * 32) * 32));
    };

    f32x16 acc[MT][NTL], tot[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
    auto product = [&](int pa, int pb, bool fresh) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                f32x16 c = acc[i][j];
                if (fresh) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = 0.f;
                }
                acc[i][j] = TSNET_MFMA_BF16(af[pa][i], bf[pb][j], c);
            }
    };

    // step kc (u = kc mod 4, static): A(kc) is in LDS stage u&1, A(kc+1) in register set (u+1)&1, B(kc) in bf
    auto step = [&](int kc, int u) __attribute__((always_inline)) {
        __syncthreads();                                     // stage u&1 complete; stage (u+1)&1 no longer read
        frag_a(u & 1);
        store_a((u + 1) & 1, (u + 1) & 1);                   // A(kc+1): loaded during step kc-1
        load_a(kc + 2, u & 1);                               // register set u&1 held A(kc), already in LDS
        if (NP == 3) {
            product(2, 0, u == 0);                           // chains of 4 k-groups counted from k = 0 (conv_x3.hpp)
            product(1, 0, false);
        }
        product(0, 0, NP == 1 && u == 0);
        load_b(0, kc + 1);
        if (NP == 3) {
            product(1, 1, false);
            product(0, 1, false);
            load_b(1, kc + 1);
            product(0, 2, false);
            load_b(2, kc + 1);
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
#pragma unroll
    for (int p = 0; p < NP; ++p) load_b(p, 0);
    const int nst = a.nchunks;
    int kc = 0;
    for (; kc + 4 <= nst; kc += 4) { step(kc, 0); step(kc + 1, 1); step(kc + 2, 2); step(kc + 3, 3); }
    if (kc < nst) {                                          // 1..3 trailing k-groups: a last, partial chain
        step(kc, 0);
        if (kc + 1 < nst) step(kc + 1, 1);
        if (kc + 2 < nst) step(kc + 2, 2);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) tot[i][j] += acc[i][j];
    }

    const int hw = a.Ho * a.Wo;
    const int img0 = m0 / hw;
    x3_epilogue<BN, WARPS_M, WARPS_N, MT, NTL>(a, tot, smem_raw, tid, wave, n0, (size_t)img0 * (hw / BM) + (m0 - img0 * hw) / BM,
                                               [&](int l) { const int m = m0 + l; return m < a.M ? m : -1; });
}

}  // namespace tsnet
