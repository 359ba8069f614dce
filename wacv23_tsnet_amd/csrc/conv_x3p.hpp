// conv_x3p.hpp -- bf16x3 3x3 stride-1 convolution with an input PATCH resident in LDS (no im2col re-fetch).
//
// Why (tools/x3_ablate.py, profiles/round1_notes.md): conv_x3.hpp is bound by its A-operand LDS-DMAs.  An
// implicit GEMM fetches every input pixel once per tap -- 9 x 128 pixel rows of 32 B per 16-channel slab, each a
// separate 128-byte line for the texture path -- and with the MFMA work 2.67x shorter than in fp32 those gathers
// set the pace (159 TF; 219 TF with the A DMAs removed, B DMAs alone are free).  Here the workgroup's output tile
// is a 4 x 32 pixel rectangle of one image; for a 16-channel slab it stages the (4+2) x (32+2) input patch, halo
// included (reflection or zero padding resolved per lane at DMA time), ONCE, and the nine taps are nine shifted
// views of that patch: the tap offset is an immediate in the ds_read address.  204 pixel fetches replace 1152.
//
// K order is therefore slab-major: step s = cc*9 + tap (cc = 16-channel slab), weights chunk kc = tap*(Cin/16)+cc
// of the unchanged conv_x3 packing.  Two-level accumulation folds every 4 steps of THIS order; a layer always runs
// on the same kernel, so results stay independent of batch size and tile count.
//
// LDS: two patch stages (3 planes x 7 KiB: 224 pixel slots x 32 B, swizzled octets) + a ring of three B stages
// (3 planes x BN x 32 B) + 1 KiB scratch for surplus zero-fill DMAs = 79 KiB at BN = 128: two workgroups per CU.
// vmcnt bookkeeping: every wave issues 3 B DMAs per step plus one patch DMA in taps 0..5 (6 per slab: 2 blocks x
// 3 planes); the unrolled tap index makes every wait count a compile-time constant.
#pragma once
#include "conv_x3.hpp"

namespace tsnet {

constexpr int kPatchRows = 4, kPatchCols = 32;            // output rectangle of one tile (BM = 128 positions)

// one output tile: rows tile_m*128 .. +127 (a 4 x 32 rectangle), columns n0 .. n0+BN-1
// ABL (tools/x3_ablate.py only; non-zero computes garbage): bit0 no DMA in the loop, bit1 no vmcnt/barrier, bit2 no
// fold, bit3 fragments read from LDS once instead of every step
template <int BN, int WARPS_M, int WARPS_N, bool FOLD, int ABL = 0>
__device__ __forceinline__ void x3p_tile(const X3Args& a, unsigned char* smem_raw, const int tile_m, const int n0) {
    constexpr int BM = kPatchRows * kPatchCols;
    constexpr int NW = WARPS_M * WARPS_N;
    static_assert(NW == 4, "four waves: patch blocks are dealt w, w+4");
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 32, NTL = WN / 32;
    constexpr int PC = kPatchCols + 2, PP = (kPatchRows + 2) * PC;   // 34, 204 patch pixels
    constexpr int PBLK = (PP + 31) / 32;                             // 7 DMA blocks of 32 pixels per plane
    static_assert(PBLK <= 8, "two blocks per wave");
    constexpr int PLANE_P = PBLK * 1024, PATCH_BYTES = 3 * PLANE_P;
    constexpr int TB = BN / 32;                                      // B blocks per plane (<= 4: one per wave)
    static_assert(TB <= NW, "one B block per wave");
    constexpr int PLANE_B = BN * 32, BSTAGE = 3 * PLANE_B;
    constexpr int OFF_B = 2 * PATCH_BYTES, OFF_SCRATCH = OFF_B + 3 * BSTAGE;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int wrow = wave / WARPS_N;
    const int wn0 = (wave % WARPS_N) * WN;
    const int li = lane & 31, lh = lane >> 5;
    // tile_m -> (image, row block, column block); stride 1 / pad 1: input and output grids coincide
    const int tcols = a.Wo / kPatchCols, tper = (a.Ho / kPatchRows) * tcols;
    const int img = tile_m / tper, tin = tile_m - img * tper;
    const int oy0 = (tin / tcols) * kPatchRows, ox0 = (tin % tcols) * kPatchCols;
    const int ncc = a.Cin >> 4;                                      // 16-channel slabs

    const int C2 = a.Cin - a.Csplit;
    const size_t plane1 = (size_t)a.N * a.H * a.W * a.Csplit, plane2 = (size_t)a.x2_nmod * a.H * a.W * C2;
    const size_t planew = (size_t)((a.nchunks + 1) / 2 * 2) * a.Npad * 16;
    tsnet_rsrc_t rs1[3], rs2[3], rsw[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        rs1[p] = tsnet_make_rsrc(a.x + p * plane1, (unsigned)(plane1 * 2));
        rs2[p] = tsnet_make_rsrc(a.x2 ? a.x2 + p * plane2 : a.x, a.x2 ? (unsigned)(plane2 * 2) : 0u);
        rsw[p] = tsnet_make_rsrc(a.w + p * planew, (unsigned)(planew * 2));
    }
    const tsnet_lds_t lds0 = TSNET_LDS_BASE(smem_raw);

    // ---- patch DMA geometry: this wave fills pixel blocks b = wave and wave + 4; lane -> (pixel b*32 + lane/2,
    //      physical octet lane&1), logical octet swizzled by bit 3 of the pixel slot so that the 16 lanes of a
    //      ds_read_b128 phase (consecutive slots) spread over all banks
    unsigned vP1[2], vP2[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int pp = (wave + 4 * r) * 32 + (lane >> 1);
        const int oct_log = (lane & 1) ^ ((pp >> 3) & 1);
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
        vP1[r] = ok ? (unsigned)(((img * a.H * a.W + pix) * a.Csplit + oct_log * 8) * 2) : kOOB;
        vP2[r] = ok ? (unsigned)((((img % a.x2_nmod) * a.H * a.W + pix) * C2 + oct_log * 8) * 2) : kOOB;
    }
    const unsigned vB = (unsigned)(lane * 16);
    const tsnet_lds_t scratch = lds0 + OFF_SCRATCH;

    // one patch DMA: slot q = 0..5 of slab cn -> plane q%3, block wave + 4*(q/3)
    auto issue_patch = [&](int cn, int q) __attribute__((always_inline)) {
        const int p = q % 3, r = q / 3;
        const int c0 = cn << 4;
        const bool second = c0 >= a.Csplit;                          // wave-uniform
        const unsigned so = (unsigned)((second ? c0 - a.Csplit : c0) * 2);
        const int b = wave + 4 * r;
        const tsnet_lds_t dst = b < PBLK ? lds0 + (cn & 1) * PATCH_BYTES + p * PLANE_P + b * 1024 : scratch;
        if (second) TSNET_BUF_DMA16(rs2[p], vP2[r], so, dst);
        else TSNET_BUF_DMA16(rs1[p], vP1[r], so, dst);
    };
    // the three B DMAs of step (cc, t) into ring stage `stage`; past the end of K the descriptor returns zeros
    auto issue_b = [&](int cc, int t, int stage) __attribute__((always_inline)) {
        const int kc = t * ncc + cc;
        const unsigned so = (unsigned)((kc * a.Npad + n0 + wave * 32) * 32);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            if (TB == NW || wave < TB) {
                TSNET_BUF_DMA16(rsw[p], vB, so, lds0 + OFF_B + stage * BSTAGE + p * PLANE_B + wave * 1024);
            } else {
                const unsigned oob = kOOB;
                TSNET_BUF_DMA16(rsw[p], oob, 0u, scratch);
            }
        }
    };

    f32x16 acc[MT][NTL], tot[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

    const int p0 = wrow * MT * PC + li;                      // patch slot of (first row of this wave, column li), tap (0,0)
    const int b_off = (wn0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16;

    // ops issued per step: 3 B + (t < 6).  Step s needs B(s), issued first thing in step s-2:
    // outstanding allowance = patch op of step s-2 + everything of step s-1.
    auto step = [&](int cc, int t) __attribute__((always_inline)) {
        const int n_after = 3 + (((t + 7) % 9) < 6 ? 1 : 0) + (((t + 8) % 9) < 6 ? 1 : 0);
        if (!(ABL & 2)) {
            if (n_after == 3) TSNET_VMCNT(3); else if (n_after == 4) TSNET_VMCNT(4); else TSNET_VMCNT(5);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        const int ky = t / 3, kx = t - ky * 3;
        const unsigned char* pbase = smem_raw + ((ABL & 8) ? 0 : (cc & 1)) * PATCH_BYTES;
        const unsigned char* bbase = smem_raw + OFF_B + ((ABL & 8) ? 0 : (t % 3)) * BSTAGE;
        F4 af[3][MT], bf[3][NTL];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int slot = (ABL & 8) ? p0 + i * PC : p0 + (i + ky) * PC + kx;
            const int off = ((slot << 1) | (lh ^ ((slot >> 3) & 1))) << 4;
#pragma unroll
            for (int p = 0; p < 3; ++p) af[p][i] = *reinterpret_cast<const F4*>(pbase + p * PLANE_P + off);
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < NTL; ++j) bf[p][j] = *reinterpret_cast<const F4*>(bbase + p * PLANE_B + j * 1024 + b_off);
        if (!(ABL & 1)) {
            const int t2 = (t + 2) % 9;
            issue_b(cc + (t + 2 >= 9 ? 1 : 0), t2, t2 % 3);
            if (t < 6) issue_patch(cc + 1, t);
        }
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
                    acc[i][j] = TSNET_MFMA_BF16(af[PA[q]][i], bf[PB[q]][j], acc[i][j]);
        if (FOLD && !(ABL & 4) && ((cc + t + 1) & 3) == 0) {                       // step index 9*cc + t == cc + t (mod 4); wave-uniform
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
                    tot[i][j] += acc[i][j];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
        }
    };

    // prologue in steady-state order: patch(0), B(0,0), B(0,1)
#pragma unroll
    for (int q = 0; q < 6; ++q) issue_patch(0, q);
    issue_b(0, 0, 0);
    issue_b(0, 1, 1);
    for (int cc = 0; cc < ncc; ++cc) {
        step(cc, 0); step(cc, 1); step(cc, 2);
        step(cc, 3); step(cc, 4); step(cc, 5);
        step(cc, 6); step(cc, 7); step(cc, 8);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) tot[i][j] += acc[i][j];    // the last, partial chain
    TSNET_VMCNT(0);

    const int m_img = img * a.Ho * a.Wo;
    x3_epilogue<BN, WARPS_M, WARPS_N, MT, NTL>(a, tot, smem_raw, tid, wave, n0, (size_t)img * tper + tin,
                                               [&](int l) { return m_img + (oy0 + (l >> 5)) * a.Wo + ox0 + (l & 31); });
}

// ---------------------------------------------------------------------------------------------------------------
// x3q: the same patch convolution with the WEIGHT fragments fetched straight into registers.
//
// Why (tools/x3_ablate.py patch, profiles/round1_notes.md): x3p_tile moves 14.6 KB of LDS-DMA per 0.52 MFLOP; at its
// 232 TF that is 6.5 TB/s, the chip's measured LDS-DMA fill ceiling (6.4-6.8 TB/s), and 82 % of it is the weight
// tile.  A weight fragment is already in the layout the MFMA wants (pack_weights_x3_kernel), it is read by exactly
// the waves that multiply with it, and a wave's six 1 KiB fragments per step are contiguous in memory: a plain
// buffer_load_dwordx4 through the vector L1 (64 B/clk/CU, 4x the DMA rate) delivers them with no LDS round trip.
// That removes the B ring (36 KiB), its 12 ds_reads per step (half of the LDS read traffic), the per-step barrier
// (the patch is the only shared data: one barrier per 16-channel slab) and every inline-asm load: the patch is
// staged through registers (buffer load -> ds_write) so that hipcc counts every vmcnt itself.
//
// Register reuse instead of double buffering: the six products of a step are ordered so that each operand plane
// retires early -- (lo,hi) (mid,hi) (hi,hi) (mid,mid) (hi,mid) (hi,lo): weights-hi is free after 3 products,
// weights-mid after 5 -- and the fragment of the NEXT step is loaded into the same registers right after the
// last use; every load has >= 12 MFMAs (>= 384 matrix-pipe cycles, twice that with two waves per SIMD) to land.
// Fold points are static (after taps 3 and 8 of every slab: chains of 64 and 80 products); the first product of
// a chain takes C = 0, so a fold costs 32 v_pk_add and no zeroing.
#ifndef TSNET_BUF_LOAD16
typedef __amdgpu_buffer_rsrc_t tsnet_brsrc_t;
__device__ __forceinline__ tsnet_brsrc_t tsnet_make_brsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
#define TSNET_BUF_LOAD16(rsrc, voff, soff) __builtin_bit_cast(F4, __builtin_amdgcn_raw_buffer_load_b128((rsrc), (int)(voff), (int)(soff), 0))
#endif

// QABL (tools/x3_ablate.py q only; non-zero computes garbage): bit0 no patch staging, bit1 no slab barrier, bit2 no fold,
// bit3 weight fragments loaded once, bit4 A fragments read once, bit5 plane 1 unused (two planes, three products)
// NP = 1: bf16-operand mode (BASELINE.json configs[2] / [4]): only the hi plane -- x rounded to bf16 -- is read and multiplied (one
// product per k-group); accumulation, epilogue and statistics are unchanged (fp32 / fp64).
template <int BN, int WARPS_M, int WARPS_N, int QABL = 0, int NP = 3>
__device__ __forceinline__ void x3q_tile(const X3Args& a, unsigned char* smem_raw, const int tile_m, const int n0) {
    constexpr int BM = kPatchRows * kPatchCols;
    constexpr int NW = WARPS_M * WARPS_N;
    static_assert(NW == 4, "four waves: patch blocks are dealt w, w+4");
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 32, NTL = WN / 32;
    constexpr int PC = kPatchCols + 2, PP = (kPatchRows + 2) * PC;
    constexpr int PBLK = (PP + 31) / 32;
    constexpr int PLANE_P = PBLK * 1024, PATCH_BYTES = 3 * PLANE_P;
    constexpr int OFF_SCRATCH = 2 * PATCH_BYTES;                     // 1 KiB sink for the wave whose second block does not exist

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

    // patch staging geometry (same LDS image as x3p_tile): this wave owns pixel blocks wave and wave + 4
    unsigned vP1[2], vP2[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int pp = (wave + 4 * r) * 32 + (lane >> 1);
        const int oct_log = (lane & 1) ^ ((pp >> 3) & 1);
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
        vP1[r] = ok ? (unsigned)(((img * a.H * a.W + pix) * a.Csplit + oct_log * 8) * 2) : kOOB;
        vP2[r] = ok ? (unsigned)((((img % a.x2_nmod) * a.H * a.W + pix) * C2 + oct_log * 8) * 2) : kOOB;
    }
    // entry q = 0..5 of slab cn: plane q%3 of block wave + 4*(q/3); branch-free source select
    auto patch_load = [&](int cn, int q) __attribute__((always_inline)) {
        const int p = q % 3, r = q / 3;
        const int c0 = cn << 4;
        const bool second = c0 >= a.Csplit;                          // wave-uniform
        const unsigned so = (unsigned)((second ? c0 - a.Csplit : c0) * 2);
        const tsnet_brsrc_t rs = second ? rs2[p] : rs1[p];
        return TSNET_BUF_LOAD16(rs, second ? vP2[r] : vP1[r], so);
    };
    auto patch_store = [&](int cn, int q, const F4& v) __attribute__((always_inline)) {
        const int p = q % 3, r = q / 3;
        const int b = wave + 4 * r;
        unsigned char* dst = smem_raw + (b < PBLK ? (cn & 1) * PATCH_BYTES + p * PLANE_P + b * 1024 : OFF_SCRATCH) + lane * 16;
        *reinterpret_cast<F4*>(dst) = v;
    };

    // weight fragments: lane (li, lh) takes the 16 bytes of column wn0 + j*32 + li, logical octet lh
    const unsigned vB = (unsigned)((wn0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16);
    F4 af[3][MT], bf[3][NTL];
    auto load_b = [&](int p, int cc, int t) __attribute__((always_inline)) {
        const int kc = t * ncc + cc;
#pragma unroll
        for (int j = 0; j < NTL; ++j) bf[p][j] = TSNET_BUF_LOAD16(rsw[p], vB, (unsigned)((kc * a.Npad + n0 + j * 32) * 32));
    };
    const int p0 = wrow * MT * PC + li;
    auto load_a = [&](int p, int cc, int t) __attribute__((always_inline)) {
        const int ky = t / 3, kx = t - ky * 3;
        const unsigned char* pbase = smem_raw + (cc & 1) * PATCH_BYTES + p * PLANE_P;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int slot = p0 + (i + ky) * PC + kx;
            af[p][i] = *reinterpret_cast<const F4*>(pbase + (((slot << 1) | (lh ^ ((slot >> 3) & 1))) << 4));
        }
    };

    f32x16 acc[MT][NTL], tot[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

    // one product of planes (pa, pb) over the wave tile; `fresh` starts a new chain from C = 0
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
    auto step = [&](int cc, int t) __attribute__((always_inline)) {
        const bool fresh = t == 0 || t == 4;                 // chains: taps 0..3 and 4..8 of the slab
        const bool last = t == 8;
        const int nc = last ? cc + 1 : cc, nt = last ? 0 : t + 1;
        F4 stage;
        constexpr bool TWO = (QABL & 32) != 0;               // timing proxy of a two-plane / three-product scheme: plane 1 unused
        constexpr bool ONE = NP == 1;
        const bool stage_on = t < 6 && !(QABL & 1) && !(TWO && (t % 3) == 1) && !(ONE && (t % 3) != 0);
        if (stage_on) stage = patch_load(cc + 1, t);
        if (!ONE) product(2, 0, fresh && !(QABL & 4));       // lo  * hi
        if (!last && !(QABL & 16) && !ONE) load_a(2, nc, nt);
        if (!TWO && !ONE) product(1, 0, false);              // mid * hi
        product(0, 0, ONE && fresh);                         // hi  * hi
        if (!(QABL & 8)) load_b(0, nc, nt);
        if (!TWO && !ONE) product(1, 1, false);              // mid * mid
        if (!last && !(QABL & 16) && !TWO && !ONE) load_a(1, nc, nt);
        if (!TWO && !ONE) product(0, 1, false);              // hi  * mid
        if (!(QABL & 8) && !TWO && !ONE) load_b(1, nc, nt);
        if (!ONE) product(0, 2, false);                      // hi  * lo
        if (!last && !(QABL & 16)) load_a(0, nc, nt);
        if (!(QABL & 8) && !ONE) load_b(2, nc, nt);
        if (stage_on) patch_store(cc + 1, t, stage);
        if ((t == 3 || t == 8) && !(QABL & 4)) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) tot[i][j] += acc[i][j];
        }
    };

    // prologue: patch of slab 0, then the fragments of step (0, 0)
#pragma unroll
    for (int q = 0; q < 6; ++q) if (NP == 3 || (q % 3) == 0) patch_store(0, q, patch_load(0, q));
#pragma unroll
    for (int p = 0; p < NP; ++p) load_b(p, 0, 0);
    for (int cc = 0; cc < ncc; ++cc) {
        if (!(QABL & 2)) __syncthreads();                    // patch(cc) complete and visible; slab cc-1 fully read
        if (!(QABL & 16) || cc == 0) {
#pragma unroll
            for (int p = 0; p < NP; ++p) load_a(p, cc, 0);
        }
        step(cc, 0); step(cc, 1); step(cc, 2);
        step(cc, 3); step(cc, 4); step(cc, 5);
        step(cc, 6); step(cc, 7); step(cc, 8);
    }

    const int m_img = img * a.Ho * a.Wo;
    x3_epilogue<BN, WARPS_M, WARPS_N, MT, NTL>(a, tot, smem_raw, tid, wave, n0, (size_t)img * tper + tin,
                                               [&](int l) { return m_img + (oy0 + (l >> 5)) * a.Wo + ox0 + (l & 31); });
}

// XCD-aware block -> work item: consecutive items stay on one XCD (its L2 then holds their shared operands)
__device__ __forceinline__ int x3p_item(int bid, int nitems) {
    const int q = nitems >> 3, r = nitems & 7, xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

template <int BN, int WARPS_M, int WARPS_N, int QABL = 0, int NP = 3>
__global__ __launch_bounds__(256)
void conv_x3q_kernel(X3Args a) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int bid = x3p_item(blockIdx.x, a.tiles_m * a.tiles_n);
    const int tile_m = bid / a.tiles_n;
    x3q_tile<BN, WARPS_M, WARPS_N, QABL, NP>(a, smem_raw, tile_m, (bid - tile_m * a.tiles_n) * BN);
}

template <int BN, int WARPS_M, int WARPS_N, bool FOLD = true, int ABL = 0>
__global__ __launch_bounds__(256)
void conv_x3p_kernel(X3Args a) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int bid = x3p_item(blockIdx.x, a.tiles_m * a.tiles_n);
    const int tile_m = bid / a.tiles_n;
    x3p_tile<BN, WARPS_M, WARPS_N, FOLD, ABL>(a, smem_raw, tile_m, (bid - tile_m * a.tiles_n) * BN);
}

// Mixed launch for layers whose 128 x 128 unit count is a half-integer multiple of the CU count (the 384-tile
// residual convolutions at batch 4: half the CUs would run two tiles, half one -- 75 % of the machine).  The last
// `a.tiles_n`-major `nsplit` units are cut into two 128 x 64 halves; blocks [0, nbig) run whole units, the rest
// halves, so with two resident blocks per CU every CU gets one whole and one half unit.  Per output element the
// K order and fold points are those of the plain kernel: results are bit-identical.
__global__ __launch_bounds__(256)
void conv_x3p_mixed_kernel(X3Args a, int nbig) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    if ((int)blockIdx.x < nbig) {
        const int u = x3p_item(blockIdx.x, nbig);
        const int tile_m = u / a.tiles_n;
        x3p_tile<128, 2, 2, true>(a, smem_raw, tile_m, (u - tile_m * a.tiles_n) * 128);
    } else {
        const int nsmall = (int)gridDim.x - nbig;
        const int h = x3p_item((int)blockIdx.x - nbig, nsmall);
        const int u = nbig + (h >> 1);
        const int tile_m = u / a.tiles_n;
        x3p_tile<64, 2, 2, true>(a, smem_raw, tile_m, (u - tile_m * a.tiles_n) * 128 + (h & 1) * 64);
    }
}

}  // namespace tsnet
