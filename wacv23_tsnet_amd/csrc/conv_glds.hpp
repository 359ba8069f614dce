// conv_glds.hpp -- NHWC implicit-GEMM convolution, LDS-DMA pipelined (gfx950 / CDNA4).
//
// Same GEMM view and numerics as conv_igemm.hpp (exact-fp32 v_mfma_f32_32x32x2_f32, two-level
// accumulation), but both operands travel HBM/L2 -> LDS by `global_load_lds_dwordx4` (LDS-DMA):
// no staging VGPRs, no ds_write pass, and a 4-stage LDS ring with COUNTED s_waitcnt vmcnt so three
// K-chunks of loads stay in flight across the one barrier per chunk (cdna_hip_programming.md
// section 5 "Pipelining across barriers", T3+T4).  Measured motivation (profiles/round1_notes.md):
// the register-staged kernel loses 23 % to exposed global-load latency and 10 % to ds_write+barrier.
//
// Consequences of LDS-DMA for the design:
//   * the DMA cannot transform data, so the input must already be the activation the reference's
//     conv sees: InstanceNorm+ReLU are materialised by norm_act_kernel (one HBM-bound pass, ~1 % of
//     the forward).  Reflection / zero padding and the channel concat stay in the address
//     computation; zero padding, ragged rows and padded K taps read a zero page so every lane
//     always issues (the vmcnt accounting needs a fixed number of loads per wave per chunk).
//   * the LDS image is lane-linear per wave instruction (dest = wave-uniform base + lane*16), so
//     the bank-conflict swizzle is applied on the SOURCE address and again on the read (rule 21):
//     image [row][4 k-quads], physical quad = logical quad ^ ((row>>2)&3); a 16-lane ds_read_b128
//     group (16 distinct rows, same logical quad) then hits 16 distinct 16-byte slots.
//   * weights are pre-packed in exactly that image order, so a B tile is one contiguous,
//     fully coalesced 1 KiB read per wave instruction.
#pragma once
#include <hip/hip_runtime.h>

#include "conv_igemm.hpp"

namespace tsnet {

struct GldsArgs {
    const float* x;         // source 0, NHWC (N,H,W,Csplit)
    const float* x2;        // source 1 (channels >= Csplit) or null
    const float* zero_page; // >= 64 zero bytes, 16-byte aligned
    const float* w;         // packed [K/16][Npad][4 quads (swizzled)][4]
    const float* bias;
    float* y;
    double* stat_part;      // null, or (N, tiles_per_img, Cout, 2) per-tile sum / sum-of-squares of y (conv_dma only)
    const float* addend;    // null, or NHWC (add_nmod, Ho, Wo, Cout) added to y before statistics (conv_dma only)
    int add_nmod;           // image index into addend = img % add_nmod
    int N, H, W, Cin, cin_log2, Csplit, x2_nmod;
    int Ho, Wo, Cout, Npad;
    int stride, pad, reflect, taps, nchunks, M;
    int act, out_nchw, composite, fore_x0, fore_x1;
    float bg[3];
    int tiles_m, tiles_n;
};

// One LDS-DMA instruction: every lane fetches 16 bytes from its own global address `g`; the wave's
// 64 x 16 B land at LDS byte address `lds_wave_base` (wave-uniform, goes through M0) + lane*16.
// It is issued through inline asm on purpose: hipcc cannot prove that the DMA's destination stage
// differs from the stage the next ds_read touches and would put `s_waitcnt vmcnt(0)` in front of
// every ds_read (measured: the pipeline then drains every chunk).  As an asm statement the DMA is
// invisible to the compiler's waitcnt bookkeeping; completion is tracked by TSNET_VMCNT below, as
// cdna_hip_programming.md section 5.7 prescribes (M0 saved/restored inside the same statement).
// TSNET_GLDS16 is the single hook tests/emu predefines to run this kernel on the CPU.
#ifndef TSNET_GLDS16
#define TSNET_GLDS16(g, lds_u32)                                                                         \
    do {                                                                                                 \
        unsigned keep__;                                                                                 \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep__) : "v"(g), "s"(lds_u32) : "memory");                                 \
    } while (0)
#define TSNET_LDS_ADDR(p) __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) unsigned char*)(p)))
#endif

// s_waitcnt vmcnt(n) only (expcnt / lgkmcnt left unconstrained); gfx9 simm16 encoding
#define TSNET_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | ((((n) >> 4) & 3) << 14))

// ABL (tools/conv_ablate.py only; any non-zero value computes garbage): bit0 no DMA in the loop,
// bit1 no vmcnt/barrier, bit2 no two-level flush, bit3 no ds_reads (MFMA-only loop).
template <int KS, int BM, int BN, int WARPS_M, int WARPS_N, int FLUSH, int ABL = 0>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N)
void conv_glds_kernel(GldsArgs a) {
    constexpr int BK = 16, KQ = 4, NSTAGE = 4;
    constexpr int NW = WARPS_M * WARPS_N;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 32, NTL = WN / 32;
    constexpr int IA = BM * KQ / 64 / NW;       // A DMA instructions per wave per chunk
    constexpr int IB = BN * KQ / 64 / NW;       // B DMA instructions per wave per chunk
    constexpr int LPC = IA + IB;                // loads per wave per chunk
    static_assert((BM * KQ) % (64 * NW) == 0 && (BN * KQ) % (64 * NW) == 0, "tile must split into whole wave DMAs");
    static_assert(LPC * (NSTAGE - 1) < 64, "vmcnt is 6 bits");
    constexpr int STAGE_F4 = (BM + BN) * KQ;    // float4 per stage

    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    F4* ring = reinterpret_cast<F4*>(smem_raw);  // [NSTAGE][A: BM*4 | B: BN*4] float4

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave / WARPS_N) * WM;
    const int wn0 = (wave % WARPS_N) * WN;
    const int li = lane & 31, lh = lane >> 5;

    const int ntiles = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware tile id (see conv_igemm.hpp)
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- DMA geometry: A instruction j of this wave covers image float4 (j*NW+wave)*64 + lane,
    //      i.e. row (that >> 2), physical quad lane&3, logical quad = physical ^ ((row>>2)&3)
    int a_img[IA], a_img2[IA], a_oy[IA], a_ox[IA], a_kq[IA];
    bool a_rowok[IA];
#pragma unroll
    for (int j = 0; j < IA; ++j) {
        const int row = (j * NW + wave) * 16 + (lane >> 2);
        a_kq[j] = (lane & 3) ^ ((row >> 2) & 3);
        const int m = m0 + row;
        a_rowok[j] = m < a.M;
        const int mm = a_rowok[j] ? m : 0;
        const int hw = a.Ho * a.Wo;
        const int img = mm / hw;
        const int rem = mm - img * hw;
        const int oy = rem / a.Wo;
        a_img[j] = img;
        a_img2[j] = img % a.x2_nmod;
        a_oy[j] = oy * a.stride - a.pad;
        a_ox[j] = (rem - oy * a.Wo) * a.stride - a.pad;
    }
    const int C2 = a.Cin - a.Csplit;

    auto issue_chunk = [&](int kc, int stage) {
        F4* sA = ring + stage * STAGE_F4;
        F4* sB = sA + BM * KQ;
        const bool second = ((kc * BK) & (a.Cin - 1)) >= a.Csplit;   // wave-uniform: chunk never straddles the split
        const float* src = second ? a.x2 : a.x;
        const int cs = second ? C2 : a.Csplit;
#pragma unroll
        for (int j = 0; j < IA; ++j) {
            const int k = kc * BK + a_kq[j] * 4;
            const int tap = k >> a.cin_log2;
            const int c = k & (a.Cin - 1);
            const int ky = tap / KS, kx = tap - ky * KS;
            int iy = a_oy[j] + ky, ix = a_ox[j] + kx;
            bool ok = a_rowok[j] && tap < a.taps;
            if (a.reflect) {
                iy = iy < 0 ? -iy : iy;
                iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
                ix = ix < 0 ? -ix : ix;
                ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
            } else {
                ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            }
            const int img = second ? a_img2[j] : a_img[j];
            const float* p = src + ((size_t)((img * a.H + iy) * a.W + ix)) * cs + (second ? c - a.Csplit : c);
            const float* gp = ok ? p : a.zero_page;
            TSNET_GLDS16(gp, TSNET_LDS_ADDR(sA + (j * NW + wave) * 64));
        }
        const float* wb = a.w + ((size_t)kc * a.Npad + n0) * (KQ * 4);
#pragma unroll
        for (int j = 0; j < IB; ++j) {
            const float* gp = wb + ((j * NW + wave) * 64 + lane) * 4;
            TSNET_GLDS16(gp, TSNET_LDS_ADDR(sB + (j * NW + wave) * 64));
        }
    };

    f32x16 acc[MT][NTL], tot[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

    // read-side geometry: row/col = w?0 + t*32 + li, logical quad (2*s+lh); swizzle depends on li only
    const int rswz = (li >> 2) & 3;
    const int q_s0 = (lh ^ rswz), q_s1 = ((2 + lh) ^ rswz);
    const int a_base = (wm0 + li) * KQ, b_base = BM * KQ + (wn0 + li) * KQ;   // float4 index inside a stage

    // ---- prologue: NSTAGE-1 chunks in flight.  Chunks past the end re-read the last chunk into a
    //      stage nobody consumes: they keep the per-wave vmcnt arithmetic uniform.
    const int nch = a.nchunks;
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) issue_chunk(s < nch ? s : nch - 1, s);

    for (int kc = 0; kc < nch; ++kc) {
        const F4* st = ring + (((ABL & 8) ? 0 : kc) % NSTAGE) * STAGE_F4;
        // chunk kc landed when at most (NSTAGE-2) newer chunks of this wave are outstanding
        if (!(ABL & 2)) {
        TSNET_VMCNT(LPC * (NSTAGE - 2));
        asm volatile("" ::: "memory");         // compiler-level fence only: no LDS access may cross the barrier
        __builtin_amdgcn_s_barrier();          // every wave's share of chunk kc landed; stage (kc-1)%NSTAGE is free again
        asm volatile("" ::: "memory");
        }
        F4 af[2][MT], bf[2][NTL];
        if ((ABL & 8) && kc > 0) { st = ring; }
#pragma unroll
        for (int i = 0; i < MT; ++i) af[0][i] = st[a_base + i * 32 * KQ + q_s0];
#pragma unroll
        for (int j = 0; j < NTL; ++j) bf[0][j] = st[b_base + j * 32 * KQ + q_s0];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[1][i] = st[a_base + i * 32 * KQ + q_s1];
#pragma unroll
        for (int j = 0; j < NTL; ++j) bf[1][j] = st[b_base + j * 32 * KQ + q_s1];
        if (!(ABL & 1)) {   // refill the stage freed by the barrier (after this chunk's ds_reads in program order)
            const int nk = kc + NSTAGE - 1;
            issue_chunk(nk < nch ? nk : nch - 1, nk % NSTAGE);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s][i].v[e], bf[s][j].v[e], acc[i][j], 0, 0, 0);
        if (FLUSH > 0 && !(ABL & 4) && ((kc + 1) % FLUSH) == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
                    tot[i][j] += acc[i][j];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
        }
    }
    TSNET_VMCNT(0);   // drain the tail DMAs before the block may exit

    // ---- epilogue (identical to conv_igemm.hpp)
    const int hw = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const int n = n0 + wn0 + j * 32 + li;
            if (n >= a.Cout) continue;
            const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.M) continue;
                float v = (tot[i][j][r] + acc[i][j][r]) + bv;
                if (a.act == 1) v = tanhf(v);
                if (a.out_nchw) {
                    const int img = m / hw;
                    const int rem = m - img * hw;
                    if (a.composite) {
                        const int ox = rem % a.Wo;
                        if (ox < a.fore_x0 || ox >= a.fore_x1) v = a.bg[n];
                    }
                    a.y[((size_t)img * a.Cout + n) * hw + rem] = v;
                } else {
                    a.y[(size_t)m * a.Cout + n] = v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Wave-specialised variant: NL loader waves issue every LDS-DMA of the workgroup, the
// WARPS_M x WARPS_N consumer waves only ds_read + MFMA.
//
// Why (profiles/round1_notes.md, ablation of conv_glds_kernel): removing the barrier / vmcnt wait
// changes nothing, removing the DMA *issue* from the MFMA-issuing waves gives +24 % -- one LDS-DMA
// instruction holds its wave's issue for ~60-185 cycles (MI355X_MICROARCH.md, "LDS-DMA piece"),
// and a wave issues in order, so the MFMAs queued behind it wait.  A separate wave issues VMEM in
// parallel with another wave's MFMA (different issue ports), so the cost leaves the critical path.
// Loader addressing is incremental: one base pointer per image row per tap, +16 floats per chunk.
// Protocol per chunk (all waves execute the same number of s_barrier):
//   loader  : wait own vmcnt so chunk kc has landed -> s_barrier -> issue chunk kc+3 into the stage
//             the consumers finished reading before that barrier
//   consumer: s_barrier -> ds_read chunk kc -> 32 MFMA
template <int KS, int BM, int BN, int WARPS_M, int WARPS_N, int NL, int FLUSH, bool SMALL_CIN>
__global__ __launch_bounds__(64 * (WARPS_M * WARPS_N + NL))
void conv_glds_ws_kernel(GldsArgs a) {
    constexpr int BK = 16, KQ = 4, NSTAGE = 4;
    constexpr int NC = WARPS_M * WARPS_N;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 32, NTL = WN / 32;
    constexpr int TA = BM / 16, TB = BN / 16;       // DMA instructions per chunk for A and B (1 KiB each)
    static_assert(TA % NL == 0 && TB % NL == 0, "loaders must split the DMA list evenly");
    constexpr int IA = TA / NL, IB = TB / NL, LPC = IA + IB;
    static_assert(LPC * (NSTAGE - 1) < 64, "vmcnt is 6 bits");
    constexpr int STAGE_F4 = (BM + BN) * KQ;

    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    F4* ring = reinterpret_cast<F4*>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    const int ntiles = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nch = a.nchunks;

    if (wave >= NC) {
        // =================================================================== loader wave
        const int ld = wave - NC;
        const int kq_phys = lane & 3;
        int r_img[IA], r_img2[IA], r_oy[IA], r_ox[IA], r_kq[IA];
        bool r_ok[IA];
#pragma unroll
        for (int j = 0; j < IA; ++j) {
            const int row = (j * NL + ld) * 16 + (lane >> 2);
            r_kq[j] = kq_phys ^ ((row >> 2) & 3);
            const int m = m0 + row;
            r_ok[j] = m < a.M;
            const int mm = r_ok[j] ? m : 0;
            const int hw = a.Ho * a.Wo;
            const int img = mm / hw;
            const int rem = mm - img * hw;
            const int oy = rem / a.Wo;
            r_img[j] = img;
            r_img2[j] = img % a.x2_nmod;
            r_oy[j] = oy * a.stride - a.pad;
            r_ox[j] = (rem - oy * a.Wo) * a.stride - a.pad;
        }
        const int C2 = a.Cin - a.Csplit;
        const int cpt_log2 = a.cin_log2 - 4;          // chunks per tap = Cin/16 (Cin >= 16 unless SMALL_CIN)
        const float* p1[IA];                           // per-row pixel base in source 0 / source 1 for the current tap
        const float* p2[IA];
        bool pok[IA];
        int cur_tap = -1;

        auto issue = [&](int kc, int stage) {
            F4* sA = ring + stage * STAGE_F4;
            F4* sB = sA + BM * KQ;
            if (SMALL_CIN) {
                // stem (Cin = 8): a chunk spans two taps, so the tap is per lane; recompute every time
#pragma unroll
                for (int j = 0; j < IA; ++j) {
                    const int k = kc * BK + r_kq[j] * 4;
                    const int tap = k >> a.cin_log2;
                    const int c = k & (a.Cin - 1);
                    const int ky = tap / KS, kx = tap - ky * KS;
                    int iy = r_oy[j] + ky, ix = r_ox[j] + kx;
                    bool ok = r_ok[j] && tap < a.taps;
                    if (a.reflect) {
                        iy = iy < 0 ? -iy : iy;
                        iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
                        ix = ix < 0 ? -ix : ix;
                        ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
                    } else {
                        ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                    }
                    const float* p = a.x + ((size_t)((r_img[j] * a.H + iy) * a.W + ix)) * a.Csplit + c;
                    const float* gp = ok ? p : a.zero_page;
                    TSNET_GLDS16(gp, TSNET_LDS_ADDR(sA + (j * NL + ld) * 64));
                }
            } else {
                const int tap = kc >> cpt_log2;                   // wave-uniform
                const int c0 = (kc << 4) & (a.Cin - 1);
                if (tap != cur_tap) {                              // new tap: rebuild the per-row base pointers
                    cur_tap = tap;
                    const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
                    for (int j = 0; j < IA; ++j) {
                        int iy = r_oy[j] + ky, ix = r_ox[j] + kx;
                        bool ok = r_ok[j] && tap < a.taps;
                        if (a.reflect) {
                            iy = iy < 0 ? -iy : iy;
                            iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
                            ix = ix < 0 ? -ix : ix;
                            ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
                        } else {
                            ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                        }
                        iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);
                        ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
                        pok[j] = ok;
                        p1[j] = a.x + ((size_t)((r_img[j] * a.H + iy) * a.W + ix)) * a.Csplit + r_kq[j] * 4;
                        p2[j] = a.x2 ? a.x2 + ((size_t)((r_img2[j] * a.H + iy) * a.W + ix)) * C2 + r_kq[j] * 4 : a.zero_page;
                    }
                }
                const bool second = c0 >= a.Csplit;
#pragma unroll
                for (int j = 0; j < IA; ++j) {
                    const float* p = second ? p2[j] + (c0 - a.Csplit) : p1[j] + c0;
                    const float* gp = pok[j] ? p : a.zero_page;
                    TSNET_GLDS16(gp, TSNET_LDS_ADDR(sA + (j * NL + ld) * 64));
                }
            }
            const float* wb = a.w + ((size_t)kc * a.Npad + n0) * (KQ * 4);
#pragma unroll
            for (int j = 0; j < IB; ++j) {
                const float* gp = wb + ((j * NL + ld) * 64 + lane) * 4;
                TSNET_GLDS16(gp, TSNET_LDS_ADDR(sB + (j * NL + ld) * 64));
            }
        };

#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s) issue(s < nch ? s : nch - 1, s);
        for (int kc = 0; kc < nch; ++kc) {
            TSNET_VMCNT(LPC * (NSTAGE - 2));        // this loader's share of chunk kc has landed
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int nk = kc + NSTAGE - 1;
            issue(nk < nch ? nk : nch - 1, nk % NSTAGE);
        }
        TSNET_VMCNT(0);
        return;
    }

    // ======================================================================= consumer wave
    const int wm0 = (wave / WARPS_N) * WM;
    const int wn0 = (wave % WARPS_N) * WN;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[MT][NTL], tot[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

    const int rswz = (li >> 2) & 3;
    const int q_s0 = (lh ^ rswz), q_s1 = ((2 + lh) ^ rswz);
    const int a_base = (wm0 + li) * KQ, b_base = BM * KQ + (wn0 + li) * KQ;

    for (int kc = 0; kc < nch; ++kc) {
        const F4* st = ring + (kc % NSTAGE) * STAGE_F4;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();          // the loaders' chunk kc is in LDS; see protocol above
        asm volatile("" ::: "memory");
        F4 af[2][MT], bf[2][NTL];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[0][i] = st[a_base + i * 32 * KQ + q_s0];
#pragma unroll
        for (int j = 0; j < NTL; ++j) bf[0][j] = st[b_base + j * 32 * KQ + q_s0];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[1][i] = st[a_base + i * 32 * KQ + q_s1];
#pragma unroll
        for (int j = 0; j < NTL; ++j) bf[1][j] = st[b_base + j * 32 * KQ + q_s1];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s][i].v[e], bf[s][j].v[e], acc[i][j], 0, 0, 0);
        if (FLUSH > 0 && ((kc + 1) % FLUSH) == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
                    tot[i][j] += acc[i][j];
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                }
        }
    }

    const int hw = a.Ho * a.Wo;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const int n = n0 + wn0 + j * 32 + li;
            if (n >= a.Cout) continue;
            const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m >= a.M) continue;
                float v = (tot[i][j][r] + acc[i][j][r]) + bv;
                if (a.act == 1) v = tanhf(v);
                if (a.out_nchw) {
                    const int img = m / hw;
                    const int rem = m - img * hw;
                    if (a.composite) {
                        const int ox = rem % a.Wo;
                        if (ox < a.fore_x0 || ox >= a.fore_x1) v = a.bg[n];
                    }
                    a.y[((size_t)img * a.Cout + n) * hw + rem] = v;
                } else {
                    a.y[(size_t)m * a.Cout + n] = v;
                }
            }
        }
    }
}

// OIHW -> [K/16][Npad][4 physical quads][4], physical quad p of column n holds logical quad p ^ ((n>>2)&3)
// The packed layer may take a window [cin_off, cin_off+cin_real) of the parameter's cin_total input channels
// (FuseNet's first conv is split into its source half and its shared target half).
__global__ void pack_weights_glds_kernel(const float* __restrict__ w, float* __restrict__ out,
                                         int cout, int cin_real, int cin_pad, int ks, int kpad, int npad, int cin_total, int cin_off) {
    const size_t total = (size_t)kpad * npad;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 3;
        const int p = (idx >> 2) & 3;
        const size_t rest = idx >> 4;
        const int n = (int)(rest % npad);
        const int kc = (int)(rest / npad);
        const int k = kc * 16 + (p ^ ((n >> 2) & 3)) * 4 + e;
        const int tap = k / cin_pad, c = k - tap * cin_pad;
        float v = 0.f;
        if (tap < ks * ks && c < cin_real && n < cout) {
            const int ky = tap / ks, kx = tap - ky * ks;
            v = w[(((size_t)n * cin_total + cin_off + c) * ks + ky) * ks + kx];
        }
        out[idx] = v;
    }
}

}  // namespace tsnet
