// conv_x3.hpp -- fp32-accurate implicit-GEMM convolution on the bf16 MFMA via a 3-way operand split.
//
// Every fp32 operand x is stored as three bf16 planes with x = hi + mid + lo to 2^-27 (round-to-nearest
// residual split, split3.hpp).  A product a*b is evaluated as the six bf16 products
//   lo*hi + hi*lo + mid*mid + mid*hi + hi*mid + hi*hi          (dropped terms are <= 2^-24 |a||b|)
// on v_mfma_f32_32x32x16_bf16 (exact bf16 products, fp32 accumulate), 6 MFMAs per 16-deep k-group:
// 12 matrix-pipe cycles per k instead of 32 on the fp32 MFMA, i.e. a 2.67x higher ceiling (417 TF
// fp32-equivalent).  tools/probes/bf16x3_probe.hip measured the accuracy on gfx950: mean |err| vs fp64
// 7.8e-7 against 6.5e-7 for the fp32 MFMA chain at K=4608 -- the same error class, which the 1e-3
// parity budget needs.  Accumulation is two-level exactly like conv_dma.hpp (fold every 64 products).
//
// Structure = conv_dma.hpp: buffer-descriptor LDS-DMA (hardware zero fill for padding), ring of LDS
// stages with counted vmcnt, one barrier per chunk, source-side swizzle.  Differences: the planes are
// separate tensors (3 DMAs where the fp32 kernel has 1, but 2-byte elements), an LDS row of a plane is
// 32 B per 16 k (two 16-byte octets, swizzled by (row>>3)&1), and a wave instruction covers 32 rows.
#pragma once
#include <hip/hip_runtime.h>

#include "conv_dma.hpp"
#include "split3.hpp"

namespace tsnet {

struct X3Args {
    const unsigned short* x;    // source 0 planes: (3, N,H,W,Csplit) bf16
    const unsigned short* x2;   // source 1 planes or null: (3, x2_nmod,H,W,Cin-Csplit)
    const unsigned short* w;    // packed planes: [3][K/16][Npad][2 swizzled octets][8] bf16
    const float* bias;
    float* y;                   // fp32 NHWC output (raw conv result)
    unsigned short* y3;         // null, or split planes of y (3, N,Ho,Wo,Cout) for a conv that feeds a conv directly
    double* stat_part;
    const float* addend; int add_nmod;
    int N, H, W, Cin, cin_log2, Csplit, x2_nmod;
    int Ho, Wo, Cout, Npad;
    int stride, pad, reflect, taps, nchunks, M;   // nchunks in units of 16 k
    int tiles_m, tiles_n;
    // optional: the last workgroup to deliver the statistics of an (image, channel tile) finalises them itself
    // (alpha = rstd, beta = -mean*rstd), replacing the in_finalize2 launch; null = off
    float* fin_alpha; float* fin_beta; int* fin_counter; int fin_S; float fin_eps;
    // optional: max |y| of the launch, as the bit pattern of a non-negative float (order-independent atomic max: deterministic).  The
    // fp16 x 2 kernels (conv_h2.hpp) derive the operand scale of a tensor without an a-priori bound from it; null = off
    unsigned* amax_out;    // null, or amax_out[image] <- max |y| of that image (float bits, atomic max); needs Ho*Wo % BM == 0
};

#ifndef TSNET_DRAIN_VMEM
#define TSNET_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

#ifndef TSNET_MFMA_BF16
typedef __bf16 tsnet_bf16x8 __attribute__((ext_vector_type(8)));
#define TSNET_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tsnet_bf16x8, a), __builtin_bit_cast(tsnet_bf16x8, b), c, 0, 0, 0)
#endif

// Shared epilogue of the bf16x3 kernels: bias, optional per-pixel addend, fp64 InstanceNorm partial sums of the tile,
// fp32 store and/or split-plane store.  m_of(l) maps the local row l of the tile to the output position m (or -1).
template <int BN, int WARPS_M, int WARPS_N, int MT, int NTL, typename Args, typename MOf>
__device__ __forceinline__ void x3_epilogue(const Args& a, f32x16 (&tot)[MT][NTL], unsigned char* smem_raw, int tid, int wave, int n0,
                                            size_t stat_tile, MOf m_of) {
    constexpr int WM = MT * 32, WN = NTL * 32;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WARPS_N) * WM, wn0 = (wave % WARPS_N) * WN;
    const int hw = a.Ho * a.Wo;
    double csum[NTL], csq[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) { csum[j] = 0.0; csq[j] = 0.0; }
    float vmax = 0.f;
    const size_t yplane = (size_t)a.M * a.Cout;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const int n = n0 + wn0 + j * 32 + li;
            const bool nok = n < a.Cout;
            const float bv = (a.bias && nok) ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_of(wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
                const bool mok = m >= 0;
                float v = tot[i][j][r] + bv;
                if (a.addend && nok && mok) {
                    const int img = m / hw;
                    v += a.addend[((size_t)(img % a.add_nmod) * hw + (m - img * hw)) * a.Cout + n];
                }
                if (a.stat_part && mok) { csum[j] += (double)v; csq[j] += (double)v * (double)v; }
                if (!nok || !mok) continue;
                vmax = __builtin_fmaxf(vmax, __builtin_fabsf(v));
                if (a.y) a.y[(size_t)m * a.Cout + n] = v;
                if (a.y3) {
                    unsigned short sh, sm, sl;
                    split3_scalar(v, sh, sm, sl);
                    const size_t o = (size_t)m * a.Cout + n;
                    a.y3[o] = sh; a.y3[yplane + o] = sm; a.y3[2 * yplane + o] = sl;
                }
            }
        }
    }
    if (a.amax_out) { const int mf = m_of(0); tsnet_publish_amax(a.amax_out + (mf < 0 ? 0 : mf / hw), vmax); }   // the tile lies inside one image (host-checked)
    if (a.stat_part) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem_raw);
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const double s2 = csum[j] + __shfl_xor(csum[j], 32);
            const double q2 = csq[j] + __shfl_xor(csq[j], 32);
            if (lh == 0) {
                double* o = red + ((size_t)(wave / WARPS_N) * BN + wn0 + j * 32 + li) * 2;
                o[0] = s2; o[1] = q2;
            }
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.Cout) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int wmi = 0; wmi < WARPS_M; ++wmi) { s += red[((size_t)wmi * BN + tid) * 2]; q += red[((size_t)wmi * BN + tid) * 2 + 1]; }
            double* o = a.stat_part + (stat_tile * a.Cout + n0 + tid) * 2;
            if (a.fin_counter) {            // device-scope write-through: another XCD's workgroup may read them
                __hip_atomic_store(o, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 1, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                o[0] = s; o[1] = q;
            }
        }
        if (a.fin_counter) {
            // Arrival counter per (image, channel tile).  The partials above are
            // acknowledged at device scope (vmcnt(0)) before this workgroup counts itself, so the workgroup that
            // reads fin_S - 1 sees all of them.  It sums them in in_finalize2_kernel's order (four interleaved
            // groups, then g0+g1+g2+g3): bit-identical to the separate kernel, run-to-run deterministic.
            const int S = a.fin_S;
            const int img = (int)(stat_tile / (size_t)S);
            int* counter = a.fin_counter + (size_t)img * ((a.Npad + 31) / 32) + n0 / 32;   // 32 = narrowest tile
            int* flag = reinterpret_cast<int*>(smem_raw + 8192);
            // Hand-off form (MI355X_MICROARCH.md, "valid forms"): 8-byte agent-scope atomics on BOTH sides (write-through sc1 stores
            // above, sc1 loads below: never served from a stale L1 / another XCD's L2), the producer's stores drained before it
            // counts itself.  The drain is inline asm on purpose: the compiler may drop a builtin s_waitcnt it can prove redundant.
            TSNET_DRAIN_VMEM();
            __syncthreads();
            if (tid == 0) *flag = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (*flag == S - 1) {
                if (tid < BN && n0 + tid < a.Cout) {
                    double gs[4] = {0, 0, 0, 0}, gq[4] = {0, 0, 0, 0};
                    for (int t = 0; t < S; ++t) {
                        const double* p = a.stat_part + (((size_t)img * S + t) * a.Cout + n0 + tid) * 2;
                        gs[t & 3] += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        gq[t & 3] += __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    double sm = gs[0], sq = gq[0];
                    for (int k = 1; k < 4; ++k) { sm += gs[k]; sq += gq[k]; }
                    const double mean = sm / hw;
                    double var = sq / hw - mean * mean;
                    if (var < 0) var = 0;
                    const float al = 1.0f / sqrtf((float)var + a.fin_eps);
                    a.fin_alpha[(size_t)img * a.Cout + n0 + tid] = al;
                    a.fin_beta[(size_t)img * a.Cout + n0 + tid] = -((float)mean) * al;
                }
                if (tid == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
        }
    }
}

// KC = 16-deep k-groups per ring stage (1 or 2); NSTAGE = ring depth
// ABL (tools/x3_ablate.py only; non-zero computes garbage): bit0 no DMA in the loop, bit1 no vmcnt/barrier,
// bit2 no fold, bit3 no ds_reads (fragments loaded once), bit4 no B-operand DMAs,
// bit5 no A-operand DMAs, bit6 A DMAs read contiguous KiBs instead of gathering 32 B per row.
template <int KS, int BM, int BN, int WARPS_M, int WARPS_N, int KC, int NSTAGE, bool SMALL_CIN, int ABL = 0>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N)
void conv_x3_kernel(X3Args a) {
    constexpr int NW = WARPS_M * WARPS_N;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 32, NTL = WN / 32;
    // DMA list of one (k-group, plane): TA = BM/32 A blocks then TB = BN/32 B blocks, 1 KiB each.  Wave w serves
    // entries e = r*NW + w, r < R; entries past the list are zero-fill DMAs into a scratch KiB so that every wave
    // issues the same, compile-time number of loads per ring step (the vmcnt arithmetic needs that).
    constexpr int TA = BM / 32, TB = BN / 32, E = TA + TB;
    constexpr int R = (E + NW - 1) / NW;
    constexpr int RA = (TA + NW - 1) / NW;                // entries r < RA may be A blocks
    constexpr int LPC = R * 3 * KC;
    static_assert(LPC * (NSTAGE - 1) < 64, "vmcnt is 6 bits");
    constexpr int PLANE_A = BM * 32, PLANE_B = BN * 32;   // bytes of one plane of one k-group
    constexpr int GROUP_BYTES = 3 * (PLANE_A + PLANE_B);
    constexpr int STAGE_BYTES = KC * GROUP_BYTES + 1024;  // + scratch slot for surplus DMAs

    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int wm0 = (wave / WARPS_N) * WM;
    const int wn0 = (wave % WARPS_N) * WN;
    const int li = lane & 31, lh = lane >> 5;

    const int ntiles = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nst = (a.nchunks + KC - 1) / KC;            // ring steps

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

    // ---- per-lane geometry of the A row blocks this wave serves (block t = r*NW + wave; row = lane>>1,
    //      physical octet = lane&1, logical octet swizzled so rows r and r+8 of a 16-lane read group differ)
    const int row_in = lane >> 1;
    const int oct_phys = lane & 1;
    const int oct_log = oct_phys ^ ((row_in >> 3) & 1);
    int g_pix[RA], g_pix2[RA], g_oy[RA], g_ox[RA];
    bool g_ok[RA];
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int m = m0 + (r * NW + wave) * 32 + row_in;
        g_ok[r] = m < a.M && (r * NW + wave) < TA;
        const int mm = g_ok[r] ? m : 0;
        const int hw = a.Ho * a.Wo;
        const int img = mm / hw;
        const int rem = mm - img * hw;
        const int oy = rem / a.Wo;
        g_pix[r] = img * a.H * a.W;
        g_pix2[r] = (img % a.x2_nmod) * a.H * a.W;
        g_oy[r] = oy * a.stride - a.pad;
        g_ox[r] = (rem - oy * a.Wo) * a.stride - a.pad;
    }
    unsigned vA1[RA], vA2[RA];
    const int cpt_log2 = SMALL_CIN ? 0 : a.cin_log2 - 4;
    int cur_tap = -1;
    // byte offsets of (row, tap) inside one plane of source 0 / source 1; kOOB = read zeros
    auto tap_offsets = [&](int tap, int c_lane) {
        const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            int iy = g_oy[r] + ky, ix = g_ox[r] + kx;
            bool ok = g_ok[r] && tap < a.taps;
            if (a.reflect) {
                iy = iy < 0 ? -iy : iy;
                iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
                ix = ix < 0 ? -ix : ix;
                ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
            } else {
                ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            }
            const int pix = iy * a.W + ix;
            vA1[r] = ok ? (unsigned)(((g_pix[r] + pix) * a.Csplit + c_lane) * 2) : kOOB;
            vA2[r] = ok ? (unsigned)(((g_pix2[r] + pix) * C2 + c_lane) * 2) : kOOB;
        }
    };
    const unsigned vB = (unsigned)(lane * 16);              // weights are packed in image order already

    // issue all DMAs of ring step `st` (k-groups st*KC .. st*KC+KC-1) into ring stage `stage`
    auto issue_step = [&](int st, int stage) {
        const tsnet_lds_t ls = lds0 + stage * STAGE_BYTES;
#pragma unroll
        for (int g = 0; g < KC; ++g) {
            const int kc = st * KC + g;                     // 16-deep k-group (past the end: everything reads zeros)
            const tsnet_lds_t lg = ls + g * GROUP_BYTES;
            unsigned so_a = 0;
            bool second = false;
            if (SMALL_CIN) {                                // Cin = 8: the two octets of a row are two different taps
                const int k = kc * 16 + oct_log * 8;
                tap_offsets(k >> a.cin_log2, k & (a.Cin - 1));
            } else {
                const int tap = kc >> cpt_log2;             // wave-uniform
                const int c0 = (kc << 4) & (a.Cin - 1);
                if (tap != cur_tap) { cur_tap = tap; tap_offsets(tap, oct_log * 8); }
                second = c0 >= a.Csplit;
                so_a = (unsigned)((second ? c0 - a.Csplit : c0) * 2);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int e = r * NW + wave;            // wave-uniform list entry
                    if (r < RA && e < TA) {
                        const tsnet_lds_t dst = lg + p * PLANE_A + e * 1024;
                        if (ABL & 32) continue;
                        if (ABL & 64) { TSNET_BUF_DMA16(rs1[p], vB, so_a, dst); continue; }
                        if (second) TSNET_BUF_DMA16(rs2[p], vA2[r < RA ? r : 0], so_a, dst);
                        else TSNET_BUF_DMA16(rs1[p], vA1[r < RA ? r : 0], so_a, dst);
                    } else if (e < E) {
                        if (ABL & 16) continue;
                        const int tb = e - TA;
                        const tsnet_lds_t dst = lg + 3 * PLANE_A + p * PLANE_B + tb * 1024;
                        const unsigned so = (unsigned)((kc * a.Npad + n0 + tb * 32) * 32);
                        TSNET_BUF_DMA16(rsw[p], vB, so, dst);
                    } else {
                        const unsigned oob = kOOB;
                        TSNET_BUF_DMA16(rsw[p], oob, 0u, ls + KC * GROUP_BYTES);     // surplus slot: zero fill into the scratch KiB
                    }
                }
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

    // read side: lane (li, lh) needs octet lh (k = 8*lh .. 8*lh+7) of row/col w?0 + t*32 + li
    const int oct_r = lh ^ ((li >> 3) & 1);
    const int a_off = (wm0 + li) * 32 + oct_r * 16;         // byte offset inside a plane
    const int b_off = (wn0 + li) * 32 + oct_r * 16;

    // one ring step; inlined at NSTAGE call sites so the stage indices are constants.  The fmaf chain is folded
    // into the running total every 4 k-groups (64 products) counted from k = 0 -- the SAME fold points for every
    // tile shape and ring depth, so a layer's result does not depend on which tile the heuristic picked.
    auto step = [&](int st, int stage, int refill_stage) {
        if (!(ABL & 2)) {
            TSNET_VMCNT(LPC * (NSTAGE - 2));
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        const unsigned char* sbase = smem_raw + ((ABL & 8) ? 0 : stage) * STAGE_BYTES;
#pragma unroll
        for (int g = 0; g < KC; ++g) {
            const unsigned char* gb = sbase + g * GROUP_BYTES;
            F4 af[3][MT], bf[3][NTL];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < MT; ++i) af[p][i] = *reinterpret_cast<const F4*>(gb + p * PLANE_A + i * 1024 + a_off);
#pragma unroll
                for (int j = 0; j < NTL; ++j) bf[p][j] = *reinterpret_cast<const F4*>(gb + 3 * PLANE_A + p * PLANE_B + j * 1024 + b_off);
            }
            if (g == 0 && !(ABL & 1)) issue_step(st + NSTAGE - 1, refill_stage);
            // six products per tile: lo*hi, mid*hi, hi*hi, mid*mid, hi*mid, hi*lo -- the order of conv_x3r.hpp (each
            // weight plane retires early there); the two kernels must agree bit for bit
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 0, 0, 1, 1, 2};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
                        acc[i][j] = TSNET_MFMA_BF16(af[PA[q]][i], bf[PB[q]][j], acc[i][j]);
            if (!(ABL & 4) && (((st * KC + g) + 1) & 3) == 0) {          // wave-uniform
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
    };

    static_assert(NSTAGE == 3 || NSTAGE == 4, "ring depth");
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) issue_step(s, s);     // steps past the end read zeros (OOB / K-padded weights)
    for (int st = 0; st < nst; st += NSTAGE) {
        step(st, 0, NSTAGE - 1);
        if (st + 1 < nst) step(st + 1, 1, 0);
        if (st + 2 < nst) step(st + 2, 2, 1);
        if (NSTAGE == 4 && st + 3 < nst) step(st + 3, 3, 2);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j) tot[i][j] += acc[i][j];    // the last, partial chain
    TSNET_VMCNT(0);

    // ---- epilogue (as conv_dma.hpp, plus the optional split-plane copy of the output)
    const int hw = a.Ho * a.Wo;
    const int img0 = m0 / hw;
    x3_epilogue<BN, WARPS_M, WARPS_N, MT, NTL>(a, tot, smem_raw, tid, wave, n0, (size_t)img0 * (hw / BM) + (m0 - img0 * hw) / BM,
                                               [&](int l) { const int m = m0 + l; return m < a.M ? m : -1; });
}

// OIHW fp32 -> three bf16 planes in image order: plane p, k-group kc, column n, physical octet o, element e:
//   out[p][((kc*Npad + n)*2 + o)*8 + e] = part_p( W[k = kc*16 + (o ^ ((n>>3)&1))*8 + e][n] )
__global__ void pack_weights_x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
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
        unsigned short sh, sm, sl;
        split3_scalar(v, sh, sm, sl);
        out[idx] = sh; out[plane + idx] = sm; out[2 * plane + idx] = sl;
    }
}

// fp32 NHWC tensor -> three bf16 planes (stand-alone converter; producers normally write planes directly)
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, unsigned short* __restrict__ out, size_t total4) {
    const size_t plane = total4 * 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x)
        split3_store(reinterpret_cast<const float4*>(x)[i], out + i * 4, out + plane + i * 4, out + 2 * plane + i * 4);
}

}  // namespace tsnet
