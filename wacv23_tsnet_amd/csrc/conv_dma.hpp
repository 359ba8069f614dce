// conv_dma.hpp -- NHWC implicit-GEMM convolution on the exact-fp32 MFMA with a lean LDS-DMA pipeline.
//
// Third generation of the conv kernel (history and measurements: profiles/round1_notes.md):
//   conv_igemm.hpp  register-staged loads                      ~90 TF   (global-load latency exposed)
//   (removed)       global_load_lds ring, per-lane 64-bit math ~100 TF  (MFMA pipe 70 % busy: the DMA
//                   path doubled the VALU instruction count; ~6 non-MFMA instructions per MFMA)
//   this file       buffer_load ... lds with a buffer descriptor: the hardware adds
//                   base + voffset(VGPR, per lane, changes once per filter tap) + soffset(SGPR, the
//                   per-chunk channel advance), and out-of-range offsets return ZERO, which is
//                   exactly zero padding / ragged rows / padded K -- no zero page, no selects, no
//                   64-bit address arithmetic.  A chunk costs each wave 4 DMAs of 3 instructions,
//                   8 ds_read_b128, 32 MFMA and one barrier.
// Other ingredients (unchanged numerics: exact fp32 products, fmaf chains of <= 64 products folded
// into a running total): 4-stage LDS ring with counted vmcnt; K loop unrolled by 4 so ring stages,
// LDS addresses and the fold points are compile-time constants and the first MFMA after a fold takes
// C = 0 as an inline constant (no accumulator re-zeroing); source-side XOR swizzle of the k-quads
// for conflict-free ds_read_b128 (0 bank conflicts measured).
#pragma once
#include <hip/hip_runtime.h>

#include "conv_igemm.hpp"

namespace tsnet {

struct DmaArgs {
    const float* x;         // source 0, NHWC (N,H,W,Csplit)
    const float* x2;        // source 1 (channels >= Csplit) or null
    const float* w;         // packed [K/16][Npad][4 k-quads (swizzled)][4]
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

// s_waitcnt vmcnt(n) only (expcnt / lgkmcnt left unconstrained); gfx9 simm16 encoding
#define TSNET_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | ((((n) >> 4) & 3) << 14))

// ---- the hardware hook (tests/emu predefines these four names to run the kernel on the CPU) ----
#ifndef TSNET_BUF_DMA16
typedef unsigned tsnet_rsrc_t __attribute__((ext_vector_type(4)));   // 128-bit buffer descriptor (SGPRs)
typedef unsigned tsnet_lds_t;                                         // LDS byte address (wave-uniform)
__device__ __forceinline__ tsnet_rsrc_t tsnet_make_rsrc(const void* p, unsigned bytes) {
    const unsigned long long addr = (unsigned long long)p;
    tsnet_rsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)addr);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32) & 0xFFFFu);   // stride 0: raw buffer
    r.z = __builtin_amdgcn_readfirstlane(bytes);                              // num_records (bytes)
    r.w = 0x00020000u;
    return r;
}
#define TSNET_LDS_BASE(p) __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) unsigned char*)(p)))
// One LDS-DMA: lane l fetches 16 B at base + voff[l] + soff (zeros if out of range) into LDS at lds + 16*l.
// m0 carries the LDS address; s_nop covers the m0->DMA and SGPR->VMEM wait states (section 5.7 of the guide).
#define TSNET_BUF_DMA16(rsrc, voff, soff, lds) \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds), "v"(voff), "s"(rsrc), "s"(soff) : "memory")
#define TSNET_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

constexpr unsigned kOOB = 0x80000000u;   // voffset of a lane that must read zeros (tensors are < 2 GiB, checked on the host)

template <int KS, int BM, int BN, int WARPS_M, int WARPS_N, bool SMALL_CIN>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N)
void conv_dma_kernel(DmaArgs a) {
    constexpr int KQ = 4, NSTAGE = 4;
    constexpr int NW = WARPS_M * WARPS_N;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 32, NTL = WN / 32;
    constexpr int IA = BM * KQ / 64 / NW, IB = BN * KQ / 64 / NW, LPC = IA + IB;
    static_assert((BM * KQ) % (64 * NW) == 0 && (BN * KQ) % (64 * NW) == 0, "tile must split into whole wave DMAs");
    static_assert(LPC * (NSTAGE - 1) < 64, "vmcnt is 6 bits");
    constexpr int STAGE_F4 = (BM + BN) * KQ;
    constexpr int STAGE_BYTES = STAGE_F4 * 16;

    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const F4* ring = reinterpret_cast<const F4*>(smem_raw);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
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
    const int nch = a.nchunks;

    // ---- descriptors and wave-uniform LDS destinations
    const int C2 = a.Cin - a.Csplit;
    const tsnet_rsrc_t rs1 = tsnet_make_rsrc(a.x, (unsigned)((size_t)a.N * a.H * a.W * a.Csplit * 4));
    const tsnet_rsrc_t rs2 = tsnet_make_rsrc(a.x2 ? a.x2 : a.x, a.x2 ? (unsigned)((size_t)a.x2_nmod * a.H * a.W * C2 * 4) : 0u);
    const tsnet_rsrc_t rsw = tsnet_make_rsrc(a.w, (unsigned)((size_t)((nch + 1) / 2 * 2) * a.Npad * 64));
    const tsnet_lds_t lds0 = TSNET_LDS_BASE(smem_raw);

    // ---- per-lane geometry of the A rows this lane feeds (fixed over the K loop)
    int g_pix[IA], g_pix2[IA], g_oy[IA], g_ox[IA], g_kq[IA];
    bool g_ok[IA];
#pragma unroll
    for (int j = 0; j < IA; ++j) {
        const int row = (j * NW + wave) * 16 + (lane >> 2);
        g_kq[j] = (lane & 3) ^ ((row >> 2) & 3);
        const int m = m0 + row;
        g_ok[j] = m < a.M;
        const int mm = g_ok[j] ? m : 0;
        const int hw = a.Ho * a.Wo;
        const int img = mm / hw;
        const int rem = mm - img * hw;
        const int oy = rem / a.Wo;
        g_pix[j] = img * a.H * a.W;
        g_pix2[j] = (img % a.x2_nmod) * a.H * a.W;
        g_oy[j] = oy * a.stride - a.pad;
        g_ox[j] = (rem - oy * a.Wo) * a.stride - a.pad;
    }
    unsigned vA1[IA], vA2[IA], vB[IB];
#pragma unroll
    for (int j = 0; j < IB; ++j) vB[j] = (unsigned)(((j * NW + wave) * 64 + lane) * 16);
    const int cpt_log2 = SMALL_CIN ? 0 : a.cin_log2 - 4;      // chunks per tap = Cin/16
    int cur_tap = -1;

    // byte offsets of (row j, tap) inside source 0 / source 1; kOOB where the tap falls into zero
    // padding, beyond the last row or beyond the last tap
    auto tap_offsets = [&](int tap) {
        const int ky = tap / KS, kx = tap - ky * KS;
#pragma unroll
        for (int j = 0; j < IA; ++j) {
            int iy = g_oy[j] + ky, ix = g_ox[j] + kx;
            bool ok = g_ok[j] && tap < a.taps;
            if (a.reflect) {
                iy = iy < 0 ? -iy : iy;
                iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
                ix = ix < 0 ? -ix : ix;
                ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
            } else {
                ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            }
            const int pix = iy * a.W + ix;
            vA1[j] = ok ? (unsigned)(((g_pix[j] + pix) * a.Csplit + g_kq[j] * 4) * 4) : kOOB;
            vA2[j] = ok ? (unsigned)(((g_pix2[j] + pix) * C2 + g_kq[j] * 4) * 4) : kOOB;
        }
    };

    // issue the DMAs of chunk kc into ring stage `stage` (compile-time constant at every call site)
    auto issue_chunk = [&](int kc, int stage) {
        const tsnet_lds_t la = lds0 + stage * STAGE_BYTES + wave * 1024;
        if (SMALL_CIN) {
            // stem: Cin < 16, a chunk spans several taps, so the tap is per lane
#pragma unroll
            for (int j = 0; j < IA; ++j) {
                const int k = kc * 16 + g_kq[j] * 4;
                const int tap = k >> a.cin_log2, c = k & (a.Cin - 1);
                const int ky = tap / KS, kx = tap - ky * KS;
                int iy = g_oy[j] + ky, ix = g_ox[j] + kx;
                bool ok = g_ok[j] && tap < a.taps;
                if (a.reflect) {
                    iy = iy < 0 ? -iy : iy;
                    iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
                    ix = ix < 0 ? -ix : ix;
                    ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
                } else {
                    ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                }
                const unsigned v = ok ? (unsigned)(((g_pix[j] + iy * a.W + ix) * a.Csplit + c) * 4) : kOOB;
                TSNET_BUF_DMA16(rs1, v, 0u, la + j * NW * 1024);
            }
        } else {
            const int tap = kc >> cpt_log2;                       // wave-uniform
            const int c0 = (kc << 4) & (a.Cin - 1);
            if (tap != cur_tap) { cur_tap = tap; tap_offsets(tap); }
            if (c0 < a.Csplit) {
                const unsigned so = (unsigned)(c0 * 4);
#pragma unroll
                for (int j = 0; j < IA; ++j) TSNET_BUF_DMA16(rs1, vA1[j], so, la + j * NW * 1024);
            } else {
                const unsigned so = (unsigned)((c0 - a.Csplit) * 4);
#pragma unroll
                for (int j = 0; j < IA; ++j) TSNET_BUF_DMA16(rs2, vA2[j], so, la + j * NW * 1024);
            }
        }
        const unsigned sob = (unsigned)((kc * a.Npad + n0) * 64);
        const tsnet_lds_t lb = la + BM * KQ * 16;
#pragma unroll
        for (int j = 0; j < IB; ++j) TSNET_BUF_DMA16(rsw, vB[j], sob, lb + j * NW * 1024);
    };

    f32x16 acc[MT][NTL], tot[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

    // read side: row/col = w?0 + t*32 + li, logical quad 2*s+lh, swizzle from li only
    const int rswz = (li >> 2) & 3;
    const int q_s0 = (lh ^ rswz), q_s1 = ((2 + lh) ^ rswz);
    const int a_base = (wm0 + li) * KQ, b_base = BM * KQ + (wn0 + li) * KQ;

    // one K-chunk: wait for its DMAs, barrier, read fragments, refill the stage freed by the barrier,
    // 32 MFMA.  FIRST: the accumulators were just folded, so the first MFMA of every tile takes C = 0.
    auto step = [&](int kc, int stage, int refill_stage, bool first) {
        TSNET_VMCNT(LPC * (NSTAGE - 2));       // this wave's share of chunk kc has landed
        asm volatile("" ::: "memory");         // (compiler fence: no LDS access may cross the barrier)
        __builtin_amdgcn_s_barrier();          // everybody's share landed; everybody is done with refill_stage
        asm volatile("" ::: "memory");
        const F4* st = ring + stage * STAGE_F4;
        F4 af[2][MT], bf[2][NTL];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[0][i] = st[a_base + i * 32 * KQ + q_s0];
#pragma unroll
        for (int j = 0; j < NTL; ++j) bf[0][j] = st[b_base + j * 32 * KQ + q_s0];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[1][i] = st[a_base + i * 32 * KQ + q_s1];
#pragma unroll
        for (int j = 0; j < NTL; ++j) bf[1][j] = st[b_base + j * 32 * KQ + q_s1];
        {
            const int nk = kc + NSTAGE - 1;    // past-the-end chunks re-read the last one: vmcnt stays uniform
            issue_chunk(nk < nch ? nk : nch - 1, refill_stage);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j) {
                        if (first && s == 0 && e == 0) {
                            f32x16 z;
#pragma unroll
                            for (int r = 0; r < 16; ++r) z[r] = 0.f;
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s][i].v[e], bf[s][j].v[e], z, 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s][i].v[e], bf[s][j].v[e], acc[i][j], 0, 0, 0);
                        }
                    }
    };

    issue_chunk(0, 0);
    issue_chunk(1 < nch ? 1 : nch - 1, 1);
    issue_chunk(2 < nch ? 2 : nch - 1, 2);
    for (int kc = 0; kc < nch; kc += NSTAGE) {
        step(kc, 0, 3, true);
        if (kc + 1 < nch) step(kc + 1, 1, 0, false);
        if (kc + 2 < nch) step(kc + 2, 2, 1, false);
        if (kc + 3 < nch) step(kc + 3, 3, 2, false);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) tot[i][j] += acc[i][j];    // fold <= 64 products into the running total
    }
    TSNET_VMCNT(0);   // drain the tail DMAs before the block may exit

    // ---- epilogue: bias, activation, store.  D layout: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
    // With a.stat_part the InstanceNorm statistics of the output are produced here as well (the host
    // only asks for it when every row of a tile belongs to one image, Ho*Wo % BM == 0): per column the
    // wave sums its WM rows in fp64 (so the statistics do not depend on the tile shape), the WARPS_M waves combine through LDS
    // in a fixed order, one (sum, sumsq) pair per (image, tile, channel) goes to HBM; in_finalize2_kernel
    // reduces the tiles.  This removes the separate read pass over every conv output.
    const int hw = a.Ho * a.Wo;
    double csum[NTL], csq[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) { csum[j] = 0.0; csq[j] = 0.0; }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const int n = n0 + wn0 + j * 32 + li;
            const bool nok = n < a.Cout;
            const float bv = (a.bias && nok) ? a.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = tot[i][j][r] + bv;
                if (a.addend && nok && m < a.M) {
                    const int img = m / hw;
                    v += a.addend[((size_t)(img % a.add_nmod) * hw + (m - img * hw)) * a.Cout + n];
                }
                if (a.stat_part && m < a.M) { csum[j] += (double)v; csq[j] += (double)v * (double)v; }
                if (!nok || m >= a.M) continue;
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
    if (a.stat_part) {
        __syncthreads();                                   // the ring is dead: reuse LDS for the cross-wave reduction
        double* red = reinterpret_cast<double*>(smem_raw);   // [WARPS_M][BN][2]
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const double s2 = csum[j] + __shfl_xor(csum[j], 32);   // the other 4-row groups of the same column
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
            const int img = m0 / hw;
            const int tile_in_img = (m0 - img * hw) / BM;
            const int tiles_per_img = hw / BM;
            double* o = a.stat_part + (((size_t)img * tiles_per_img + tile_in_img) * a.Cout + n0 + tid) * 2;
            o[0] = s; o[1] = q;
        }
    }
}

// OIHW -> [K/16][Npad][4 physical quads][4], physical quad p of column n holds logical quad p ^ ((n>>2)&3)
// The packed layer may take a window [cin_off, cin_off+cin_real) of the parameter's cin_total input channels
// (FuseNet's first conv is split into its source half and its shared target half).
__global__ void pack_weights_dma_kernel(const float* __restrict__ w, float* __restrict__ out,
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
