// flow_persist.hpp -- flow_kernel_p: the correlation -> softmax(100*) -> soft-argmax sweep of flow_warp.hpp for LARGE maps.
//
// Replaces (model/TSNet.py:336-365, the loop over sources): the two masked torch.bmm, F.softmax(100*., dim=2), get_grid + torch.matmul, for
// all K sources of a driving frame in one launch.  BASELINE.json configs[4] per GPU: one driving frame, five sources, 64 x 64 positions.
//
// Why a second kernel.  flow_kernel gives every (source, batch element, 64 targets) its own workgroup: at configs[4] that is 320 workgroups
// of ~170 us on 256 CUs -- two rounds, the second a quarter full (346 us, measured) -- and every workgroup fetches its 128 KiB of target
// planes for ONE source image.  Here a workgroup KEEPS its 64 target positions (128 KiB of LDS) for all K sources of its batch element and
// sweeps 1 / G of every source image, G = 256 / target tiles (a power of two, flow_args.hpp flowp_plan): tiles x G workgroups = one per CU,
// all finishing together (249 us).  The 32 workgroups of an XCD share g and the batch element (items (g, b, t), t fastest, dealt to the
// XCDs in runs): they take the same source slice at the same time -- 2 MiB at configs[4], resident in that XCD's L2.
//
//   per source slice   each wave sweeps its source pairs exactly as in flow_kernel (source fragments global -> registers two groups ahead,
//                      target fragments from LDS, three fp16 products per step on v_mfma_f32_32x32x16_f16); the 64 mask values of a pair
//                      are fetched by the wave itself (one per lane, under the MFMA sweep) into its own LDS row -- no workgroup-wide mask
//                      table, so any slice length fits.  The epilogue reads mask and grid row four sources at a time and needs no range
//                      selects (P % 64 == 0, w % 4 == 0 are conditions of this form).
//   end of a slice     the two half-waves of a column merge by shuffle, the eight waves through LDS (one barrier per slice, two buffers),
//                      wave 0 merges in wave order.  G == 1: that is the flow.  G > 1: the state goes to global memory as two 8-byte
//                      agent-scope atomic stores per column.
//   end of the kernel  G > 1: wave 0 drains its stores and counts the workgroup on an arrival counter per target tile (conv_epilogue's
//                      hand-off: 8-byte agent-scope atomics on both sides); the last of the G workgroups merges, for every source, the G
//                      partial states in the order g = 0 .. G-1 -- run-to-run deterministic -- and resets the counter.
//
// THIS FILE IS COMPILED WITHOUT THE SLP VECTORIZER (build.py UNIT_FLAGS, tests/test_isa.py).  With it the epilogue becomes packed fp32
// arithmetic (96 v_pk_fma_f32, 114 v_pk_mul_f32, 79 v_pk_add_f32 beside the other wave's MFMAs) and the kernel returns, run to run, different
// and wrong flows for 1-2 % of the target columns 16..31 of a tile -- always the LOW half of a packed pair, lanes 16..31 and 48..63; the
// contributions of four consecutive sources missing, or their grid coordinate wrong (profiles/round4_flow_cfg4.txt).  Same LDS and memory
// instructions in both builds; the scalar build is exact run after run.  Not isolated further (a hazard the compiler does not know, or the
// hardware's): the MFMA kernels of this library do not use packed fp32 arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_common.hpp"
#include "flow_args.hpp"

namespace tsnet {

__device__ __forceinline__ unsigned long long flow_pack2(float a, float b) {
    return (unsigned long long)__builtin_bit_cast(unsigned, a) | ((unsigned long long)__builtin_bit_cast(unsigned, b) << 32);
}
__device__ __forceinline__ void flow_unpack2(unsigned long long v, float& a, float& b) {
    a = __builtin_bit_cast(float, (unsigned)(v & 0xffffffffull));
    b = __builtin_bit_cast(float, (unsigned)(v >> 32));
}

// grid = B * (P / 64) * G (1-D), block = 512.  OPT (tools): bit 1 = skip the exp / accumulate pass (ablation: computes garbage).
template <int OPT>
__global__ __launch_bounds__(64 * kFlowWaves) void flow_kernel_p(FlowArgs a) {
    constexpr int NT = 2, NTH = 64 * kFlowWaves;
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int KC = (a.C + 31) / 32 * 2;
    const int TBYTES = NT * KC * 2048;
    const int npair = a.P >> 6, nps = npair / a.S, spg = a.S / a.G, tiles = npair;       // pairs per slice; slices per workgroup
    F4* sRed = reinterpret_cast<F4*>(smem_raw + TBYTES);                                   // [2][kFlowWaves][64]
    float* sMw = reinterpret_cast<float*>(smem_raw + TBYTES + 2 * kFlowWaves * 64 * 16);   // [kFlowWaves][64]
    float* sGx = sMw + kFlowWaves * 64;
    float* sGy = sGx + ((a.w + 3) & ~3);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int item = xcd_item(blockIdx.x, (int)gridDim.x);
    const int t = item % tiles, gb = item / tiles;
    const int b = gb % a.B, g = gb / a.B;
    const int tb0 = t * NT;

    {   // the workgroup's target fragments: one contiguous region of the plane buffer
        const F4* gp = reinterpret_cast<const F4*>(a.tq + ((size_t)(b * (a.P >> 5) + tb0) * KC) * 1024);
        F4* d = reinterpret_cast<F4*>(smem_raw);
        const int cnt = TBYTES / 16;
        for (int i0 = 0; i0 < cnt; i0 += 8 * NTH) {
            F4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * NTH + tid; v[u] = gp[i < cnt ? i : tid]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * NTH + tid; if (i < cnt) d[i] = v[u]; }
        }
    }
    for (int p = tid; p < a.w; p += NTH) sGx[p] = a.gx[p];
    for (int p = tid; p < a.h; p += NTH) sGy[p] = a.gy[p];
    __syncthreads();

    // mask factor x un-scale per target column: mf * 2^-28 = fma(mt * 2^-28, ms, ((1 - mt) * 2^-28) * (1 - ms)) -- powers of two commute with rounding
    float mtu[NT], omtu[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int tt = (tb0 + j) * 32 + li;
        const int ty = tt / a.w, tx = tt - ty * a.w;
        const float m = a.tar_bbox[(size_t)b * a.H * a.W + (size_t)(ty * a.sy) * a.W + tx * a.sx];
        mtu[j] = m * kFlowUnscale; omtu[j] = (1.0f - m) * kFlowUnscale;
    }
    const float inv_w = 1.0f / (float)a.w;
    const unsigned char* tbase = smem_raw + lane * 16;
    float* ms_w = sMw + wave * 64;

    // (source, slice) items: every slice is a complete reduction of its own -- waves in wave order -- whatever G is
    for (int it = 0; it < a.K * spg; ++it) {
        const int s_idx = it / spg, slice = g * spg + (it - s_idx * spg);
        const int n = s_idx * a.B + b;
        const int buf = it & 1;
        const int sp_base = slice * nps;
        const float* sb = a.src_bbox[s_idx] + (size_t)b * a.H * a.W;
        float m_run[NT], l_run[NT], ax[NT], ay[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) { m_run[j] = -3.0e38f; l_run[j] = 0.f; ax[j] = 0.f; ay[j] = 0.f; }
        const unsigned char* sbase = reinterpret_cast<const unsigned char*>(a.sq) + ((size_t)n * (a.P >> 5) * KC) * 2048 + lane * 16;
        for (int sp = sp_base + wave; sp < sp_base + nps; sp += kFlowWaves) {
            // this lane's share of the pair's source mask (F.interpolate(nearest)): source sp * 64 + lane; lands under the MFMA sweep
            float msl;
            {
                const int p = sp * 64 + lane;
                const int py = p / a.w, px = p - py * a.w;
                msl = sb[(size_t)(py * a.sy) * a.W + px * a.sx];
            }
            const unsigned char* ap = sbase + (size_t)(sp * 2) * KC * 2048;     // source blocks 2 sp, 2 sp + 1: KC * 2 KiB each
            f32x16 acc[2][NT];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            F4 af[3][2][2][2];                                                  // [set][step of the group][source block][plane]
            F4 bf[2][NT][2];                                                    // [set][target block][plane]
            auto load_a = [&](int set, int gg) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl)
                            af[set][u][i][pl] = *reinterpret_cast<const F4*>(ap + ((size_t)(i * KC + gg * 2 + u) * 2 + pl) * 1024);
            };
            auto load_b = [&](int set, int kc) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) bf[set][j][pl] = *reinterpret_cast<const F4*>(tbase + ((j * KC + kc) * 2 + pl) * 1024);
            };
            auto mfmas = [&](int sa, int u, int sbt) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        acc[i][j] = TSNET_MFMA_F16(af[sa][u][i][1], bf[sbt][j][0], acc[i][j]);      // lo * hi
                        acc[i][j] = TSNET_MFMA_F16(af[sa][u][i][0], bf[sbt][j][1], acc[i][j]);      // hi * lo
                        acc[i][j] = TSNET_MFMA_F16(af[sa][u][i][0], bf[sbt][j][0], acc[i][j]);      // hi * hi
                    }
            };
            // groups of two steps; source fragments two groups ahead (three register sets), as in flow_kernel
            const int ngrp = KC >> 1;
            load_a(0, 0);
            load_a(1, ngrp > 1 ? 1 : 0);
            load_b(0, 0);
            // (no sched_barrier here, unlike conv_h2: pinning the eight source loads of a group to its top costs 7 % -- 266 against 249 us.
            // The sweep does not wait on load latency; what it waits on is the rate at which 8 KiB per wave and group arrive, and the
            // scheduler's own placement spreads those loads between the MFMAs)
            auto group = [&](int gg, int S) __attribute__((always_inline)) {
                load_a((S + 2) % 3, gg + 2 < ngrp ? gg + 2 : gg);               // past the end: re-reads a valid group (unused)
                load_b(1, 2 * gg + 1);
                mfmas(S, 0, 0);
                load_b(0, 2 * gg + 2 < KC ? 2 * gg + 2 : 0);
                mfmas(S, 1, 1);
            };
            int gg = 0;
            for (; gg + 3 <= ngrp; gg += 3) { group(gg, 0); group(gg + 1, 1); group(gg + 2, 2); }
            if (gg < ngrp) group(gg, 0);
            if (gg + 1 < ngrp) group(gg + 1, 1);
#if defined(TSNET_FLOWP_PROBE) && (TSNET_FLOWP_PROBE & 1)
            // tools/probes/slp_probe_*.py: idle states between the last MFMA and the first VALU read of its accumulators
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#endif

            ms_w[lane] = msl;                 // the wave's own row: the LDS operations of one wave execute in order
            TSNET_WAVE_SYNC();
#if defined(TSNET_FLOWP_PROBE) && (TSNET_FLOWP_PROBE & 2)
            // ... or: everything outstanding (LDS, memory) landed, then idle states, before the packed arithmetic of the epilogue
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#endif
            // D[row = source][col = target]: lane owns target li of each block, source rows i*32 + 8*q + e + 4*lh (r = 4 q + e): four
            // consecutive sources per (i, q)
            float mx[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) mx[j] = -3.0e38f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const F4 ms4 = *reinterpret_cast<const F4*>(ms_w + 4 * lh + i * 32 + 8 * q);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ms = ms4.v[e], oms = 1.0f - ms;
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const float mfu = __builtin_fmaf(mtu[j], ms, omtu[j] * oms);
                            const float lg = 100.0f * (acc[i][j][q * 4 + e] * mfu);
                            acc[i][j][q * 4 + e] = lg;
                            mx[j] = lg > mx[j] ? lg : mx[j];
                        }
                    }
                }
            TSNET_WAVE_SYNC();                // (emulator: every lane has read the row before the next pair overwrites it)
            float m_new[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                m_new[j] = mx[j] > m_run[j] ? mx[j] : m_run[j];
                const float sc = expf(m_run[j] - m_new[j]);
                l_run[j] *= sc; ax[j] *= sc; ay[j] *= sc;
                m_run[j] = m_new[j];
            }
            if (!(OPT & 2)) {
                const int sg0 = sp * 64 + 4 * lh;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int sg = sg0 + i * 32 + 8 * q;                       // a multiple of 4: the four sources share a grid row (w % 4 == 0)
                        const int py = (int)(((float)sg + 0.5f) * inv_w), px = sg - py * a.w;      // exact for sg < 2^20
                        const float gyv = sGy[py];
                        const F4 gx4 = *reinterpret_cast<const F4*>(sGx + px);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int j = 0; j < NT; ++j) {
                                const float ex = TSNET_FAST_EXP(acc[i][j][q * 4 + e] - m_new[j]);   // v_exp_f32: see flow_kernel
                                l_run[j] += ex;
                                ax[j] = __builtin_fmaf(ex, gx4.v[e], ax[j]);
                                ay[j] = __builtin_fmaf(ex, gyv, ay[j]);
                            }
                    }
            } else {
#pragma unroll
                for (int j = 0; j < NT; ++j) { l_run[j] += 1.0f; ax[j] += acc[0][j][0]; ay[j] += acc[1][j][0]; }
            }
        }
        // the two half-waves of a column (lower half first), then the eight waves in order
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float mo = __shfl_xor(m_run[j], 32), lo = __shfl_xor(l_run[j], 32), xo = __shfl_xor(ax[j], 32), yo = __shfl_xor(ay[j], 32);
            const float m0 = lh ? mo : m_run[j], l0 = lh ? lo : l_run[j], x0 = lh ? xo : ax[j], y0 = lh ? yo : ay[j];
            const float m1 = lh ? m_run[j] : mo, l1 = lh ? l_run[j] : lo, x1 = lh ? ax[j] : xo, y1 = lh ? ay[j] : yo;
            const float M = m0 > m1 ? m0 : m1;
            const float s0 = expf(m0 - M), s1 = expf(m1 - M);
            F4 o;
            o.v[0] = M; o.v[1] = l0 * s0 + l1 * s1; o.v[2] = x0 * s0 + x1 * s1; o.v[3] = y0 * s0 + y1 * s1;
            if (lh == 0) sRed[(buf * kFlowWaves + wave) * 64 + j * 32 + li] = o;
        }
        __syncthreads();                      // merge buffer `buf` complete; wave 0 reads it while the others sweep the next slice (buffer buf ^ 1)
        if (tid < 64) {
            float M = -3.0e38f;
            for (int wv = 0; wv < kFlowWaves; ++wv) { const float v = sRed[(buf * kFlowWaves + wv) * 64 + tid].v[0]; M = v > M ? v : M; }
            float L = 0.f, X = 0.f, Y = 0.f;
            for (int wv = 0; wv < kFlowWaves; ++wv) {
                const F4 pr = sRed[(buf * kFlowWaves + wv) * 64 + tid];
                const float sc = expf(pr.v[0] - M);
                L += pr.v[1] * sc; X += pr.v[2] * sc; Y += pr.v[3] * sc;
            }
            // the slice's state, device-scope write-through: the workgroup that merges may sit on another XCD
            unsigned long long* o = a.part + ((((size_t)n * tiles + t) * a.S + slice) * 64 + tid) * 2;
            __hip_atomic_store(o, flow_pack2(M, L), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o + 1, flow_pack2(X, Y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid < 64) {
        // every slice state this workgroup owes is stored and drained; with G > 1 the workgroups of a target tile count themselves and the
        // last one merges (conv_epilogue's hand-off: 8-byte agent-scope atomics on both sides).  The merge runs over the S slices in slice
        // order, whoever computed them: the same association for every G
        TSNET_DRAIN_VMEM();
        int* counter = a.cnt + b * tiles + t;
        float arrived = (float)(a.G - 1);
        if (a.G > 1) {
            arrived = 0.f;
            if (lane == 0) arrived = (float)__hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) arrived += __shfl_xor(arrived, off);
        }
        if ((int)arrived == a.G - 1) {
            for (int s_idx = 0; s_idx < a.K; ++s_idx) {
                const int n = s_idx * a.B + b;
                const unsigned long long* pp = a.part + (((size_t)n * tiles + t) * a.S * 64 + tid) * 2;
                float M = -3.0e38f;
                for (int q = 0; q < a.S; ++q) {
                    float pm, pl;
                    flow_unpack2(__hip_atomic_load(pp + (size_t)q * 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), pm, pl);
                    M = pm > M ? pm : M;
                }
                float L = 0.f, X = 0.f, Y = 0.f;
                for (int q = 0; q < a.S; ++q) {
                    float pm, pl, px, py;
                    flow_unpack2(__hip_atomic_load(pp + (size_t)q * 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), pm, pl);
                    flow_unpack2(__hip_atomic_load(pp + (size_t)q * 128 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), px, py);
                    const float sc = expf(pm - M);
                    L += pl * sc; X += px * sc; Y += py * sc;
                }
                float* f = a.flow + ((size_t)n * a.P + t * 64 + tid) * 2;
                f[0] = X / L;
                f[1] = Y / L;
            }
            if (a.G > 1 && lane == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace tsnet
