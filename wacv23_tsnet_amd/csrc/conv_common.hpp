// conv_common.hpp -- what every convolution kernel of the forward shares: the operand format, the hardware hooks, the argument block,
// the epilogue (bias, fp64 InstanceNorm partials, in-kernel finalize, fp32 store) and the weight packers.
//
// Operand format (tools/probes/split_probe.hip).  An fp32 value x scaled by a power of two s is held as two fp16
// numbers  hi = rne(x*s),  lo = rne(x*s - hi):  the residual of a round-to-nearest hi has at most 12 significant bits left and lo rounds
// away at most the last one, so hi + lo = x*s up to 2^-24 |x*s| -- the rounding class of fp32 itself.  A product a*w is the three exact
// fp16 products  lo*hi + hi*lo + hi*hi  (dropped: lo*lo <= 2^-24 |a||w|, random sign) on v_mfma_f32_32x32x16_f16, fp32 accumulate in
// two levels (chains of 64-80 k, then a running total).  The power-of-two scales (exact) keep |x*s| < 65504: activations are bounded by
// construction (an InstanceNorm output is <= sqrt(HW-1) in magnitude) or their producer publishes max |x| per image; weights are scaled
// per layer from their maximum.  The epilogue multiplies by 2^-(sa+sw) (exact) before bias, statistics and store.
// bf16-operand mode (tsnet_cfg.operand_mode = 1, BASELINE.json configs[2] / [4]): ONE bf16 plane per operand, one product, no scales.
#pragma once
#include <hip/hip_runtime.h>

namespace tsnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
struct alignas(16) F4 { float v[4]; };

constexpr unsigned kOOB = 0x80000000u;                    // voffset of a lane that must read zeros (tensors are < 2 GiB, checked on the host)
constexpr int kPatchCols = 32;                            // a patch tile is PR x 32 output pixels of one image (PR = 2 or 4 rows)
constexpr int kPatchRows = 4;

// ---- hardware hooks (tests/emu predefines these names to run the kernels on the CPU) ----
#ifndef TSNET_BUF_LOAD16
typedef __amdgpu_buffer_rsrc_t tsnet_brsrc_t;
__device__ __forceinline__ tsnet_brsrc_t tsnet_make_brsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
// 16-byte buffer load: lanes whose offset lies outside the descriptor read zeros (zero padding and ragged tiles cost no branch)
#define TSNET_BUF_LOAD16(rsrc, voff, soff) __builtin_bit_cast(F4, __builtin_amdgcn_raw_buffer_load_b128((rsrc), (int)(voff), (int)(soff), 0))
#endif
#ifndef TSNET_BUF_LOAD8
struct alignas(8) F2 { float v[2]; };
// 8-byte buffer load (four bf16 of a channel quad in the bf16-storage mode), same out-of-range rule
#define TSNET_BUF_LOAD8(rsrc, voff, soff) __builtin_bit_cast(F2, __builtin_amdgcn_raw_buffer_load_b64((rsrc), (int)(voff), (int)(soff), 0))
#endif
#ifndef TSNET_UNIFORM
#define TSNET_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif
#ifndef TSNET_WAVE_SYNC
#define TSNET_WAVE_SYNC() __builtin_amdgcn_wave_barrier()     // the lanes of a wave run in lock-step; the emulator's do not
#endif
#ifndef TSNET_DRAIN_VMEM
#define TSNET_DRAIN_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
#ifndef TSNET_MFMA_F16
typedef _Float16 tsnet_f16x8 __attribute__((ext_vector_type(8)));
#define TSNET_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(tsnet_f16x8, a), __builtin_bit_cast(tsnet_f16x8, b), c, 0, 0, 0)
#endif
#ifndef TSNET_MFMA_BF16
typedef __bf16 tsnet_bf16x8 __attribute__((ext_vector_type(8)));
#define TSNET_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tsnet_bf16x8, a), __builtin_bit_cast(tsnet_bf16x8, b), c, 0, 0, 0)
#endif
// (a, b) already scaled -> packed hi pair and packed lo pair of the fp16 x 2 split.  Three instructions for two elements:
// v_cvt_pk_f16_f32 rounds both to nearest even; v_fma_mix{lo,hi}_f16 evaluates a*1.0 - hi with the fp16 operand widened exactly (the
// difference is exact in fp32: hi is a within half an ulp_16) and rounds it to fp16 into the chosen half.  Bit-identical to
// (_Float16)(a - (float)(_Float16)a); the compiler's own lowering of that expression takes eight instructions (cvt, cvt back, sub, cvt,
// pack).  The mixhi statement opens with a wait state: mixlo wrote half of the same register (partial-write forwarding hazard).
#ifndef TSNET_SPLIT_PAIR
#define TSNET_SPLIT_PAIR(a, b, hw, lw)                                                                                              \
    do {                                                                                                                            \
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hw) : "v"(a), "v"(b));                                                             \
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(lw) : "v"(a), "v"(hw));                      \
        asm("s_nop 0\n\tv_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lw) : "v"(b), "v"(hw));           \
    } while (0)
#endif
// Two pairs at once, the six instructions interleaved so that each mixhi stands one instruction behind the mixlo of its register: the wait
// state of the hazard without an s_nop (an issue slot beside MFMA waves: conv_w1's producers).  Same bits as two TSNET_SPLIT_PAIR.
#ifndef TSNET_SPLIT_2PAIRS
#define TSNET_SPLIT_2PAIRS(a0, b0, a1, b1, h0, l0, h1, l1)                                                       \
    asm("v_cvt_pk_f16_f32 %0, %4, %5\n\t"                                                                        \
        "v_cvt_pk_f16_f32 %1, %6, %7\n\t"                                                                        \
        "v_fma_mixlo_f16 %2, %4, 1.0, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"                                  \
        "v_fma_mixlo_f16 %3, %6, 1.0, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"                                  \
        "v_fma_mixhi_f16 %2, %5, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"                                  \
        "v_fma_mixhi_f16 %3, %7, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"                                       \
        : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1)                                                              \
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1))
#endif
// The same split of (a0, b0, a1, b1) * s for a power-of-two s held in a register: v_fma_mix evaluates a * s (exact) and rounds once to fp16,
// so hi = rne16(a s) costs one instruction per VALUE where multiply + v_cvt_pk_f16_f32 cost three per pair -- eight instructions for two
// pairs instead of ten -- and lo = rne16(a s - hi) as before (the difference is exact in fp32).  Same bits as TSNET_SPLIT_2PAIRS on the
// products (tests: the emulator evaluates the C form; the GPU op tests compare against it through the fp64 reference and across kernels).
// Every mixhi stands at least one instruction behind the mixlo of its register (partial-write forwarding hazard, as above).
#ifndef TSNET_SPLIT_2PAIRS_SCALED
#define TSNET_SPLIT_2PAIRS_SCALED(a0, b0, a1, b1, s, h0, l0, h1, l1)                                              \
    asm("v_fma_mixlo_f16 %0, %4, %8, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\t"                                       \
        "v_fma_mixlo_f16 %1, %6, %8, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\t"                                       \
        "v_fma_mixhi_f16 %0, %5, %8, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\t"                                       \
        "v_fma_mixhi_f16 %1, %7, %8, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n\t"                                       \
        "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"                                     \
        "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"                                     \
        "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"                                     \
        "v_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"                                          \
        : "=&v"(h0), "=&v"(h1), "=&v"(l0), "=&v"(l1)                                                               \
        : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(s))
#endif
#ifndef TSNET_FAST_EXP
#define TSNET_FAST_EXP(x) __expf(x)
#endif
#ifndef TSNET_SETPRIO
#define TSNET_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
// The compiler may not assume it knows x beyond this point: what is computed from x stays where the source puts it (an epilogue inside a
// tile loop: hoisted out of the loop, its lane-derived addresses would live -- and spill -- across the K loop).
#ifndef TSNET_OPAQUE_V
#define TSNET_OPAQUE_V(x) asm volatile("" : "+v"(x))
#endif

// XCD-aware block -> work item: consecutive items stay on one XCD (its L2 then holds their shared operands).  Bijective for any count.
__device__ __forceinline__ int xcd_item(int bid, int nitems) {
    const int q = nitems >> 3, r = nitems & 7, xcd = bid & 7, loc = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// block -> (tile_m, tile_n).  Blocks are dealt to the XCDs round-robin (block b runs on XCD b % 8: observed, used for speed only) and every
// XCD has its own L2.  gn = 0: XCD x takes the x-th run of consecutive tiles (N fastest).  gn > 0: the XCDs form a (8 / gn) x gn grid over
// the tile matrix -- XCD (mg, ng) takes the M-tiles of block row mg and the N-tiles of block column ng, N fastest inside -- so an XCD
// streams only 1 / gn of the weights and gn / 8 of the activations through its L2 (weights x 8 / gn + activations x gn cross the fabric
// instead of weights x 8).  The host passes gn > 0 only when it divides evenly (tiles_n % gn == 0, tiles_m % (8 / gn) == 0).
__device__ __forceinline__ void tile_of_block(int bid, int tiles_m, int tiles_n, int gn, int& tile_m, int& tile_n) {
    if (gn > 0) {
        const int xcd = bid & 7, loc = bid >> 3;
        const int tn = tiles_n / gn, tm = tiles_m / (8 / gn);
        tile_m = (xcd / gn) * tm + loc / tn;
        tile_n = (xcd % gn) * tn + loc % tn;
    } else {
        const int item = xcd_item(bid, tiles_m * tiles_n);
        tile_m = item / tiles_n;
        tile_n = item - tile_m * tiles_n;
    }
}

__device__ __forceinline__ unsigned short bf16_rne(float f) {
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);     // finite inputs only (activations / weights)
}
// two floats -> one dword of two bf16 (round to nearest even), low half = a.  gfx950 has the instruction; the integer form above costs
// five VALU operations per element, and the bf16-operand kernels -- a third of the MFMA work of the fp16 x 2 ones -- are bound by exactly
// that staging arithmetic.  Same bits as bf16_rne for finite inputs (tests/emu runs the integer form).
#ifndef TSNET_CVT_PK_BF16
__device__ __forceinline__ unsigned tsnet_cvt_pk_bf16(float a, float b) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#define TSNET_CVT_PK_BF16(a, b) tsnet_cvt_pk_bf16((a), (b))
#endif

// max |v| of the workgroup's values -> ONE atomic per workgroup (atomics on one address serialise at ~12 ns each; non-negative
// floats order like their bit patterns, and a max is order-independent: deterministic).  Every thread of the workgroup must call it.
__device__ __forceinline__ void tsnet_publish_amax(unsigned* slot, float m) {
    __shared__ float wave_max[16];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(m, off); m = o > m ? o : m; }
    __syncthreads();                                     // a previous use of wave_max in this workgroup is over
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (int)((blockDim.x + 63) >> 6);
        for (int i = 1; i < nw; ++i) m = wave_max[i] > m ? wave_max[i] : m;
        __hip_atomic_fetch_max(slot, __builtin_bit_cast(unsigned, m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct ConvArgs {
    const float* x;            // (N,H,W,Csplit) fp32 NHWC
    const float* x2;           // null, or the channels >= Csplit: (x2_nmod,H,W,Cin-Csplit), image index n % x2_nmod (torch.cat formed on load)
    const float* in_alpha;     // null, or (N*Cin): x*alpha+beta on load (the producer's InstanceNorm statistics)
    const float* in_beta;
    int in_relu;
    float in_scale, in_unscale;               // 2^sa, 2^-sa from the host's a-priori bound, or:
    const unsigned* in_amax; float in_bound_add;   // in_amax[image] = float bits of max |x| published by x's producer; bound = that + in_bound_add
    const unsigned short* w;   // operand planes [NPL][K/16][Npad][2 swizzled octets][8] of w * 2^sw (fp16 hi, lo) or of w (one bf16 plane)
    const float* w_unscale;    // device scalar 2^-sw (lives in the packed weight buffer: replicas receive it with the broadcast), or null
    const float* bias;
    float* y;                  // (N,Ho,Wo,Cout) fp32 NHWC, raw convolution output
    double* stat_part;         // null, or fp64 partial (sum, sum of squares) per (tile, channel)
    const float* addend; int add_nmod;        // y += addend[img % add_nmod] (the shared half of a split convolution)
    int N, H, W, Cin, cin_log2, Csplit, x2_nmod;
    int Ho, Wo, Cout, Npad;
    int stride, pad, reflect, taps, nchunks, M;      // nchunks in units of 16 k
    int tiles_m, tiles_n, tpi;                       // tpi = tiles per image (an M tile never straddles two images)
    int xcd_gn;                                      // 0: consecutive tiles per XCD; else the 8 XCDs form a (8 / xcd_gn) x xcd_gn grid over (M, N) tiles
    // optional: the last workgroup to deliver the statistics of an (image, channel tile) finalises them itself
    // (alpha = rstd, beta = -mean*rstd), replacing the in_finalize2 launch; null = off
    float* fin_alpha; float* fin_beta; int* fin_counter; int fin_S; float fin_eps;
    unsigned* amax_out;        // null, or amax_out[image] <- max |y| of that image (float bits, atomic max)
    // bf16 STORAGE (tsnet_cfg.operand_mode = 2, bf16-operand kernels only): x / y hold bf16 instead of fp32 -- same shapes, half the bytes.
    // The statistics still come from the fp32 accumulators; the consumer widens exactly (bf16 -> fp32 is a shift).
    int x_bf16, y_bf16;
    // conv_w1 only: tiles per workgroup (a chunk of consecutive tiles, the next tile's first periods produced under the current one's last:
    // conv_w1.hpp), and whether the LDS holds two transform tables (a chunk may then cross from one image into the next)
    int w1_chunk, w1_tab2;
};

// eight consecutive bf16 channels (one 16-byte vector) -> two float4, exactly
__device__ __forceinline__ void bf16x8_widen(const F4& p, F4 (&o)[2]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned w = __builtin_bit_cast(unsigned, p.v[e]);
        o[e >> 1].v[(e & 1) * 2] = __builtin_bit_cast(float, w << 16);
        o[e >> 1].v[(e & 1) * 2 + 1] = __builtin_bit_cast(float, w & 0xFFFF0000u);
    }
}

// The eight channels at fp32 byte offset (voff + soff) of an activation tensor -- or, xb16, of the same tensor stored as bf16 (half the
// offsets, one load).  A lane whose voff is kOOB reads zeros either way (kOOB / 2 still lies past any descriptor: tensors are < 2 GiB).
__device__ __forceinline__ void load_x_octet(const tsnet_brsrc_t& rs, bool xb16, unsigned voff, unsigned soff, F4 (&o)[2]) {
    if (xb16) {
        const F4 p = TSNET_BUF_LOAD16(rs, voff >> 1, soff >> 1);
        bf16x8_widen(p, o);
    } else {
        o[0] = TSNET_BUF_LOAD16(rs, voff, soff);
        o[1] = TSNET_BUF_LOAD16(rs, voff, soff + 16u);
    }
}

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

// x (already scaled) -> (hi, lo) fp16 bit patterns (scalar form: weight packing)
__device__ __forceinline__ void split_h2(float v, unsigned& hi, unsigned& lo) {
    const _Float16 h = (_Float16)v;                              // round to nearest even
    const _Float16 l = (_Float16)(v - (float)h);                 // exact residual, then RNE
    hi = (unsigned)__builtin_bit_cast(unsigned short, h);
    lo = (unsigned)__builtin_bit_cast(unsigned short, l);
}

// eight consecutive channels (two float4, already scaled) -> one 16-byte octet per plane
__device__ __forceinline__ void split_h2_octet(const F4& x0, const F4& x1, F4& H, F4& L) {
    unsigned hw[4], lw[4];
    // two pairs per statement: six instructions, no wait state (the per-pair form costs an s_nop each -- four issue slots per octet in the
    // staging code of every MFMA wave: 20 - 38 s_nop per K-loop iteration of conv_h2 / conv_h2d, tools/isa_mix.py)
    TSNET_SPLIT_2PAIRS(x0.v[0], x0.v[1], x0.v[2], x0.v[3], hw[0], lw[0], hw[1], lw[1]);
    TSNET_SPLIT_2PAIRS(x1.v[0], x1.v[1], x1.v[2], x1.v[3], hw[2], lw[2], hw[3], lw[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) { H.v[e] = __builtin_bit_cast(float, hw[e]); L.v[e] = __builtin_bit_cast(float, lw[e]); }
}

// eight consecutive channels -> one octet of bf16 (round to nearest even): the bf16-operand mode's single plane
__device__ __forceinline__ void bf16_octet(const F4& x0, const F4& x1, F4& H) {
    unsigned hw[4];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        hw[e] = TSNET_CVT_PK_BF16(x0.v[2 * e], x0.v[2 * e + 1]);
        hw[2 + e] = TSNET_CVT_PK_BF16(x1.v[2 * e], x1.v[2 * e + 1]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) H.v[e] = __builtin_bit_cast(float, hw[e]);
}

// The input transform of a staged octet: t = x*alpha + beta (table pre-multiplied by the operand scale: a power of two, exact, and
// fl(x*(al*s) + be*s) == s * fl(x*al + be)), ReLU, zero padding re-imposed AFTER the transform (a padded pixel is zero in the conv's
// input, not beta: keep = 0 there, 1 elsewhere); without a producer InstanceNorm: t = x*s (a padded slot loaded zeros).
// Branch-free on purpose (relu_floor = 0 or -inf): a wave-uniform branch here would cut the unrolled K loop into basic blocks, and the
// register allocator then spills across them.  KEEP = false: the layer pads by reflection (every staged slot is a real pixel; slots
// past the patch are never read), so the multiply is dropped -- x * 1.0f is exact, the results are the same bits either way.
template <bool AFFINE, bool KEEP = true>
__device__ __forceinline__ void transform_octet(const F4 (&sx)[2], const float* ta, int Cin, float in_scale, float relu_floor, float keep, F4 (&t)[2]) {
    if (AFFINE) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const F4 al = *reinterpret_cast<const F4*>(ta + q * 4), be = *reinterpret_cast<const F4*>(ta + Cin + q * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = __builtin_fmaxf(__builtin_fmaf(sx[q].v[e], al.v[e], be.v[e]), relu_floor);
                t[q].v[e] = KEEP ? v * keep : v;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) t[q].v[e] = __builtin_fmaxf(sx[q].v[e] * in_scale, relu_floor);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Shared epilogue: bias, optional per-pixel addend, fp64 InstanceNorm partial sums of the tile (fixed order: deterministic), fp32 store (last).
// m_of(l) maps the local row l of the tile to the output position m (or -1: a row past the end of the image).
// STAT_WAVE: the wave that reduces the tile's statistics partials and hands them off (BN <= 64: one wave does it alone).  conv_w1 names a
// producer wave -- idle in the epilogue -- so that no MFMA wave waits for the store drain and the arrival counter's round trip.
template <int BN, int WARPS_M, int WARPS_N, int MT, int NTL, int STAT_WAVE = 0, typename MOf>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&tot)[MT][NTL], unsigned char* smem_raw, int tid, int wave, int n0,
                                              size_t stat_tile, MOf m_of, const bool active = true) {
    constexpr int WM = MT * 32, WN = NTL * 32;
    const int lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wm0 = (wave / WARPS_N) * WM, wn0 = (wave % WARPS_N) * WN;
    const int hw = a.Ho * a.Wo;
    double csum[NTL], csq[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) { csum[j] = 0.0; csq[j] = 0.0; }
    float vmax = 0.f;
    // pass 1: the values (bias, addend) in place, their statistics and maximum.  The stores come LAST (pass 2 below): the statistics
    // hand-off drains this wave's outstanding stores before the workgroup counts itself, and with the tile's 32 - 64 KiB of output in
    // flight that drain was the longest wait of the epilogue -- exposed in full where one workgroup owns the CU (conv_w1).
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
                const bool mok = active && m >= 0;
                float v = tot[i][j][r] + bv;
                if (a.addend && nok && mok) {
                    const int img = m / hw;
                    v += a.addend[((size_t)(img % a.add_nmod) * hw + (m - img * hw)) * a.Cout + n];
                }
                tot[i][j][r] = v;
                if (a.stat_part && mok) { csum[j] += (double)v; csq[j] += (double)v * (double)v; }
                if (nok && mok) vmax = __builtin_fmaxf(vmax, __builtin_fabsf(v));
            }
        }
    }
    if (a.amax_out) tsnet_publish_amax(a.amax_out + stat_tile / (size_t)a.tpi, vmax);        // a tile lies inside one image
    if (a.stat_part) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem_raw);
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const double s2 = csum[j] + __shfl_xor(csum[j], 32);
            const double q2 = csq[j] + __shfl_xor(csq[j], 32);
            if (lh == 0 && active) {
                double* o = red + ((size_t)(wave / WARPS_N) * BN + wn0 + j * 32 + li) * 2;
                o[0] = s2; o[1] = q2;
            }
        }
        __syncthreads();
        const int stid = tid - 64 * STAT_WAVE;                   // the reducing threads count from the statistics wave
        if (stid >= 0 && stid < BN && n0 + stid < a.Cout) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int wmi = 0; wmi < WARPS_M; ++wmi) { s += red[((size_t)wmi * BN + stid) * 2]; q += red[((size_t)wmi * BN + stid) * 2 + 1]; }
            double* o = a.stat_part + (stat_tile * a.Cout + n0 + stid) * 2;
            if (a.fin_counter) {            // device-scope write-through: another XCD's workgroup may read them
                __hip_atomic_store(o, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(o + 1, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                o[0] = s; o[1] = q;
            }
        }
        if (a.fin_counter) {
            // Arrival counter per (image, channel tile).  Hand-off form (MI355X_MICROARCH.md, "valid forms"): 8-byte agent-scope atomics
            // on BOTH sides (write-through sc1 stores above, sc1 loads below: never served from a stale L1 / another XCD's L2), every
            // storing wave's stores drained before the workgroup counts itself.  The drain is inline asm on purpose: the compiler may drop
            // a builtin s_waitcnt it can prove redundant.  The workgroup that reads fin_S - 1 sums the partials in a fixed order
            // (four interleaved groups, then g0+g1+g2+g3): run-to-run deterministic.
            const int S = a.fin_S;
            const int img = (int)(stat_tile / (size_t)S);
            const int ntn = (a.Npad + 31) / 32;                                                    // 32 = narrowest tile
            int* counter = a.fin_counter + (size_t)img * ntn + n0 / 32;
            const bool mine = stid >= 0 && stid < BN && n0 + stid < a.Cout;
            // Entries in order t = 0 .. cnt-1 into four interleaved sums (t & 3), then g0 + g1 + g2 + g3.  The loads of EIGHT entries are issued
            // before the first addition: a loop of one load pair per iteration compiles to a memory round trip per entry (s_waitcnt vmcnt(0)
            // before every v_add_f64 -- 8 to 32 serial round trips in the last workgroup of every launch up to round 5).  Entries past cnt
            // re-read the last one and contribute +0.0 (x + 0.0 == x): the association does not depend on the batching.
            auto fold = [&](const double* p0, int cnt, size_t step, double& sm, double& sq) __attribute__((always_inline)) {
                double gs[4] = {0, 0, 0, 0}, gq[4] = {0, 0, 0, 0};
                for (int t0 = 0; t0 < cnt; t0 += 8) {
                    double v[8], w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const double* p = p0 + (size_t)(t0 + u < cnt ? t0 + u : cnt - 1) * step;
                        v[u] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        w[u] = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const bool in = t0 + u < cnt;
                        gs[u & 3] += in ? v[u] : 0.0;
                        gq[u & 3] += in ? w[u] : 0.0;
                    }
                }
                sm = gs[0]; sq = gq[0];
#pragma unroll
                for (int k = 1; k < 4; ++k) { sm += gs[k]; sq += gq[k]; }
            };
            auto write_ab = [&](double sm, double sq) __attribute__((always_inline)) {
                const double mean = sm / hw;
                double var = sq / hw - mean * mean;
                if (var < 0) var = 0;
                const float al = 1.0f / sqrtf((float)var + a.fin_eps);
                a.fin_alpha[(size_t)img * a.Cout + n0 + stid] = al;
                a.fin_beta[(size_t)img * a.Cout + n0 + stid] = -((float)mean) * al;
            };
            auto finalize = [&]() __attribute__((always_inline)) {                                 // all S tile partials of the image
                if (mine) {
                    double sm, sq;
                    fold(a.stat_part + (((size_t)img * S) * a.Cout + n0 + stid) * 2, S, (size_t)a.Cout * 2, sm, sq);
                    write_ab(sm, sq);
                }
                if (stid == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            };
            if (BN <= 64) {
                // One wave stored every partial of this tile (tid < BN): it drains its own stores, counts the workgroup and -- if it is the
                // last to arrive -- finalises, all in program order and WITHOUT a workgroup barrier; the other waves go straight on to their
                // output stores.  (On a CU that one workgroup owns -- conv_w1 -- nothing else hides the counter's round trip.)
                if (stid >= 0 && stid < 64) {
                    auto arrive = [&](int* c) __attribute__((always_inline)) {
                        TSNET_DRAIN_VMEM();
                        float arrived = 0.f;
                        if (lane == 0) arrived = (float)__hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int off = 32; off > 0; off >>= 1) arrived += __shfl_xor(arrived, off);      // lane 0's value to every lane (the others hold 0)
                        return (int)arrived;
                    };
                    if (arrive(counter) == S - 1) finalize();
                }
            } else {
                int* flag = reinterpret_cast<int*>(smem_raw + 8192);
                auto arrive = [&](int* c) __attribute__((always_inline)) {                         // every thread of the workgroup calls it
                    TSNET_DRAIN_VMEM();
                    __syncthreads();
                    if (stid == 0) *flag = __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __syncthreads();
                    return *flag;
                };
                if (arrive(counter) == S - 1) finalize();                                         // workgroup-uniform
            }
        }
    }
    // pass 2: the stores (fp32, or bf16 in the bf16-storage mode)
    unsigned short* const y16 = reinterpret_cast<unsigned short*>(a.y);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) {
            const int n = n0 + wn0 + j * 32 + li;
            if (n >= a.Cout) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_of(wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
                if (!(active && m >= 0)) continue;
                if (a.y_bf16) y16[(size_t)m * a.Cout + n] = bf16_rne(tot[i][j][r]);
                else a.y[(size_t)m * a.Cout + n] = tot[i][j][r];
            }
        }
    }
}

}  // namespace tsnet
