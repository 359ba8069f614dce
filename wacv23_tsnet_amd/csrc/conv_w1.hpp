// conv_w1.hpp -- the 3 x 3 / stride-1 / pad-1 convolution (ResnetBlocks, FuseNet, decoder up-convolutions: TSNet.py:27,42,147; 83 % of the
// forward's FLOPs) with the Winograd F(2, 3) transform ALONG X: two adjacent output pixels of a row are computed from four transformed
// input columns with FOUR products per tap row instead of six -- two thirds of the MFMA work of the direct form (conv_h2.hpp), which is
// what bounds these layers (three fp16 MFMA products per fp32 product on a power-limited matrix pipe).
//
//   V_p[y][j] = sum_i Bt[p][i] d[y][2j-1+i]   p = 0..3:  d0-d2,  d1+d2,  d2-d1,  d1-d3        input transform  (one add / value, fp32)
//   U_p[ky]   = sum_kx G[p][kx] g[ky][kx]              g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2     filter transform (host, fp64, at pack time)
//   M_p[y][j] = sum_ky sum_c V_p[y+ky][j][c] U_p[ky][c]                                       4 GEMMs, K = 3 Cin: the MFMA work
//   out[y][2j] = M0 + M1 + M2,   out[y][2j+1] = M1 - M2 - M3                                  output transform (epilogue, fp32)
//
// Numerics: the transform acts on the fp32 activation AFTER the producer's InstanceNorm + ReLU and BEFORE the fp16 x 2 split (sums of fp16
// numbers need more than fp16's bits), U is rounded once to fp32 and split like any weight; products, chains (two 16-channel slabs) and
// the running total are conv_h2's.  tools/probes/winograd_probe.py: error against fp64 0.96 x the direct kernel's at the ResnetBlock shape
// (the 2-D transform: 1.11 x; both inside the 1.5 x line).
//
// A tile = 4 x 32 output pixels (64 column pairs) x 64 output channels, FOURTEEN waves, one workgroup per CU (768 tiles on the ResnetBlock
// layers at the headline batch = three rounds; the 192 tiles of a single frame = one):
//   * eight CONSUMERS (waves 0..7): wave = (position p, 32-channel half), wave tile 64 pairs x 32 channels of ONE position, acc + total =
//     64 VGPRs.  The K loop runs in PERIODS of two 16-channel slabs (six steps, one barrier, one accumulation chain); an A fragment = two
//     consecutive V rows (32 pairs) of the wave's position: rows (ky, ky+1) and (ky+2, ky+3) for the two 32-pair halves -- five distinct
//     fragments serve the six (half, ky) uses of a slab; weight fragments two steps ahead (three register sets); s_setprio 2;
//   * six PRODUCERS (waves 8..13) turn the input patch into V: a period's V has 12 wave-sized items (input row, half of the pairs), two per
//     producer; lane = (pair, channel quad): the 8 lanes of a pixel read its 32 channels as ONE 128-byte line (a lane per (pixel, octet)
//     touched 32 - 64 lines per load, and the texture path -- shared with the weight fragments -- bound the first forms of this kernel);
//     four input pixels per lane, fetched one item ahead (reflection / zero padding in the offsets); IN + ReLU; per position one add, the
//     split, one ds_write_b64 per plane;
//   * V lives in LDS, three stages of [slab][position][plane][octet][6 rows x 16 pairs] x 16 B (50 KiB each, region pitches padded so
//     that a producer's 16-lane write group spans all banks): period s is consumed while period s+2 is produced, so the first A fragments
//     of period s+1 are fetched BEFORE the barrier (no exposed LDS latency at a period's start);
//   * epilogue: the four positions of a pair meet through LDS (64 KiB over the V stages), consumer (row, half) forms its 32 pixels x 32
//     channels and runs the shared conv_epilogue (bias, addend, fp64 statistics, in-kernel finalize, amax); the producers keep the
//     barriers company;
//   * <= 128 VGPRs (two SIMDs carry four of the fourteen waves; tests/test_isa.py), 154 - 158 KiB of LDS.
// How it got here, form by form with measurements: DESIGN.md section 4.4.
#pragma once
#include <type_traits>

#include "conv_common.hpp"

namespace tsnet {

constexpr int kW1Prod = 6;                                                        // producer waves (transform): two items of a period each
constexpr int kW1Waves = 8 + kW1Prod;                                             // eight consumers (MFMA) + the producers
constexpr int kW1Reg = 96 * 16 + 64;                                              // one (slab, position, plane, octet) region of V + its bank pad
constexpr int kW1Stage(int npl) { return 2 * (4 * npl * 2 * kW1Reg + 32); }       // bytes of one V stage: two slabs
constexpr int w1_lds_bytes(int Cin, int npl = 2, int tables = 1) {
    const int v = 3 * kW1Stage(npl), ex = 4 * 64 * 64 * 4 + 2048;                  // V stages | output exchange
    return (v > ex ? v : ex) + tables * 2 * ((Cin + 31) / 32 * 32) * 4;           // + the transform table(s): whole periods of 32 channels
}

#ifdef TSNET_TOOLS
// tools build: per-wave time stamps of one launch (OPT bit 9), read back by tsnet_w1_prof_read (conv_w1_launch.cpp); 16 slots per wave
constexpr int kW1ProfSlots = 16, kW1ProfTiles = 1024;
static __device__ unsigned long long g_w1_prof[kW1ProfTiles * kW1Waves * kW1ProfSlots];
#define TSNET_W1_STAMP(slot)                                                                                                   \
    do {                                                                                                                        \
        if ((OPT & 512) && (threadIdx.x & 63) == 0 && blockIdx.x < kW1ProfTiles)                                                \
            g_w1_prof[((size_t)blockIdx.x * kW1Waves + (threadIdx.x >> 6)) * kW1ProfSlots + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
// the K loop's barrier, with the time this wave spent waiting at it summed into `acc_` (tools build, OPT bit 9)
#define TSNET_W1_LOOP_BARRIER(acc_)                                                        \
    do {                                                                                    \
        if (OPT & 512) {                                                                    \
            const unsigned long long t0_ = __builtin_amdgcn_s_memrealtime();                \
            __syncthreads();                                                                \
            acc_ += __builtin_amdgcn_s_memrealtime() - t0_;                                 \
        } else {                                                                            \
            __syncthreads();                                                                \
        }                                                                                   \
    } while (0)
#define TSNET_W1_PUT(slot, v)                                                                                                   \
    do {                                                                                                                        \
        if ((OPT & 512) && (threadIdx.x & 63) == 0 && blockIdx.x < kW1ProfTiles)                                                \
            g_w1_prof[((size_t)blockIdx.x * kW1Waves + (threadIdx.x >> 6)) * kW1ProfSlots + (slot)] = (v);                      \
    } while (0)
#else
#define TSNET_W1_STAMP(slot) do { } while (0)
#define TSNET_W1_LOOP_BARRIER(acc_) __syncthreads()
#define TSNET_W1_PUT(slot, v) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------
// THE ARITHMETIC of the kernel, written once: the chunk path below and the one-tile path (conv_w1_one.hpp) differ in how a workgroup is
// scheduled -- prologue, tiles per workgroup, where the next fetch comes from -- and share these three pieces, so a change to the
// transform, the split, the product order or the output transform is made in one place (the two kernels used to carry a copy each).
//
// (1) Producer: four input pixels x four channels of one column pair -> the four Winograd positions, split, one ds_write_b64 per plane.
//     b[q] = pixel q of the pair as fetched (zeros for a padded pixel); pad[q] = that pixel lies in the ZERO padding of an InstanceNorm-ed
//     input (re-zeroed after the affine transform: a padded pixel is zero, not beta); ta = the image's table at the item's first channel
//     (alpha * s, then beta * s at + cp32); scale = the operand scale s of a raw input.
//     A raw input WITHOUT ReLU (the residual stream, the upsampled maps: NO_RELU) is transformed unscaled and scaled inside the split --
//     s is a power of two, (d0 - d2) s == d0 s - d2 s exactly, and v_fma_mix takes the factor as an operand (TSNET_SPLIT_2PAIRS_SCALED):
//     16 multiplies fewer per item in the waves whose instruction count bounds the K loop (60 -> 52 VALU per item).
template <int NPROD, bool AFFINE, bool ZPAD_KEEP, bool NO_RELU, int POSB, int PLANE_V>
__device__ __forceinline__ void w1_put_item(const F4 (&b)[4], const bool (&pad)[4], const float* ta, const int cp32, const float scale,
                                            const float relu_floor, unsigned char* dst) {
    struct alignas(8) U2 { unsigned x, y; };
    constexpr bool LATE_SCALE = !AFFINE && NO_RELU && NPROD != 1;
    F4 d[4];
    if (AFFINE) {
        const F4 al = *reinterpret_cast<const F4*>(ta), be = *reinterpret_cast<const F4*>(ta + cp32);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = __builtin_fmaxf(__builtin_fmaf(b[q].v[e], al.v[e], be.v[e]), relu_floor);
                d[q].v[e] = (ZPAD_KEEP && pad[q]) ? 0.f : v;
            }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) d[q].v[e] = LATE_SCALE ? b[q].v[e] : (NO_RELU ? b[q].v[e] * scale : __builtin_fmaxf(b[q].v[e] * scale, relu_floor));
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d0 = d[0].v[e], d1 = d[1].v[e], d2 = d[2].v[e], d3 = d[3].v[e];
            v[e] = p == 0 ? d0 - d2 : (p == 1 ? d1 + d2 : (p == 2 ? d2 - d1 : d1 - d3));
        }
        if (NPROD == 1) {
            U2 h;
            h.x = TSNET_CVT_PK_BF16(v[0], v[1]);
            h.y = TSNET_CVT_PK_BF16(v[2], v[3]);
            *reinterpret_cast<U2*>(dst + p * POSB) = h;
        } else {
            U2 h, l;
            if (LATE_SCALE) TSNET_SPLIT_2PAIRS_SCALED(v[0], v[1], v[2], v[3], scale, h.x, l.x, h.y, l.y);
            else TSNET_SPLIT_2PAIRS(v[0], v[1], v[2], v[3], h.x, l.x, h.y, l.y);
            *reinterpret_cast<U2*>(dst + p * POSB) = h;
            *reinterpret_cast<U2*>(dst + p * POSB + PLANE_V) = l;
        }
    }
}

// (2) Consumer: the MFMA products of one step (tap row, slab) on the wave's two 32-pair halves: lo * hi, hi * lo, hi * hi in this order into
//     the period's chain (`fresh`: the chain starts here), or the single bf16 product.  a0 / a1: the A fragments of the two halves, [plane].
template <int NPROD, int NPL>
__device__ __forceinline__ void w1_step_products(f32x16 (&acc)[2], const F4 (&a0)[NPL], const F4 (&a1)[NPL], const F4 (&bw)[NPL], const bool fresh) {
    auto product = [&](int pa, int pb, bool fr) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x16 c = acc[i];
            if (fr) {
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = 0.f;
            }
            const F4& A = i ? a1[pa] : a0[pa];
            if (NPROD == 1) acc[i] = TSNET_MFMA_BF16(A, bw[pb], c);
            else acc[i] = TSNET_MFMA_F16(A, bw[pb], c);
        }
    };
    if (NPROD == 1) {
        product(0, 0, fresh);
    } else {
        product(NPL - 1, 0, fresh);          // lo * hi
        product(0, NPL - 1, false);          // hi * lo
        product(0, 0, false);                // hi * hi
    }
}

// (3) Output transform: the four positions M0..M3 of the column pairs of one output row half (m: the pair-major exchange region at this lane's
//     first pair and channel, pstride floats between positions) -> the lane's 16 accumulator rows = 8 pairs x (even pixel, odd pixel).
__device__ __forceinline__ void w1_output_transform(const float* m, const int pstride, const int lh, f32x16& out) {
#pragma unroll
    for (int r2 = 0; r2 < 8; ++r2) {                                 // accumulator rows 2 r2, 2 r2 + 1 = pixels (x, x + 1) of one pair
        const int x = ((2 * r2) & 3) + 8 * ((2 * r2) >> 2) + 4 * lh;
        const float* q = m + (size_t)(x >> 1) * 64;
        const float m0 = q[0], m1 = q[pstride], m2 = q[2 * pstride], m3 = q[3 * pstride];
        out[2 * r2] = (m0 + m1) + m2;
        out[2 * r2 + 1] = (m1 - m2) - m3;
    }
}

// A workgroup runs a CHUNK of a.w1_chunk consecutive tiles (1, 2 or 3: run_conv picks it so that the chunks fill the chip in whole rounds).
// Measured on the one-tile form (tools/w1_timeline.py, ResnetBlock layer at the headline batch): of a tile's 44 - 45 us, 35 are the K loop;
// 3.2 - 3.8 us pass before its first MFMA (the table, then V(0) and V(1): the consumers wait at the prologue barrier), 4.5 - 5 us after its
// last (exchange, output transform, statistics hand-off, stores), 0.4 us between two workgroups on a CU -- and the producers spend the last
// two periods of every tile transforming channels past the end of K into stages nobody reads.  Inside a chunk those two periods produce
// V(0) and V(1) of the NEXT tile instead (its pixels, its image's table: double-buffered when the image changes inside a chunk), the
// consumers' last steps fetch the next tile's first weight and V fragments, and the K loop of tile j + 1 starts right behind the epilogue
// of tile j: the prologue is paid once per chunk.  The epilogue of a tile that has a successor may use only the stage the tile's last
// period was read from (the other two hold the successor's V(0), V(1)): the four positions then meet in two passes of 32 KiB (rows 0 - 1,
// rows 2 - 3); the last tile of a chunk uses the whole region in one pass, as the one-tile form did.  Same arithmetic per output
// element in every chunk size: bit-identical results (tests/test_emu_ops.py, tests/test_gpu_ops.py).
//
// OPT bit 0: the layer zero-pads an InstanceNorm-ed input (padded pixels re-zeroed after the affine transform); bit 1: raw input, no ReLU.  Tools build (ablations,
// compute garbage): bit 4 no producer work in the loop, bit 5 weight fragments loaded once, bit 6 A fragments read once, bit 7 no barrier;
// bit 9: time stamps
struct W1Tile {                 // what differs between the tiles of a chunk (wave-uniform)
    int img, oy0, ox0, n0, tin;
    float in_scale, in_unscale;
};

template <int NPROD, bool AFFINE, int OPT = 0>
__device__ __forceinline__ void w1_chunk(const ConvArgs& a, unsigned char* smem_raw, const int wg, const int nrun) {
    static_assert(NPROD == 1 || NPROD == 3, "one (bf16 operands) or three products");
    constexpr int NPL = NPROD == 1 ? 1 : 2;
    constexpr bool ONE_LEVEL = NPROD == 1;
    constexpr bool ZPAD_KEEP = (OPT & 1) != 0;
    constexpr bool NO_RELU = (OPT & 2) != 0;                          // a raw input without ReLU (the residual stream): no max in the producer
    constexpr int BD = 3;                                            // weight register sets: fragments BD - 1 steps ahead
    // one (slab, position, plane, octet) region: 6 rows x 16 pairs x 16 B, + 64 B so that the two octet regions a producer's 16-lane write
    // group spans fall on different banks (and + 32 B per slab for the same reason)
    constexpr int REG = kW1Reg;
    constexpr int PLANE_V = 2 * REG, POSB = NPL * PLANE_V, SLABB = 4 * POSB + 32, STAGE = 2 * SLABB;
    static_assert(STAGE == kW1Stage(NPL), "stage size");
    static_assert(NPL == 1 || STAGE >= 4 * 32 * 64 * 4 + 4096, "a stage holds half the output exchange and the epilogue's reductions (one plane: chunks of one tile only)");
    constexpr int OFF_END = 3 * STAGE, OFF_EX_END = 4 * 64 * 64 * 4 + 2048;
    constexpr int OFF_TAB = OFF_END > OFF_EX_END ? OFF_END : OFF_EX_END;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    TSNET_W1_STAMP(0);
    const int pos = wave >> 1, nt = wave & 1;                        // K loop: this wave's Winograd position and 32-channel half
    const int wn0 = nt * 32;
    const int li = lane & 31, lh = lane >> 5;

    const int tcols = a.Wo / kPatchCols, tper = (a.Ho / kPatchRows) * tcols;
    const int ncc = a.Cin >> 4;
    const int npp = (ncc + 1) >> 1;                                  // periods of two 16-channel slabs (an odd count: the last slab is all zeros)
    const int cp32 = (a.Cin + 31) & ~31;                             // the transform table covers whole periods: zeros past Cin (an odd slab count)
    // The j-th tile of this workgroup's chunk.  One tile per workgroup: the XCD-aware block -> tile map every convolution kernel uses.  Chunks
    // run ALONG M (consecutive spatial tiles of one channel tile), channel tiles fastest across the workgroups of an XCD: at any time the
    // CUs of an XCD then share patches and weight slices through its L2 exactly as single tiles do (a chunk along the channel tiles re-reads
    // its patch from beyond the L2 for every tile: measured slower on the 512-tile layers).  run_conv checks the divisibility.
    auto tile_at = [&](int j) __attribute__((always_inline)) {
        W1Tile t;
        int tile_m, tile_n;
        if (nrun == 1) {
            tile_of_block(wg, a.tiles_m, a.tiles_n, a.xcd_gn, tile_m, tile_n);
        } else {
            const int xcd = wg & 7, loc = wg >> 3;
            const int gn = a.xcd_gn > 0 ? a.xcd_gn : 1;              // no grid: every XCD owns whole rows of the tile matrix
            const int tn = a.tiles_n / gn, tm = a.xcd_gn > 0 ? a.tiles_m / (8 / gn) : a.tiles_m / 8;
            tile_m = (a.xcd_gn > 0 ? xcd / gn : xcd) * tm + (loc / tn) * nrun + j;
            tile_n = (a.xcd_gn > 0 ? xcd % gn : 0) * tn + loc % tn;
        }
        t.img = tile_m / tper; t.tin = tile_m - t.img * tper;
        t.oy0 = (t.tin / tcols) * kPatchRows; t.ox0 = (t.tin % tcols) * kPatchCols;
        t.n0 = tile_n * 64;
        t.in_scale = a.in_scale; t.in_unscale = a.in_unscale;
        if (NPROD != 1 && a.in_amax) {                               // |V| <= 2 max|x|: one bit of head-room more than the direct form
            h2_device_scale(a.in_amax + t.img, a.in_bound_add, t.in_scale, t.in_unscale);
            t.in_scale *= 0.5f; t.in_unscale *= 2.0f;
        }
        return t;
    };
    W1Tile T = tile_at(0);

    const size_t planew = (size_t)((a.nchunks + 1) / 2 * 2) * a.Npad * 16;
    const tsnet_brsrc_t rsx = tsnet_make_brsrc(a.x, (unsigned)((size_t)a.N * a.H * a.W * a.Cin * 4));
    tsnet_brsrc_t rsw[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) rsw[p] = tsnet_make_brsrc(a.w + p * planew, (unsigned)(planew * 2));

    // transform table of an image: [cp32] alpha * s, then [cp32] beta * s; two of them when a.w1_tab2 (a chunk may then cross into the next image)
    float* const tab0 = reinterpret_cast<float*>(smem_raw + OFF_TAB);
    auto tab_fill = [&](float* tb, const W1Tile& t, int first, int stride) __attribute__((always_inline)) {
        for (int c = first; c < cp32; c += stride) {
            const bool ok = c < a.Cin;
            tb[c] = ok ? a.in_alpha[(size_t)t.img * a.Cin + c] * t.in_scale : 0.f;
            tb[cp32 + c] = ok ? a.in_beta[(size_t)t.img * a.Cin + c] * t.in_scale : 0.f;
        }
    };
    // the first tile's table in two halves -- its loads, then (behind whatever the caller puts in flight meanwhile) its stores: loads return in
    // order, so the table must be REQUESTED before the prologue's items, or its stores wait for them too.  Two entries per thread cover 2 x 896
    // channels (run_conv admits nothing wider)
    struct TabRegs { float a0, b0, a1, b1; };
    auto tab_request = [&](const W1Tile& t) __attribute__((always_inline)) {
        const int c0 = tid, c1 = tid + 64 * kW1Waves;
        TabRegs r;
        r.a0 = c0 < a.Cin ? a.in_alpha[(size_t)t.img * a.Cin + c0] : 0.f; r.b0 = c0 < a.Cin ? a.in_beta[(size_t)t.img * a.Cin + c0] : 0.f;
        r.a1 = c1 < a.Cin ? a.in_alpha[(size_t)t.img * a.Cin + c1] : 0.f; r.b1 = c1 < a.Cin ? a.in_beta[(size_t)t.img * a.Cin + c1] : 0.f;
        return r;
    };
    auto tab_store = [&](float* tb, const W1Tile& t, const TabRegs& r) __attribute__((always_inline)) {
        const int c0 = tid, c1 = tid + 64 * kW1Waves;
        if (c0 < cp32) { tb[c0] = r.a0 * t.in_scale; tb[cp32 + c0] = r.b0 * t.in_scale; }
        if (c1 < cp32) { tb[c1] = r.a1 * t.in_scale; tb[cp32 + c1] = r.b1 * t.in_scale; }
    };
    // (the first tile's table is filled by every thread, inside the two roles below: the producers put their first fetches in flight before it)
    f32x16 tot[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][r] = 0.f;
    unsigned long long bar_wait = 0;                                 // (tools build: time spent at the K loop's barriers)
    const bool consumer = wave < 8;
    const int mq = consumer ? wave >> 1 : 0;                         // epilogue role: output row mq of the tile, channel half nt

    // ---- the epilogue of one tile (every wave runs it: the producers keep the barriers company).  `two_pass`: the tile has a successor in
    // the chunk, whose V(0) and V(1) sit in the other two stages -- only the stage at st_free may be used
    auto epilogue = [&](const W1Tile& t, const int st_free, const bool two_pass) __attribute__((always_inline)) {
        const float unscale = a.w_unscale ? t.in_unscale * a.w_unscale[0] : t.in_unscale;
        __syncthreads();                                             // every read of the tile's last stage has been issued
        // the epilogue sits inside the tile loop: its lane-derived addresses are computed here, per tile, not hoisted across the K loop
        int tid_e = tid;
        TSNET_OPAQUE_V(tid_e);
        const int li = tid_e & 31, lh = (tid_e >> 5) & 1;
        f32x16 out[1][1];
        unsigned char* const ebase = two_pass ? smem_raw + st_free : smem_raw;
        float* const ex = reinterpret_cast<float*>(ebase);           // one pass: [position][pair 64][channel 64]; two: [position][pair 32][channel 64]
        auto transform = [&](const float* m, const int pstride) __attribute__((always_inline)) { w1_output_transform(m, pstride, lh, out[0][0]); };
        if (!two_pass) {
            if (consumer) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pair = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        ex[(pos * 64 + pair) * 64 + wn0 + li] = tot[i][r] * unscale;          // exact: power of two
                    }
            }
            __syncthreads();
            if (consumer) transform(ex + (size_t)(mq * 16) * 64 + wn0 + li, 64 * 64);
            __syncthreads();                                         // the shared epilogue reuses the region for its reductions
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {                            // rows 2 h, 2 h + 1 of the tile = pairs 32 h .. 32 h + 31
                if (consumer) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int pair = (r & 3) + 8 * (r >> 2) + 4 * lh;
                        ex[(pos * 32 + pair) * 64 + wn0 + li] = tot[h][r] * unscale;
                    }
                }
                __syncthreads();
                if (consumer && (mq >> 1) == h) transform(ex + (size_t)((mq & 1) * 16) * 64 + wn0 + li, 32 * 64);
                __syncthreads();
            }
        }
        const int m_img = t.img * a.Ho * a.Wo;
        conv_epilogue<64, 4, 2, 1, 1, 8>(a, out, ebase, tid_e, consumer ? wave : nt, t.n0, (size_t)t.img * tper + t.tin,
                                      [&](int l) { return m_img + (t.oy0 + (l >> 5)) * a.Wo + t.ox0 + (l & 31); }, consumer);
    };

    if (wave >= 8) {
        // ================= producers (waves 8..13): V of period pp + 2 while the consumers run period pp =================
        // A period's V has 6 rows x 16 pairs x 8 channel quads (32 channels) = 12 wave-sized items (row, half of the pairs); producer w owns
        // items NIT w .. NIT w + NIT - 1 (NIT = 2).  lane -> (pair (lane >> 3) of the half, quad lane & 7): the 8 lanes of a pixel read its 32 channels as ONE
        // 128-byte line -- a wave's load touches 8 lines, all of them whole (a lane per (pixel, octet) touches 32+ lines for the same bytes, and
        // the texture path, shared with the consumers' weight fragments, was what bound the first forms of this kernel).  Per item: the four
        // input pixels of the pair (columns ox0 - 1 + 2 pair + q of input row oy0 - 1 + row; reflection / zero padding in the offsets), fetched
        // one item ahead; IN + ReLU; per position: one add, the split, one ds_write_b64 per plane.  Periods past the tile's last one are the
        // first periods of the chunk's next tile (`nxt`); behind the last tile of the chunk nothing is produced.
        constexpr int NIT = 12 / kW1Prod;                            // items per producer and period
        static_assert(NIT * kW1Prod == 12 && (NIT & 1) == 0, "an even number of items per producer (the fetch buffers alternate)");
        const int pw = wave - 8, quad = lane & 7;
        const int psl = quad >> 2, poct = (quad >> 1) & 1, psub = quad & 1;
        struct Offs { unsigned vP[NIT][4]; };                       // byte offset of (pixel, channel quad), or kOOB for a padded pixel
        int ldst[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = NIT * pw + it;
            ldst[it] = psl * SLABB + poct * REG + ((item >> 1) * 16 + (item & 1) * 8 + (lane >> 3)) * 16 + psub * 8;
        }
        auto offsets = [&](const W1Tile& t, const bool real, Offs& o) __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int item = NIT * pw + it, prow = item >> 1, ppair = (item & 1) * 8 + (lane >> 3);
                int iy = t.oy0 - 1 + prow;
                bool rok = real;
                if (a.reflect) {
                    iy = iy < 0 ? -iy : iy;
                    iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
                } else {
                    rok = rok && iy >= 0 && iy < a.H;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int ix = t.ox0 - 1 + 2 * ppair + q;
                    bool ok = rok;
                    if (a.reflect) {
                        ix = ix < 0 ? -ix : ix;
                        ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
                    } else {
                        ok = ok && ix >= 0 && ix < a.W;
                    }
                    o.vP[it][q] = ok ? (unsigned)(((t.img * a.H * a.W + iy * a.W + ix) * a.Cin + quad * 4) * 4) : kOOB;
                }
            }
        };
        Offs cur, nxt;
        W1Tile Tn = T;
        bool has_next = nrun > 1;
        offsets(T, true, cur);
        offsets(T, false, nxt);                                      // (nothing behind this tile yet: zeros)
        // the offsets of a tile's LAST period: with an odd slab count its second slab lies past Cin -- those lanes read zeros, like the weights
        // of that slab (the transform table holds zeros there too)
        unsigned curT[NIT][4];
        auto tail_of = [&](const Offs& o) __attribute__((always_inline)) {
            const bool cok = (npp - 1) * 32 + quad * 4 < a.Cin;
#pragma unroll
            for (int it = 0; it < NIT; ++it)
#pragma unroll
                for (int q = 0; q < 4; ++q) curT[it][q] = cok ? o.vP[it][q] : kOOB;
        };
        tail_of(cur);
        const float* tab_cur = tab0;
        const float* tab_nxt = tab0;
        const float relu_floor = a.in_relu ? 0.f : -__builtin_inff();
        F4 sx[2][4];                                                 // two items in turn: four pixels x four channels each
        // Which tile an item belongs to is decided where the loop is written, not per load: the period loop below is split at the tile
        // boundary (every select, compare and branch in here is an issue slot taken from the MFMA waves of the same SIMD; the first form
        // spent 71 scalar instructions and 27 s_nop per two items).  SRC: 0 = the current tile, 1 = its last period, 2 = the next tile.
        using SrcCur = std::integral_constant<int, 0>;
        using SrcTail = std::integral_constant<int, 1>;
        using SrcNxt = std::integral_constant<int, 2>;
        auto v_fetch = [&](auto SRC, int pe, int it, F4 (&b)[4]) __attribute__((always_inline)) {
            constexpr int src = decltype(SRC)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                b[q] = TSNET_BUF_LOAD16(rsx, src == 2 ? nxt.vP[it][q] : (src == 1 ? curT[it][q] : cur.vP[it][q]), (unsigned)(pe * 128));
        };
        auto v_put = [&](auto SRC, int pe, int it, int st, const F4 (&b)[4]) __attribute__((always_inline)) {   // transform + split + store of a fetched item into the stage at st
            constexpr bool nx = decltype(SRC)::value == 2;
            const int c0 = pe * 32 + quad * 4;
            bool pad[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) pad[q] = ZPAD_KEEP && (nx ? nxt.vP[it][q] : cur.vP[it][q]) == kOOB;
            w1_put_item<NPROD, AFFINE, ZPAD_KEEP, NO_RELU, POSB, PLANE_V>(b, pad, (nx ? tab_nxt : tab_cur) + c0, cp32, nx ? Tn.in_scale : T.in_scale, relu_floor,
                                                                        smem_raw + st + ldst[it]);
        };
        // an item by its period pq counted from the current tile (pq >= npp: the next tile's), the source picked by branches: the prologue
        // and tiles of fewer than five periods (the hot layers have 16 and 32)
        auto fetch_any = [&](int pq, int it, F4 (&b)[4]) __attribute__((always_inline)) {
            if (pq >= npp) v_fetch(SrcNxt{}, pq - npp, it, b);
            else if (pq == npp - 1) v_fetch(SrcTail{}, pq, it, b);
            else v_fetch(SrcCur{}, pq, it, b);
        };
        auto put_any = [&](int pq, int it, int st, const F4 (&b)[4]) __attribute__((always_inline)) {
            if (pq >= npp) v_put(SrcNxt{}, pq - npp, it, st, b);
            else v_put(SrcCur{}, pq, it, st, b);
        };
        // the item stream (period, item): each item is fetched while its predecessor is transformed; two buffers in turn.  The chunk's
        // prologue -- V(0), V(1) of its first tile, the only ones the consumers wait for -- fetches its four items at once (one exposed
        // memory latency instead of two).
        static_assert(NIT == 2, "the prologue and the period loop are written for two items per producer");
        {
            F4 sp[2][4];                                             // with sx: four buffers, the prologue's four items in flight at once
            TabRegs tr{};
            if (AFFINE) tr = tab_request(T);
            fetch_any(0, 0, sx[0]); fetch_any(0, 1, sx[1]); fetch_any(1, 0, sp[0]); fetch_any(1, 1, sp[1]);
            if (AFFINE) {
                tab_store(tab0, T, tr);
                __syncthreads();                                     // the first tile's table (the consumers fill their share)
            }
            if (has_next) { Tn = tile_at(1); offsets(Tn, true, nxt); }
            put_any(0, 0, 0, sx[0]);
            fetch_any(2, 0, sx[0]);                                  // item 0 of period 2 stays in flight in buffer 0
            put_any(0, 1, 0, sx[1]);
            if (1 < npp || has_next) { put_any(1, 0, STAGE, sp[0]); put_any(1, 1, STAGE, sp[1]); }
        }
        TSNET_W1_STAMP(1);
        __syncthreads();                                             // (the consumers' prologue barrier)
        TSNET_W1_STAMP(2);
        int st_wr = 2 * STAGE;
        // consumer period pp: item (pp + 2, 1) fetched, (pp + 2, 0) written, (pp + 3, 0) fetched, (pp + 2, 1) written.  L1 / L2: where the
        // two fetched items live; PUT: where the written period lives; pe*: their period numbers inside their own tile
        auto iter = [&](auto L1, auto L2, auto PUT, int pe1, int pe2, int pep, bool put_ok) __attribute__((always_inline)) {
            if (!(OPT & 128)) TSNET_W1_LOOP_BARRIER(bar_wait);       // every read of the stage produced next has been issued
            if (!(OPT & 16)) {
                v_fetch(L1, pe1, 1, sx[1]);
                if (put_ok) v_put(PUT, pep, 0, st_wr, sx[0]);
                v_fetch(L2, pe2, 0, sx[0]);
                if (put_ok) v_put(PUT, pep, 1, st_wr, sx[1]);
            }
            st_wr = st_wr == 2 * STAGE ? 0 : st_wr + STAGE;
        };
        for (int j = 0; j < nrun; ++j) {
            if (AFFINE && a.w1_tab2 && has_next && Tn.img != T.img) {
                // the next tile lies in another image: its table into the buffer the current tile does not use -- last read (two tiles ago at
                // the latest) before this tile's first barrier, first read behind the barrier of period npp - 2 >= 1
                float* tb = tab0 + (tab_cur == tab0 ? 2 * cp32 : 0);
                tab_fill(tb, Tn, tid - 64 * 8, 64 * kW1Prod);
                tab_nxt = tb;
            }
            if (npp >= 5) {
                for (int pp = 0; pp < npp - 4; ++pp) iter(SrcCur{}, SrcCur{}, SrcCur{}, pp + 2, pp + 3, pp + 2, true);
                iter(SrcCur{}, SrcTail{}, SrcCur{}, npp - 2, npp - 1, npp - 2, true);
                iter(SrcTail{}, SrcNxt{}, SrcCur{}, npp - 1, 0, npp - 1, true);
                iter(SrcNxt{}, SrcNxt{}, SrcNxt{}, 0, 1, 0, has_next);
                iter(SrcNxt{}, SrcNxt{}, SrcNxt{}, 1, 2, 1, has_next);
            } else {
                for (int pp = 0; pp < npp; ++pp) {
                    if (!(OPT & 128)) TSNET_W1_LOOP_BARRIER(bar_wait);
                    if (!(OPT & 16)) {
                        fetch_any(pp + 2, 1, sx[1]);
                        if (pp + 2 < npp || has_next) put_any(pp + 2, 0, st_wr, sx[0]);
                        fetch_any(pp + 3, 0, sx[0]);
                        if (pp + 2 < npp || has_next) put_any(pp + 2, 1, st_wr, sx[1]);
                    }
                    st_wr = st_wr == 2 * STAGE ? 0 : st_wr + STAGE;
                }
            }
            if (j < 3) TSNET_W1_STAMP(3 + 3 * j);
            // the stage of the tile's last period: the one written three periods back = the one about to be written
            epilogue(T, st_wr, has_next);
            if (j < 3) TSNET_W1_STAMP(5 + 3 * j);
            // on to the next tile: its offsets become the current ones
            T = Tn; cur = nxt; tab_cur = tab_nxt;
            tail_of(cur);
            has_next = j + 2 < nrun;
            if (has_next) Tn = tile_at(j + 2);
            offsets(Tn, has_next, nxt);
        }
    } else {
        // ================= consumers (waves 0..7): the K loop =================
        TabRegs tr{};
        if (AFFINE) tr = tab_request(T);
        const unsigned vB = (unsigned)((wn0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16);
        F4 af[2][5][NPL], bf[BD][NPL];                               // [slab of the period][first row of the row pair][plane], [step % BD][plane]
        // Weight fragment of (tap row ky, slab cc) of this wave's position: byte offset ((ky * 4 + pos) * ncc + cc) * Npad + n0) * 32 =
        // ky * wA + cc * wB + wtile.  The K loop keeps the running offset of the period's first slab (`wsp`, + 2 wB per period) and adds
        // compile-time multiples of wA / wB per step: the loop was SALU-heavy (62 scalar instructions per period beside 36 MFMAs; the SIMD
        // hides ~5 issue slots per MFMA, MI355X_MICROARCH.md) -- every instruction here is an issue slot the matrix pipe does not get.
        bool has_next = nrun > 1;
        W1Tile Tn = T;
        if (has_next) Tn = tile_at(1);
        const int wB = a.Npad * 32, wA = 4 * ncc * wB;
        auto wtile = [&](const W1Tile& t) __attribute__((always_inline)) { return pos * ncc * wB + t.n0 * 32; };
        int wsp = wtile(T), wnext = wtile(Tn);                      // first slab of the current period; first slab of the chunk's next tile
        auto load_bs = [&](int set, unsigned vo, int soff) __attribute__((always_inline)) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) bf[set][p] = TSNET_BUF_LOAD16(rsw[p], vo, (unsigned)soff);
        };
        const unsigned char* abase = smem_raw + pos * POSB + lh * REG + li * 16;
        auto load_f = [&](int sl, int f, int st) __attribute__((always_inline)) { // rows (f, f + 1) of slab sl of the stage at byte offset st
#pragma unroll
            for (int p = 0; p < NPL; ++p) af[sl][f][p] = *reinterpret_cast<const F4*>(abase + st + sl * SLABB + p * PLANE_V + f * 256);
        };
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // One period = two slabs = six steps (t: slab t / 3, tap row t % 3) between two barriers = one accumulation chain.  st_cur is read
        // now, st_nxt = the next period (complete before this period's barrier: its first fragments are fetched at the last step).  Step
        // (sl, ky) uses fragments ky and ky + 2 of slab sl; weights BD - 1 steps ahead.
        auto period = [&](int pp, int st_cur, int st_nxt, bool first) __attribute__((always_inline)) {
            if (!(OPT & 128)) TSNET_W1_LOOP_BARRIER(bar_wait);       // the next period's V complete
            // the fragments fetched in this period: steps 2..5 of it (slab 0: tap row 2; slab 1 -- all zeros when the slab count is odd and this
            // is the last period) and steps 0, 1 of the next period (the chunk's next tile behind the last period; nothing behind the last tile)
            const bool last = pp == npp - 1;
            const unsigned vo1 = 2 * pp + 1 < ncc ? vB : kOOB, von = (!last || has_next) ? vB : kOOB;
            const int wsn = last ? wnext : wsp + 2 * wB;
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int sl = t / 3, ky = t % 3;
                if (!(OPT & 32)) {
                    static_assert(BD == 3, "fragments two steps ahead");
                    const int t2 = t + 2, ky2 = t2 % 3;             // the step fetched for: slab t2 / 3 of this period (2: the next period's first)
                    if (t2 < 3) load_bs(t2 % BD, vB, wsp + ky2 * wA);
                    else if (t2 < 6) load_bs(t2 % BD, vo1, wsp + wB + ky2 * wA);
                    else load_bs(t2 % BD, von, wsn + ky2 * wA);
                }
                if (!(OPT & 64)) {
                    if (ky == 0) { load_f(sl, 1, st_cur); load_f(sl, 3, st_cur); }
                    if (ky == 1) load_f(sl, 4, st_cur);
                    if (t == 2) { load_f(1, 0, st_cur); load_f(1, 2, st_cur); }
                    if (t == 5) { load_f(0, 0, st_nxt); load_f(0, 2, st_nxt); }
                }
                const bool fresh = (!ONE_LEVEL && t == 0) || (ONE_LEVEL && t == 0 && first);
                w1_step_products<NPROD, NPL>(acc, af[sl][ky], af[sl][ky + 2], bf[t % BD], fresh);
                __builtin_amdgcn_sched_barrier(0);                   // loads stay ahead of their use (conv_h2.hpp)
            }
            if (!ONE_LEVEL) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tot[i][r] += acc[i][r];
            }
        };
        load_bs(0, vB, wsp); load_bs(1, vB, wsp + wA);            // steps 0, 1 of the first period: requested BEFORE the table barrier -- cold from
                                                                     // HBM they need the whole prologue to arrive (behind it they cost a single-round
                                                                     // launch on cold weights 7 us: 54.8 against 47.4 us for one frame's layer)
        if (AFFINE) {
            tab_store(tab0, T, tr);
            __syncthreads();                                         // the first tile's table
        }
        TSNET_W1_STAMP(1);
        __syncthreads();                                             // V(0), V(1) complete
        TSNET_W1_STAMP(2);
        load_f(0, 0, 0); load_f(0, 2, 0);
        if (OPT & 32) load_bs(BD - 1, vB, wsp + wB + 2 * wA);
        if (OPT & 64) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int f = 0; f < 5; ++f) load_f(sl, f, 0);
        }
        int st0 = 0, st1 = STAGE;
        for (int j = 0; j < nrun; ++j) {
            TSNET_SETPRIO(2);                                        // MFMA issue ahead of the producers' VALU streams on the same SIMD
            for (int pp = 0; pp < npp; ++pp) {
                period(pp, st0, st1, pp == 0);
                st0 = st1; st1 = st1 == 2 * STAGE ? 0 : st1 + STAGE;
                wsp += 2 * wB;
            }
            if (ONE_LEVEL) { tot[0] = acc[0]; tot[1] = acc[1]; }
            TSNET_SETPRIO(0);
            if (j < 3) TSNET_W1_STAMP(3 + 3 * j);
            // the stage the last period was read from: two behind the one the next period will be read from
            epilogue(T, st1 == 2 * STAGE ? 0 : st1 + STAGE, has_next);
            if (j < 3) TSNET_W1_STAMP(5 + 3 * j);
            T = Tn; wsp = wnext;
            has_next = j + 2 < nrun;
            if (has_next) { Tn = tile_at(j + 2); wnext = wtile(Tn); }
            if (!ONE_LEVEL) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tot[i][r] = 0.f;
            }
        }
    }
    TSNET_W1_STAMP(14);
    TSNET_W1_PUT(12, bar_wait);
#ifdef TSNET_TOOLS
    if ((OPT & 512) && (threadIdx.x & 63) == 0 && blockIdx.x < kW1ProfTiles) {     // where the workgroup ran: HW_ID (CU / SE), XCC_ID
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_w1_prof[((size_t)blockIdx.x * kW1Waves + (threadIdx.x >> 6)) * kW1ProfSlots + 15] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

template <int NPROD, bool AFFINE, int OPT = 0>
__global__ __launch_bounds__(64 * kW1Waves, 1)
void conv_w1_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    // One chunk of a.w1_chunk tiles per workgroup, the hardware's dynamic dispatch over the CUs.  (A persistent form -- 256 workgroups each
    // walking three tiles, no overlap between them -- measured 5 % slower in the forward: a workgroup that starts late behind the side lane's
    // kernels carries its whole run; chunks are dealt like tiles.)
    w1_chunk<NPROD, AFFINE, OPT>(a, smem_raw, (int)blockIdx.x, a.w1_chunk > 1 ? a.w1_chunk : 1);
}

}  // namespace tsnet
