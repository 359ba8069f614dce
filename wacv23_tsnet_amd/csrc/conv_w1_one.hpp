// conv_w1_one.hpp -- conv_w1's ONE-TILE path: a workgroup runs exactly one tile (launches of a single round -- one driving frame -- and every
// layer whose tile count does not fill the CUs in whole rounds of chunks).  This is round 4's kernel (same tile, same waves, same LDS layout,
// same arithmetic and association: BIT-IDENTICAL to the chunk path of conv_w1.hpp, tests/test_gpu_ops.py / test_emu_ops.py) with round 5's
// fixes (zeros staged past Cin, the no-wait-state split).  It exists beside the chunk path because the latter pays for what chunks need --
// two tiles' offsets per producer, a table that may change image, items requested four at a time -- with a prologue of 5.9 us before the
// first MFMA where this one needs 3.9: same-box A/B on a one-frame ResnetBlock layer 43.1 against 41.8 us (54.5 against 47.5 on cold
// weights), B = 1 forward 2.030 against 1.916 ms.  run_conv sends a launch here when its chunk size is 1.
#pragma once
#include "conv_w1.hpp"

namespace tsnet {

template <int NPROD, bool AFFINE, int OPT = 0>
__device__ __forceinline__ void w1_tile_one(const ConvArgs& a, unsigned char* smem_raw, const int tile_m, const int n0) {
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
    constexpr int OFF_END = 3 * STAGE, OFF_EX_END = 4 * 64 * 64 * 4 + 2048;
    constexpr int OFF_TAB = OFF_END > OFF_EX_END ? OFF_END : OFF_EX_END;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int pos = wave >> 1, nt = wave & 1;                        // K loop: this wave's Winograd position and 32-channel half
    const int wn0 = nt * 32;
    const int li = lane & 31, lh = lane >> 5;

    const int tcols = a.Wo / kPatchCols, tper = (a.Ho / kPatchRows) * tcols;
    const int img = tile_m / tper, tin = tile_m - img * tper;
    const int oy0 = (tin / tcols) * kPatchRows, ox0 = (tin % tcols) * kPatchCols;
    const int ncc = a.Cin >> 4;
    const int npp = (ncc + 1) >> 1;                                  // periods of two 16-channel slabs (an odd count: the last slab is all zeros)
    float in_scale = a.in_scale, in_unscale = a.in_unscale;
    if (NPROD != 1 && a.in_amax) {                                   // |V| <= 2 max|x|: one bit of head-room more than the direct form
        h2_device_scale(a.in_amax + img, a.in_bound_add, in_scale, in_unscale);
        in_scale *= 0.5f; in_unscale *= 2.0f;
    }

    const size_t planew = (size_t)((a.nchunks + 1) / 2 * 2) * a.Npad * 16;
    const tsnet_brsrc_t rsx = tsnet_make_brsrc(a.x, (unsigned)((size_t)a.N * a.H * a.W * a.Cin * 4));
    tsnet_brsrc_t rsw[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) rsw[p] = tsnet_make_brsrc(a.w + p * planew, (unsigned)(planew * 2));

    const int cp32 = (a.Cin + 31) & ~31;                             // the table covers whole periods: zeros past Cin (an odd slab count)
    float* tab = reinterpret_cast<float*>(smem_raw + OFF_TAB);       // [cp32] alpha*s, then [cp32] beta*s
    if (AFFINE) {
        for (int c = tid; c < cp32; c += 64 * kW1Waves) {
            const bool ok = c < a.Cin;
            tab[c] = ok ? a.in_alpha[(size_t)img * a.Cin + c] * in_scale : 0.f;
            tab[cp32 + c] = ok ? a.in_beta[(size_t)img * a.Cin + c] * in_scale : 0.f;
        }
        __syncthreads();
    }
    f32x16 tot[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][r] = 0.f;

    if (wave >= 8) {
        // ================= producers (waves 8..13): V of period pp + 2 while the consumers run period pp =================
        // A period's V has 6 rows x 16 pairs x 8 channel quads (32 channels) = 12 wave-sized items (row, half of the pairs); producer w owns
        // items NIT w .. NIT w + NIT - 1 (NIT = 2).  lane -> (pair (lane >> 3) of the half, quad lane & 7): the 8 lanes of a pixel read its 32 channels as ONE
        // 128-byte line -- a wave's load touches 8 lines, all of them whole (a lane per (pixel, octet) touches 32+ lines for the same bytes, and
        // the texture path, shared with the consumers' weight fragments, was what bound the first forms of this kernel).  Per item: the four
        // input pixels of the pair (columns ox0 - 1 + 2 pair + q of input row oy0 - 1 + row; reflection / zero padding in the offsets), fetched
        // one item ahead; IN + ReLU; per position: one add, the split, one ds_write_b64 per plane.
        constexpr int NIT = 12 / kW1Prod;                            // items per producer and period
        static_assert(NIT * kW1Prod == 12 && (NIT & 1) == 0, "an even number of items per producer (the fetch buffers alternate)");
        const int pw = wave - 8, quad = lane & 7;
        const int psl = quad >> 2, poct = (quad >> 1) & 1, psub = quad & 1;
        unsigned vP[NIT][4];
        int ldst[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int item = NIT * pw + it, prow = item >> 1, ppair = (item & 1) * 8 + (lane >> 3);
            int iy = oy0 - 1 + prow;
            bool rok = true;
            if (a.reflect) {
                iy = iy < 0 ? -iy : iy;
                iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
            } else {
                rok = iy >= 0 && iy < a.H;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int ix = ox0 - 1 + 2 * ppair + q;
                bool ok = rok;
                if (a.reflect) {
                    ix = ix < 0 ? -ix : ix;
                    ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
                } else {
                    ok = ok && ix >= 0 && ix < a.W;
                }
                vP[it][q] = ok ? (unsigned)(((img * a.H * a.W + iy * a.W + ix) * a.Cin + quad * 4) * 4) : kOOB;
            }
            ldst[it] = psl * SLABB + poct * REG + (prow * 16 + ppair) * 16 + psub * 8;
        }
        const float relu_floor = a.in_relu ? 0.f : -__builtin_inff();
        F4 sx[2][4];                                                 // two items in turn: four pixels x four channels each
        auto v_load = [&](int pq, int it, int buf) __attribute__((always_inline)) {   // period pq; channels past Cin (the second slab of an odd count) read zeros, like the weights of that slab
            const bool cok = pq * 32 + quad * 4 < a.Cin;
#pragma unroll
            for (int q = 0; q < 4; ++q) sx[buf][q] = TSNET_BUF_LOAD16(rsx, cok ? vP[it][q] : kOOB, (unsigned)(pq * 128));
        };
        auto v_item = [&](int pq, int it, int st, int buf) __attribute__((always_inline)) {   // transform + split + store of a fetched item into the stage at st
            const int c0 = pq * 32 + quad * 4;
            bool pad[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) pad[q] = ZPAD_KEEP && vP[it][q] == kOOB;
            w1_put_item<NPROD, AFFINE, ZPAD_KEEP, NO_RELU, POSB, PLANE_V>(sx[buf], pad, tab + (c0 < cp32 ? c0 : 0) /* (periods past the tile: never read) */, cp32, in_scale,
                                                                        relu_floor, smem_raw + st + ldst[it]);
        };
        // the item stream (period, item): each item is fetched while its predecessor is transformed; two buffers in turn
        v_load(0, 0, 0);
#pragma unroll
        for (int u = 0; u < 2 * NIT; ++u) {                          // V(0), V(1); item 0 of period 2 left in flight in buffer 0
            v_load((u + 1) / NIT, (u + 1) % NIT, (u + 1) & 1);
            v_item(u / NIT, u % NIT, (u / NIT) * STAGE, u & 1);
        }
        __syncthreads();                                             // (the consumers' prologue barrier)
        int st_wr = 2 * STAGE;
        for (int pp = 0; pp < npp; ++pp) {
            if (!(OPT & 128)) __syncthreads();                       // every read of the stage produced next has been issued
            if (!(OPT & 16)) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    v_load(it == NIT - 1 ? pp + 3 : pp + 2, it == NIT - 1 ? 0 : it + 1, (it & 1) ^ 1);
                    v_item(pp + 2, it, st_wr, it & 1);
                }
            }
            st_wr = st_wr == 2 * STAGE ? 0 : st_wr + STAGE;
        }
    } else {
        // ================= consumers (waves 0..7): the K loop =================
        TSNET_SETPRIO(2);                                            // MFMA issue ahead of the producers' VALU streams on the same SIMD
        const unsigned vB = (unsigned)((wn0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16);
        F4 af[2][5][NPL], bf[BD][NPL];                               // [slab of the period][first row of the row pair][plane], [step % BD][plane]
        auto load_b = [&](int set, int cc, int ky) __attribute__((always_inline)) {   // a slab past the last one, or past the end of K: zeros
            const int kc = (ky * 4 + pos) * ncc + cc;
            const unsigned vo = cc < ncc ? vB : kOOB;
#pragma unroll
            for (int p = 0; p < NPL; ++p) bf[set][p] = TSNET_BUF_LOAD16(rsw[p], vo, (unsigned)((kc * a.Npad + n0) * 32));
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
        // now, st_nxt = period pp + 1 (complete before this period's barrier: its first fragments are fetched at the last step).  Step
        // (sl, ky) uses fragments ky and ky + 2 of slab sl; weights BD - 1 steps ahead.
        auto period = [&](int pp, int st_cur, int st_nxt) __attribute__((always_inline)) {
            if (!(OPT & 128)) __syncthreads();                       // V(pp + 1) complete
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const int sl = t / 3, ky = t % 3;
                if (!(OPT & 32)) load_b((t + BD - 1) % BD, 2 * pp + (t + BD - 1) / 3, (t + BD - 1) % 3);
                if (!(OPT & 64)) {
                    if (ky == 0) { load_f(sl, 1, st_cur); load_f(sl, 3, st_cur); }
                    if (ky == 1) load_f(sl, 4, st_cur);
                    if (t == 2) { load_f(1, 0, st_cur); load_f(1, 2, st_cur); }
                    if (t == 5) { load_f(0, 0, st_nxt); load_f(0, 2, st_nxt); }
                }
                const bool fresh = !ONE_LEVEL && t == 0;
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
#pragma unroll
        for (int i = 0; i < BD - 1; ++i) load_b(i, i / 3, i % 3);
        __syncthreads();                                             // V(0), V(1) complete
        load_f(0, 0, 0); load_f(0, 2, 0);
        if (OPT & 32) load_b(BD - 1, 1, 2);
        if (OPT & 64) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int f = 0; f < 5; ++f) load_f(sl, f, 0);
        }
        int st0 = 0, st1 = STAGE;
        for (int pp = 0; pp < npp; ++pp) {
            period(pp, st0, st1);
            st0 = st1; st1 = st1 == 2 * STAGE ? 0 : st1 + STAGE;
        }
        if (ONE_LEVEL) { tot[0] = acc[0]; tot[1] = acc[1]; }
        TSNET_SETPRIO(0);
    }

    // ---- output transform: the four positions of a pair meet through LDS
    const float unscale = a.w_unscale ? in_unscale * a.w_unscale[0] : in_unscale;
    __syncthreads();                                                 // every stage has been read
    float* ex = reinterpret_cast<float*>(smem_raw);                  // [position][pair 64][channel 64]
    const bool consumer = wave < 8;
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
    const int mq = consumer ? wave >> 1 : 0;                         // epilogue role: output row mq of the tile, channel half nt
    f32x16 out[1][1];
    w1_output_transform(ex + (size_t)(mq * 16) * 64 + wn0 + li, 64 * 64, lh, out[0][0]);
    __syncthreads();                                                 // the shared epilogue reuses the region for its reductions
    const int m_img = img * a.Ho * a.Wo;
    conv_epilogue<64, 4, 2, 1, 1>(a, out, smem_raw, tid, consumer ? wave : nt, n0, (size_t)img * tper + tin,
                                  [&](int l) { return m_img + (oy0 + (l >> 5)) * a.Wo + ox0 + (l & 31); }, consumer);
}

template <int NPROD, bool AFFINE, int OPT = 0>
__global__ __launch_bounds__(64 * kW1Waves, 1)
void conv_w1_one_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    int tile_m, tile_n;
    tile_of_block(blockIdx.x, a.tiles_m, a.tiles_n, a.xcd_gn, tile_m, tile_n);
    w1_tile_one<NPROD, AFFINE, OPT>(a, smem_raw, tile_m, tile_n * 64);
}

}  // namespace tsnet
