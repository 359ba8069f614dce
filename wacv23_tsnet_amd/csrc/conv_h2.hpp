// conv_h2.hpp -- the PATCH convolution kernels: the input patch of an output rectangle lives in LDS, the taps are shifted views of it.
//   h2_tile   3x3 / stride 1 / pad 1 (ResnetBlocks, FuseNet, decoder up-convolutions: 83 % of the FLOPs; TSNet.py:27,42,147)
//   h2s_tile  7x7 stems at 8 input channels (TSNet.py:66 with label_nc = 2)
//   h2d_tile  3x3 / stride 2 / zero pad 1 downsampling (TSNet.py:70)
// fp32 input, the producer's InstanceNorm + ReLU applied while the patch is staged, fp16 x 2 operands / three MFMA products (or one bf16
// plane / one product), fp32 accumulate in two levels, fp64 statistics in the epilogue (conv_common.hpp).
//
// Data flow of an h2 tile (PR x 32 output pixels of one image, BN output channels, four waves):
//   * per 16-channel slab the (PR+2) x (32+2) input patch is fetched ONCE as fp32 (two 16-byte buffer loads per thread and round,
//     reflection / zero padding resolved in the lane's offset), transformed in registers (x*alpha+beta, ReLU, *s, split: conv_common.hpp)
//     and written to LDS as two fp16 planes; the nine taps are nine shifted views (immediate offsets of the ds_read);
//   * the LDS image is octet-planar ([plane][channel octet][pixel slot], 16-byte entries): a fragment read of 16 consecutive lanes covers
//     256 consecutive bytes -- conflict-free without a swizzle -- and one address register serves all (row, tap) combinations;
//   * weight fragments (fragment order, contiguous per wave: pack_weights_kernel) go straight into registers two steps ahead (three
//     register sets); A fragments one step ahead (two sets; the slab loop is unrolled by two because nine taps flip the parity);
//   * DEEP (OPT bit 3; launches of at most one workgroup per CU -- a single driving frame): nothing else on the CU hides a load's
//     latency, so weight fragments are fetched EIGHT steps ahead (nine register sets, one per tap) and both staging rounds of the next
//     slab are in flight for five / six steps; same K order, chains and fold points, hence the same bits;
//   * one __syncthreads per slab; every load is compiler-visible (the only inline asm is the three-instruction split).
// K order is slab-major (slab, tap); chains = taps 0..3 and 4..8 of a slab, folded into the running total.  Every tile shape runs the
// same chains per output element: bit-identical convolution results (tested).
#pragma once
#include "conv_common.hpp"

namespace tsnet {

// NPROD = 3: lo*hi, hi*lo, hi*hi;  NPROD = 4: lo*lo first (kept for the accuracy comparison in the op tests);  NPROD = 1: bf16 operands.
// Tile shapes (PR rows x BN channels, waves WARPS_M x WARPS_N, wave tile (PR/WARPS_M * 32) x (BN/WARPS_N)):
//   4 x 64  (2 x 2, 64 x 32)   768 tiles = three per CU on the ResnetBlock layers at the headline batch
//   4 x 128 (2 x 2, 64 x 64)   half the LDS / L1 bytes per MFMA, two workgroups per CU (FuseNet, decoder)
//   4 x 32  (4 x 1, 32 x 32)   launches with few tiles (one driving frame)
//   2 x 128 (1 x 4, 64 x 32)   the patch is staged once per 128 instead of 64 output channels (0.67 x the staging work per MFMA)
// OPT: bit 0 = the layer zero-pads an InstanceNorm-ed input (the padded pixels are re-zeroed after the affine transform; reflection
// needs no such multiply); bits 3 + 4 = DEEP prefetch with two K groups (24: the single-frame tiles of the product).  Tools build only
// (tools/h2_variants.py): bit 1 = one chain per slab, bit 2 = one chain per four slabs (the product folds every two), bits 3 / 4 alone,
// bit 5 = weight fragments five steps ahead (six register sets).
// HABL (tools/h2_variants.py; non-zero computes garbage): bit0 no patch staging in the loop, bit1 weight fragments loaded once, bit2 A
// fragments read once, bit3 no fold, bit4 no slab barrier.
template <int PR, int BN, int WARPS_M, int WARPS_N, int NPROD, bool AFFINE, int HABL = 0, int OPT = 0>
__device__ __forceinline__ void h2_tile(const ConvArgs& a, unsigned char* smem_raw, const int tile_m, const int n0) {
    constexpr int BM = PR * kPatchCols;
    constexpr int NW = WARPS_M * WARPS_N;
    static_assert(NW == 4, "four waves: patch blocks are dealt w, w+4");
    static_assert(NPROD == 1 || NPROD == 3 || NPROD == 4, "one (bf16 operands), three or four products");
    constexpr int NPL = NPROD == 1 ? 1 : 2;                          // operand planes
    constexpr bool DEEP = (OPT & 8) != 0;
    constexpr int BD = DEEP ? 9 : ((OPT & 32) ? 6 : 3);              // weight register sets: fragments are fetched BD - 1 steps ahead (3, 6, 9: divisors of the 18 steps of a slab pair)
    constexpr bool ONE_LEVEL = NPROD == 1;                            // bf16 operands (2^-9 each): one fp32 chain over all of K -- the second
                                                                     // level buys nothing below the operand rounding and costs 16 VGPRs per tile
    constexpr bool ZPAD_KEEP = (OPT & 1) != 0;
    constexpr int CH = (OPT & 2) ? 1 : ((OPT & 4) ? 4 : 2);          // slabs per accumulation chain
    constexpr int KG = (OPT & 16) ? 2 : 1;                           // K groups: 4 KG waves, group g runs the slabs g, g + KG, ... of the tile
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 32, NTL = WN / 32;
    static_assert(WARPS_M * MT == PR, "a wave covers MT whole patch rows");
    constexpr int NRW = MT + 2;                                      // patch rows under a wave's MT output rows
    constexpr int PC = kPatchCols + 2, PP = (PR + 2) * PC;           // 34 columns; 204 (PR = 4) or 136 (PR = 2) patch pixels
    constexpr int PBLK = (PP + 31) / 32;                             // 7 / 5 blocks of 32 pixel slots
    static_assert(PBLK <= 8, "two staging rounds");
    constexpr int REGION = PBLK * 512;                               // one octet region: 32 PBLK slots x 16 B
    constexpr int PLANE_P = 2 * REGION, PATCH_BYTES = NPL * PLANE_P;
    constexpr int OFF_SINK = 2 * 2 * PLANE_P;                        // 2 KiB sink for a wave whose second pixel block does not exist (branch-free staging)
    constexpr int GROUP_BYTES = OFF_SINK + 2048;                     // stages + sink of one K group
    constexpr int OFF_TAB = KG * GROUP_BYTES;                        // (alpha*s, beta*s) table of the image: 2 x Cin floats (fixed offset in both modes)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_all = TSNET_UNIFORM(tid >> 6);
    const int kg = KG == 1 ? 0 : wave_all >> 2;                      // this wave's K group
    // ... and its place in the group's 2 x 2 / 4 x 1 / 1 x 4 wave grid.  (One group: no mask -- "& 3" hides from the compiler that the
    // readfirstlane value IS the wave index and costs the 168-VGPR tiles 11 scratch operations per two slabs: tools/isa_check.py.)
    const int wave = KG == 1 ? wave_all : (wave_all & 3);
    unsigned char* const gbase = smem_raw + kg * GROUP_BYTES;        // the group's own patch stages
    const int wrow = wave / WARPS_N;
    const int wn0 = (wave % WARPS_N) * WN;
    const int li = lane & 31, lh = lane >> 5;

    const int tcols = a.Wo / kPatchCols, tper = (a.Ho / PR) * tcols;
    const int img = tile_m / tper, tin = tile_m - img * tper;
    const int oy0 = (tin / tcols) * PR, ox0 = (tin % tcols) * kPatchCols;
    const int ncc = a.Cin >> 4;
    float in_scale = a.in_scale, in_unscale = a.in_unscale;
    if (NPROD != 1 && a.in_amax) h2_device_scale(a.in_amax + img, a.in_bound_add, in_scale, in_unscale);

    const size_t planew = (size_t)((a.nchunks + 1) / 2 * 2) * a.Npad * 16;
    const bool xb16 = NPROD == 1 && a.x_bf16;                        // bf16 storage: the input tensor holds bf16 (conv_common.hpp load_x_octet)
    const tsnet_brsrc_t rsx = tsnet_make_brsrc(a.x, (unsigned)((size_t)a.N * a.H * a.W * a.Cin * (xb16 ? 2 : 4)));
    tsnet_brsrc_t rsw[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) rsw[p] = tsnet_make_brsrc(a.w + p * planew, (unsigned)(planew * 2));

    // ---- staging geometry: this wave owns pixel blocks wave and wave + 4 (the second one may not exist: wave-uniform skip);
    //      lane -> (slot b*32 + (lane & 31), octet lane >> 5): the 8 lanes of a ds_write_b128 group write 128 consecutive bytes.
    const int oct = lane >> 5;
    unsigned vP[2];
    float vM[2];                                                     // 1, or 0 for a zero-padded / unused slot
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int pp = (wave + 4 * r) * 32 + (lane & 31);
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
        vP[r] = ok ? (unsigned)(((img * a.H * a.W + pix) * a.Cin + oct * 8) * 4) : kOOB;
        vM[r] = ok ? 1.f : 0.f;
    }
    float* tab = reinterpret_cast<float*>(smem_raw + OFF_TAB);       // [Cin] alpha*s, then [Cin] beta*s
    if (AFFINE) {
        for (int c = tid; c < a.Cin; c += 256 * KG) {
            tab[c] = a.in_alpha[(size_t)img * a.Cin + c] * in_scale;
            tab[a.Cin + c] = a.in_beta[(size_t)img * a.Cin + c] * in_scale;
        }
        __syncthreads();
    }
    const float relu_floor = a.in_relu ? 0.f : -__builtin_inff();    // branch-free ReLU switch
    // slab indices below are LOCAL to the K group (stage parity, loop counter); slab_of() is the channel slab they stand for
    auto slab_of = [&](int lc) __attribute__((always_inline)) { return lc * KG + kg; };
    F4 sx[2][2];                                                     // staging registers [round][half octet] (DEEP: both rounds live at once)
    auto stage_load_x = [&](int cn, int r) __attribute__((always_inline)) {
        if (NPROD == 1) load_x_octet(rsx, xb16, vP[r], (unsigned)(slab_of(cn) * 64), sx[r]);
        else {
#pragma unroll
            for (int q = 0; q < 2; ++q) sx[r][q] = TSNET_BUF_LOAD16(rsx, vP[r], (unsigned)(slab_of(cn) * 64 + q * 16));
        }
    };
    // par = stage the slab is written to (its local index & 1, a compile-time constant at every call site)
    auto stage_store = [&](int cn, int par, int r) __attribute__((always_inline)) {
        const int b = wave + 4 * r;
        F4 t[2];
        const float* ta = tab + (slab_of(cn) < ncc ? slab_of(cn) * 16 : 0) + oct * 8;   // past the last slab: any valid entry (result unused)
        transform_octet<AFFINE, ZPAD_KEEP>(sx[r], ta, a.Cin, in_scale, relu_floor, vM[r], t);
        // a block past the patch does not exist: its wave writes into the sink (wave-uniform select, no branch)
        unsigned char* dst = gbase + (b < PBLK ? par * PATCH_BYTES + oct * REGION + b * 512 : OFF_SINK + oct * 512) + (lane & 31) * 16;
        F4 Hh, Ll;
        if (NPROD == 1) {
            bf16_octet(t[0], t[1], Hh);
            *reinterpret_cast<F4*>(dst) = Hh;
        } else {
            split_h2_octet(t[0], t[1], Hh, Ll);
            *reinterpret_cast<F4*>(dst) = Hh;
            *reinterpret_cast<F4*>(dst + (b < PBLK ? PLANE_P : 1024)) = Ll;
        }
    };

    // ---- fragments.  Weights: lane (li, lh) takes the 16 bytes of column wn0 + j*32 + li, logical octet lh.
    // Step s of a slab is tap column kx = s / 3, tap row ky = s % 3 (weight tap ky*3 + kx): under one kx the MT output rows of a wave read
    // the MT + 2 patch rows ar[0 .. MT+1], row i + ky for output row i -- every row fragment is read ONCE per tap column and serves up to
    // three taps (12 instead of 18 LDS reads per plane and slab on a two-row wave tile, in the registers of the two alternating sets it replaces).
    const unsigned vB = (unsigned)((wn0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16);
    // A wave tile of more than two rows (MT = 4: the four waves side by side over the channels, every weight fragment loaded once per workgroup)
    // cannot roll its rows through one ring -- rows 2 .. MT-1 of a column are still read at its last tap row -- so the row fragments are
    // double-buffered by tap column instead: the whole next column is fetched, two rows per step, while the current one is used.
    constexpr bool DB = MT > 2;
    static_assert(!DB || NRW == 6, "the double-buffered form fetches two rows of the next tap column per step");
    F4 ar[DB ? 2 : 1][NRW][NPL], bf[BD][NPL][NTL];                   // [tap-column parity][patch row][plane], [register set][plane][tile]
    auto tap_of = [](int s) __attribute__((always_inline)) { return (s % 3) * 3 + s / 3; };
    // Weight fragment of (tap, slab): byte offset ((tap * ncc + slab) * Npad + n0 + 32 j) * 32 = tap * wT + slab * wB + n0 * 32 + 1024 j.  The K
    // loop keeps the running offset of its current slab (`wsl`, + KG slabs per iteration) and adds compile-time multiples of wT / wB per
    // step: re-deriving it per step cost ~5 scalar instructions each -- 96 SALU beside the 72 MFMAs of a bf16 slab pair (tools/isa_mix.py),
    // in a loop that is bound by instruction issue.
    const int wB = a.Npad * 32, wT = ncc * wB;
    int wsl = slab_of(0) * wB + n0 * 32;
    auto load_b = [&](int set, int dslab, int s) __attribute__((always_inline)) {     // dslab: 0 = the loop's current slab, 1 = the one behind it; past the end of K the descriptor returns zeros
        const int soff = wsl + dslab * KG * wB + tap_of(s) * wT;
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int j = 0; j < NTL; ++j) bf[set][p][j] = TSNET_BUF_LOAD16(rsw[p], vB + j * 1024u, (unsigned)soff);      // (+ 1024 j rides in the instruction's immediate offset)
    };
    const unsigned char* abase = gbase + lh * REGION + (wrow * MT * PC + li) * 16;
    auto load_row = [&](int r, int par, int kx) __attribute__((always_inline)) {
        const unsigned char* pbase = abase + par * PATCH_BYTES;
#pragma unroll
        for (int p = 0; p < NPL; ++p) ar[DB ? (kx & 1) : 0][r][p] = *reinterpret_cast<const F4*>(pbase + p * PLANE_P + (r * PC + kx) * 16);
    };

    f32x16 acc[MT][NTL], tot[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

    auto product = [&](int kx, int ky, int sb, int pa, int pb, bool fresh) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                f32x16 c = acc[i][j];
                if (fresh) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = 0.f;
                }
                if (NPROD == 1) acc[i][j] = TSNET_MFMA_BF16(ar[DB ? (kx & 1) : 0][i + ky][pa], bf[sb][pb][j], c);
                else acc[i][j] = TSNET_MFMA_F16(ar[DB ? (kx & 1) : 0][i + ky][pa], bf[sb][pb][j], c);
            }
    };
    // step s of local slab cc (stage par): B(cc, s) sits in set s % BD, the rows of its tap column in ar; issues first B of BD - 1 steps ahead
    // and the patch row(s) the NEXT step needs: row ky + MT of this column (ky < 2); rows 0 .. MT-2 of the next column at ky = 1 (their last
    // reader was ky = 0), row MT-1 at ky = 2.  first / last: the step opens / closes an accumulation chain.
    auto step = [&](int cc, int par, int s, bool first, bool last) __attribute__((always_inline)) {
        const int kx = s / 3, ky = s % 3;
        const bool fresh = !ONE_LEVEL && first;
        const int s2 = (s + BD - 1) % 9;
        if (!(HABL & 2)) load_b((par * 9 + s + BD - 1) % BD, s + BD - 1 >= 9 ? 1 : 0, s2);
        if (!(HABL & 4)) {
            if (DB) {
                if (kx < 2) { load_row(2 * ky, par, kx + 1); load_row(2 * ky + 1, par, kx + 1); }     // the next tap column, into the other buffer
            } else {
                if (ky < 2) load_row(ky + MT, par, kx);
                if (kx < 2 && ky == 1) {
#pragma unroll
                    for (int r = 0; r + 1 < MT; ++r) load_row(r, par, kx + 1);
                }
                if (kx < 2 && ky == 2) load_row(MT - 1, par, kx + 1);
            }
        }
        // staging of slab cc+1: round 0 fetched at step 0 and written at step 2, round 1 fetched at step 3 and written at step 5 (DEEP: fetched
        // at steps 0 and 1, written at steps 5 and 7).  Past the last slab the loads run into the next pixel's channels or return zeros:
        // written to the idle stage, never read.
        if (s == 0 && !(HABL & 1)) stage_load_x(cc + 1, 0);
        if (s == (DEEP ? 1 : 3) && !(HABL & 1)) stage_load_x(cc + 1, 1);
        const int SB = (par * 9 + s) % BD;
        if (NPROD == 1) {
            product(kx, ky, SB, 0, 0, fresh);                        // bf16 * bf16
        } else {
            if (NPROD == 4) product(kx, ky, SB, NPL - 1, NPL - 1, fresh);    // lo * lo
            product(kx, ky, SB, NPL - 1, 0, fresh && NPROD == 3);    // lo * hi
            product(kx, ky, SB, 0, NPL - 1, false);                  // hi * lo
            product(kx, ky, SB, 0, 0, false);                        // hi * hi
        }
        if (s == (DEEP ? 5 : 2) && !(HABL & 1)) stage_store(cc + 1, par ^ 1, 0);
        if (s == (DEEP ? 7 : 5) && !(HABL & 1)) stage_store(cc + 1, par ^ 1, 1);
        // Nothing crosses a step boundary in the scheduler: the loads above stay ONE step (weights: BD - 1 steps) ahead of their first use.
        // Left free, the scheduler sinks every load to its use to save registers (127 VGPRs, s_waitcnt vmcnt(0) before each MFMA group,
        // 181 us on the ResnetBlock layer instead of 160).
        __builtin_amdgcn_sched_barrier(0);
        if (!ONE_LEVEL && last && !(HABL & 8)) {             // element by element: a vector add would be selected as v_pk_add_f32
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tot[i][j][r] += acc[i][j][r];
        }
    };
    // a slab: nine steps between two barriers; open / close: the slab is the first / last of its accumulation chain
    auto slab = [&](int cc, int par, bool open, bool close) __attribute__((always_inline)) {
        if (!(HABL & 16)) __syncthreads();                           // patch(cc) complete and visible; slab cc-1 fully read
        if ((!(HABL & 4) || cc == 0) && !(DB && (HABL & 8) && cc > 0)) {        // (HABL 8 on the double-buffered form: without the exposed fetch of a slab's first tap column)
#pragma unroll
            for (int r = 0; r < (DB ? NRW : MT); ++r) load_row(r, par, 0);
        }
#pragma unroll
        for (int s = 0; s < 9; ++s) step(cc, par, s, open && s == 0, close && s == 8);
        wsl += KG * wB;
    };

    // prologue: patch of slab 0, weight fragments of the first BD - 1 steps
    stage_load_x(0, 0); stage_load_x(0, 1);
#pragma unroll
    for (int i = 0; i < BD - 1; ++i) load_b(i, 0, i);
    stage_store(0, 0, 0); stage_store(0, 0, 1);
    if (HABL & 2) load_b(2, 0, 2);
    if (HABL & 4) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NRW; ++r) load_row(r, 0, 0);
    }
    // K order: slab-major, (kx, ky) inside a slab.  Chains: CH = 2 slabs (18 taps x 16 channels = 288 k, 54 MFMAs), folded into the running
    // total; a trailing pair / single slab forms a shorter chain.  With 32 slabs (512 channels) that is 16 folds of partial sums an eighth
    // the size of the total: fewer roundings at the total's magnitude than a fold per half slab, for a quarter of the fold instructions.
    const int nloc = ncc / KG;                                       // the launcher passes KG = 2 only for an even slab count
    int cc = 0;
    if (CH == 4) {
        for (; cc + 4 <= nloc; cc += 4) { slab(cc, 0, true, false); slab(cc + 1, 1, false, false); slab(cc + 2, 0, false, false); slab(cc + 3, 1, false, true); }
    }
    if (CH == 1) {
        for (; cc + 2 <= nloc; cc += 2) { slab(cc, 0, true, true); slab(cc + 1, 1, true, true); }
    } else {
        for (; cc + 2 <= nloc; cc += 2) { slab(cc, 0, true, false); slab(cc + 1, 1, false, true); }
    }
    if (cc < nloc) slab(cc, 0, true, true);
    if (ONE_LEVEL) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) tot[i][j] = acc[i][j];
    }
    if (KG == 2) {
        // total = P0 + P1 (the groups' partial totals, each a sequential fold of its own chains): group 1 hands its registers over through
        // LDS in lane order (conflict-free), group 0 adds them and runs the epilogue; group 1 only keeps the barriers company
        __syncthreads();                                             // every stage has been read
        float* fold = reinterpret_cast<float*>(smem_raw) + (size_t)wave * (MT * NTL * 16 * 64) + lane;
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) fold[((i * NTL + j) * 16 + r) * 64] = tot[i][j][r];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) tot[i][j][r] += fold[((i * NTL + j) * 16 + r) * 64];
        }
    }

    const float unscale = a.w_unscale ? in_unscale * a.w_unscale[0] : in_unscale;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[i][j][r] *= unscale;    // exact: power of two
    const int m_img = img * a.Ho * a.Wo;
    conv_epilogue<BN, WARPS_M, WARPS_N, MT, NTL>(a, tot, smem_raw, tid, wave, n0, (size_t)img * tper + tin,
                                                 [&](int l) { return m_img + (oy0 + (l >> 5)) * a.Wo + ox0 + (l & 31); }, kg == 0);
}

// LDS bytes of an h2 tile: two stages x two planes x two octet regions + the sink + the transform table (two-plane offsets in every mode)
constexpr int h2_lds_bytes(int PR, int Cin, int KG = 1) { return KG * (2 * 2 * 2 * (((PR + 2) * (kPatchCols + 2) + 31) / 32) * 512 + 2048) + 2 * Cin * 4; }

// workgroups per CU the tile is built for
template <int PR, int BN, int WARPS_M, int WARPS_N, int NPROD, int OPT>
constexpr int h2_wgs_per_cu() { return (OPT & 16) ? 1 : ((BN / WARPS_N) * (PR / WARPS_M) * ((OPT & 8) ? 2 : 1) <= (NPROD == 1 ? 128 : 64) ? ((OPT & 64) ? 4 : 3) : 2); }

template <int PR, int BN, int WARPS_M, int WARPS_N, int NPROD, bool AFFINE, int HABL = 0, int OPT = 0>
__global__ __launch_bounds__((OPT & 16) ? 512 : 256, (h2_wgs_per_cu<PR, BN, WARPS_M, WARPS_N, NPROD, OPT>()))
void conv_h2_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    int tile_m, tile_n;
    tile_of_block(blockIdx.x, a.tiles_m, a.tiles_n, a.xcd_gn, tile_m, tile_n);
    h2_tile<PR, BN, WARPS_M, WARPS_N, NPROD, AFFINE, HABL, OPT>(a, smem_raw, tile_m, tile_n * BN);
}

// ---------------------------------------------------------------------------------------------------------------
// h2s: the 7 x 7 stems at 8 input channels (TSNet.py:66 with label_nc = 2: image 3 + label 2 + coordinates 3, or label 2 + coordinates 3
// padded to 8) as a PATCH kernel.  As an implicit GEMM (conv_h2r, SMALL_CIN) the stem gathers 128 rows x 16 k of fp32 from L1 for every
// k-step -- 49 taps re-read every input pixel 49 times, and with only 64 output channels there is little MFMA work per gathered byte
// (217 us for 39.5 GFLOP).  Here the (4+6) x (32+6) x 8-channel patch of a 4 x 32 output rectangle is fetched ONCE (12 KB of fp32),
// scaled, split into two fp16 planes of one 16-byte octet per pixel, and the 25 k-steps (two taps per 16-deep k-group, the 50th tap has
// zero weights) read it through shifted views: no barrier and no global A traffic inside the loop.
// A fragment of step s: lane (li, lh) supplies output pixel li of a row and k-half lh = tap 2s + lh, i.e. patch slot
// (row + ky) * 38 + li + kx of THAT tap.  Tap 2s + 1 is one slot right of tap 2s, or -- when tap 2s is the last of its row -- 32 slots
// on: one per-lane base, a wave-uniform tap offset and lh x delta per step.
template <int NPROD>
__device__ __forceinline__ void h2s_tile(const ConvArgs& a, unsigned char* smem_raw, const int tile_m, const int n0) {
    constexpr int BN = 64, WARPS_M = 2, WARPS_N = 2, MT = 2, NTL = 1;
    constexpr int NPL = NPROD == 1 ? 1 : 2;
    constexpr int PC = kPatchCols + 6, PR = kPatchRows + 6, PP = PR * PC;       // 38 x 10 = 380 patch pixels
    constexpr int PLANE_S = 384 * 16;                                             // one plane: a 16-byte octet per pixel slot
    static_assert(NPROD == 1 || NPROD == 3, "one (bf16 operands) or three products");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int wrow = wave / WARPS_N;
    const int wn0 = (wave % WARPS_N) * 32;
    const int li = lane & 31, lh = lane >> 5;

    const int tcols = a.Wo / kPatchCols, tper = (a.Ho / kPatchRows) * tcols;
    const int img = tile_m / tper, tin = tile_m - img * tper;
    const int oy0 = (tin / tcols) * kPatchRows, ox0 = (tin % tcols) * kPatchCols;
    float in_scale = a.in_scale, in_unscale = a.in_unscale;
    if (NPROD != 1 && a.in_amax) h2_device_scale(a.in_amax + img, a.in_bound_add, in_scale, in_unscale);

    const size_t planew = (size_t)((a.nchunks + 1) / 2 * 2) * a.Npad * 16;
    const tsnet_brsrc_t rsx = tsnet_make_brsrc(a.x, (unsigned)((size_t)a.N * a.H * a.W * 8 * 4));
    tsnet_brsrc_t rsw[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) rsw[p] = tsnet_make_brsrc(a.w + p * planew, (unsigned)(planew * 2));

    // ---- weight fragments first (their latency hides behind the patch staging), two steps ahead afterwards
    const unsigned vB = (unsigned)((wn0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16);
    F4 af[2][NPL][MT], bf[3][NPL];
    auto load_b = [&](int set, int kc) __attribute__((always_inline)) {           // past the end of K the descriptor returns zeros
#pragma unroll
        for (int p = 0; p < NPL; ++p) bf[set][p] = TSNET_BUF_LOAD16(rsw[p], vB, (unsigned)((kc * a.Npad + n0) * 32));
    };
    load_b(0, 0);
    load_b(1, 1);

    // ---- patch staging: thread t takes pixel slots t and t + 256 (380 in all); reflection padding resolved in the address
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int pp = tid + r * 256;
        const int pr = pp / PC, pc = pp - pr * PC;
        int iy = oy0 - 3 + pr, ix = ox0 - 3 + pc;
        iy = iy < 0 ? -iy : iy;
        iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
        ix = ix < 0 ? -ix : ix;
        ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
        const unsigned v = pp < PP ? (unsigned)((((img * a.H + iy) * a.W) + ix) * 32) : kOOB;
        F4 x0 = TSNET_BUF_LOAD16(rsx, v, 0u), x1 = TSNET_BUF_LOAD16(rsx, v, 16u);
#pragma unroll
        for (int e = 0; e < 4; ++e) { x0.v[e] *= in_scale; x1.v[e] *= in_scale; }
        if (pp < 384) {                                               // slots 380..383 hold zeros (never read, kept finite)
            F4 Hh, Ll;
            if (NPROD == 1) {
                bf16_octet(x0, x1, Hh);
                *reinterpret_cast<F4*>(smem_raw + pp * 16) = Hh;
            } else {
                split_h2_octet(x0, x1, Hh, Ll);
                *reinterpret_cast<F4*>(smem_raw + pp * 16) = Hh;
                *reinterpret_cast<F4*>(smem_raw + PLANE_S + pp * 16) = Ll;
            }
        }
    }
    __syncthreads();

    // per-lane address of step st: slot of tap 2 st for lh = 0; for lh = 1 the next tap = one slot right, or the first slot of the next patch
    // row when tap 2 st ends its row, or (last step: the 50th tap does not exist, its weights are zero) the same slot again
    const int base = (wrow * MT * PC + li) * 16;
    // (slot offset of tap 2 st, and how far its partner tap 2 st + 1 lies) per step: a table in constant memory, read with scalar loads --
    // the loop over the chains is rolled, and deriving tap row and column of a runtime step (t0 / 7, t0 % 7) cost 84 scalar instructions per
    // iteration beside 36 MFMAs (tools/isa_mix.py): every one an issue slot the MFMAs did not get
    struct StepTab { int off[25], delta[25]; };
    static constexpr StepTab kStep = [] {
        StepTab t{};
        for (int st = 0; st < 25; ++st) {
            const int t0 = 2 * st, ky = t0 / 7, kx = t0 - ky * 7;
            t.off[st] = (ky * PC + kx) * 16;
            t.delta[st] = st >= 24 ? 0 : (kx == 6 ? (PC - 6) * 16 : 16);
        }
        return t;
    }();
    auto load_a = [&](int set, int st) __attribute__((always_inline)) {
        const unsigned char* b = smem_raw + base + kStep.off[st] + lh * kStep.delta[st];      // st: wave-uniform
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int p = 0; p < NPL; ++p) af[set][p][i] = *reinterpret_cast<const F4*>(b + p * PLANE_S + i * PC * 16);
    };

    f32x16 acc[MT], tot[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; tot[i][0][r] = 0.f; }
    auto product = [&](int sa, int sb, int pa, int pb, bool fresh) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            f32x16 c = acc[i];
            if (fresh) {
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = 0.f;
            }
            if (NPROD == 1) acc[i] = TSNET_MFMA_BF16(af[sa][pa][i], bf[sb][pb], c);
            else acc[i] = TSNET_MFMA_F16(af[sa][pa][i], bf[sb][pb], c);
        }
    };
    // step st = 6 c + j: A(st) in set j & 1, B(st) in set j % 3 (six steps per chain keep both rotations static); issues A(st + 1), B(st + 2) first
    auto step = [&](int st, int j) __attribute__((always_inline)) {
        load_b((j + 2) % 3, st + 2);
        load_a((j + 1) & 1, st + 1 < 25 ? st + 1 : 24);
        const int SA = j & 1, SB = j % 3;
        if (NPROD == 1) {
            product(SA, SB, 0, 0, j == 0);
        } else {
            product(SA, SB, 1, 0, j == 0);                            // lo * hi
            product(SA, SB, 0, 1, false);                             // hi * lo
            product(SA, SB, 0, 0, false);                             // hi * hi
        }
    };
    auto fold = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i) tot[i][0] += acc[i];
    };
    load_a(0, 0);
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {                                     // four chains of six k-groups, then the 25th
        const int s0 = 6 * c;
        step(s0, 0); step(s0 + 1, 1); step(s0 + 2, 2); step(s0 + 3, 3); step(s0 + 4, 4); step(s0 + 5, 5);
        fold();
    }
    step(24, 0);
    fold();

    const float unscale = a.w_unscale ? in_unscale * a.w_unscale[0] : in_unscale;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][0][r] *= unscale;                     // exact: power of two
    const int m_img = img * a.Ho * a.Wo;
    __syncthreads();                                                  // the epilogue reuses the patch region for its reduction
    conv_epilogue<BN, WARPS_M, WARPS_N, MT, NTL>(a, tot, smem_raw, tid, wave, n0, (size_t)img * tper + tin,
                                                 [&](int l) { return m_img + (oy0 + (l >> 5)) * a.Wo + ox0 + (l & 31); });
}

constexpr int kH2sLds = 2 * 384 * 16 + 2048;     // two planes of 384 slots (+ room for the epilogue's flag word at 8192)

template <int NPROD>
__global__ __launch_bounds__(256, 3)
void conv_h2s_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int bid = xcd_item(blockIdx.x, a.tiles_m * a.tiles_n);
    const int tile_m = bid / a.tiles_n;
    h2s_tile<NPROD>(a, smem_raw, tile_m, (bid - tile_m * a.tiles_n) * 64);
}

// ---------------------------------------------------------------------------------------------------------------
// h2d: the encoders' 3 x 3 / stride-2 / zero-pad-1 downsampling convolutions (TSNet.py:70) as a patch kernel -- h2_tile with the patch
// geometry of stride 2.  A 4 x 32 output rectangle reads the (2*4+1) x (2*32+1) = 9 x 65 input patch; per 16-channel slab it is fetched once
// (585 pixels x 64 B), transformed and split like h2_tile's, and written to LDS with its columns DE-INTERLEAVED by parity: row pitch 68
// slots = 33 even columns, then (from slot 36) 32 odd ones.  Output column x under tap column kx reads input column 2x + kx: kx = 0 -> even slot x,
// kx = 1 -> odd slot x, kx = 2 -> even slot x + 1, so the 32 lanes of a fragment read 32 consecutive 16-byte slots (conflict-free) and
// row / tap shifts are immediates, exactly as in the stride-1 kernel.  Against the implicit GEMM (conv_h2r) on these layers: half the
// staged elements (the im2col tile holds every input element 2.25 times), no global A traffic per k-step, one barrier per slab instead of
// one per k-step.
// NW = 4: 128 x 64 tile, waves 2 (M) x 2 (N), five staging rounds per slab.  NW = 8 (512 threads): 128 x 128 tile, waves 2 (M) x 4 (N)
// with the SAME 64 x 32 wave tile -- one patch shared by eight waves (a four-wave 128-wide tile needs a 64 x 64 wave tile, which does not
// fit the register file next to the five-round staging): half the staging work and half the L2 -> L1 activation traffic per MFMA, three
// staging rounds per slab.  Both 4-row shapes hold 80 KiB of LDS (two stages of a 9 x 68-slot patch): ONE workgroup per CU, two waves per
// SIMD -- their MFMA pipe is 0.20 busy.  PR = 2 (NW = 4, BN = 128): a 2 x 32 output rectangle, waves 1 (M) x 4 (N), patch 5 x 65: 44 KiB and
// 168 VGPRs -> THREE workgroups per CU (twelve waves), three staging rounds; 11 % more patch pixels per output than the 4-row tile.
// K order, chains and fold points are h2_tile's in every shape: bit-identical results.
// DEEP (two-row tile only; round 6): the schedule of a launch that cannot fill the chip -- a single frame's 64 .. 384 workgroups, one or two per
// CU.  With weight fragments two steps ahead and each staging round fetched two steps before it is written, a lone workgroup pays one memory round
// trip per STEP (72 us for 16 slabs on 64 CUs: ~1000 cycles per step of 6 MFMAs); three co-resident workgroups hide that for each other, one does
// not.  DEEP keeps the weight fragments of eight steps in flight (nine register sets: the tap index is the set index) and fetches all three
// staging rounds of the next slab in its first three steps, writing them in its last three: ~230 VGPRs, two workgroups per CU.  Same
// arithmetic, same bits.
template <int BN, int NWV, int NPROD, bool AFFINE, int PR = kPatchRows, bool DEEP = false>
__device__ __forceinline__ void h2d_tile(const ConvArgs& a, unsigned char* smem_raw, const int tile_m, const int n0) {
    constexpr int BM = PR * kPatchCols;
    constexpr int WARPS_M = PR / 2, WARPS_N = NWV / WARPS_M;
    static_assert((NWV == 4 || NWV == 8) && (PR == 2 || PR == 4), "four or eight waves, two output rows per wave");
    static_assert(NPROD == 1 || NPROD == 3, "one (bf16 operands) or three products");
    constexpr int NPL = NPROD == 1 ? 1 : 2;
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 32, NTL = WN / 32;
    static_assert(MT == 2 && NTL == 1, "wave tile 64 x 32");
    constexpr int PCI = 2 * kPatchCols + 1, PRI = 2 * PR + 1, PP = PRI * PCI;              // 65 x 9 = 585 (65 x 5 = 325) patch pixels
    constexpr int RP = 68, ODD0 = 36;                                // row pitch in slots: 33 even columns, pad, 32 odd columns from slot 36 on: the 8 lanes of a
                                                                     // ds_write_b128 group (4 even + 4 odd pixels) then hit disjoint banks (36 * 16 B = 16 banks mod 32)
    constexpr int REGION = PRI * RP * 16;                            // one octet region: 594 slots x 16 B
    constexpr int PLANE_P = 2 * REGION, PATCH_BYTES = NPL * PLANE_P;
    constexpr int OFF_SINK = 2 * 2 * PLANE_P;                        // 2 KiB sink for pixel slots that do not exist (branch-free staging)
    constexpr int OFF_TAB = OFF_SINK + 2048;
    constexpr int NBLK = (PP + 31) / 32;                             // 19 (11) blocks of 32 pixels
    constexpr int NR = (NBLK + NWV - 1) / NWV;                       // staging rounds: 5 (four waves, four rows) or 3 (eight waves; two rows)
    static_assert(NR == 5 || NR == 3, "staging schedules exist for five and three rounds");
    static_assert(!DEEP || NR == 3, "the deep schedule is the two-row tile's");
    constexpr int BD = DEEP ? 9 : 3;                                 // weight register sets: fragments are fetched BD - 1 steps ahead

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = TSNET_UNIFORM(tid >> 6);
    const int wrow = wave / WARPS_N;
    const int wn0 = (wave % WARPS_N) * WN;
    const int li = lane & 31, lh = lane >> 5;

    const int tcols = a.Wo / kPatchCols, tper = (a.Ho / PR) * tcols;
    const int img = tile_m / tper, tin = tile_m - img * tper;
    const int oy0 = (tin / tcols) * PR, ox0 = (tin % tcols) * kPatchCols;
    const int ncc = a.Cin >> 4;
    float in_scale = a.in_scale, in_unscale = a.in_unscale;
    if (NPROD != 1 && a.in_amax) h2_device_scale(a.in_amax + img, a.in_bound_add, in_scale, in_unscale);

    const size_t planew = (size_t)((a.nchunks + 1) / 2 * 2) * a.Npad * 16;
    const bool xb16 = NPROD == 1 && a.x_bf16;                        // bf16 storage: the input tensor holds bf16
    const tsnet_brsrc_t rsx = tsnet_make_brsrc(a.x, (unsigned)((size_t)a.N * a.H * a.W * a.Cin * (xb16 ? 2 : 4)));
    tsnet_brsrc_t rsw[NPL];
#pragma unroll
    for (int p = 0; p < NPL; ++p) rsw[p] = tsnet_make_brsrc(a.w + p * planew, (unsigned)(planew * 2));

    // ---- staging geometry: round r of this wave = pixel block wave + NWV r (slots past the patch go to the sink: no branch in the K loop);
    //      lane -> (pixel b*32 + (lane & 31), octet lane >> 5)
    const int oct = lane >> 5;
    unsigned vP[NR];
    float vM[NR];
    auto slot_of = [&](int r) __attribute__((always_inline)) {       // LDS byte offset of the lane's slot inside an octet region (recomputed, not kept), or -1
        const int pp = (wave + NWV * r) * 32 + (lane & 31);
        const int pr = pp / PCI, pc = pp - pr * PCI;
        return pp < PP ? (pr * RP + ((pc & 1) ? ODD0 + (pc >> 1) : (pc >> 1))) * 16 : -1;
    };
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int pp = (wave + NWV * r) * 32 + (lane & 31);
        const int pr = pp / PCI, pc = pp - pr * PCI;
        const int iy = 2 * oy0 - 1 + pr, ix = 2 * ox0 - 1 + pc;
        const bool ok = pp < PP && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;       // zero padding
        vP[r] = ok ? (unsigned)(((img * a.H * a.W + iy * a.W + ix) * a.Cin + oct * 8) * 4) : kOOB;
        vM[r] = ok ? 1.f : 0.f;
    }
    float* tab = reinterpret_cast<float*>(smem_raw + OFF_TAB);       // [Cin] alpha*s, then [Cin] beta*s
    if (AFFINE) {
        for (int c = tid; c < a.Cin; c += 64 * NWV) {
            tab[c] = a.in_alpha[(size_t)img * a.Cin + c] * in_scale;
            tab[a.Cin + c] = a.in_beta[(size_t)img * a.Cin + c] * in_scale;
        }
        __syncthreads();
    }
    const float relu_floor = a.in_relu ? 0.f : -__builtin_inff();
    F4 sx[DEEP ? NR : 1][2];                                         // staging registers: one round of x in flight (DEEP: all three)
    auto stage_load_x = [&](int cn, int r) __attribute__((always_inline)) {
        F4 (&sxr)[2] = sx[DEEP ? r : 0];
        if (NPROD == 1) load_x_octet(rsx, xb16, vP[r], (unsigned)(cn * 64), sxr);
        else {
#pragma unroll
            for (int q = 0; q < 2; ++q) sxr[q] = TSNET_BUF_LOAD16(rsx, vP[r], (unsigned)(cn * 64 + q * 16));
        }
    };
    auto stage_store = [&](int cn, int r) __attribute__((always_inline)) {
        F4 t[2];
        const float* ta = tab + (cn < ncc ? cn * 16 : 0) + oct * 8;
        transform_octet<AFFINE>(sx[DEEP ? r : 0], ta, a.Cin, in_scale, relu_floor, vM[r], t);
        const int so = slot_of(r);
        const bool real = so >= 0;
        unsigned char* dst = smem_raw + (real ? (cn & 1) * PATCH_BYTES + oct * REGION + so : OFF_SINK + (lane & 63) * 16);
        F4 Hh, Ll;
        if (NPROD == 1) {
            bf16_octet(t[0], t[1], Hh);
            *reinterpret_cast<F4*>(dst) = Hh;
        } else {
            split_h2_octet(t[0], t[1], Hh, Ll);
            *reinterpret_cast<F4*>(dst) = Hh;
            *reinterpret_cast<F4*>(dst + (real ? PLANE_P : 1024)) = Ll;
        }
    };

    // ---- fragments
    const unsigned vB = (unsigned)((wn0 + li) * 32 + (lh ^ ((li >> 3) & 1)) * 16);
    F4 af[2][NPL][MT], bf[BD][NPL][NTL];
    const int wB = a.Npad * 32, wT = ncc * wB;                       // (the running weight offset of h2_tile: one add per step instead of a re-derivation)
    int wsl = n0 * 32;
    auto load_b = [&](int set, int dslab, int t) __attribute__((always_inline)) {
        const int soff = wsl + dslab * wB + t * wT;
#pragma unroll
        for (int p = 0; p < NPL; ++p)
#pragma unroll
            for (int j = 0; j < NTL; ++j) bf[set][p][j] = TSNET_BUF_LOAD16(rsw[p], vB + j * 1024u, (unsigned)soff);      // (+ 1024 j rides in the instruction's immediate offset)
    };
    const unsigned char* abase = smem_raw + lh * REGION + (2 * wrow * MT * RP + li) * 16;
    auto load_a = [&](int set, int cc, int t) __attribute__((always_inline)) {
        const int ky = t / 3, kx = t - ky * 3;
        const unsigned char* pbase = abase + (cc & 1) * PATCH_BYTES;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                af[set][p][i] = *reinterpret_cast<const F4*>(pbase + p * PLANE_P + ((2 * i + ky) * RP + (kx == 1 ? ODD0 : (kx >> 1))) * 16);
    };

    f32x16 acc[MT][NTL], tot[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }
    auto product = [&](int sa, int sb, int pa, int pb, bool fresh) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                f32x16 c = acc[i][j];
                if (fresh) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c[r] = 0.f;
                }
                if (NPROD == 1) acc[i][j] = TSNET_MFMA_BF16(af[sa][pa][i], bf[sb][pb][j], c);
                else acc[i][j] = TSNET_MFMA_F16(af[sa][pa][i], bf[sb][pb][j], c);
            }
    };
    // staging of slab cc + 1 over the nine taps of slab cc, one round in flight at a time:
    //   five rounds: load at taps 0 1 3 4 6, store at taps 1 3 4 6 8;   three rounds: load at taps 0 3 6, store at taps 2 5 8
    //   (a store precedes the next load issued in its tap)
    auto step = [&](int cc, int t, int SA) __attribute__((always_inline)) {
        const bool fresh = t == 0 || t == 4;
        const int t2 = (t + BD - 1) % 9;
        load_b(t2 % BD, t + BD - 1 >= 9 ? 1 : 0, t2);
        if (t < 8) load_a(SA ^ 1, cc, t + 1);
        if (DEEP) {
            if (t < 3) stage_load_x(cc + 1, t);
        } else if (NR == 5) {
            if (t == 1) stage_store(cc + 1, 0);
            if (t == 3) stage_store(cc + 1, 1);
            if (t == 4) stage_store(cc + 1, 2);
            if (t == 6) stage_store(cc + 1, 3);
            if (t == 0) stage_load_x(cc + 1, 0);
            if (t == 1) stage_load_x(cc + 1, 1);
            if (t == 3) stage_load_x(cc + 1, 2);
            if (t == 4) stage_load_x(cc + 1, 3);
            if (t == 6) stage_load_x(cc + 1, 4);
        } else {
            if (t == 0) stage_load_x(cc + 1, 0);
            if (t == 3) stage_load_x(cc + 1, 1);
            if (t == 6) stage_load_x(cc + 1, 2);
        }
        const int SB = t % BD;
        if (NPROD == 1) {
            product(SA, SB, 0, 0, fresh);
        } else {
            product(SA, SB, NPL - 1, 0, fresh);                      // lo * hi
            product(SA, SB, 0, NPL - 1, false);                      // hi * lo
            product(SA, SB, 0, 0, false);                            // hi * hi
        }
        if (DEEP) {
            if (t >= 6) stage_store(cc + 1, t - 6);
        } else if (NR == 5) {
            if (t == 8) stage_store(cc + 1, 4);
        } else {
            if (t == 2) stage_store(cc + 1, 0);
            if (t == 5) stage_store(cc + 1, 1);
            if (t == 8) stage_store(cc + 1, 2);
        }
        if (t == 3 || t == 8) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NTL; ++j) tot[i][j] += acc[i][j];
        }
    };
    auto slab = [&](int cc, int S0) __attribute__((always_inline)) {
        __syncthreads();                                             // patch(cc) complete and visible; slab cc-1 fully read
        load_a(S0, cc, 0);
        step(cc, 0, S0); step(cc, 1, S0 ^ 1); step(cc, 2, S0);
        step(cc, 3, S0 ^ 1); step(cc, 4, S0); step(cc, 5, S0 ^ 1);
        step(cc, 6, S0); step(cc, 7, S0 ^ 1); step(cc, 8, S0);
        wsl += wB;
    };

    // prologue: patch of slab 0, weight fragments of steps (0,0) and (0,1)
#pragma unroll
    for (int r = 0; r < NR; ++r) { stage_load_x(0, r); stage_store(0, r); }
#pragma unroll
    for (int i = 0; i < BD - 1; ++i) load_b(i, 0, i);
    int cc = 0;
    for (; cc + 2 <= ncc; cc += 2) { slab(cc, 0); slab(cc + 1, 1); }
    if (cc < ncc) slab(cc, 0);

    const float unscale = a.w_unscale ? in_unscale * a.w_unscale[0] : in_unscale;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) tot[i][j][r] *= unscale;
    const int m_img = img * a.Ho * a.Wo;
    __syncthreads();                                                 // the epilogue reuses the patch region
    conv_epilogue<BN, WARPS_M, WARPS_N, MT, NTL>(a, tot, smem_raw, tid, wave, n0, (size_t)img * tper + tin,
                                                 [&](int l) { return m_img + (oy0 + (l >> 5)) * a.Wo + ox0 + (l & 31); });
}

// two stages x two planes x two octet regions + sink (+ 2 Cin floats x 2 of the table, added by the launcher)
constexpr int h2d_lds_bytes(int PR) { return 2 * 2 * 2 * (2 * PR + 1) * 68 * 16 + 2048; }

template <int BN, int NWV, int NPROD, bool AFFINE, int PR = kPatchRows, bool DEEP = false>
__global__ __launch_bounds__(64 * NWV, DEEP ? 2 : ((NWV == 8 || PR == 2) ? 3 : 2))  // waves per SIMD: eight waves = 1.5 workgroups' worth; two rows: three workgroups of four (DEEP: two)
void conv_h2d_kernel(ConvArgs a) {
    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    const int bid = xcd_item(blockIdx.x, a.tiles_m * a.tiles_n);
    const int tile_m = bid / a.tiles_n;
    h2d_tile<BN, NWV, NPROD, AFFINE, PR, DEEP>(a, smem_raw, tile_m, (bid - tile_m * a.tiles_n) * BN);
}

}  // namespace tsnet
