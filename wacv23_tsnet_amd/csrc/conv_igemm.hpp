// conv_igemm.hpp -- NHWC implicit-GEMM convolution on the exact-fp32 MFMA (gfx950 / CDNA4).
//
// Replaces on the reference hot path: every nn.Conv2d together with the nn.ReflectionPad2d /
// zero padding in front of it and the consumer side of the nn.InstanceNorm2d + nn.ReLU that
// precedes it (model/TSNet.py:27,42 ResnetBlock; :66,70 Encoder; :139,147,152 Decoder; :193 FuseNet).
//
// GEMM view:  Y[m][n] = sum_k A[m][k] * Wt[k][n],  m = (img, oy, ox), n = cout, k = (ky, kx, cin).
//   * A is never materialised (im2col-free): the loader computes the NHWC address of every
//     16-byte (4-channel) unit, applies reflection / zero padding in the address, and applies the
//     producer's InstanceNorm + ReLU as one FMA + max on the way into LDS (x' = max(alpha*x+beta,0)).
//     A second source tensor supplies the upper channel range (torch.cat on the channel axis,
//     TSNet.py:163,196) so concatenations are never written to HBM.
//   * Wt is pre-packed once at load time as [K/4][Npad][4] so that a (k-quad, 32 cout) MFMA
//     operand is one 16-byte LDS read per lane.
//   * v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulate (bitwise an fmaf chain) --
//     required by the 1e-3 output parity budget.  Each lane holds
//     A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; reading one float4 of 4 consecutive k per lane feeds
//     4 MFMAs (k-order inside the sum is permuted consistently for A and B).
//   * LDS tiles are [BK/4][rows+pad] float4: the row pad makes both the 8-lane-group
//     ds_write_b128 of the loader and the 16-lane-group ds_read_b128 of the MFMA feed
//     conflict-free (MI355X_MICROARCH.md "LDS").  Two stages, one barrier per K-chunk; the global
//     loads of chunk c+1 are issued before the MFMAs of chunk c and land in registers meanwhile.
#pragma once
#include <hip/hip_runtime.h>

namespace tsnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* x;         // source 0, NHWC (N,H,W,Csplit)
    const float* x2;        // source 1 (channels >= Csplit), NHWC (x2_nmod,H,W,Cin-Csplit); may be null
    const float* in_alpha;  // (N*Cin) or null: x' = alpha*x+beta
    const float* in_beta;
    const float* w;         // packed [Kpad/4][Npad][4]
    const float* bias;      // (Cout) or null
    float* y;               // NHWC (N,Ho,Wo,Cout)  or NCHW (N,Cout,Ho,Wo) when out_nchw
    int N, H, W, Cin;       // Cin = total (padded, power of two) input channels
    int cin_log2;
    int Csplit;             // channels taken from x (== Cin when x2 is null)
    int x2_nmod;            // image index into x2 = n % x2_nmod
    int Ho, Wo, Cout, Npad;
    int stride, pad, reflect;
    int taps;               // ks*ks
    int nchunks;            // ceil(taps*Cin / BK)
    int M;                  // N*Ho*Wo
    int in_relu;
    int act;                // 0 none, 1 tanh
    int out_nchw;
    int composite;          // pose epilogue: out = fore ? v : bg[c]  (TSNet_pose.py:416-417)
    int fore_x0, fore_x1;
    float bg[3];
    int tiles_m, tiles_n;
};

struct alignas(16) F4 { float v[4]; };

__device__ __forceinline__ F4 ld4(const float* p) { return *reinterpret_cast<const F4*>(p); }

template <int KS, int BM, int BN, int BK, int WARPS_M, int WARPS_N, int FLUSH>
__global__ __launch_bounds__(64 * WARPS_M * WARPS_N)
void conv_igemm_kernel(ConvArgs a) {
    constexpr int NT = 64 * WARPS_M * WARPS_N;
    constexpr int KQ = BK / 4;                 // float4 units along K per chunk
    constexpr int WM = BM / WARPS_M, WN = BN / WARPS_N;
    constexpr int MT = WM / 32, NTL = WN / 32;
    constexpr int UA = (BM * KQ + NT - 1) / NT;  // A units per thread (last one predicated when BM*KQ % NT != 0)
    constexpr int UB = (BN * KQ + NT - 1) / NT;  // B units per thread (last one predicated when BN*KQ < NT)
    constexpr int RSTEP = NT / KQ;             // row step between a thread's A units
    static_assert(BM % (32 * WARPS_M) == 0 && BN % (32 * WARPS_N) == 0, "tile/wave mismatch");
    constexpr int PAD = 8 / KQ;                // row pad (in float4) that keeps ds_write_b128 8-lane groups on distinct slots
    constexpr int LDA = BM + PAD, LDB = BN + PAD;  // float4 row pitch

    HIP_DYNAMIC_SHARED(__attribute__((aligned(16))) unsigned char, smem_raw)
    F4* sA = reinterpret_cast<F4*>(smem_raw);                 // [2][KQ][LDA]
    F4* sB = sA + 2 * KQ * LDA;                               // [2][KQ][LDB]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm0 = (wave / WARPS_N) * WM;
    const int wn0 = (wave % WARPS_N) * WN;
    const int li = lane & 31, lh = lane >> 5;

    // XCD-aware tile id: consecutive tile ids (n fastest, sharing an A tile) stay on one XCD's L2.
    const int ntiles = a.tiles_m * a.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = ntiles >> 3, r = ntiles & 7, xcd = bid & 7, loc = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tile_m = bid / a.tiles_n, tile_n = bid - tile_m * a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread A-unit geometry (fixed over the K loop)
    const int ak4 = tid % KQ;
    int a_img[UA], a_img2[UA], a_oy[UA], a_ox[UA];
    bool a_rowok[UA];
#pragma unroll
    for (int j = 0; j < UA; ++j) {
        const int row = tid / KQ + j * RSTEP;
        const int m = m0 + row;
        a_rowok[j] = m < a.M && row < BM;
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
    const bool has_norm = a.in_alpha != nullptr;

    F4 ra[UA], rb[UB], ral[UA], rbe[UA];
    bool rok[UA];

    // Loads are unconditional (out-of-range units read a clamped, valid address and are zeroed at
    // store time) so the K loop stays one basic block and the accumulators never leave their
    // registers; the only branches are wave-uniform (source select, has_norm).
    auto load_chunk = [&](int kc) {
        const int k = kc * BK + ak4 * 4;
        const int tap = k >> a.cin_log2;
        const int c = k & (a.Cin - 1);
        const int ky = tap / KS, kx = tap - ky * KS;
        const bool tapok = tap < a.taps;
        // a chunk never straddles the channel split (Csplit is a multiple of BK whenever x2 is set)
        const bool second = ((kc * BK) & (a.Cin - 1)) >= a.Csplit;
        const float* src = second ? a.x2 : a.x;
        const int cs = second ? C2 : a.Csplit;
        const int coff = second ? c - a.Csplit : c;
#pragma unroll
        for (int j = 0; j < UA; ++j) {
            int iy = a_oy[j] + ky, ix = a_ox[j] + kx;
            bool ok = a_rowok[j] && tapok;
            if (a.reflect) {
                iy = iy < 0 ? -iy : iy;
                iy = iy >= a.H ? 2 * (a.H - 1) - iy : iy;
                ix = ix < 0 ? -ix : ix;
                ix = ix >= a.W ? 2 * (a.W - 1) - ix : ix;
            } else {
                ok = ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            }
            iy = iy < 0 ? 0 : (iy >= a.H ? a.H - 1 : iy);     // clamp: keeps padded-K / ragged-M loads in bounds
            ix = ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix);
            rok[j] = ok;
            const int img = second ? a_img2[j] : a_img[j];
            ra[j] = ld4(src + ((size_t)((img * a.H + iy) * a.W + ix)) * cs + coff);
            if (has_norm) {
                ral[j] = ld4(a.in_alpha + a_img[j] * a.Cin + c);
                rbe[j] = ld4(a.in_beta + a_img[j] * a.Cin + c);
            }
        }
        // B: packed weights, fully coalesced
#pragma unroll
        for (int j = 0; j < UB; ++j) {
            const int idx = tid + j * NT;
            if (idx < BN * KQ) {
                const int k4 = idx / BN, n = idx - k4 * BN;
                rb[j] = ld4(a.w + ((size_t)(kc * KQ + k4) * a.Npad + n0 + n) * 4);
            }
        }
    };

    auto store_chunk = [&](int buf) {
        F4* dA = sA + buf * KQ * LDA;
        F4* dB = sB + buf * KQ * LDB;
#pragma unroll
        for (int j = 0; j < UA; ++j) {
            F4 v = ra[j];
            if (has_norm) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = __builtin_fmaf(v.v[e], ral[j].v[e], rbe[j].v[e]);
                    v.v[e] = a.in_relu ? (t > 0.f ? t : 0.f) : t;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v.v[e] = rok[j] ? v.v[e] : 0.f;
            if ((BM * KQ) % NT == 0 || tid / KQ + j * RSTEP < BM) dA[ak4 * LDA + tid / KQ + j * RSTEP] = v;
        }
#pragma unroll
        for (int j = 0; j < UB; ++j) {
            const int idx = tid + j * NT;
            if (idx < BN * KQ) {
                const int k4 = idx / BN, n = idx - k4 * BN;
                dB[k4 * LDB + n] = rb[j];
            }
        }
    };

    // Two-level accumulation: the MFMA chain `acc` is folded into `tot` every FLUSH chunks
    // (FLUSH*BK products), so no fp32 fmaf chain is longer than that and the rounding error grows
    // like sqrt(FLUSH*BK)+sqrt(K/(FLUSH*BK)) instead of sqrt(K) (K is up to 9216 here).
    f32x16 acc[MT][NTL], tot[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; tot[i][j][r] = 0.f; }

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    for (int kc = 0; kc < a.nchunks; ++kc) {
        const int buf = kc & 1;
        const bool more = kc + 1 < a.nchunks;
        if (more) load_chunk(kc + 1);

        const F4* cA = sA + buf * KQ * LDA;
        const F4* cB = sB + buf * KQ * LDB;
#pragma unroll
        for (int k8 = 0; k8 < KQ / 2; ++k8) {
            F4 af[MT], bf[NTL];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = cA[(k8 * 2 + lh) * LDA + wm0 + i * 32 + li];
#pragma unroll
            for (int j = 0; j < NTL; ++j) bf[j] = cB[(k8 * 2 + lh) * LDB + wn0 + j * 32 + li];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NTL; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].v[e], bf[j].v[e], acc[i][j], 0, 0, 0);
        }

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

        if (more) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias, activation, store.  D layout: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
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

// Re-pack OIHW fp32 weights into the kernel's [Kpad/4][Npad][4] layout, k = tap*Cin_pad + c.
// Channels c >= cin_real and couts >= cout are zero (channel padding of the 8/32-wide stems and
// of the 3-wide RGB head).
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ out,
                                    int cout, int cin_real, int cin_pad, int ks, int kpad, int npad, int cin_total, int cin_off) {
    const size_t total = (size_t)kpad * npad;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = idx & 3;
        const size_t q = idx >> 2;
        const int n = (int)(q % npad);
        const int k = (int)(q / npad) * 4 + e;
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
