// flow_args.hpp -- what the two flow kernels (flow_warp.hpp: flow_kernel; flow_persist.hpp: flow_kernel_p) and the host share: operand
// plane geometry, the argument block, LDS sizes and the plan that picks the kernel.  No kernels here: included by two translation units.
#pragma once
#include <cstddef>

namespace tsnet {

constexpr int kFlowWaves = 8;
constexpr float kFlowScale = 16384.0f;                       // 2^14: |v| <= 1 -> |hi| <= 2^14, lo keeps 11 more bits down to |v| ~ 2e-5
constexpr float kFlowUnscale = 1.0f / (16384.0f * 16384.0f);
inline int flow_ppad(int P) { return (P + 63) / 64 * 64; }
inline int flow_ksteps(int C) { return (C + 31) / 32 * 2; }  // 16-channel steps, padded to an even count
inline size_t flow_plane_halves(int N, int P, int C) { return (size_t)N * flow_ppad(P) * flow_ksteps(C) * 16 * 2; }
// LDS bytes of flow_kernel<NT>: max(target planes, merge buffer) + the source-mask row
inline size_t flow_lds_bytes(int NT, int P, int C) {
    const size_t t = (size_t)NT * flow_ksteps(C) * 2048, r = (size_t)2 * kFlowWaves * NT * 32 * 16;
    return (t > r ? t : r) + (size_t)flow_ppad(P) * 4;
}
inline size_t flow_lds_bytes(int NT, int h, int w, int C) { return flow_lds_bytes(NT, h * w, C) + (size_t)(((w + 3) & ~3) + ((h + 3) & ~3)) * 4; }


struct FlowArgs {
    const unsigned short* tq; // target planes of B images (l2norm_split_kernel)
    const unsigned short* sq; // source planes of NB images, n = s*B + b
    const float* tar_bbox;    // (B, H, W)
    const float* src_bbox[8]; // per source (B, H, W)
    const float* gx;          // (w) linspace(-1,1,w)
    const float* gy;          // (h)
    float* flow;              // (NB, P, 2)
    int B, P, C, h, w, H, W, sy, sx;
    // flow_kernel_p only
    unsigned long long* part; // [NB][tiles][S][64][2]: (max, sum) and (x, y) of a slice's softmax state, two floats per 8-byte word
    int* cnt;                 // [B][tiles] arrival counters, zero between launches
    int K, G, S;              // sources per batch element; workgroups per target tile; slices per source image (flowp_slices: S % G == 0)
};

// LDS bytes of flow_kernel_p (64 targets per workgroup): target planes, two merge buffers [kFlowWaves][64][4] floats, a 64-float mask row
// per wave, gx, gy
inline size_t flowp_lds_bytes(int h, int w, int C) {
    return (size_t)2 * flow_ksteps(C) * 2048 + (size_t)2 * kFlowWaves * 64 * 16 + (size_t)kFlowWaves * 256 + (size_t)(((w + 3) & ~3) + ((h + 3) & ~3)) * 4;
}
// SLICES of a source image in flow_kernel_p: a function of the MAP alone (the largest power of two <= 8 that leaves every wave a whole
// source pair per slice).  The reduction tree of a target column -- waves of a slice in wave order, then the slices in slice order -- is
// built on them, so a frame's flow is the same bits in any batch (it used to be built on G, which depends on the batch).
inline int flowp_slices(int h, int w) {
    const int npair = (h * w) / 64;
    int S = 1;
    while (S < 8 && npair % (S * 2 * kFlowWaves) == 0) S *= 2;
    return S;
}
// Workgroups per target tile of flow_kernel_p (one workgroup per CU when the target tiles alone do not fill the chip; a workgroup sweeps
// S / G slices of every source), or 0 where the form does not apply: small maps (flow_kernel fills the chip there), ragged maps, LDS.
inline int flowp_plan(int B, int h, int w, int C) {
    const int P = h * w;
    if (P < 2048 || P % 64 || w % 4) return 0;
    const int npair = P / 64, tiles = B * npair, S = flowp_slices(h, w);
    int G = 1;
    while (G < S && tiles * G * 2 <= 256) G *= 2;
    if ((npair / S) % kFlowWaves || flowp_lds_bytes(h, w, C) > (size_t)160 * 1024) return 0;
    return G;
}
inline size_t flowp_part_words(int NB, int P, int S) { return (size_t)NB * (P / 64) * S * 64 * 2; }   // 8-byte words: one state per (image, tile, slice, column)

}  // namespace tsnet
