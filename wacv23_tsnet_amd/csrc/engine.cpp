// engine.cpp -- host side of the TS-Net forward engine behind include/tsnet_abi.h.
//
// Compiled with hipcc for gfx950 (the product) and, by tests only, with a host compiler against tests/emu's HIP emulation headers so
// that the launch geometry / indexing logic of every kernel can be checked against the oracle without a GPU.  There is no CPU fallback
// in the product: every entry point launches HIP kernels and reports HIP failures through tsnet_last_error.
//
// ONE schedule (reference: model/TSNet.py:309-407), NHWC fp32 activations throughout, the K sources batched into one launch per layer
// (image index n = s*B + b):
//   pack_input -> img_enc (stem 7x7, 3 stride-2 convs, 9 ResnetBlocks)           [A2, A4]
//   pack_input -> lbl_enc (stem + 3 stride-2 convs)                              [A3]      (side stream)
//   l2norm x2 -> flow_kernel -> warp_mean_kernel                                 [A5, A6]  (side stream)
//   fuse conv1 (per-source half + shared target half) -> conv2 -> residual+mean -> 1x1 conv   [A7, A6]
//   dec map 1x1 (cat on load) -> ResnetBlocks -> 3x (upsample, conv) -> 7x7+tanh [A8, A9]
// Every convolution reads the producer's RAW fp32 output and applies that producer's InstanceNorm + ReLU while it stages its operand
// tile; operands enter the matrix pipe as fp16 x 2 splits (three products) or, with tsnet_cfg.operand_mode = 1, as one bf16 plane.
// Which kernel a layer runs on (the Winograd-along-x kernel of conv_w1.hpp for the ResnetBlock / FuseNet layers and the first two
// up-convolutions, the patch kernels of conv_h2.hpp where the output splits into 4 x 32 rectangles, the general implicit GEMM of
// conv_g64.hpp / conv_h2r.hpp elsewhere) and which tile it takes depend on the layer and the frame size only -- never on the batch, never
// on the environment.  Two launch parameters DO follow the batch -- conv_w1's tiles per workgroup and nothing else -- and are bit-neutral by
// construction (the same chains per output element in every chunk size): a frame's result is the same bits alone (B = 1) and in any batch
// (tests/test_gpu_forward.py::test_single_frame_forward).
// InstanceNorm statistics: fp64 partial sums in the producing conv's epilogue, finalised there by the last-arriving workgroup
// (or by in_finalize2 when an image has many tiles).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/tsnet_abi.h"
#include "kernels.hpp"
#include "lmfit.hpp"
#include "flow_warp.hpp"
#include "head_conv.hpp"
#include "norm_elementwise.hpp"
#include "pack_weights.hpp"
#include "postproc.hpp"
#include "raster.hpp"
#include "train_extras.hpp"

using namespace tsnet;

namespace {

thread_local std::string g_op_error;
int64_t g_launch_counters[4] = {0, 0, 0, 0};   // conv launches: [0] patch kernels (conv_h2.hpp), [1] general kernel (conv_h2r.hpp), [2] RGB head;
                                               // [3] = tile code (rows * 1000 + width, + 20000 two K groups) of the last ResnetBlock-class convolution launch
std::string g_create_error;

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess) {                                                                    \
            char buf__[512];                                                                        \
            snprintf(buf__, sizeof buf__, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            throw std::runtime_error(buf__);                                                        \
        }                                                                                           \
    } while (0)

#ifdef TSNET_TOOLS
int g_tools_knob[8] = {3, 0, 0, 0, 0, 0, 0, 0};     // tools build only (tsnet_tools_set): [0] largest conv_w1 chunk the launcher may choose
#endif
struct ArgError : std::runtime_error { using std::runtime_error::runtime_error; };
struct WeightError : std::runtime_error { using std::runtime_error::runtime_error; };

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
// CUs of the current device (MI355X: 256 in 8 XCDs -- the XCD count is part of the kernels' block -> tile maps, conv_common.hpp; the CU count
// only enters launch heuristics: how many tiles fill the chip in whole rounds).  Asked once; a device that reports nothing counts as 256.
inline int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v < 8) v = 256;
        cus = v;
    }
    return cus;
}
inline int ilog2(int x) { int l = 0; while ((1 << l) < x) ++l; return l; }
inline int next_pow2(int x) { int p = 4; while (p < x) p <<= 1; return p; }

// ------------------------------------------------------------------------------------------------
// timing of launch classes (bench.py's roofline object)
struct Timing {
    bool on = false;
    struct Rec { int cls; hipEvent_t a, b; };
    std::vector<Rec> recs;
    double ms[TSNET_TIMING_CLASSES] = {0};
    int64_t launches[TSNET_TIMING_CLASSES] = {0};
    void begin(int cls, hipStream_t s) {
        if (!on) return;
        Rec r{cls, nullptr, nullptr};
        HIP_TRY(hipEventCreate(&r.a));
        HIP_TRY(hipEventCreate(&r.b));
        HIP_TRY(hipEventRecord(r.a, s));
        recs.push_back(r);
    }
    void end(hipStream_t s) {
        if (!on) return;
        HIP_TRY(hipEventRecord(recs.back().b, s));
    }
    void collect() {
        for (auto& r : recs) {
            HIP_TRY(hipEventSynchronize(r.b));
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
            ms[r.cls] += t;
            launches[r.cls] += 1;
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
        recs.clear();
    }
};

struct Ctx {            // what a launch helper needs
    hipStream_t stream = nullptr;
    Timing* timing = nullptr;
    int lane = 0;       // 0 = the caller's stream; 1 = the engine's side stream (own statistics scratch, see tsnet_forward)
};

// Scope guard of a fork onto the engine's side stream: if the scope is left before the explicit join (an exception between fork
// and join), the caller's stream still waits for whatever the side lane has in flight.
struct SideJoin {
    hipStream_t side, main; hipEvent_t ev; bool done = false;
    ~SideJoin() { if (!done && side && hipEventRecord(ev, side) == hipSuccess) (void)hipStreamWaitEvent(main, ev, 0); }
};

struct TimeScope {
    Ctx& c;
    TimeScope(Ctx& c_, int cls) : c(c_) { if (c.timing) c.timing->begin(cls, c.stream); }
    ~TimeScope() { if (c.timing) { try { c.timing->end(c.stream); } catch (...) {} } }
};

inline void check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + " launch failed: " + hipGetErrorString(e));
}

inline int ew_grid(size_t total, int block = 256) {
    size_t g = (total + block - 1) / block;
    if (g > 256 * 8) g = 256 * 8;    // grid-stride above ~8 blocks per CU
    if (g < 1) g = 1;
    return (int)g;
}

void launch_finalize_impl(const double* part, float* alpha, float* beta, int N, int C, int S, int HW, hipStream_t s);
inline void launch_finalize(const double* part, float* alpha, float* beta, int N, int C, int S, int HW, hipStream_t s) { launch_finalize_impl(part, alpha, beta, N, C, S, HW, s); }

// pack_input_kernel: grid.x blocks per image (grid.y = images), sized so the whole grid stays near 8 blocks per CU
inline int pack_grid(int hw) {
    int g = (hw + 255) / 256;
    if (g > 128) g = 128;
    return g < 1 ? 1 : g;
}
// ------------------------------------------------------------------------------------------------
// convolution layer description + launch
struct ConvLayer {
    std::string name;       // state_dict prefix, e.g. "img_enc.model.1"
    std::string wparam, bparam;   // parameter names feeding this layer (bparam empty = no bias)
    int cin_real = 0, cin_pad = 0, cout = 0, ks = 1, stride = 1, pad = 0, reflect = 0;
    int cin_total = 0, cin_off = 0;   // window of the parameter's input channels this layer consumes
    int kpad = 0, npad = 0;
    int form = 0;                     // 0: the filter as it is (ks x ks taps); 1: its Winograd F(2,3)-along-x transform, 3 x 4 "taps" (conv_w1.hpp)
    size_t w_off = 0, b_off = 0;          // offsets into the packed buffer: operand planes (16-bit units from the plane section), bias (floats)
    const unsigned short* wq = nullptr;   // device: operand planes [planes][K/16][Npad][2 octets][8] (conv_common.hpp pack_weights_kernel)
    const float* w_unscale = nullptr;     // device scalar 2^-sw (in the packed buffer: replicas receive it with the broadcast); null in bf16 mode
    const float* bias = nullptr;          // device (cout)
};

// device allocations of an operator entry point: freed on every exit path (a rejected tile / variant throws out of run_conv routinely)
struct DevBufs {
    std::vector<void*> p;
    template <typename T> T* alloc(size_t bytes) { void* q = nullptr; HIP_TRY(hipMalloc(&q, bytes)); p.push_back(q); return static_cast<T*>(q); }
    ~DevBufs() { for (void* q : p) (void)hipFree(q); }
    DevBufs() = default;
    DevBufs(const DevBufs&) = delete;
    DevBufs& operator=(const DevBufs&) = delete;
};

constexpr size_t kFinCounterInts = 65536;   // size of the arrival-counter arrays of the in-kernel statistics finalize
constexpr int kFinGroup = 32;               // most tiles per image the last-arriving workgroup folds itself (conv_epilogue); above: in_finalize2
// doubles a convolution's statistics scratch must hold: (sum, sum of squares) per (tile, channel)
inline size_t stat_part_doubles(size_t N, size_t tpi, size_t Cout) { return N * tpi * Cout * 2; }
constexpr int KPAD_ALIGN = 32;   // packed weights are K-padded to an even number of 16-deep chunks (fragment prefetch runs up to two past the end)

inline int conv_kpad(int ks, int cin_pad) { return round_up(ks * ks * cin_pad, KPAD_ALIGN); }
inline int conv_kpad_w1(int cin_pad) { return round_up(12 * cin_pad, KPAD_ALIGN); }    // Winograd-along-x form: tap row ky x position p
inline int conv_npad(int cout) { return cout >= 128 ? round_up(cout, 128) : round_up(cout, 64); }
inline int conv_cin_pad(int cin) { return cin <= 8 ? 8 : round_up(cin, 16); }     // 8 (two taps per k-group) or a multiple of 16

struct ConvCall {
    const float* x = nullptr; const float* x2 = nullptr;      // x2: channels >= csplit come from a second tensor (torch.cat on load), image n % x2_nmod
    int csplit = 0, x2_nmod = 1;
    const float* alpha = nullptr; const float* beta = nullptr; int relu = 0;   // x*alpha+beta (+ReLU) on load, or the raw tensor
    float bound = 0.f;          // max |operand| after the transform (InstanceNorm output: sqrt(HW); residual stream: (blocks+1) sqrt(HW))
    const unsigned* in_amax = nullptr; float bound_add = 0.f;   // or: bound = max |x| published on the device by x's producer + bound_add
    int N = 0, H = 0, W = 0;
    float* y = nullptr;
    const float* addend = nullptr; int add_nmod = 1;
    double* stat_part = nullptr; mutable int stat_S = 0;        // stat_S out: partials per image; 0 = none; -1 = finalised in the kernel
    float* fin_alpha = nullptr; float* fin_beta = nullptr; int* fin_counter = nullptr;
    unsigned* amax_out = nullptr;   // publish max |y| per image (operand scale of a consumer without an a-priori bound)
    int x_bf16 = 0, y_bf16 = 0; // bf16 storage mode (bf16-operand kernels only): the input / output tensor holds bf16
    int nprod = 3;              // products per k-group: 3 (4 adds lo*lo), or 1 = bf16 operands
    int kernel = 0;             // 0 = the layer's own kernel class, 1 = force the general kernel, 2 = require a patch kernel (op tests)
    int tile = 0;               // 0 = heuristic; else rows * 1000 + width of the patch tile (32, 64, 128 = 4 rows; 2128 = 2 x 128; + 20000 = its
                                // two-K-group form, 4 x 32 and 4 x 64 only), or 64 / 128 for the others
    int abl = 0, opt = 0;       // tools build: ablation / experiment masks of h2_tile
    int xcd_gn = -1;            // -1 = the launcher's choice; 0 = consecutive tiles per XCD; 1, 2, 4, 8 = XCD grid columns over the N tiles (tile_of_block)
    int tclass = TSNET_T_CONV;
};

// power-of-two operand scale: |x| <= bound  ->  |x * 2^sa| <= 2^15 < 65504 (fp16 max)
inline int h2_scale_log2(float bound) {
    if (!(bound > 0.f) || !std::isfinite(bound)) throw ArgError("conv: the operand bound must be positive and finite");
    int e = 0;
    (void)std::frexp(bound, &e);          // bound = m * 2^e, m in [0.5, 1)  ->  bound <= 2^e
    int sa = 15 - e;
    if (sa > 24) sa = 24;
    if (sa < -24) sa = -24;
    return sa;
}

// Kernel class of a layer at a frame size.  The patch kernels sum K slab-major, the general one tap-major: the class must depend on the
// layer and the geometry alone, never on the batch (a sample's result is the same bits in any batch, B = 1 included).
enum { K_GENERAL = 0, K_H2 = 1, K_H2S = 2, K_H2D = 3, K_W1 = 4, K_H2S32 = 5 };
// a layer packed in the Winograd-along-x form runs conv_w1 and nothing else: 3 x 3 / stride 1 / pad 1 on frames of whole 4 x 32 tiles
inline size_t w1_lds_bytes_host(int Cin, int tables) {       // conv_w1.hpp w1_lds_bytes for the two-plane stages
    return 3 * (size_t)(2 * (4 * 2 * 2 * (96 * 16 + 64) + 32)) + (size_t)tables * 2 * ((Cin + 31) / 32 * 32) * 4;
}
inline bool w1_eligible(const ConvLayer& L, int H, int W) {
    // (the three V stages + the transform table of 2 Cin floats must fit the CU's 160 KiB beside the epilogue's 64 B of static LDS)
    return L.ks == 3 && L.stride == 1 && L.pad == 1 && L.cin_pad >= 16 && (L.cin_pad & 15) == 0 && H >= 4 && W >= 32 && H % kPatchRows == 0 && W % kPatchCols == 0 &&
           w1_lds_bytes_host(L.cin_pad, 1) + 256 <= 160 * 1024;
}
// eligible_only: what the layer CAN run on (an explicit request, op tests / tools); otherwise what the forward runs it on
inline int conv_class(const ConvLayer& L, int H, int W, bool two_sources, bool transform, int rows = kPatchRows, bool eligible_only = false, bool bf16 = false) {
    const int Ho = (H + 2 * L.pad - L.ks) / L.stride + 1, Wo = (W + 2 * L.pad - L.ks) / L.stride + 1;
    const bool s1 = L.ks == 3 && L.stride == 1 && L.pad == 1 && L.cin_pad >= 16 && (L.cin_pad & 15) == 0 && H >= 2 && W >= 2;
    if (two_sources || Ho <= 0 || Wo <= 0 || Wo % kPatchCols) return K_GENERAL;
    const bool s2 = L.ks == 3 && L.stride == 2 && L.pad == 1 && !L.reflect && L.cin_pad >= (eligible_only ? 16 : 128) && (L.cin_pad & 15) == 0 && H == 2 * Ho && W == 2 * Wo;
    if (rows == 2 && Ho % 2 == 0) {                              // a 2-row tile (requested explicitly, or the stride-2 layers' own choice)
        if (s1) return K_H2;
        if (s2) return K_H2D;
    }
    if (Ho % kPatchRows) return K_GENERAL;
    if (s1) return K_H2;
    if (L.ks == 7 && L.stride == 1 && L.pad == 3 && L.reflect && L.cin_pad == 8 && !transform && H >= 4 && W >= 4) return K_H2S;
    // the pose model's stems (31 / 28 channels padded to 32): their own patch kernel since round 6 (653 + 231 us on the general kernel at configs[3])
    if (L.ks == 7 && L.stride == 1 && L.pad == 3 && L.reflect && L.cin_pad == 32 && !transform && H >= 4 && W >= 4) return K_H2S32;
    // stride 2: the patch kernel from 128 input channels on (117 / 125 us on the 128 -> 256 / 256 -> 512 layers against 132 / 142 us for the
    // general kernel); with 64 channels the K loop is four slabs long and the general kernel's smaller per-tile prologue wins (147 vs 157 us)
    // bf16 operands: a third of the MFMA work per staged byte -- the patch tiles' five-round staging binds (459 / 378 us on 128 -> 256 /
    // 256 -> 512 at 24 images against 180 us for the general kernel's 128-wide tile: profiles/round4_bf16_layers.txt)
    if (s2 && (eligible_only || !bf16)) return K_H2D;
    return K_GENERAL;
}

void run_conv(Ctx& ctx, const ConvLayer& L, const ConvCall& c) {
    const bool bf16 = c.nprod == 1;
    if (!L.wq) throw ArgError("conv: layer has no packed weights");
    if (L.ks != 1 && L.ks != 3 && L.ks != 7) throw ArgError("conv: kernel size must be 1, 3 or 7");
    if (L.cin_pad != 8 && (L.cin_pad < 16 || (L.cin_pad & 15))) throw ArgError("conv: input channels must be 8 or a multiple of 16 (pad with zeros)");
    ConvArgs g{};
    g.x = c.x; g.x2 = c.x2; g.in_alpha = c.alpha; g.in_beta = c.alpha ? c.beta : nullptr; g.in_relu = c.relu;
    if (c.alpha && !c.beta) throw ArgError("conv: alpha without beta");
    const int sa = (bf16 || c.in_amax) ? 0 : h2_scale_log2(L.form == 1 ? 2.f * c.bound : c.bound);   // Winograd: |V| <= 2 max |x|
    g.in_scale = std::ldexp(1.0f, sa); g.in_unscale = std::ldexp(1.0f, -sa);
    g.in_amax = bf16 ? nullptr : c.in_amax; g.in_bound_add = c.bound_add; g.amax_out = c.amax_out;
    g.w = L.wq; g.w_unscale = bf16 ? nullptr : L.w_unscale; g.bias = L.bias; g.y = c.y;
    g.stat_part = c.stat_part; g.addend = c.addend; g.add_nmod = c.add_nmod > 0 ? c.add_nmod : 1;
    g.x_bf16 = c.x_bf16; g.y_bf16 = c.y_bf16;
    if ((c.x_bf16 || c.y_bf16) && !bf16) throw ArgError("conv: bf16 storage goes with bf16 operands");
    if (c.x_bf16 && c.x2) throw ArgError("conv: bf16 storage of a concatenated input is not built");
    g.N = c.N; g.H = c.H; g.W = c.W; g.Cin = L.cin_pad; g.cin_log2 = ilog2(L.cin_pad);
    g.Csplit = c.x2 ? c.csplit : L.cin_pad; g.x2_nmod = c.x2_nmod > 0 ? c.x2_nmod : 1;
    g.Ho = (c.H + 2 * L.pad - L.ks) / L.stride + 1; g.Wo = (c.W + 2 * L.pad - L.ks) / L.stride + 1;
    g.Cout = L.cout; g.Npad = L.npad; g.stride = L.stride; g.pad = L.pad; g.reflect = L.reflect;
    g.taps = L.form == 1 ? 12 : L.ks * L.ks; g.nchunks = (g.taps * g.Cin + 15) / 16; g.M = c.N * g.Ho * g.Wo;
    if (c.N < 1 || g.Ho < 1 || g.Wo < 1) throw ArgError("conv: empty tensor");
    if (L.reflect && (L.pad >= c.H || L.pad >= c.W)) throw ArgError("conv: reflection pad needs pad < input size");
    if (c.x2 && ((g.Csplit & 15) || g.Csplit <= 0 || g.Csplit >= g.Cin)) throw ArgError("conv: channel split must be a multiple of 16 inside the channel range");
    if ((double)c.N * c.H * c.W * g.Csplit * 4 >= 2147483648.0 || (double)g.x2_nmod * c.H * c.W * (g.Cin - g.Csplit) * 4 >= 2147483648.0 ||
        (double)g.M * L.cout >= 2147483647.0 || (double)L.kpad * L.npad * 2 >= 2147483648.0)
        throw ArgError("conv: tensor too large for 32-bit buffer offsets");
    if ((size_t)2 * g.Cin * 4 > 32 * 1024) throw ArgError("conv: too many input channels for the transform table");
    const int hw = g.Ho * g.Wo;
    if (L.form == 1 && (c.x2 || !w1_eligible(L, c.H, c.W))) throw ArgError("conv(w1): the layer is packed in the Winograd form, which needs a single source and whole 4 x 32 tiles");
    if (L.form == 1 && c.nprod == 4) throw ArgError("conv(w1): 1 or 3 products");
    const int cls = L.form == 1 ? K_W1 : c.kernel == 1 ? K_GENERAL : conv_class(L, c.H, c.W, c.x2 != nullptr, c.alpha != nullptr, c.tile % 10000 >= 1000 ? c.tile % 10000 / 1000 : kPatchRows, c.kernel == 2, bf16);
    if (c.kernel == 2 && cls == K_GENERAL) throw ArgError("conv: this layer / frame size has no patch kernel");
    g.fin_alpha = c.fin_alpha; g.fin_beta = c.fin_beta; g.fin_eps = 1e-5f;
    TimeScope ts(ctx, c.tclass);
    auto set_tiles = [&](int rows_per_tile_m, int bn) {     // rows_per_tile_m: output positions of an M tile (patch: PR * 32; general: 128)
        g.tpi = (hw + rows_per_tile_m - 1) / rows_per_tile_m;
        g.tiles_m = c.N * g.tpi; g.tiles_n = (g.Cout + bn - 1) / bn;
        g.fin_S = g.tpi;
        // few tiles per image: the last workgroup of each (image, channel tile) finalises the statistics (conv_epilogue).  Many tiles per
        // image (the 64^2 .. 256^2 layers): the in_finalize2 launch spreads the fold over C / 16 x N workgroups.  Round 6 measured the
        // alternative -- a two-level in-kernel finalize, groups of 32 tiles then groups -- against it in one process: 5.062 vs 5.035 ms
        // per forward for the launch (profiles/round6_ab_twolevel.txt): the last group's serial tail (6 memory round trips + 2 hand-offs
        // on the stem) costs more than a kernel boundary + a 4 us kernel, once that kernel's loads are batched.
        g.fin_counter = (c.stat_part && c.fin_counter && g.tpi <= kFinGroup && (size_t)g.N * ((g.Npad + 31) / 32) <= kFinCounterInts) ? c.fin_counter : nullptr;
    };
    // 256 CUs each run ceil(tiles / 256) tiles (co-resident workgroups share the MFMA pipe): minimise that count x tile area / efficiency
    auto wide_pays = [&](long tm, double gain) {
        const double c64 = (double)((tm * ((g.Cout + 63) / 64) + 255) / 256) * 64.0;
        const double c128 = (double)((tm * ((g.Cout + 127) / 128) + 255) / 256) * 128.0 / gain;
        return g.Npad % 128 == 0 && g.Cout > 64 && c128 < c64;
    };
    try {
        if (cls == K_W1) {
            if (c.opt) throw ArgError("conv(w1): no experiment variants");
            set_tiles(kPatchRows * kPatchCols, 64);
            // weight planes far beyond an XCD's L2 (FuseNet's 1024 -> 1024: 50 MB): a 2 x 4 XCD grid streams a quarter of them per XCD (603 -> 565 us)
            int gn = c.xcd_gn < 0 ? (((double)L.kpad * L.npad * 4 > 16e6) ? 4 : 0) : c.xcd_gn;
            if (gn && ((gn != 1 && gn != 2 && gn != 4 && gn != 8) || g.tiles_n % gn || g.tiles_m % (8 / gn))) {
                if (c.xcd_gn >= 0) throw ArgError("conv(w1): the XCD grid does not divide the tile matrix");
                gn = 0;
            }
            g.xcd_gn = gn;
            // Tiles per workgroup (conv_w1.hpp): chunks of 3 or 2 consecutive spatial tiles of one channel tile -- the prologue of every tile
            // but a chunk's first disappears under its predecessor's last two periods.  Every XCD owns whole rows of the tile matrix (or its
            // cell of the XCD grid) and the chunk size must divide their count.  A chunk that crosses into the next image needs the second
            // transform table in LDS (it fits up to 512 input channels; a raw input has no table); without it chunks stay inside an image.
            // Chosen where the chunks fill the CUs in whole rounds -- so the chunk size DOES depend on the batch; what keeps a frame's bits
            // independent of its batch is that every chunk size runs the same chains per output element (bit-identical, tested at the
            // forward's shapes and under random ones).  c.tile = 1, 2, 3 forces a size (op tests, tools).
            {
                const int n = g.tiles_m * g.tiles_n, npp = ((g.Cin >> 4) + 1) / 2;
                const int rows = gn ? g.tiles_m / (8 / gn) : (g.tiles_m % 8 ? 0 : g.tiles_m / 8);
                const bool tab2 = !c.alpha || w1_lds_bytes_host(g.Cin, 2) + 256 <= 160 * 1024;
                // (four periods at least: the fetch stream reads the next tile's periods 0..2 without the last period's tail mask -- with three
                // periods and an odd slab count, period 2 would stage another pixel's channels past Cin)
                auto fits = [&](int cc) { return cc == 1 || (c.nprod != 1 && npp >= 4 && rows > 0 && rows % cc == 0 && (tab2 || g.tpi % cc == 0)); };
                int cc = 1;
                if (c.tile >= 1 && c.tile <= 3) {
                    if (!fits(c.tile)) throw ArgError("conv(w1): this chunk size does not fit the layer");
                    cc = c.tile;
                } else {
                    for (int t : {3, 2}) if (fits(t) && n % (device_cus() * t) == 0) { cc = t; break; }
                }
#ifdef TSNET_TOOLS
                if (!c.tile && cc > g_tools_knob[0]) cc = 1;          // tools/forward_ab.py: the same forward with and without chunks, one process
#endif
                g.w1_chunk = cc; g.w1_tab2 = (c.alpha && tab2 && cc > 1) ? 1 : 0;
            }
            launch_conv_w1(g, c.nprod, c.abl, ctx.stream);
            ++g_launch_counters[0];
            if (c.tclass == TSNET_T_CONV_RES) g_launch_counters[3] = 4064 + 30000;      // 4 x 32 pixels x 64 channels, Winograd form
        } else if (cls == K_H2) {
            int pr = 4, bn = 0, opt = c.opt;
            if (c.tile) {
                const int tc = c.tile % 10000, mode = c.tile / 10000;          // mode 2: deep prefetch + two K groups; 1 / 3: one of the two (tools build)
                pr = tc >= 1000 ? tc / 1000 : 4; bn = tc % 1000;
                opt |= mode == 1 ? 8 : (mode == 2 ? 24 : (mode == 3 ? 16 : 0));
            }
            if (bn == 0) {
                // 128-wide tiles (wave tile 64 x 64) move half the LDS / L1 bytes per MFMA: measured 1.07-1.15x per unit area without the
                // fused transform, 1.03x with it; the 384-tile ResnetBlock layers at batch 4 are the case where 768 64-wide tiles = exactly
                // three per CU win.  Small M (one driving frame: the decoder's ResnetBlocks are 64 tiles of 128 x 64 on 256 CUs): 32-wide
                // tiles double the number of workgroups; each stages the same patch but runs half the MFMA chain.
                const long tm = (long)c.N * (hw / 128);
                bn = wide_pays(tm, c.alpha ? 1.03 : 1.10) ? 128 : 64;
                // bf16 operands: one product per step -- the 64-wide tile is latency-bound (mfma_util 0.19, 81 % of the wave cycles
                // waiting at configs[4]); the 128-wide one runs three workgroups per CU as well (no second accumulator level) and wins
                // whenever it still fills the chip
                if (bf16 && g.Npad % 128 == 0 && g.Cout > 64 && tm * ((g.Cout + 127) / 128) >= 256) bn = 128;
                // (Round 3 gave a forward of ONE frame two-K-group tiles here -- eight waves, total = P0 + P1: another association of the same
                // chains, so a frame run alone differed from its copy inside a batch in the last bits.  Since round 5 every layer that form
                // paid on runs conv_w1 in every batch (the decoder's second up-convolution included: 29.7 against 34.0 us for one frame, equal
                // at B = 4), and the forward no longer takes it: a frame's bits do not depend on its batch, B = 1 included.  The tile codes
                // 20032 / 20064 stay available to tsnet_op_conv2d.)
            }
            // bf16 operands, the 4 x 128 tile: the four waves SIDE BY SIDE (1 x 4, wave tile 128 x 32) instead of 2 x 2 (64 x 64) -- every weight
            // fragment is loaded once per workgroup instead of twice (the tile was bound by its weight loads: 72 of 80 vector loads per slab),
            // the A rows double-buffered by tap column: 120.5 -> 107.0 us on the 512 -> 512 layer, 96.1 -> 86.3 us on the decoder's first
            // up-convolution (profiles/round6_h2_1x4.txt).  Same K order and chains: the same bits as the 2 x 2 form (tests).  Tile code 3128
            // forces it, 128 (or 4128) the 2 x 2 form.
            const bool side_by_side = pr == 3 || (!c.tile && bf16 && pr == 4 && bn == 128);
            if (pr == 3) pr = 4;
            if (side_by_side && (!bf16 || bn != 128)) throw ArgError("conv(h2): the 1 x 4 wave grid is the bf16 4 x 128 tile's");
            if ((pr != 2 && pr != 4) || g.Ho % pr) throw ArgError("conv(h2): the output height must be a multiple of the tile's rows (2 or 4)");
            if (g.Npad % bn) throw ArgError("conv(h2): the tile width must divide the padded output width");
            set_tiles(pr * kPatchCols, bn);
            {
                // weight planes far beyond the 4 MiB of an XCD's L2 (FuseNet: 37.7 MB): a 2 x 4 XCD grid streams a quarter of them per XCD
                // (-2..3 % on the 1024 -> 1024 layer, nothing on the 512 -> 512 ones: profiles/round3_conv_variants.txt)
                int gn = c.xcd_gn < 0 ? (((double)L.kpad * L.npad * 4 > 16e6) ? 4 : 0) : c.xcd_gn;
                if (gn && ((gn != 1 && gn != 2 && gn != 4 && gn != 8) || g.tiles_n % gn || g.tiles_m % (8 / gn))) {
                    if (c.xcd_gn >= 0) throw ArgError("conv(h2): the XCD grid does not divide the tile matrix");
                    gn = 0;
                }
                g.xcd_gn = gn;
            }
            launch_conv_h2(g, side_by_side ? 5 : pr, bn, c.nprod, c.abl, opt, ctx.stream);
            ++g_launch_counters[0];
            if (c.tclass == TSNET_T_CONV_RES) g_launch_counters[3] = (side_by_side ? 3 : pr) * 1000 + bn + ((opt & 16) ? 20000 : 0);
        } else if (cls == K_H2S) {
            set_tiles(128, 64);
            launch_conv_h2s(g, c.nprod, ctx.stream);
            ++g_launch_counters[0];
        } else if (cls == K_H2S32) {
            set_tiles(128, 64);
            launch_conv_h2s32(g, c.nprod, ctx.stream);
            ++g_launch_counters[0];
        } else if (cls == K_H2D) {
            // two rows x 128 columns (four waves, 44 KiB of LDS: three workgroups per CU) wherever the layer is 128 channels wide: 125 -> 99 us
            // on 256 -> 512, 117 -> 109 us on 128 -> 256 at batch 4, equal or better down to one frame (profiles/round3_conv_variants.txt);
            // the four-row shapes hold 80 KiB and run one workgroup per CU
            if (c.tile / 10000 > 1) throw ArgError("conv(h2d): the stride-2 tiles have no two-K-group form (1xxxx: the deep schedule of the 2 x 128 tile)");
            const int tc = c.tile % 10000;
            int pr = tc >= 1000 ? tc / 1000 : 4, bn = tc % 1000;          // a code without a width (0, 2000, 4000) leaves the width to the heuristic
            if (bn == 0) {
                bn = 64;
                if (tc < 1000 && g.Npad % 128 == 0 && g.Cout > 64 && g.Ho % 2 == 0) { pr = 2; bn = 128; }
            }
            if (pr != 2 && pr != 4) throw ArgError("conv(h2d): tiles have two or four output rows");
            if (g.Npad % bn) throw ArgError("conv(h2d): the tile width must divide the padded output width");
            if (g.Ho % pr) throw ArgError("conv(h2d): the output height must be a multiple of the tile's rows");
            set_tiles(pr * kPatchCols, bn);
            // A launch of at most two workgroups per CU (a single frame's: 64 .. 384 tiles) runs the deep schedule -- nothing else hides a lone
            // workgroup's memory round trips (72 -> us on the 64-tile launch).  Same bits: the choice may follow the batch.  Tile code 12128 forces
            // it, 2128 the plain schedule.
            const bool deep = c.tile ? c.tile / 10000 == 1 : (pr == 2 && bn == 128 && c.nprod == 3 && (long)g.tiles_m * g.tiles_n <= 2L * device_cus());
            launch_conv_h2d(g, pr, bn, c.nprod, deep, ctx.stream);
            ++g_launch_counters[0];
        } else {
            if (c.abl || c.opt) throw ArgError("conv: experiment variants exist for the 3x3 / stride-1 patch kernel only");
            const long tm = (long)c.N * ((hw + 127) / 128);
            // 64-deep K steps (conv_g64.hpp: the same bits, a quarter of the barriers and address computations, whole cache lines per
            // load) wherever the layer allows: 1 x 1 / 3 x 3, input channels (and the concat split) multiples of 64, 128-wide output.  In the
            // forward: the two 1 x 1 convolutions, the 64 -> 128 stride-2 layer and, with bf16 operands, every stride-2 layer.  Tile codes
            // 3064 / 3128 (with kernel = general) request it with 64 / 128 rows; 64 / 128 request conv_h2r with that width.
            const bool g64_ok = (L.ks == 1 || L.ks == 3) && (g.Cin & 63) == 0 && (!c.x2 || (g.Csplit & 63) == 0) && g.Npad % 128 == 0 && g.Cout > 64 &&
                                (c.nprod == 1 || c.nprod == 3);
            if (c.tile == 3064 || c.tile == 3128 || (!c.tile && g64_ok)) {
                if (!g64_ok) throw ArgError("conv(g64): needs a 1 x 1 / 3 x 3 layer, input channels in multiples of 64 and more than 64 output channels");
                // rows per tile: 64 in every batch (with statistics the tile decides the partial sums' grouping, so it may never depend on the
                // batch: a frame's bits do not)
                // (measured, tools/g64_variants.py: the 128-row tile -- eight waves at 140 VGPRs: one workgroup per CU -- loses everywhere with
                // fp16 x 2 operands, 208 vs 156 us on the 64 -> 128 stride-2 layer, and ties with bf16 operands)
                const int bm = c.tile ? c.tile - 3000 : 64;
                set_tiles(bm, 128);
                launch_conv_g64(g, L.ks, bm, c.nprod, ctx.stream);
                ++g_launch_counters[1];
            } else {
            int bn = c.tile ? c.tile : ((L.ks == 3 && L.cin_pad >= 16 && wide_pays(tm, 1.15)) ? 128 : 64);
            if (bn != 64 && bn != 128) throw ArgError("conv(h2r): the tile width must be 64 or 128");
            if (g.Npad % bn) throw ArgError("conv(h2r): the tile width must divide the padded output width");
            set_tiles(128, bn);
            launch_conv_h2r(g, L.ks, bn, c.nprod, ctx.stream);
            ++g_launch_counters[1];
            }
        }
    } catch (const std::invalid_argument& e) { throw ArgError(e.what()); }
    check_launch("conv");
    c.stat_S = !c.stat_part ? 0 : (g.fin_counter ? -1 : g.tpi);
}

// InstanceNorm statistics -> (alpha, beta); `part` must hold N*64*C*2 doubles
void run_stats(Ctx& ctx, const float* x, int N, int HW, int C, double* part, float* alpha, float* beta) {
    if (C & 3) throw ArgError("instnorm: C must be a multiple of 4");
    TimeScope ts(ctx, TSNET_T_STATS);
    const int cq = C / 4, cols = cq < 256 ? cq : 256, R = 256 / cols;
    int S = (HW + R * 8 - 1) / (R * 8);
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    const int rps = (HW + S - 1) / S;
    S = (HW + rps - 1) / rps;
    StatsArgs sa{x, part, HW, C, S, rps};
    hipLaunchKernelGGL(in_stats_partial_kernel, dim3(S, N, (cq + 255) / 256), dim3(256), 0, ctx.stream, sa);
    check_launch("in_stats_partial");
    launch_finalize(part, alpha, beta, N, C, S, HW, ctx.stream);
}

// stage 2 of the statistics: with more than a handful of partials per image the 16-group kernel (one serial walk per (image, channel)
// is latency-bound: 39 us for the 64 partials of the FuseNet join)
void launch_finalize_impl(const double* part, float* alpha, float* beta, int N, int C, int S, int HW, hipStream_t s) {
    if (S > 8) {
        hipLaunchKernelGGL(in_finalize2_kernel, dim3((C + kFin2Ch - 1) / kFin2Ch, N), dim3(256), 0, s, part, alpha, beta, C, S, HW, 1e-5f);
    } else {
        const int NC = N * C;
        hipLaunchKernelGGL(in_finalize_kernel, dim3((NC + 255) / 256), dim3(256), 0, s, part, alpha, beta, NC, C, S, HW, 1e-5f);
    }
    check_launch("in_finalize");
}

// y = x + add[n % add_nmod] with the InstanceNorm statistics of y -> (alpha, beta); `part` must hold N*64*C*2 doubles
void run_add_stats(Ctx& ctx, const float* x, const float* add, int add_nmod, float* y, int N, int HW, int C, double* part, float* alpha, float* beta) {
    if (C & 3) throw ArgError("add_stats: C must be a multiple of 4");
    TimeScope ts(ctx, TSNET_T_STATS);
    const int cq = C / 4, cols = cq < 256 ? cq : 256, R = 256 / cols;
    int S = (HW + R * 8 - 1) / (R * 8);
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    const int rps = (HW + S - 1) / S;
    S = (HW + rps - 1) / rps;
    AddStatsArgs sa{x, add, y, part, HW, C, S, rps, add_nmod > 0 ? add_nmod : 1};
    hipLaunchKernelGGL(add_stats_partial_kernel, dim3(S, N, (cq + 255) / 256), dim3(256), 0, ctx.stream, sa);
    check_launch("add_stats_partial");
    // (its finalize stays a launch: 64 splits x 1024 channels per image are a megabyte of partials -- one last-arriving workgroup per image
    // would read them at a single block's rate, in_finalize2 spreads them over 64 x N)
    launch_finalize(part, alpha, beta, N, C, S, HW, ctx.stream);
}

void run_norm_act(Ctx& ctx, const float* x, const float* alpha, const float* beta, int relu, const float* resid,
                  int N, int HW, int C, float* y) {
    if (C & 3) throw ArgError("norm_act: C must be a multiple of 4");
    TimeScope ts(ctx, TSNET_T_ELEMWISE);
    NormActArgs a{x, alpha, beta, resid, y, HW, C, relu, (size_t)N * HW * C / 4};
    if ((double)HW * C / 4 >= 4294967295.0 || N > 65535) throw ArgError("norm_act: tensor too large");
    size_t per_img4 = (size_t)HW * C / 4, gx = (per_img4 + 255) / 256, cap = std::max<size_t>(1, 4096 / (size_t)N);
    if (gx > cap) gx = cap;
    hipLaunchKernelGGL(norm_act_kernel, dim3((unsigned)gx, N), dim3(256), 0, ctx.stream, a);
    check_launch("norm_act");
}

void run_upsample(Ctx& ctx, const float* x, const float* alpha, const float* beta, int relu, int N, int H, int W, int C, float* y, int x_bf16 = 0, int y_bf16 = 0) {
    if (C & 3) throw ArgError("upsample: C must be a multiple of 4");
    TimeScope ts(ctx, TSNET_T_UPSAMPLE);
    UpsampleArgs a{x, alpha, beta, y, N, H, W, C, relu, x_bf16, y_bf16};
    if (2 * H > 65535 || N > 65535) throw ArgError("upsample: tensor too large");
    const size_t row4 = (size_t)2 * W * C / 4;
    hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)std::min<size_t>((row4 + 255) / 256, 64), 2 * H, N), dim3(256), 0, ctx.stream, a);
    check_launch("upsample2x");
}

// RGB head (head_conv.hpp): head_conv3_kernel for widths that are multiples of 16 (every reference caller: ngf = 64), the
// one-pixel-per-thread form for narrower nets.  The choice depends on the width alone.
void launch_head(const HeadArgs& ha, int hh, int ww, int B, hipStream_t s, int force_rows = 0) {
    if (ha.C % (2 * kHead3Ch) != 0) {
        const int tiles = ((ww + kHeadT - 1) / kHeadT) * ((hh + kHeadT - 1) / kHeadT);
        hipLaunchKernelGGL(head_conv_kernel, dim3(tiles, B), dim3(256), 0, s, ha);
    } else {
        // Tile rows by the number of workgroups (head_conv.hpp): the tallest tile that still gives every CU one -- 32 rows from B = 4 on
        // at 256^2, 16 at B = 2, 8 for one frame (84 -> 3x us there: eight waves on 64 CUs queue at the LDS pipe).  Same bits for every choice.
        const int tiles_x = (ww + kHead3T - 1) / kHead3T;
        auto go = [&](auto tr) {
            constexpr int TR = decltype(tr)::value;
            const size_t lds = (size_t)(2 * (head3_patch_f4(TR) + kHead3WtsF4) + ha.C / 2) * sizeof(float4);
            ensure_dynamic_lds(reinterpret_cast<const void*>(head_conv3_kernel<TR>), lds);
            hipLaunchKernelGGL(head_conv3_kernel<TR>, dim3(tiles_x * ((hh + TR - 1) / TR), B), dim3(TR * 16), lds, s, ha);
        };
        const long cus = device_cus();
        if (force_rows && force_rows != 8 && force_rows != 16 && force_rows != 32) throw ArgError("head: tile rows 8, 16 or 32");
        const int rows = force_rows ? force_rows : ((long)B * tiles_x * ((hh + 31) / 32) >= cus ? 32 : ((long)B * tiles_x * ((hh + 15) / 16) >= cus ? 16 : 8));
        if (rows == 32) go(std::integral_constant<int, 32>{});
        else if (rows == 16) go(std::integral_constant<int, 16>{});
        else go(std::integral_constant<int, 8>{});
    }
    check_launch("head_conv");
    ++g_launch_counters[2];
}

// F.normalize over channels of N images of P positions -> the flow kernel's fp16 (hi, lo) operand planes (flow_plane_halves(N, P, C) halves)
void run_l2norm_split(Ctx& ctx, const float* x, unsigned short* q, int N, int P, int C) {
    if (C & 7) throw ArgError("l2norm: C must be a multiple of 8");
    TimeScope ts(ctx, TSNET_T_ELEMWISE);
    const int Ppad = flow_ppad(P);
    hipLaunchKernelGGL(l2norm_split_kernel, dim3((unsigned)(((size_t)N * Ppad + 3) / 4)), dim3(256), 0, ctx.stream, x, q, N, P, Ppad, C, flow_ksteps(C));
    check_launch("l2norm_split");
}

// variant (tsnet_op_flow_k; the forward passes 0): 1 = flow_kernel whatever the plan says; 2 = flow_kernel_p without the exp pass (tools build)
void run_flow(Ctx& ctx, FlowArgs a, int NB, int variant = 0) {
    if (a.C & 7) throw ArgError("flow: C must be a multiple of 8");
    if (variant < 0 || variant > 2) throw ArgError("flow: variant 0, 1 or 2");
    TimeScope ts(ctx, TSNET_T_FLOW);
    const size_t budget = 160 * 1024;
    // large maps (a.part / a.cnt supplied): persistent target tiles, the K sources of a batch element swept in G slices (flow_persist.hpp)
    const int G = (a.part && a.cnt && NB % a.B == 0 && NB / a.B <= 8 && variant != 1) ? flowp_plan(a.B, a.h, a.w, a.C) : 0;
    if (G) {
        a.K = NB / a.B; a.G = G; a.S = flowp_slices(a.h, a.w);
        try {
            launch_flow_p(a, variant, ctx.stream);
        } catch (const std::invalid_argument& e) { throw ArgError(e.what()); }
        check_launch("flow_p");
        return;
    }
    // 64 target positions per workgroup while their planes fit in LDS beside the mask row (C <= 512 at P <= 4096), else 32
    const int NT = flow_lds_bytes(2, a.h, a.w, a.C) <= budget ? 2 : 1;
    const size_t lds = flow_lds_bytes(NT, a.h, a.w, a.C);
    if (lds > budget) throw ArgError("flow: feature width / position count exceed the LDS budget");
    try {
        launch_flow(a, NT, lds, (unsigned)(flow_ppad(a.P) / (32 * NT) * NB), ctx.stream);
    } catch (const std::invalid_argument& e) { throw ArgError(e.what()); }
    check_launch("flow");
}

void run_warp(Ctx& ctx, const float* src, const float* flow, float* out, int B, int K, int h, int w, int C) {
    TimeScope ts(ctx, TSNET_T_WARP);
    WarpArgs a{src, flow, out, B, K, h, w, C};
    hipLaunchKernelGGL(warp_mean_kernel, dim3(ew_grid((size_t)B * h * w * C / 4)), dim3(256), 0, ctx.stream, a);
    check_launch("warp_mean");
}

// weights of one layer -> operand planes + the un-scale factor; `stage` holds the OIHW parameter on the device (form 1: its transform,
// kernel 3 x 4)
void pack_layer(const float* stage, const ConvLayer& L, unsigned short* planes_out, int planes, float scale, hipStream_t s) {
    hipLaunchKernelGGL(pack_weights_kernel, dim3(ew_grid((size_t)L.kpad * L.npad)), dim3(256), 0, s, stage, planes_out, scale, planes,
                       L.cout, L.cin_real, L.cin_pad, L.ks, L.form == 1 ? 4 : L.ks, L.kpad, L.npad, L.cin_total > 0 ? L.cin_total : L.cin_real, L.cin_off);
    check_launch("pack_weights");
}

// Winograd F(2,3) filter transform along x (conv_w1.hpp): (O, I, 3, 3) -> (O, I, 3, 4),  U_p[ky] = sum_kx G[p][kx] g[ky][kx] with
// G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], evaluated in fp64 and rounded once to fp32
std::vector<float> winograd_x_filters(const std::vector<float>& w) {
    if (w.size() % 9) throw ArgError("winograd filter transform: not a 3 x 3 kernel");
    std::vector<float> u(w.size() / 9 * 12);
    for (size_t i = 0; i < w.size() / 3; ++i) {                     // one (o, i, ky) row of three taps -> four positions
        const double g0 = w[3 * i], g1 = w[3 * i + 1], g2 = w[3 * i + 2];
        u[4 * i] = (float)g0;
        u[4 * i + 1] = (float)(0.5 * (g0 + g1 + g2));
        u[4 * i + 2] = (float)(0.5 * (g0 - g1 + g2));
        u[4 * i + 3] = (float)g2;
    }
    return u;
}

// torch.linspace(-1, 1, n) in float32: step = (end-start)/(n-1); first half start+i*step, second half
// end-(n-1-i)*step (ATen's symmetric formula).
void linspace_pm1(int n, float* out) {
    if (n == 1) { out[0] = -1.f; return; }
    const float start = -1.f, end = 1.f;
    const float step = (end - start) / (float)(n - 1);
    const int half = n / 2;
    // ATen evaluates both halves with a fused multiply-add (e.g. linspace(-1,1,7)[3] == -2.98e-8, not 0)
    for (int i = 0; i < n; ++i) out[i] = i < half ? fmaf(step, (float)i, start) : fmaf(-step, (float)(n - i - 1), end);
}

// Encoder.coord_conv channels: xx = 2*(j/(w-1))-1, yy likewise, rr = sqrt(xx*xx+yy*yy); layout (H,W,3).
void coord_table(int H, int W, float* out) {
    for (int i = 0; i < H; ++i) {
        volatile float ys = (float)i / (float)(H - 1);
        volatile float yy = 2.f * ys - 1.f;
        for (int j = 0; j < W; ++j) {
            volatile float xs = (float)j / (float)(W - 1);
            volatile float xx = 2.f * xs - 1.f;
            volatile float x2 = xx * xx;
            volatile float y2 = yy * yy;
            volatile float ss = x2 + y2;
            float* o = out + ((size_t)i * W + j) * 3;
            o[0] = xx; o[1] = yy; o[2] = sqrtf(ss);
        }
    }
}

}  // namespace

// ================================================================================================
struct tsnet_engine {
    tsnet_cfg cfg{};
    std::string err;
    bool finalized = false;
    int C = 0, h = 0, w = 0, P = 0, K = 0, Bmax = 0;
    int cp_img = 0, cp_lbl = 0;
    int np = 3;                           // MFMA products per k-group: 3 (fp16 x 2 operands), or 1 = bf16-operand mode (cfg.operand_mode)
    bool st16 = false;                    // cfg.operand_mode = 2: bf16 operands AND bf16 storage of the large conv-to-conv activations (the encoder's
                                          // 256^2 .. 64^2 maps, the decoder's up-convolution outputs and upsampled inputs); statistics from fp32 accumulators

    struct Param { std::string name; std::vector<int64_t> shape; std::vector<float> host; bool loaded = false; };
    std::vector<Param> params;
    std::map<std::string, int> pindex;

    // layers
    std::vector<ConvLayer> img_enc, lbl_enc;            // stem, downs, then 2 per resblock
    ConvLayer fuse_c1_src, fuse_c1_tar;   // FuseNet's first convolution split at the channel concat: per-source half / shared target half
    ConvLayer fuse_c2, fuse_out, dec_map, dec_head;
    std::vector<ConvLayer> dec_res, dec_up;
    std::vector<ConvLayer*> all_layers;   // every layer that runs on the MFMA kernels (all but the RGB head)
    float* head_w = nullptr;              // [49][ngf][4] weights of the RGB head (head_conv.hpp)
    const float* head_bias = nullptr;

    // device memory
    float* wpack = nullptr; size_t wpack_floats = 0;    // ONE buffer: biases, RGB-head table, un-scale factors, operand planes of every layer
    float* arena = nullptr; size_t arena_floats = 0;
    float* d_coords = nullptr; float* d_gx = nullptr; float* d_gy = nullptr;
    unsigned* amax = nullptr;             // device: max |x| PER IMAGE of tensors without an a-priori bound, as float bits; reset before each
                                          // producer.  One slot per image: a sample's scale (hence its result, bit for bit) never depends on the
                                          // rest of the batch.  K*Bmax packed source inputs | Bmax packed label inputs | Bmax sg | Bmax decoder streams
    unsigned* amax_src() const { return amax; }
    unsigned* amax_tar() const { return amax + (size_t)K * Bmax; }
    unsigned* amax_sg() const { return amax + (size_t)(K + 1) * Bmax; }
    unsigned* amax_dec() const { return amax + (size_t)(K + 2) * Bmax; }

    // arena buffers (sized for Bmax)
    float *x_img = nullptr, *x_lbl = nullptr;
    std::vector<float*> raw_img, raw_lbl;
    float *X = nullptr, *Y1 = nullptr, *Y2 = nullptr;
    float *tar_fea = nullptr, *that = nullptr, *shat = nullptr, *flow = nullptr, *pg = nullptr;   // that / shat: fp16 operand planes of the flow kernel
    float *F1 = nullptr, *F2 = nullptr, *zbar = nullptr, *sg = nullptr;
    float* F1s = nullptr;                 // per-source half of FuseNet's first convolution (+ bias): computed by set_sources, cached in clip mode
    float* FT = nullptr;                  // (B,P,2C) target half of it, computed once per driving frame
    float *D = nullptr, *DY1 = nullptr, *DY2 = nullptr;
    std::vector<float*> U, R;
    float* ab[4][2] = {{nullptr}};
    float* ab_side[2][2] = {{nullptr}};   // (alpha, beta) buffers of the side lane (target-label encoder)
    int ab_rr = 0, ab_side_rr = 0;
    double* part = nullptr;
    double* part_side = nullptr;       // statistics partials / arrival counters of the side lane
    int* fin_counter = nullptr;        // arrival counters of the in-kernel statistics finalize (conv_epilogue), all zero between launches
    int* fin_counter_side = nullptr;
    int* flow_cnt = nullptr;           // arrival counters of flow_kernel_p (256 ints behind the two above), all zero between launches
    float* flow_part = nullptr;        // its partial softmax states (large maps only)
    hipStream_t side_stream = nullptr; // target-label chain of a full forward runs here, concurrently with the source encoder
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_fork2 = nullptr, ev_join2 = nullptr;
    hipEvent_t ev_done = nullptr;      // recorded behind the last launch of a forward on the caller's stream
    float* train_ws = nullptr;         // workspace of tsnet_train_extras, allocated on first use

    // clip-mode cache
    int cached_B = 0;
    float* bbox_copy = nullptr;     // (K, Bmax, H, W) device copies of the source bboxes
    int last_B = 0;
    int cur_B = 0;                        // batch of the forward being enqueued
    float src_div[TSNET_MAX_SOURCES];  // per-source image divisor (255; 1 for use_prev sources), tsnet_set_source_divisors

    Timing timing;

    void add_param(const std::string& name, std::vector<int64_t> shape) {
        pindex[name] = (int)params.size();
        Param p; p.name = name; p.shape = std::move(shape);
        params.push_back(std::move(p));
    }
    ConvLayer make_conv(const std::string& name, int cin_real, int cout, int ks, int stride, int pad, int reflect) {
        ConvLayer L; L.name = name; L.cin_real = cin_real; L.cin_pad = conv_cin_pad(cin_real); L.cout = cout; L.ks = ks;
        L.stride = stride; L.pad = pad; L.reflect = reflect;
        L.kpad = conv_kpad(ks, L.cin_pad); L.npad = conv_npad(cout);
        L.wparam = name + ".weight"; L.bparam = name + ".bias"; L.cin_total = cin_real; L.cin_off = 0;
        add_param(L.wparam, {cout, cin_real, ks, ks});
        add_param(L.bparam, {cout});
        return L;
    }
    void build_layers();
    void alloc_all(hipStream_t s);

    std::pair<float*, float*> next_ab(const Ctx& ctx) {
        if (!ctx.lane) { auto r = std::make_pair(ab[ab_rr][0], ab[ab_rr][1]); ab_rr = (ab_rr + 1) & 3; return r; }
        auto r = std::make_pair(ab_side[ab_side_rr][0], ab_side[ab_side_rr][1]); ab_side_rr ^= 1; return r;
    }
    float enc_bound() const { return (float)(cfg.enc_blocks + 1) * std::sqrt((float)P); }  // bound of the source features: |relu(IN)| <= sqrt(P), + one IN output per block
    // every convolution of the forward goes through here: the engine's operand mode, the lane's statistics scratch, the finalize
    void conv(Ctx& ctx, const ConvLayer& L, ConvCall& c) { c.nprod = np; run_conv(ctx, L, c); }
    void conv_stats(Ctx& ctx, const ConvLayer& L, ConvCall& c, int N, int HW, float* alpha, float* beta) {
        double* pt = ctx.lane ? part_side : part;
        c.stat_part = pt;
        c.fin_alpha = alpha; c.fin_beta = beta; c.fin_counter = ctx.lane ? fin_counter_side : fin_counter;
        conv(ctx, L, c);
        if (c.stat_S < 0) return;              // finalised by the last workgroups of the convolution itself
        TimeScope ts(ctx, TSNET_T_STATS);
        hipLaunchKernelGGL(in_finalize2_kernel, dim3((L.cout + kFin2Ch - 1) / kFin2Ch, N), dim3(256), 0, ctx.stream, pt, alpha, beta, L.cout, c.stat_S, HW, 1e-5f);
        check_launch("in_finalize2");
    }
    void encode(Ctx& ctx, std::vector<ConvLayer>& L, const float* xin, const unsigned* xin_amax, int N, std::vector<float*>& raw, float* out_fea, int nblocks);
    // ResnetBlock.  stream_amax == null: the residual stream Xs has the a-priori bound stream_bound (encoder: (blocks+1) sqrt(HW));
    // else its bound is measured: max |first value| published by the producer + amax_add (decoder: the stream starts at a raw conv output)
    void resblock(Ctx& ctx, const ConvLayer& c1, const ConvLayer& c2, float* Xs, float stream_bound, const unsigned* stream_amax, float amax_add,
                  float* y1, float* y2, int N, int hh, int ww);
    void set_sources(Ctx& ctx, const float* const* src_img, const float* const* src_lbl, const float* const* src_bbox, int B, hipStream_t bbox_stream = nullptr);
    void target_chain(Ctx& ctx, const float* tar_lbl, int B);
    void forward_rest(Ctx& ctx, const float* tar_bbox, float* out_rgb, float* out_flow, int B);
    void forward_target(Ctx& ctx, const float* tar_lbl, const float* tar_bbox, float* out_rgb, float* out_flow, int B) {
        target_chain(ctx, tar_lbl, B);
        forward_rest(ctx, tar_bbox, out_rgb, out_flow, B);
    }
};

void tsnet_engine::build_layers() {
    const tsnet_cfg& c = cfg;
    const int coords = c.addcoords ? 3 : 0;
    auto enc = [&](const std::string& net, int cin, int nblocks, std::vector<ConvLayer>& out) {
        const int cin_real = cin + coords;
        out.push_back(make_conv(net + ".model.1", cin_real, c.ngf, 7, 1, 3, 1));
        int idx = 4, ch = c.ngf;
        for (int i = 0; i < c.n_downsampling; ++i) {
            out.push_back(make_conv(net + ".model." + std::to_string(idx), ch, ch * 2, 3, 2, 1, 0));
            ch *= 2; idx += 3;
        }
        for (int i = 0; i < nblocks; ++i) {
            out.push_back(make_conv(net + ".model." + std::to_string(idx) + ".conv_block.1", ch, ch, 3, 1, 1, 1));
            out.push_back(make_conv(net + ".model." + std::to_string(idx) + ".conv_block.5", ch, ch, 3, 1, 1, 1));
            idx += 1;
        }
        return out[0].cin_pad;
    };
    cp_img = enc("img_enc", 3 + c.label_nc, c.enc_blocks, img_enc);
    cp_lbl = enc("lbl_enc", c.label_nc, 0, lbl_enc);
    const int fc = 2 * C;   // FuseNet width: cat of two feature maps (1024 in the reference, TSNet.py:227)
    // conv(cat(src, tar)) = conv_src(src) + conv_tar(tar): the target half is shared by the K sources (SURVEY.md 7.2, -9.66 GMAC/frame).
    // Both halves read windows of the same OIHW parameter.
    fuse_c1_src = make_conv("fuse_net.model.0.conv_block.1", fc, fc, 3, 1, 1, 1);
    fuse_c1_src.name += "[src]"; fuse_c1_src.cin_real = C; fuse_c1_src.cin_pad = conv_cin_pad(C); fuse_c1_src.cin_off = 0;
    fuse_c1_src.kpad = conv_kpad(3, fuse_c1_src.cin_pad);
    fuse_c1_tar = fuse_c1_src; fuse_c1_tar.name = "fuse_net.model.0.conv_block.1[tar]"; fuse_c1_tar.cin_off = C; fuse_c1_tar.bparam.clear();
    fuse_c2 = make_conv("fuse_net.model.0.conv_block.5", fc, fc, 3, 1, 1, 1);
    fuse_out = make_conv("fuse_net.conv", fc, fc / 2, 1, 1, 0, 0);
    dec_map = make_conv("dec.map_conv", 2 * C, C, 1, 1, 0, 0);
    int n = 0;
    for (int i = 0; i < c.n_blocks; ++i) {
        dec_res.push_back(make_conv("dec.model" + std::to_string(n) + ".0.conv_block.1", C, C, 3, 1, 1, 1));
        dec_res.push_back(make_conv("dec.model" + std::to_string(n) + ".0.conv_block.5", C, C, 3, 1, 1, 1));
        ++n;
    }
    for (int i = 0; i < c.n_downsampling; ++i) {
        const int ci = c.ngf << (c.n_downsampling - i);
        dec_up.push_back(make_conv("dec.model" + std::to_string(n) + ".2", ci, ci / 2, 3, 1, 1, 1));
        ++n;
    }
    dec_head = make_conv("dec.model" + std::to_string(n) + ".1", c.ngf, 3, 7, 1, 3, 1);
    // ---- Winograd-along-x form (conv_w1.hpp) for the 3 x 3 / stride-1 layers it wins on, where the frame splits into whole 4 x 32 tiles: the
    // ResnetBlocks (127 / 133 us against 153 / 150 us at the headline batch, 39 against 50 us for one frame), FuseNet (266 / 87 / 565 us
    // against 293 / 101 / 618), the decoder's first up-convolution (86 against 101 us); profiles/round4_conv_variants.txt.  The later
    // up-convolutions (128^2: equal; 256^2: 120 against 111 us) and the bf16-operand mode keep the direct kernel.  A layer has ONE packed
    // form and therefore one kernel in every batch.
    auto to_w1 = [&](ConvLayer& L, int hh, int ww) {
        if (np == 1 || !w1_eligible(L, hh, ww)) return;
        L.form = 1; L.kpad = conv_kpad_w1(L.cin_pad);
    };
    for (size_t i = 1 + c.n_downsampling; i < img_enc.size(); ++i) to_w1(img_enc[i], h, w);
    to_w1(fuse_c1_src, h, w); to_w1(fuse_c1_tar, h, w); to_w1(fuse_c2, h, w);
    for (auto& L : dec_res) to_w1(L, h, w);
    if (!dec_up.empty()) to_w1(dec_up[0], 2 * h, 2 * w);
    if (dec_up.size() > 1) to_w1(dec_up[1], 4 * h, 4 * w);      // equal at B = 4 (102.7 against 103.4 us), 29.7 against 34 us for one frame; the third (256^2 x 128 -> 64) stays direct: 118 against 140 us
    for (auto& L : img_enc) all_layers.push_back(&L);
    for (auto& L : lbl_enc) all_layers.push_back(&L);
    all_layers.push_back(&fuse_c1_src); all_layers.push_back(&fuse_c1_tar);
    all_layers.push_back(&fuse_c2); all_layers.push_back(&fuse_out);
    all_layers.push_back(&dec_map);
    for (auto& L : dec_res) all_layers.push_back(&L);
    for (auto& L : dec_up) all_layers.push_back(&L);
}

void tsnet_engine::alloc_all(hipStream_t s) {
    // ---- packed weights: ONE contiguous buffer (a single RCCL broadcast replicates a model): biases, the RGB-head table, the per-layer
    // un-scale factors 2^-sw, then the operand planes of every layer -- fp16 (hi, lo) of w * 2^sw, or one bf16 plane in the bf16-operand
    // mode.  Nothing else is packed: what a replica receives is exactly what its kernels read.
    const int planes = np == 1 ? 1 : 2;
    size_t off = 0;
    for (ConvLayer* L : all_layers) { L->b_off = off; off += (size_t)round_up(L->cout, 4); }
    dec_head.b_off = off; off += 4;
    off = (off + 63) / 64 * 64;            // 256-byte aligned sections
    const size_t head_off = off;
    off += (size_t)49 * cfg.ngf * 4 + 64;  // + one 16-channel row: the narrow-net head kernel reads whole 16-channel rows of the last tap
    off = (off + 63) / 64 * 64;
    const size_t tab_off = off;
    off += round_up((int)all_layers.size(), 64);
    off = (off + 63) / 64 * 64;
    const size_t planes_off = off;
    size_t o16 = 0;
    for (ConvLayer* L : all_layers) { L->w_off = o16; o16 += (size_t)planes * L->kpad * L->npad; o16 = (o16 + 127) / 128 * 128; }
    off += (o16 + 1) / 2;                  // two 16-bit values per float slot
    wpack_floats = off;
    HIP_TRY(hipMalloc((void**)&wpack, wpack_floats * sizeof(float)));
    HIP_TRY(hipMemsetAsync(wpack, 0, wpack_floats * sizeof(float), s));
    size_t max_w = 0;
    for (ConvLayer* L : all_layers) max_w = std::max(max_w, (size_t)L->cout * L->cin_total * L->ks * (L->form == 1 ? 4 : L->ks));
    max_w = std::max(max_w, (size_t)3 * cfg.ngf * 49);
    float* stage = nullptr;
    HIP_TRY(hipMalloc((void**)&stage, max_w * sizeof(float)));
    unsigned short* planes_base = reinterpret_cast<unsigned short*>(wpack + planes_off);
    for (size_t i = 0; i < all_layers.size(); ++i) {
        ConvLayer* L = all_layers[i];
        const Param& pw = params[pindex[L->wparam]];
        // the filter as the kernel consumes it: itself, or its Winograd transform along x (fp64 on the host, rounded once)
        std::vector<float> wino;
        if (L->form == 1) wino = winograd_x_filters(pw.host);
        const std::vector<float>& wsrc = L->form == 1 ? wino : pw.host;
        // per layer a power-of-two scale from the largest |weight|: |w * 2^sw| <= 2^15
        float mx = 0.f;
        for (float v : wsrc) { const float av = std::fabs(v); if (av > mx) mx = av; }
        if (!std::isfinite(mx)) throw WeightError("parameter '" + L->wparam + "' holds a non-finite value");
        const int sw = (np != 1 && mx > 0.f) ? h2_scale_log2(mx) : 0;
        const float unscale = std::ldexp(1.0f, -sw);
        HIP_TRY(hipMemcpyAsync(stage, wsrc.data(), wsrc.size() * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(wpack + tab_off + i, &unscale, sizeof(float), hipMemcpyHostToDevice, s));
        pack_layer(stage, *L, planes_base + L->w_off, planes, std::ldexp(1.0f, sw), s);
        if (!L->bparam.empty()) {
            const Param& pb = params[pindex[L->bparam]];
            HIP_TRY(hipMemcpyAsync(wpack + L->b_off, pb.host.data(), pb.host.size() * sizeof(float), hipMemcpyHostToDevice, s));
        }
        HIP_TRY(hipStreamSynchronize(s));   // host vectors / staging buffer reused next iteration
        L->wq = planes_base + L->w_off;
        L->w_unscale = np == 1 ? nullptr : wpack + tab_off + i;
        L->bias = L->bparam.empty() ? nullptr : wpack + L->b_off;
    }
    {                                      // RGB head: fp32 weights in the vector kernel's order + bias
        const Param& pw = params[pindex[dec_head.wparam]];
        const Param& pb = params[pindex[dec_head.bparam]];
        HIP_TRY(hipMemcpyAsync(stage, pw.host.data(), pw.host.size() * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(wpack + dec_head.b_off, pb.host.data(), pb.host.size() * sizeof(float), hipMemcpyHostToDevice, s));
        head_w = wpack + head_off;
        head_bias = wpack + dec_head.b_off;
        hipLaunchKernelGGL(pack_head_weights_kernel, dim3(64), dim3(256), 0, s, stage, head_w, cfg.ngf);
        check_launch("pack_head_weights");
        HIP_TRY(hipStreamSynchronize(s));
    }
    HIP_TRY(hipFree(stage));
    for (auto& p : params) { std::vector<float>().swap(p.host); }   // host copies no longer needed

    // ---- constant tables
    const int H = cfg.height, W = cfg.width;
    {
        std::vector<float> t((size_t)H * W * 3), gx(w), gy(h);
        coord_table(H, W, t.data());
        linspace_pm1(w, gx.data());
        linspace_pm1(h, gy.data());
        HIP_TRY(hipMalloc((void**)&d_coords, t.size() * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&d_gx, w * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&d_gy, h * sizeof(float)));
        HIP_TRY(hipMemcpy(d_coords, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_gx, gx.data(), w * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_gy, gy.data(), h * sizeof(float), hipMemcpyHostToDevice));
    }

    // ---- workspace arena: every buffer gets a fixed offset (no allocation after finalize)
    const size_t NB = (size_t)K * Bmax, B = Bmax;
    std::vector<std::pair<float**, size_t>> req;
    auto want = [&](float** p, size_t n) { req.emplace_back(p, (n + 63) & ~(size_t)63); };
    want(&x_img, NB * H * W * cp_img);
    want(&x_lbl, B * H * W * cp_lbl);
    raw_img.assign(cfg.n_downsampling + 1, nullptr);
    raw_lbl.assign(cfg.n_downsampling + 1, nullptr);
    for (int l = 0; l <= cfg.n_downsampling; ++l) {
        const size_t e = (size_t)(H >> l) * (W >> l) * (cfg.ngf << l);
        want(&raw_img[l], NB * e);
        want(&raw_lbl[l], B * e);
    }
    const size_t fe = (size_t)P * C;
    want(&X, NB * fe); want(&Y1, NB * fe); want(&Y2, NB * fe);
    want(&tar_fea, B * fe); want(&that, (flow_plane_halves((int)B, P, C) + 1) / 2); want(&shat, (flow_plane_halves((int)NB, P, C) + 1) / 2); want(&flow, NB * P * 2); want(&pg, B * fe);
    want(&F1, NB * fe * 2); want(&F2, NB * fe * 2); want(&F1s, NB * fe * 2); want(&zbar, B * fe * 2); want(&sg, B * fe); want(&FT, B * fe * 2);
    want(&D, B * fe); want(&DY1, B * fe); want(&DY2, B * fe);
    U.assign(cfg.n_downsampling, nullptr); R.assign(cfg.n_downsampling, nullptr);
    for (int i = 0; i < cfg.n_downsampling; ++i) {
        const size_t sp = (size_t)(h << (i + 1)) * (w << (i + 1));
        want(&U[i], B * sp * (C >> i));
        want(&R[i], B * sp * (C >> (i + 1)));
    }
    for (int i = 0; i < 4; ++i) { want(&ab[i][0], NB * 2 * C); want(&ab[i][1], NB * 2 * C); }
    for (int i = 0; i < 2; ++i) { want(&ab_side[i][0], NB * 2 * C); want(&ab_side[i][1], NB * 2 * C); }
    want(&bbox_copy, NB * H * W);
    if (P >= 2048 && P % 64 == 0) want(&flow_part, 2 * flowp_part_words((int)NB, P, flowp_slices(h, w)));   // flow_kernel_p: one softmax state per (image, target tile, slice, column), 8-byte words
    // InstanceNorm partials (doubles = 2 floats each): the stand-alone pass N*64*C*2, a conv epilogue N * tiles-per-image * Cout * 2
    // with tiles of at least 64 positions (ceil for ragged images), per lane
    size_t part_doubles = NB * 64 * 2 * C * 2;
    for (int l = 0; l <= cfg.n_downsampling; ++l) {
        const size_t hw = (size_t)(H >> l) * (W >> l), tpi = (hw + 63) / 64;
        part_doubles = std::max(part_doubles, stat_part_doubles(NB, tpi, (size_t)std::max(cfg.ngf << l, 2 * C)));
    }
    float *part_f = nullptr, *part_side_f = nullptr;
    want(&part_f, 2 * part_doubles); want(&part_side_f, 2 * part_doubles);
    size_t total = 0;
    for (auto& r : req) total += r.second;
    arena_floats = total;
    HIP_TRY(hipMalloc((void**)&arena, total * sizeof(float)));
    size_t o = 0;
    for (auto& r : req) { *r.first = arena + o; o += r.second; }
    part = reinterpret_cast<double*>(part_f);
    part_side = reinterpret_cast<double*>(part_side_f);
    HIP_TRY(hipMalloc((void**)&amax, (size_t)(K + 3) * Bmax * sizeof(unsigned)));
    HIP_TRY(hipMemsetAsync(amax, 0, (size_t)(K + 3) * Bmax * sizeof(unsigned), s));
    // arrival counters: one per (image, 32-channel group) of a launch; launches with more (image, group) pairs than kFinCounterInts
    // fall back to the in_finalize2 kernel (run_conv checks the index range against this size)
    HIP_TRY(hipMalloc((void**)&fin_counter, (2 * kFinCounterInts + 256) * sizeof(int)));
    HIP_TRY(hipMemsetAsync(fin_counter, 0, (2 * kFinCounterInts + 256) * sizeof(int), s));
    fin_counter_side = fin_counter + kFinCounterInts;
    flow_cnt = fin_counter + 2 * kFinCounterInts;
    // side lane (tsnet_forward): own statistics scratch, counters, (alpha, beta) pairs (above), stream and events
    HIP_TRY(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev_fork2, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev_join2, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
}

void tsnet_engine::resblock(Ctx& ctx, const ConvLayer& c1, const ConvLayer& c2, float* Xs, float stream_bound, const unsigned* stream_amax,
                            float amax_add, float* y1, float* y2, int N, int hh, int ww) {
    const int Cc = c1.cout, HW = hh * ww;
    auto s1 = next_ab(ctx);
    ConvCall a; a.x = Xs; a.bound = stream_bound; a.in_amax = stream_amax; a.bound_add = amax_add;
    a.N = N; a.H = hh; a.W = ww; a.y = y1; a.tclass = TSNET_T_CONV_RES;
    conv_stats(ctx, c1, a, N, HW, s1.first, s1.second);
    ConvCall b; b.x = y1; b.alpha = s1.first; b.beta = s1.second; b.relu = 1; b.bound = std::sqrt((float)HW);   // |IN(.)| <= sqrt(HW - 1)
    b.N = N; b.H = hh; b.W = ww; b.y = y2; b.tclass = TSNET_T_CONV_RES;
    auto s2 = next_ab(ctx);
    conv_stats(ctx, c2, b, N, HW, s2.first, s2.second);
    run_norm_act(ctx, y2, s2.first, s2.second, 0, Xs, N, HW, Cc, Xs);   // X += IN(y2)
}

// Encoder.forward (TSNet.py:52-105) on a packed NHWC input whose maximum its producer published; out_fea = final feature map
void tsnet_engine::encode(Ctx& ctx, std::vector<ConvLayer>& L, const float* xin, const unsigned* xin_amax, int N, std::vector<float*>& raw,
                          float* out_fea, int nblocks) {
    int hh = cfg.height, ww = cfg.width;
    auto st = next_ab(ctx);
    {
        ConvCall a; a.x = xin; a.in_amax = xin_amax; a.bound = 1.f; a.N = N; a.H = hh; a.W = ww; a.y = raw[0];
        a.y_bf16 = st16;                                             // bf16 storage: the stem's and the first down-convolutions' raw outputs
        conv_stats(ctx, L[0], a, N, hh * ww, st.first, st.second);
    }
    for (int l = 1; l <= cfg.n_downsampling; ++l) {
        // the downsampling convolution reads relu(IN(previous)): the transform is applied while its operand tile is staged
        auto prev = st;
        st = next_ab(ctx);
        ConvCall d; d.x = raw[l - 1]; d.alpha = prev.first; d.beta = prev.second; d.relu = 1; d.bound = std::sqrt((float)(hh * ww));
        d.N = N; d.H = hh; d.W = ww; d.y = raw[l];
        d.x_bf16 = st16; d.y_bf16 = st16 && l < cfg.n_downsampling;  // the last one feeds norm_act (32 x 32 features: fp32)
        hh /= 2; ww /= 2;
        conv_stats(ctx, L[l], d, N, hh * ww, st.first, st.second);
    }
    run_norm_act(ctx, raw[cfg.n_downsampling], st.first, st.second, 1, nullptr, N, hh * ww, C, out_fea);
    // |relu(IN(.))| <= sqrt(HW) and every block adds one more InstanceNorm output: the stream stays below (blocks + 1) sqrt(HW)
    const float bound = (float)(nblocks + 1) * std::sqrt((float)(hh * ww));
    for (int i = 0; i < nblocks; ++i)
        resblock(ctx, L[cfg.n_downsampling + 1 + 2 * i], L[cfg.n_downsampling + 2 + 2 * i], out_fea, bound, nullptr, 0.f, Y1, Y2, N, hh, ww);
}

// bbox_stream: where the K bounding-box copies of the clip cache are enqueued.  Their only reader is the flow kernel; the one-shot forward
// passes its side stream (the flow kernel's own lane: the copies are ordered ahead of it there and cost the caller's lane nothing --
// three 4.7 us copy kernels + their boundaries sat in front of the stem until round 5); null = the caller's stream (clip mode).
void tsnet_engine::set_sources(Ctx& ctx, const float* const* src_img, const float* const* src_lbl, const float* const* src_bbox, int B, hipStream_t bbox_stream) {
    cur_B = B;
    const int H = cfg.height, W = cfg.width;
    {
        TimeScope ts(ctx, TSNET_T_PACK);
        PackArgs p{};
        for (int s = 0; s < K; ++s) { p.img[s] = src_img[s]; p.lbl[s] = src_lbl[s]; p.img_div[s] = src_div[s]; }
        p.coords = cfg.addcoords ? d_coords : nullptr;
        p.out = x_img; p.S = K; p.B = B; p.H = H; p.W = W; p.L = cfg.label_nc; p.nimg = 3; p.Cp = cp_img;
        HIP_TRY(hipMemsetAsync(amax_src(), 0, (size_t)K * B * sizeof(unsigned), ctx.stream));
        p.amax_out = amax_src();
        hipLaunchKernelGGL(pack_input_kernel, dim3(pack_grid(H * W), K * B), dim3(256), 0, ctx.stream, p);
        check_launch("pack_input(img)");
        for (int s = 0; s < K; ++s)
            HIP_TRY(hipMemcpyAsync(bbox_copy + (size_t)s * Bmax * H * W, src_bbox[s], (size_t)B * H * W * sizeof(float), hipMemcpyDeviceToDevice,
                                   bbox_stream ? bbox_stream : ctx.stream));
    }
    encode(ctx, img_enc, x_img, amax_src(), K * B, raw_img, X, cfg.enc_blocks);
    run_l2norm_split(ctx, X, reinterpret_cast<unsigned short*>(shat), K * B, P, C);
    // the per-source half of FuseNet's first convolution (conv(cat(src, tar)) = conv_src(src) + conv_tar(tar), TSNet.py:195-197)
    // depends on the sources only: computed here, so a driving frame of a clip does not pay for it
    ConvCall a; a.x = X; a.bound = enc_bound(); a.N = K * B; a.H = h; a.W = w; a.y = F1s;
    conv(ctx, fuse_c1_src, a);
    cached_B = B;
}

// Everything that depends on the driving frame only: label encoder, its L2-normalised features and the target half of
// FuseNet's first convolution.  Independent of the source encoder, so a full forward runs it on the side stream.
void tsnet_engine::target_chain(Ctx& ctx, const float* tar_lbl, int B) {
    cur_B = B;
    const int H = cfg.height, W = cfg.width;
    {
        TimeScope ts(ctx, TSNET_T_PACK);
        PackArgs p{};
        p.img[0] = nullptr; p.lbl[0] = tar_lbl;
        p.coords = cfg.addcoords ? d_coords : nullptr;
        p.out = x_lbl; p.S = 1; p.B = B; p.H = H; p.W = W; p.L = cfg.label_nc; p.nimg = 0; p.Cp = cp_lbl;
        // one reset for the three per-image maxima of this forward (target input, sg, decoder stream: contiguous)
        HIP_TRY(hipMemsetAsync(amax_tar(), 0, (size_t)3 * Bmax * sizeof(unsigned), ctx.stream));
        p.amax_out = amax_tar();
        hipLaunchKernelGGL(pack_input_kernel, dim3(pack_grid(H * W), B), dim3(256), 0, ctx.stream, p);
        check_launch("pack_input(lbl)");
    }
    encode(ctx, lbl_enc, x_lbl, amax_tar(), B, raw_lbl, tar_fea, 0);
    run_l2norm_split(ctx, tar_fea, reinterpret_cast<unsigned short*>(that), B, P, C);
    ConvCall t; t.x = tar_fea; t.bound = std::sqrt((float)P); t.N = B; t.H = h; t.W = w; t.y = FT;      // shared target half of fuse conv1
    conv(ctx, fuse_c1_tar, t);
}

void tsnet_engine::forward_rest(Ctx& ctx, const float* tar_bbox, float* out_rgb, float* out_flow, int B) {
    cur_B = B;
    const int H = cfg.height, W = cfg.width, NB = K * B;
    // ---- transformation branch.  Its result (pg) is first needed by the decoder, and its kernels are latency-bound (384 workgroups):
    // with the side stream available it runs there, concurrently with the MFMA-bound synthesis branch below, and joins before dec_map.
    const bool fork = side_stream && !ctx.timing && ctx.lane == 0;
    Ctx cside; cside.stream = side_stream; cside.lane = 1;
    Ctx& cx = fork ? cside : ctx;
    SideJoin join2{fork ? side_stream : nullptr, ctx.stream, ev_join2};
    if (fork) {
        HIP_TRY(hipEventRecord(ev_fork2, ctx.stream));
        HIP_TRY(hipStreamWaitEvent(side_stream, ev_fork2, 0));
    }
    FlowArgs fa{};
    fa.tq = reinterpret_cast<const unsigned short*>(that); fa.sq = reinterpret_cast<const unsigned short*>(shat); fa.tar_bbox = tar_bbox;
    for (int s = 0; s < K; ++s) fa.src_bbox[s] = bbox_copy + (size_t)s * Bmax * H * W;
    fa.gx = d_gx; fa.gy = d_gy; fa.flow = flow;
    fa.part = reinterpret_cast<unsigned long long*>(flow_part); fa.cnt = flow_cnt;
    fa.B = B; fa.P = P; fa.C = C; fa.h = h; fa.w = w; fa.H = H; fa.W = W; fa.sy = H / h; fa.sx = W / w;
    run_flow(cx, fa, NB);
    if (out_flow)
        HIP_TRY(hipMemcpyAsync(out_flow, flow, (size_t)NB * P * 2 * sizeof(float), hipMemcpyDeviceToDevice, cx.stream));
    run_warp(cx, X, flow, pg, B, K, h, w, C);
    if (fork) HIP_TRY(hipEventRecord(ev_join2, side_stream));

    // ---- synthesis branch
    const float sqP = std::sqrt((float)P);
    {
        auto s1 = next_ab(ctx);
        auto s2 = next_ab(ctx);
        // F1 = conv_src(src) [from set_sources] + conv_tar(tar) [target chain], and its InstanceNorm statistics
        run_add_stats(ctx, F1s, FT, B, F1, NB, P, 2 * C, part, s1.first, s1.second);
        ConvCall b; b.x = F1; b.alpha = s1.first; b.beta = s1.second; b.relu = 1; b.bound = sqP;
        b.N = NB; b.H = h; b.W = w; b.y = F2;
        conv_stats(ctx, fuse_c2, b, NB, P, s2.first, s2.second);
        {
            TimeScope ts(ctx, TSNET_T_ELEMWISE);
            FuseTailArgs t2{X, tar_fea, F2, s2.first, s2.second, zbar, B, K, P, C};
            hipLaunchKernelGGL(fuse_resid_mean_kernel, dim3(ew_grid((size_t)B * P * 2 * C / 4)), dim3(256), 0, ctx.stream, t2);
            check_launch("fuse_resid_mean");
        }
        // zbar = mean over sources of cat(src_fea, tar_fea) + IN(.): bounded by enc_bound + sqrt(P).  fuse_net.conv has no norm behind it:
        // it publishes max |sg| per image for dec.map_conv's operand scale
        ConvCall c; c.x = zbar; c.bound = enc_bound() + sqP; c.N = B; c.H = h; c.W = w; c.y = sg;
        c.amax_out = amax_sg();
        conv(ctx, fuse_out, c);
    }

    // ---- decoder
    if (fork) HIP_TRY(hipStreamWaitEvent(ctx.stream, ev_join2, 0));
    join2.done = true;
    // dec.map_conv reads cat(pg, sg) (TSNet.py:163) formed on load: |pg| <= enc_bound (a convex combination of source features, zero
    // padded), |sg| <= its published maximum.  The decoder's stream starts at map_conv's raw output: no a-priori bound either, so map_conv
    // publishes max |D| and the convolutions reading the stream derive their operand scale from it on the device:
    // D_i = D_0 + (i InstanceNorm outputs), |D_i| <= max |D_0| + i sqrt(P).  (bf16-operand mode: no scales; the maxima are not read.)
    {
        ConvCall a; a.x = pg; a.x2 = sg; a.csplit = C; a.x2_nmod = B; a.N = B; a.H = h; a.W = w; a.y = D;
        a.in_amax = amax_sg(); a.bound_add = enc_bound();
        a.amax_out = amax_dec();
        conv(ctx, dec_map, a);
    }
    for (int i = 0; i < cfg.n_blocks; ++i)
        resblock(ctx, dec_res[2 * i], dec_res[2 * i + 1], D, 0.f, amax_dec(), (float)i * sqP, DY1, DY2, B, h, w);
    const float* cur = D; const float* cal = nullptr; const float* cbe = nullptr;
    int hh = h, ww = w, cc = C;
    for (int i = 0; i < cfg.n_downsampling; ++i) {
        // input of up-convolution i = bilinear x2 of relu(IN(previous)) -- a convex combination of InstanceNorm outputs, bounded by
        // sqrt(HW) of the low-resolution map; the first one upsamples the decoder stream itself: published max |D_0| + n_blocks sqrt(P)
        const float in_bound = std::sqrt((float)(hh * ww));
        run_upsample(ctx, cur, cal, cbe, cal ? 1 : 0, B, hh, ww, cc, U[i], st16 && i > 0, st16);
        hh *= 2; ww *= 2;
        cc /= 2;
        auto st = next_ab(ctx);
        ConvCall a; a.x = U[i]; a.bound = in_bound; a.N = B; a.H = hh; a.W = ww; a.y = R[i];
        a.x_bf16 = st16; a.y_bf16 = st16;
        if (!cal) { a.in_amax = amax_dec(); a.bound_add = (float)cfg.n_blocks * sqP; }
        conv_stats(ctx, dec_up[i], a, B, hh * ww, st.first, st.second);
        cur = R[i]; cal = st.first; cbe = st.second;
    }
    {
        TimeScope ts(ctx, TSNET_T_CONV);     // it is a convolution: keep it in the conv class for the roofline accounting
        HeadArgs ha{};
        ha.x = cur; ha.alpha = cal; ha.beta = cbe; ha.w = head_w; ha.bias = head_bias; ha.y = out_rgb;
        ha.N = B; ha.H = hh; ha.W = ww; ha.C = cc; ha.x_bf16 = st16 && cfg.n_downsampling > 0;
        ha.composite = cfg.pose_composite; ha.fore_x0 = 64; ha.fore_x1 = 192;          // TSNet_pose.py:279
        for (int c = 0; c < 3; ++c) ha.bg[c] = (-cfg.pose_mean[c]) / 255.0f;             // TSNet_pose.py:276
        launch_head(ha, hh, ww, B, ctx.stream);
    }
    last_B = B;
    HIP_TRY(hipEventRecord(ev_done, ctx.stream));       // what tsnet_stage_ptr orders its widening pass behind
}

// ================================================================================================
// C ABI
#define API_BEGIN(h)                        \
    if (!(h)) return TSNET_ERR_ARG;         \
    try {
#define API_END(h)                                                              \
    } catch (const ArgError& e) { (h)->err = e.what(); return TSNET_ERR_ARG; }   \
      catch (const WeightError& e) { (h)->err = e.what(); return TSNET_ERR_WEIGHT; } \
      catch (const std::bad_alloc&) { (h)->err = "out of host memory"; return TSNET_ERR_NOMEM; } \
      catch (const std::exception& e) { (h)->err = e.what(); return TSNET_ERR_HIP; } \
    return TSNET_OK;

extern "C" {

int tsnet_abi_version(void) { return TSNET_ABI_VERSION; }

int tsnet_create(const tsnet_cfg* cfg, tsnet_handle* out) {
    if (!cfg || !out) { g_create_error = "null argument"; return TSNET_ERR_ARG; }
    *out = nullptr;
    auto bad = [&](const char* m) { g_create_error = m; return TSNET_ERR_ARG; };
    if (cfg->n_source < 1 || cfg->n_source > TSNET_MAX_SOURCES) return bad("n_source must be in 1..8");
    if (cfg->label_nc < 1) return bad("label_nc must be >= 1");
    if (cfg->n_downsampling < 1 || cfg->n_downsampling > 5) return bad("n_downsampling must be in 1..5");
    if (cfg->ngf < 8 || (cfg->ngf & (cfg->ngf - 1))) return bad("ngf must be a power of two >= 8");
    if (cfg->n_blocks < 0 || cfg->enc_blocks < 0) return bad("block counts must be >= 0");
    const int ds = 1 << cfg->n_downsampling;
    if (cfg->height < ds * 2 || cfg->width < ds * 2 || cfg->height % ds || cfg->width % ds)
        return bad("height/width must be multiples of 2^n_downsampling (and at least twice that)");
    if (cfg->max_batch < 1) return bad("max_batch must be >= 1");
    if (cfg->pose_composite && (cfg->height != 256 || cfg->width != 256))
        return bad("pose composite is defined for 256x256 frames only (TSNet_pose.py:277-280)");
    if (cfg->operand_mode < 0 || cfg->operand_mode > 2) return bad("operand_mode must be 0 (fp32-class), 1 (bf16 operands) or 2 (bf16 operands + bf16 storage)");
    try {
        tsnet_engine* e = new tsnet_engine();
        e->cfg = *cfg;
        e->K = cfg->n_source; e->Bmax = cfg->max_batch;
        for (int s = 0; s < TSNET_MAX_SOURCES; ++s) e->src_div[s] = 255.0f;
        e->C = cfg->ngf << cfg->n_downsampling;
        e->h = cfg->height / ds; e->w = cfg->width / ds; e->P = e->h * e->w;
        e->np = cfg->operand_mode != 0 ? 1 : 3;
        e->st16 = cfg->operand_mode == 2;
        e->build_layers();
        *out = e;
    } catch (const std::exception& ex) { g_create_error = ex.what(); return TSNET_ERR_NOMEM; }
    return TSNET_OK;
}

const char* tsnet_last_error(tsnet_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int tsnet_num_params(tsnet_handle h) { return h ? (int)h->params.size() : TSNET_ERR_ARG; }

int tsnet_param_info(tsnet_handle h, int index, const char** name, int64_t shape_out[4], int* rank) {
    API_BEGIN(h)
    if (index < 0 || index >= (int)h->params.size()) throw ArgError("param index out of range");
    const auto& p = h->params[index];
    if (name) *name = p.name.c_str();
    if (rank) *rank = (int)p.shape.size();
    if (shape_out) for (size_t i = 0; i < 4; ++i) shape_out[i] = i < p.shape.size() ? p.shape[i] : 1;
    API_END(h)
}

int tsnet_load_weights(tsnet_handle h, const char* name, const float* data, const int64_t* shape, int rank) {
    API_BEGIN(h)
    if (h->finalized) throw ArgError("load_weights after finalize");
    if (!name || !data || !shape) throw ArgError("null argument");
    auto it = h->pindex.find(name);
    if (it == h->pindex.end()) throw WeightError(std::string("unknown parameter '") + name + "'");
    auto& p = h->params[it->second];
    bool ok = rank == (int)p.shape.size();
    for (int i = 0; ok && i < rank; ++i) ok = shape[i] == p.shape[i];
    if (!ok) {
        std::string m = std::string("shape mismatch for '") + name + "': expected (";
        for (auto d : p.shape) m += std::to_string(d) + ",";
        m += ") got (";
        for (int i = 0; i < rank; ++i) m += std::to_string(shape[i]) + ",";
        throw WeightError(m + ")");
    }
    size_t n = 1;
    for (auto d : p.shape) n *= (size_t)d;
    p.host.resize(n);
    HIP_TRY(hipMemcpy(p.host.data(), data, n * sizeof(float), hipMemcpyDefault));
    p.loaded = true;
    API_END(h)
}

int tsnet_finalize(tsnet_handle h, void* stream) {
    API_BEGIN(h)
    if (h->finalized) throw ArgError("finalize called twice");
    for (auto& p : h->params)
        if (!p.loaded) throw WeightError("parameter '" + p.name + "' was never loaded");
    h->alloc_all((hipStream_t)stream);
    h->finalized = true;
    API_END(h)
}

void tsnet_destroy(tsnet_handle h) {
    if (!h) return;
    (void)hipFree(h->fin_counter); (void)hipFree(h->train_ws); (void)hipFree(h->amax);
    if (h->side_stream) (void)hipStreamDestroy(h->side_stream);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->ev_fork2) (void)hipEventDestroy(h->ev_fork2);
    if (h->ev_join2) (void)hipEventDestroy(h->ev_join2);
    if (h->ev_done) (void)hipEventDestroy(h->ev_done);
    (void)hipFree(h->wpack); (void)hipFree(h->arena); (void)hipFree(h->d_coords); (void)hipFree(h->d_gx); (void)hipFree(h->d_gy);
    delete h;
}

int tsnet_packed_weights(tsnet_handle h, void** dev_ptr, size_t* bytes) {
    API_BEGIN(h)
    if (!h->finalized) throw ArgError("packed_weights before finalize");
    if (dev_ptr) *dev_ptr = h->wpack;
    if (bytes) *bytes = h->wpack_floats * sizeof(float);
    API_END(h)
}

static void check_forward_args(tsnet_handle h, int B) {
    if (!h->finalized) throw ArgError("forward before finalize");
    if (B < 1 || B > h->Bmax) throw ArgError("batch size outside 1..max_batch");
}

int tsnet_set_source_divisors(tsnet_handle h, const float* div, int n) {
    API_BEGIN(h)
    if (n < 0 || n > TSNET_MAX_SOURCES || (n > 0 && !div)) throw ArgError("set_source_divisors: bad argument");
    for (int s = 0; s < TSNET_MAX_SOURCES; ++s) {
        const float d = s < n ? div[s] : 255.0f;
        if (!(d > 0.f) || !std::isfinite(d)) throw ArgError("set_source_divisors: divisors must be positive and finite");
        h->src_div[s] = d;
    }
    h->cached_B = 0;                   // cached source features were encoded with the previous divisors
    API_END(h)
}

int tsnet_set_sources(tsnet_handle h, const float* const* src_img, const float* const* src_lbl,
                      const float* const* src_bbox, int B, void* stream) {
    API_BEGIN(h)
    check_forward_args(h, B);
    if (!src_img || !src_lbl || !src_bbox) throw ArgError("null source list");
    for (int s = 0; s < h->K; ++s)
        if (!src_img[s] || !src_lbl[s] || !src_bbox[s]) throw ArgError("null source tensor (need n_source entries)");
    Ctx ctx; ctx.stream = (hipStream_t)stream; ctx.timing = h->timing.on ? &h->timing : nullptr;
    h->set_sources(ctx, src_img, src_lbl, src_bbox, B);
    API_END(h)
}

int tsnet_forward_target(tsnet_handle h, const float* tar_lbl, const float* tar_bbox,
                         float* out_rgb, float* out_flow, int B, void* stream) {
    API_BEGIN(h)
    check_forward_args(h, B);
    if (!tar_lbl || !tar_bbox || !out_rgb) throw ArgError("null target/output tensor");
    if (h->cached_B != B) throw ArgError("forward_target: batch differs from the cached sources (call tsnet_set_sources first)");
    Ctx ctx; ctx.stream = (hipStream_t)stream; ctx.timing = h->timing.on ? &h->timing : nullptr;
    h->forward_target(ctx, tar_lbl, tar_bbox, out_rgb, out_flow, B);
    API_END(h)
}

int tsnet_forward(tsnet_handle h, const float* const* src_img, const float* const* src_lbl, const float* const* src_bbox,
                  const float* tar_lbl, const float* tar_bbox, float* out_rgb, float* out_flow, int B, void* stream) {
    if (h && h->finalized && h->side_stream && !h->timing.on && tar_lbl && tar_bbox && out_rgb) {
        // Full forward on two lanes: the driving-frame chain (label encoder, L2 norm, target half of FuseNet conv1 --
        // small launches that leave most CUs idle at B = 4) runs on the engine's side stream while the caller's stream
        // encodes the sources; they join before the flow kernel.  Same kernels, same arithmetic: the result is
        // bit-identical to tsnet_set_sources + tsnet_forward_target (tested).  Per-kernel timing runs them in sequence.
        API_BEGIN(h)
        check_forward_args(h, B);
        hipStream_t main = (hipStream_t)stream;
        HIP_TRY(hipEventRecord(h->ev_fork, main));
        HIP_TRY(hipStreamWaitEvent(h->side_stream, h->ev_fork, 0));
        Ctx cs; cs.stream = h->side_stream; cs.lane = 1;
        // from here on the side lane has work in flight: whatever happens below (an exception included), the caller's stream waits
        // for it before this call returns -- the side lane must not outlive the call
        SideJoin join{h->side_stream, main, h->ev_join};
        if (!src_img || !src_lbl || !src_bbox) throw ArgError("null source list");
        for (int s = 0; s < h->K; ++s)
            if (!src_img[s] || !src_lbl[s] || !src_bbox[s]) throw ArgError("null source tensor (need n_source entries)");
        h->target_chain(cs, tar_lbl, B);
        Ctx cm; cm.stream = main;
        h->set_sources(cm, src_img, src_lbl, src_bbox, B, h->side_stream);      // (the bounding-box copies ride the side lane, ahead of the flow kernel)
        HIP_TRY(hipEventRecord(h->ev_join, h->side_stream));
        HIP_TRY(hipStreamWaitEvent(main, h->ev_join, 0));
        join.done = true;
        Ctx ctx; ctx.stream = main;
        h->forward_rest(ctx, tar_bbox, out_rgb, out_flow, B);
        API_END(h)
    }
    int rc = tsnet_set_sources(h, src_img, src_lbl, src_bbox, B, stream);
    if (rc != TSNET_OK) return rc;
    return tsnet_forward_target(h, tar_lbl, tar_bbox, out_rgb, out_flow, B, stream);
}

int tsnet_train_extras(tsnet_handle h, const float* const* src_img, const float* tar_img, int B,
                       float* warp_src_img, float* losses, void* stream) {
    API_BEGIN(h)
    check_forward_args(h, B);
    if (h->last_B != B) throw ArgError("train_extras: call tsnet_forward with the same batch first (uses its flows and features)");
    if (!src_img || !tar_img || !warp_src_img || !losses) throw ArgError("train_extras: null tensor");
    const int K = h->K, H = h->cfg.height, W = h->cfg.width, hh = h->h, ww = h->w, P = h->P, C = h->C;
    for (int s = 0; s < K; ++s) if (!src_img[s]) throw ArgError("train_extras: null source image (need n_source entries)");
    if (H % hh || W % ww || H / hh != W / ww) throw ArgError("train_extras: image size must be a multiple of the feature size");
    hipStream_t st = (hipStream_t)stream;
    const int N = K * B, HW = H * W, chunks = 16, ncos = 256;
    // workspace: gen mean/std (2*N*3) + ref mean/std (2*B*3) floats, then doubles: l1 partials (N*3*chunks) + cos partials
    const size_t nf = (size_t)2 * h->K * h->Bmax * 3 + 2 * h->Bmax * 3, nfp = (nf + 3) / 4 * 4;
    const size_t nd = (size_t)h->K * h->Bmax * 3 * chunks + ncos;
    if (!h->train_ws) HIP_TRY(hipMalloc((void**)&h->train_ws, nfp * sizeof(float) + nd * sizeof(double)));
    float* gen_mean = h->train_ws; float* gen_std = gen_mean + N * 3;
    float* ref_mean = h->train_ws + 2 * h->K * h->Bmax * 3; float* ref_std = ref_mean + B * 3;
    double* l1_part = reinterpret_cast<double*>(h->train_ws + nfp); double* cos_part = l1_part + (size_t)h->K * h->Bmax * 3 * chunks;

    PatchWarpArgs pa{};
    for (int s = 0; s < K; ++s) { pa.src[s] = src_img[s]; pa.div[s] = h->src_div[s]; }
    pa.flow = h->flow; pa.out = warp_src_img; pa.K = K; pa.B = B; pa.H = H; pa.W = W; pa.h = hh; pa.w = ww; pa.down = H / hh;
    hipLaunchKernelGGL(patch_warp_kernel, dim3(ew_grid((size_t)N * HW)), dim3(256), 0, st, pa);
    check_launch("patch_warp");
    hipLaunchKernelGGL(frame_stats_kernel, dim3(3, B), dim3(256), 0, st, tar_img, 3, HW, 255.0f, ref_mean, ref_std);      // TSNet.py:329-330
    check_launch("frame_stats(tar)");
    hipLaunchKernelGGL(frame_stats_kernel, dim3(3, N), dim3(256), 0, st, warp_src_img, 3, HW, 1.0f, gen_mean, gen_std);   // :381-382
    check_launch("frame_stats(warp)");
    const bool pose = h->cfg.pose_composite != 0;       // TSNet_pose.py:399-400 composite before the L1; no alignment loss
    float bg[3];
    for (int c = 0; c < 3; ++c) bg[c] = (-h->cfg.pose_mean[c]) / 255.0f;            // TSNet_pose.py:276
    hipLaunchKernelGGL(renorm_l1_kernel, dim3(chunks, 3, N), dim3(256), 0, st, warp_src_img, tar_img, B, HW, gen_mean, gen_std, ref_mean, ref_std, l1_part,
                       W, pose ? 64 : 0, pose ? 192 : 0, bg[0], bg[1], bg[2]);
    check_launch("renorm_l1");
    if (!pose) {
        hipLaunchKernelGGL(cosine_partial_kernel, dim3(ncos), dim3(256), 0, st, h->pg, h->sg, B * P, C, cos_part);
        check_launch("cosine_partial");
    }
    hipLaunchKernelGGL(train_losses_kernel, dim3(1), dim3(64), 0, st, l1_part, K, B * 3 * chunks, (double)B * 3 * HW, cos_part, pose ? 0 : ncos, (double)B * P, losses);
    check_launch("train_losses");
    API_END(h)
}

int tsnet_stage_ptr(tsnet_handle h, const char* name, const float** dev_ptr, size_t* count) {
    API_BEGIN(h)
    if (!h->finalized || h->last_B < 1) throw ArgError("stage_ptr before a forward");
    const size_t fe = (size_t)h->P * h->C, B = h->last_B;
    std::string n = name ? name : "";
    const float* p = nullptr; size_t c = 0;
    if (n == "src_fea") { p = h->X; c = (size_t)h->K * B * fe; }
    else if (n == "tar_fea") { p = h->tar_fea; c = B * fe; }
    else if (n == "pg") { p = h->pg; c = B * fe; }
    else if (n == "sg") { p = h->sg; c = B * fe; }
    else if (n == "dec_map") { p = h->D; c = B * fe; }
    else if (n.rfind("dec_up", 0) == 0 && n.size() == 7 && n[6] >= '0' && n[6] < '0' + h->cfg.n_downsampling) {
        const int i = n[6] - '0';        // raw output of the i-th decoder up-convolution (before its InstanceNorm)
        p = h->R[i]; c = B * ((size_t)(h->h << (i + 1)) * (h->w << (i + 1)) * (h->C >> (i + 1)));
        if (h->st16) {                   // stored as bf16: widened into the (now idle, twice as large) upsampled-input buffer of the same level
            // ordered behind the last forward -- the engine's streams are non-blocking, the NULL stream orders nothing against them
            // ... and not on the caller's stream handle either, which may be gone by now: the forward leaves an event behind its last
            // launch; the engine's own side stream waits for it, widens and is drained before this call returns
            HIP_TRY(hipStreamWaitEvent(h->side_stream, h->ev_done, 0));
            hipLaunchKernelGGL(bf16_widen_kernel, dim3(ew_grid(c)), dim3(256), 0, h->side_stream, reinterpret_cast<const unsigned short*>(h->R[i]), h->U[i], c);
            check_launch("bf16_widen");
            HIP_TRY(hipStreamSynchronize(h->side_stream));
            p = h->U[i];
        }
    }
    else throw ArgError("unknown stage '" + n + "'");
    if (dev_ptr) *dev_ptr = p;
    if (count) *count = c;
    API_END(h)
}

double tsnet_forward_macs(tsnet_handle h, int B) {
    if (!h) return 0.0;
    // SURVEY.md section 8-a closed form, generalised: every conv = M*N*K_real, plus the correlation
    // (one masked GEMM per source) and the soft-argmax.
    const tsnet_cfg& c = h->cfg;
    const double H = c.height, W = c.width, P = h->P, C = h->C, K = h->K;
    auto enc = [&](double cin, int nres) {
        double t = H * W * c.ngf * cin * 49;
        for (int i = 0; i < c.n_downsampling; ++i) {
            const double ho = H / (2 << i), wo = W / (2 << i), ci = c.ngf << i;
            t += ho * wo * (2 * ci) * ci * 9;
        }
        return t + nres * 2.0 * P * C * C * 9;
    };
    const double coords = c.addcoords ? 3 : 0;
    const double fuse = 2.0 * P * (2 * C) * (2 * C) * 9 + P * C * (2 * C);
    double dec = P * C * (2 * C) + c.n_blocks * 2.0 * P * C * C * 9;
    for (int i = 0; i < c.n_downsampling; ++i) {
        const double sp = P * (double)(1 << (2 * (i + 1))), ci = C / (1 << i);
        dec += sp * (ci / 2) * ci * 9;
    }
    dec += H * W * 3 * c.ngf * 49;
    const double corr = K * P * P * C + K * P * P * 2;
    return B * (K * enc(3 + c.label_nc + coords, c.enc_blocks) + enc(c.label_nc + coords, 0) + K * fuse + dec + corr);
}

int tsnet_timing_enable(tsnet_handle h, int on) {
    API_BEGIN(h)
    h->timing.on = on != 0;
    API_END(h)
}

int tsnet_timing_read(tsnet_handle h, double ms_out[TSNET_TIMING_CLASSES], int64_t launches_out[TSNET_TIMING_CLASSES], int reset) {
    API_BEGIN(h)
    h->timing.collect();
    for (int i = 0; i < TSNET_TIMING_CLASSES; ++i) {
        if (ms_out) ms_out[i] = h->timing.ms[i];
        if (launches_out) launches_out[i] = h->timing.launches[i];
        if (reset) { h->timing.ms[i] = 0; h->timing.launches[i] = 0; }
    }
    API_END(h)
}

// ---------------------------------------------------------------------------------------------
// single operators
const char* tsnet_op_last_error(void) { return g_op_error.c_str(); }

#define OP_BEGIN try {
#define OP_END                                                                   \
    } catch (const ArgError& e) { g_op_error = e.what(); return TSNET_ERR_ARG; }  \
      catch (const std::exception& e) { g_op_error = e.what(); return TSNET_ERR_HIP; } \
    return TSNET_OK;

namespace {
// one convolution layer with its own packed weights, for the operator entry points
struct OpLayer {
    ConvLayer L;
    DevBufs mem;                  // a member: destroyed even when the constructor throws after the first allocation
    float* wd = nullptr; float* bd = nullptr; float* un = nullptr; unsigned short* wq = nullptr;
    OpLayer(const float* w_oihw, const float* bias, int Cin, int Cout, int ks, int stride, int pad, int reflect, int nprod, hipStream_t s, int form = 0) {
        L.name = "op"; L.cin_real = Cin; L.cin_pad = conv_cin_pad(Cin); L.cin_total = Cin; L.cout = Cout; L.ks = ks; L.stride = stride; L.pad = pad;
        L.reflect = reflect; L.form = form; L.kpad = form == 1 ? conv_kpad_w1(L.cin_pad) : conv_kpad(ks, L.cin_pad); L.npad = conv_npad(Cout);
        if (L.cin_pad != Cin) throw ArgError("conv2d op: Cin must be 8 or a multiple of 16 (pad channels with zeros)");
        if (form == 1 && (ks != 3 || stride != 1 || pad != 1)) throw ArgError("conv2d op: the Winograd form is for 3 x 3 / stride 1 / pad 1");
        size_t wn = (size_t)Cout * Cin * ks * ks;
        std::vector<float> hw(wn);
        HIP_TRY(hipMemcpy(hw.data(), w_oihw, wn * sizeof(float), hipMemcpyDefault));
        if (form == 1) { hw = winograd_x_filters(hw); wn = hw.size(); }
        float mx = 0.f;
        for (float v : hw) mx = std::max(mx, std::fabs(v));
        const int planes = nprod == 1 ? 1 : 2;
        const int sw = (nprod != 1 && mx > 0.f) ? h2_scale_log2(mx) : 0;
        wd = mem.alloc<float>(wn * sizeof(float));
        wq = mem.alloc<unsigned short>((size_t)L.kpad * L.npad * 2 * planes);
        un = mem.alloc<float>(sizeof(float));
        HIP_TRY(hipMemcpy(wd, hw.data(), wn * sizeof(float), hipMemcpyHostToDevice));
        const float unscale = std::ldexp(1.0f, -sw);
        HIP_TRY(hipMemcpy(un, &unscale, sizeof(float), hipMemcpyHostToDevice));
        pack_layer(wd, L, wq, planes, std::ldexp(1.0f, sw), s);
        if (bias) {
            bd = mem.alloc<float>(Cout * sizeof(float));
            HIP_TRY(hipMemcpy(bd, bias, Cout * sizeof(float), hipMemcpyDefault));
        }
        L.wq = wq; L.w_unscale = nprod == 1 ? nullptr : un; L.bias = bd;
    }
    OpLayer(const OpLayer&) = delete;
    OpLayer& operator=(const OpLayer&) = delete;
};
}  // namespace

int tsnet_op_conv2d(const float* x, int N, int H, int W, int Cin, const float* w_oihw, const float* bias, int Cout,
                    int ksize, int stride, int pad, int pad_mode, const float* in_alpha, const float* in_beta, int in_relu,
                    float bound, int nprod, int kernel, int tile, float* y, void* stream) {
    OP_BEGIN
    if (!x || !w_oihw || !y) throw ArgError("null tensor");
    if (nprod != 1 && nprod != 3 && nprod != 4) throw ArgError("conv2d op: 1 (bf16 operands), 3 or 4 products");
    hipStream_t s = (hipStream_t)stream;
    Ctx ctx; ctx.stream = s;
    OpLayer op(w_oihw, bias, Cin, Cout, ksize, stride, pad, pad_mode, nprod, s, kernel == 3 ? 1 : 0);     // kernel 3: Winograd-along-x form (conv_w1.hpp)
    ConvCall c; c.x = x; c.alpha = in_alpha; c.beta = in_beta; c.relu = in_relu; c.bound = bound; c.N = N; c.H = H; c.W = W; c.y = y;
    c.nprod = nprod; c.kernel = kernel == 3 ? 0 : kernel; c.tile = tile;       // kernel 3: tile = tiles per workgroup (0 = the launcher's choice)
    run_conv(ctx, op.L, c);
    HIP_TRY(hipStreamSynchronize(s));
    OP_END
}

int tsnet_op_conv2d_cat(const float* x, const float* x2, int N, int H, int W, int C1, int C2, int x2_nmod, const float* w_oihw, const float* bias, int Cout,
                        int ksize, int stride, int pad, int pad_mode, float bound, int nprod, float* y, void* stream) {
    OP_BEGIN
    if (!x || !x2 || !w_oihw || !y) throw ArgError("null tensor");
    hipStream_t s = (hipStream_t)stream;
    Ctx ctx; ctx.stream = s;
    OpLayer op(w_oihw, bias, C1 + C2, Cout, ksize, stride, pad, pad_mode, nprod, s);
    ConvCall c; c.x = x; c.x2 = x2; c.csplit = C1; c.x2_nmod = x2_nmod; c.bound = bound; c.N = N; c.H = H; c.W = W; c.y = y; c.nprod = nprod;
    run_conv(ctx, op.L, c);
    HIP_TRY(hipStreamSynchronize(s));
    OP_END
}

int tsnet_op_head(const float* x, int N, int H, int W, int C, const float* in_alpha, const float* in_beta, const float* w_oihw, const float* bias,
                  int composite, const float* bg, float* y, void* stream) {
    OP_BEGIN
    if (!x || !w_oihw || !bias || !y) throw ArgError("null tensor");
    if (C < 4 || (C & 3)) throw ArgError("head op: C must be a multiple of 4");
    if (in_alpha && !in_beta) throw ArgError("head op: alpha without beta");
    hipStream_t s = (hipStream_t)stream;
    const size_t wn = (size_t)3 * C * 49;
    DevBufs mem;
    float* wd = mem.alloc<float>(wn * sizeof(float));
    float* tab = mem.alloc<float>(((size_t)49 * C * 4 + 64) * sizeof(float));
    float* bd = mem.alloc<float>(4 * sizeof(float));
    HIP_TRY(hipMemsetAsync(tab, 0, ((size_t)49 * C * 4 + 64) * sizeof(float), s));
    HIP_TRY(hipMemcpy(wd, w_oihw, wn * sizeof(float), hipMemcpyDefault));
    HIP_TRY(hipMemcpy(bd, bias, 3 * sizeof(float), hipMemcpyDefault));
    hipLaunchKernelGGL(pack_head_weights_kernel, dim3(64), dim3(256), 0, s, wd, tab, C);
    check_launch("pack_head_weights");
    HeadArgs ha{};
    ha.x = x; ha.alpha = in_alpha; ha.beta = in_alpha ? in_beta : nullptr; ha.w = tab; ha.bias = bd; ha.y = y;
    ha.N = N; ha.H = H; ha.W = W; ha.C = C;
    ha.composite = composite & 1; ha.fore_x0 = 64; ha.fore_x1 = 192;
    for (int c = 0; c < 3; ++c) ha.bg[c] = bg ? bg[c] : 0.f;
    launch_head(ha, H, W, N, s, composite >> 8);               // (bits 8..: tile rows to force -- operator tests; the forward passes 0 / 1)
    HIP_TRY(hipStreamSynchronize(s));
    OP_END
}

int tsnet_op_instnorm_stats(const float* x, int N, int HW, int C, float* alpha, float* beta, void* stream) {
    OP_BEGIN
    if (!x || !alpha || !beta) throw ArgError("null tensor");
    Ctx ctx; ctx.stream = (hipStream_t)stream;
    DevBufs mem;
    double* part = mem.alloc<double>((size_t)N * 64 * C * 2 * sizeof(double));
    run_stats(ctx, x, N, HW, C, part, alpha, beta);
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    OP_END
}

int tsnet_op_norm_act(const float* x, const float* alpha, const float* beta, int relu, const float* resid,
                      int N, int HW, int C, float* y, void* stream) {
    OP_BEGIN
    if (!x || !y) throw ArgError("null tensor");
    Ctx ctx; ctx.stream = (hipStream_t)stream;
    run_norm_act(ctx, x, alpha, beta, relu, resid, N, HW, C, y);
    OP_END
}

int tsnet_op_upsample2x(const float* x, const float* alpha, const float* beta, int relu, int N, int H, int W, int C, float* y, void* stream) {
    OP_BEGIN
    if (!x || !y) throw ArgError("null tensor");
    Ctx ctx; ctx.stream = (hipStream_t)stream;
    run_upsample(ctx, x, alpha, beta, relu, N, H, W, C, y);
    OP_END
}

// tar_fea (B), src_fea / src_bbox / flow (K*B, image k*B + b).  repeat > 1 re-launches the flow kernel (identical results) for timing.
static void op_flow_impl(const float* tar_fea, const float* src_fea, const float* tar_bbox, const float* src_bbox,
                         int B, int K, int h, int w, int C, int H, int W, float* flow, int variant, int repeat, float* ms_out, hipStream_t stream) {
    if (!tar_fea || !src_fea || !tar_bbox || !src_bbox || !flow) throw ArgError("null tensor");
    if (B < 1 || K < 1 || K > 8) throw ArgError("flow op: 1 <= K <= 8 sources, B >= 1");
    if (H % h || W % w) throw ArgError("flow op: bbox size must be a multiple of the feature size");
    Ctx ctx; ctx.stream = stream;
    const int P = h * w, NB = K * B;
    DevBufs bufs;
    auto* that = bufs.alloc<unsigned short>(flow_plane_halves(B, P, C) * 2);
    auto* shat = bufs.alloc<unsigned short>(flow_plane_halves(NB, P, C) * 2);
    auto* gx = bufs.alloc<float>(w * sizeof(float));
    auto* gy = bufs.alloc<float>(h * sizeof(float));
    std::vector<float> hx(w), hy(h);
    linspace_pm1(w, hx.data()); linspace_pm1(h, hy.data());
    HIP_TRY(hipMemcpy(gx, hx.data(), w * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(gy, hy.data(), h * sizeof(float), hipMemcpyHostToDevice));
    FlowArgs fa{};
    const int G = flowp_plan(B, h, w, C);
    if (G) {
        fa.part = bufs.alloc<unsigned long long>(flowp_part_words(NB, P, flowp_slices(h, w)) * 8);
        fa.cnt = bufs.alloc<int>((size_t)B * (P / 64) * sizeof(int));
        HIP_TRY(hipMemsetAsync(fa.cnt, 0, (size_t)B * (P / 64) * sizeof(int), ctx.stream));
    }
    run_l2norm_split(ctx, tar_fea, that, B, P, C);
    run_l2norm_split(ctx, src_fea, shat, NB, P, C);
    fa.tq = that; fa.sq = shat; fa.tar_bbox = tar_bbox; fa.gx = gx; fa.gy = gy; fa.flow = flow;
    for (int k = 0; k < K; ++k) fa.src_bbox[k] = src_bbox + (size_t)k * B * H * W;
    fa.B = B; fa.P = P; fa.C = C; fa.h = h; fa.w = w; fa.H = H; fa.W = W; fa.sy = H / h; fa.sx = W / w;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ms_out) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1)); }
    run_flow(ctx, fa, NB, variant);
    if (ms_out) HIP_TRY(hipEventRecord(e0, ctx.stream));
    for (int r = 1; r < repeat; ++r) run_flow(ctx, fa, NB, variant);
    if (ms_out) HIP_TRY(hipEventRecord(e1, ctx.stream));
    HIP_TRY(hipStreamSynchronize(ctx.stream));
    if (ms_out) {
        float ms = 0.f;
        if (repeat > 1) { HIP_TRY(hipEventElapsedTime(&ms, e0, e1)); ms /= (float)(repeat - 1); }
        *ms_out = ms;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
}

int tsnet_op_flow(const float* tar_fea, const float* src_fea, const float* tar_bbox, const float* src_bbox,
                  int B, int h, int w, int C, int H, int W, float* flow, void* stream) {
    OP_BEGIN
    op_flow_impl(tar_fea, src_fea, tar_bbox, src_bbox, B, 1, h, w, C, H, W, flow, 0, 1, nullptr, (hipStream_t)stream);
    OP_END
}

int tsnet_op_flow_k(const float* tar_fea, const float* src_fea, const float* tar_bbox, const float* src_bbox,
                    int B, int K, int h, int w, int C, int H, int W, float* flow, int variant, int repeat, float* ms_out, void* stream) {
    OP_BEGIN
    op_flow_impl(tar_fea, src_fea, tar_bbox, src_bbox, B, K, h, w, C, H, W, flow, variant, repeat, ms_out, (hipStream_t)stream);
    OP_END
}

int tsnet_flow_plan(int B, int h, int w, int C) { return (B < 1 || h < 1 || w < 1 || C < 8 || (C & 7)) ? -1 : flowp_plan(B, h, w, C); }

int tsnet_op_warp(const float* src_fea, const float* flow, int B, int h, int w, int C, float* out, void* stream) {
    OP_BEGIN
    if (!src_fea || !flow || !out) throw ArgError("null tensor");
    if (C & 3) throw ArgError("warp op: C must be a multiple of 4");
    Ctx ctx; ctx.stream = (hipStream_t)stream;
    run_warp(ctx, src_fea, flow, out, B, 1, h, w, C);
    OP_END
}

int tsnet_op_warp_k(const float* src_fea, const float* flow, int B, int K, int h, int w, int C, float* out, int repeat, float* ms_out, void* stream) {
    OP_BEGIN
    if (!src_fea || !flow || !out) throw ArgError("null tensor");
    if (C & 3) throw ArgError("warp op: C must be a multiple of 4");
    if (B < 1 || K < 1 || K > TSNET_MAX_SOURCES || h < 1 || w < 1) throw ArgError("warp op: bad shape");
    Ctx ctx; ctx.stream = (hipStream_t)stream;
    if (ms_out) *ms_out = 0.f;
    run_warp(ctx, src_fea, flow, out, B, K, h, w, C);
    if (repeat > 1) {                    // launches 2 .. repeat between two events: the kernel ALONE (tools/warp_bench.py)
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, ctx.stream));
        for (int i = 1; i < repeat; ++i) run_warp(ctx, src_fea, flow, out, B, K, h, w, C);
        HIP_TRY(hipEventRecord(e1, ctx.stream));
        HIP_TRY(hipEventSynchronize(e1));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (ms_out) *ms_out = t / (float)(repeat - 1);
    }
    OP_END
}

int tsnet_frame_stats(const float* x, int B, int C, int HW, float div, float* mean, float* std_unbiased, void* stream) {
    OP_BEGIN
    if (!x || !mean || !std_unbiased) throw ArgError("frame_stats: null tensor");
    if (B < 1 || C < 1 || HW < 1 || B > 65535 || !(div > 0.f)) throw ArgError("frame_stats: bad shape or divisor");
    hipLaunchKernelGGL(frame_stats_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, x, C, HW, div, mean, std_unbiased);
    check_launch("frame_stats");
    OP_END
}

int tsnet_demo_postprocess(const float* rec, int B, int H, int W, const float* gen_mean, const float* gen_std,
                           const float* ref_mean, const float* ref_std, const float* img_mean_over_255,
                           unsigned char* out_rgb, void* stream) {
    OP_BEGIN
    if (!rec || !gen_mean || !gen_std || !ref_mean || !ref_std || !img_mean_over_255 || !out_rgb) throw ArgError("demo_postprocess: null tensor");
    if (B < 1 || H < 1 || W < 1) throw ArgError("demo_postprocess: bad shape");
    DemoPostArgs a{rec, gen_mean, gen_std, ref_mean, ref_std, {img_mean_over_255[0], img_mean_over_255[1], img_mean_over_255[2]},
                   out_rgb, B, H * W};
    hipLaunchKernelGGL(demo_post_kernel, dim3(ew_grid((size_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, a);
    check_launch("demo_post");
    OP_END
}

int tsnet_fit_face_curves(const double* keypoints, int F, double* curves) {
    OP_BEGIN
    if (!keypoints || !curves || F < 1) throw ArgError("fit_face_curves: bad argument");
    for (int f = 0; f < F; ++f)
        for (int e = 0; e < kFaceSubEdges; ++e) {
            double pts[6];
            const int n = hFaceSubEdgeTable[e][2] < 0 ? 2 : 3;
            for (int i = 0; i < n; ++i) {
                const double* k = keypoints + ((size_t)f * kFaceKeypoints + hFaceSubEdgeTable[e][i]) * 2;
                pts[2 * i] = k[0]; pts[2 * i + 1] = k[1];
            }
            lm::fit_piece(pts, n, curves + ((size_t)f * kFaceSubEdges + e) * kCurveRec);
        }
    OP_END
}

int tsnet_fit_pose_curves(const double* pts, int F, int flags, double* curves) {
    OP_BEGIN
    if (!pts || !curves || F < 1) throw ArgError("fit_pose_curves: bad argument");
    for (int f = 0; f < F; ++f) {
        const double* P = pts + (size_t)f * kPosePts * 2;
        for (int e = 0; e < kPosePrims; ++e) {
            double* rec = curves + ((size_t)f * kPosePrims + e) * kCurveRec;
            for (int i = 0; i < kCurveRec; ++i) rec[i] = 0.0;
            int ia = 0, ib = 0;
            if (!pose_primitive_points(e, flags, ia, ib)) continue;
            const double two[4] = {P[2 * ia], P[2 * ia + 1], P[2 * ib], P[2 * ib + 1]};
            if (two[0] == 0.0 || two[2] == 0.0) continue;               // `0 not in x` (connect_keypoints): a missing point
            lm::fit_piece(two, 2, rec);
        }
    }
    OP_END
}

int tsnet_raster_face(const double* keypoints, const double* curves, int F, int h, int w, int bw, unsigned char* edges, unsigned char* bbox, void* stream) {
    OP_BEGIN
    if ((!edges && !bbox) || (edges && !curves) || (bbox && !keypoints)) throw ArgError("raster_face: null tensor");
    if (F < 1 || F > 65535 || h < 1 || w < 1 || bw < 1 || (double)h * w >= 2147483647.0) throw ArgError("raster_face: bad shape");
    hipStream_t s = (hipStream_t)stream;
    if (edges) {
        HIP_TRY(hipMemsetAsync(edges, 0, (size_t)F * h * w, s));
        hipLaunchKernelGGL(face_edges_kernel, dim3(kFaceSubEdges, F), dim3(64), 0, s, curves, edges, h, w, bw);
        check_launch("face_edges");
    }
    if (bbox) {
        hipLaunchKernelGGL(face_bbox_kernel, dim3(F), dim3(256), 0, s, keypoints, bbox, h, w);
        check_launch("face_bbox");
    }
    OP_END
}

int tsnet_raster_pose(const double* pts, const double* curves, int F, int h, int w, int win_x0, int win_y0, int win_x1, int win_y1, int flags,
                      unsigned char* labels, void* stream) {
    OP_BEGIN
    if (!pts || !curves || !labels) throw ArgError("raster_pose: null tensor");
    if (F < 1 || F > 65535 || h < 1 || w < 1 || (double)h * w >= 2147483647.0) throw ArgError("raster_pose: bad shape");
    if (win_x0 < 0 || win_y0 < 0 || win_x1 > w || win_y1 > h || win_x1 <= win_x0 || win_y1 <= win_y0) throw ArgError("raster_pose: window outside the frame");
    if (((uintptr_t)labels & 3) != 0) throw ArgError("raster_pose: labels must be 4-byte aligned (and padded to a multiple of 4 bytes)");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)F * (win_y1 - win_y0) * (win_x1 - win_x0), n4 = (n + 3) / 4 * 4;
    HIP_TRY(hipMemsetAsync(labels, 0, n4, s));
    hipLaunchKernelGGL(pose_edges_kernel, dim3(kPosePrims, F), dim3(64), 0, s, pts, curves, labels, h, w, win_x0, win_y0, win_x1, win_y1, flags);
    check_launch("pose_edges");
    hipLaunchKernelGGL(pose_order_to_class_kernel, dim3(ew_grid(n)), dim3(256), 0, s, labels, n);
    check_launch("pose_order_to_class");
    OP_END
}

int tsnet_label_bbox(const unsigned char* labels, int F, int h, int w, unsigned char* bbox, void* stream) {
    OP_BEGIN
    if (!labels || !bbox) throw ArgError("label_bbox: null tensor");
    if (F < 1 || F > 65535 || h < 1 || w < 1 || (double)h * w >= 2147483647.0) throw ArgError("label_bbox: bad shape");
    hipLaunchKernelGGL(label_bbox_kernel, dim3(F), dim3(256), 0, (hipStream_t)stream, labels, bbox, h, w);
    check_launch("label_bbox");
    OP_END
}

int tsnet_resize_pad(const unsigned char* in, int F, int h, int w, const int* ytab, const int* xtab, int oh, int ow,
                     int pad_top, int pad_left, int OH, int OW, int binarise, float* out, void* stream) {
    OP_BEGIN
    if (!in || !ytab || !xtab || !out) throw ArgError("resize_pad: null tensor");
    if (F < 1 || h < 1 || w < 1 || oh < 1 || ow < 1 || pad_top < 0 || pad_left < 0 || pad_top + oh > OH || pad_left + ow > OW)
        throw ArgError("resize_pad: bad shape");
    hipLaunchKernelGGL(gather_pad_kernel, dim3(ew_grid((size_t)F * OH * OW)), dim3(256), 0, (hipStream_t)stream, in, F, h, w, ytab, xtab, oh, ow,
                       pad_top, pad_left, OH, OW, out, binarise);
    check_launch("gather_pad");
    OP_END
}

int tsnet_resize_label(const unsigned char* in, int F, int h, int w, int OH, int OW, const double* wts_rows, int lw_rows,
                       const double* wts_cols, int lw_cols, float* out, void* stream) {
    OP_BEGIN
    if (!in || !out) throw ArgError("resize_label: null tensor");
    if (F < 1 || h < 1 || w < 1 || OH < 1 || OW < 1) throw ArgError("resize_label: bad shape");
    if (lw_rows < -1 || lw_cols < -1 || lw_rows > 63 || lw_cols > 63 || (lw_rows >= 0 && !wts_rows) || (lw_cols >= 0 && !wts_cols))
        throw ArgError("resize_label: bad Gaussian kernel");
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)F * h * w;
    unsigned char *t0 = nullptr, *t1 = nullptr;
    double* dw = nullptr;
    int* mm = nullptr;
    struct Guard { unsigned char*& a; unsigned char*& b; double*& c; int*& d; ~Guard() { (void)hipFree(a); (void)hipFree(b); (void)hipFree(c); (void)hipFree(d); } } guard{t0, t1, dw, mm};
    const unsigned char* cur = in;
    // anti-aliasing passes: one per axis that shrinks (lw >= 0); the caller supplies the kernel halves (centre first) it computed in fp64
    for (int axis = 0; axis < 2; ++axis) {
        const int lw = axis ? lw_cols : lw_rows;
        const double* wts = axis ? wts_cols : wts_rows;
        if (lw < 0) continue;
        unsigned char*& dst = (cur == t0) ? t1 : t0;
        if (!dst) HIP_TRY(hipMalloc((void**)&dst, n));
        if (!dw) HIP_TRY(hipMalloc((void**)&dw, 2 * 64 * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(dw + axis * 64, wts, (lw + 1) * sizeof(double), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(gauss1d_u8_kernel, dim3(ew_grid(n)), dim3(256), 0, s, cur, dst, F, h, w, axis, dw + axis * 64, lw);
        check_launch("gauss1d_u8");
        cur = dst;
    }
    HIP_TRY(hipMalloc((void**)&mm, (size_t)2 * F * sizeof(int)));
    std::vector<int> init(2 * F);
    for (int f = 0; f < F; ++f) { init[2 * f] = 255; init[2 * f + 1] = 0; }
    HIP_TRY(hipMemcpyAsync(mm, init.data(), init.size() * sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(u8_minmax_kernel, dim3(std::min(64, (h * w + 255) / 256), F), dim3(256), 0, s, cur, F, h * w, mm);
    check_launch("u8_minmax");
    hipLaunchKernelGGL(resize_label_kernel, dim3(ew_grid((size_t)F * OH * OW)), dim3(256), 0, s, cur, F, h, w, OH, OW, mm, out);
    check_launch("resize_label");
    HIP_TRY(hipStreamSynchronize(s));                                  // host temporaries (init, the caller's kernels) and device scratch end here
    OP_END
}

int tsnet_vl2ch(const float* labels, int B, int HW, int num_classes, float* out, void* stream) {
    OP_BEGIN
    if (!labels || !out) throw ArgError("vl2ch: null tensor");
    if (B < 1 || HW < 1 || num_classes < 1) throw ArgError("vl2ch: bad shape");
    hipLaunchKernelGGL(onehot_kernel, dim3(ew_grid((size_t)B * num_classes * HW)), dim3(256), 0, (hipStream_t)stream, labels, out, B, HW, num_classes);
    check_launch("onehot");
    OP_END
}

int tsnet_bench_conv(int N, int H, int W, int Cin, int Cout, int ksize, int stride, int pad, int pad_mode, int norm,
                     int variant, int iters, float* ms_out, void* stream) {
    OP_BEGIN
    if (iters < 1 || !ms_out) throw ArgError("bench_conv: bad argument");
    hipStream_t s = (hipStream_t)stream;
    Ctx ctx; ctx.stream = s;
    // variant (-1 = the layer's own kernel and tile): bits 0-11 tile code (ConvCall::tile), bit 12 general kernel, bit 13 bf16 operands, bit 14 patch kernel,
    // bits 16-20 ablation mask, bit 21 cold weights (a fresh copy of the planes per launch), bit 22 weights five steps ahead, bit 23 two K groups, bits 24-27 experiment mask (8 = deep prefetch; the rest tools build), bits 28-30 XCD grid
    const int v = variant < 0 ? 0 : variant;
    const int nprod = (v & 8192) ? 1 : 3;
    const int Ho = (H + 2 * pad - ksize) / stride + 1, Wo = (W + 2 * pad - ksize) / stride + 1;
    const size_t xn = (size_t)N * H * W * Cin, yn = (size_t)N * Ho * Wo * Cout, wn = (size_t)Cout * Cin * ksize * ksize;
    DevBufs mem;
    float* x = mem.alloc<float>(xn * 4); float* y = mem.alloc<float>(yn * 4);
    float* al = mem.alloc<float>((size_t)N * Cin * 4); float* be = mem.alloc<float>((size_t)N * Cin * 4);
    // pseudo-random fill (not zeros: MI355X clocks higher on zero operands, cdna_hip_programming.md rule 25)
    std::vector<float> hbuf(std::max(std::max(xn, wn), (size_t)N * Cin));
    unsigned st = 12345u;
    auto fill = [&](float* d, size_t n, float scale, float off) {
        for (size_t i = 0; i < n; ++i) { st = st * 1664525u + 1013904223u; hbuf[i] = off + scale * ((float)(st >> 8) / 16777216.0f - 0.5f); }
        if (d) HIP_TRY(hipMemcpy(d, hbuf.data(), n * 4, hipMemcpyHostToDevice));
    };
    fill(x, xn, 2.f, 0.f); fill(al, (size_t)N * Cin, 1.f, 1.f); fill(be, (size_t)N * Cin, 0.5f, 0.f);
    fill(nullptr, wn, 0.1f, 0.f);
    {
        const int form = (v & 32768) ? 1 : 0;                       // bit 15: Winograd-along-x form
        OpLayer op(hbuf.data(), nullptr, Cin, Cout, ksize, stride, pad, pad_mode, nprod, s, form);
        // bit 21: COLD weights -- every launch reads another copy of the packed planes (24 copies: beyond the 256 MiB Infinity Cache for the
        // big layers), as consecutive layers of a forward do; the default re-launches one layer, whose planes stay cache-resident
        std::vector<std::unique_ptr<OpLayer>> cold;
        if (v & (1 << 21)) for (int i = 0; i < 24; ++i) cold.emplace_back(new OpLayer(hbuf.data(), nullptr, Cin, Cout, ksize, stride, pad, pad_mode, nprod, s, form));
        ConvCall c; c.x = x; c.N = N; c.H = H; c.W = W; c.y = y; c.bound = (norm & 1) ? 64.f : 1.f; c.nprod = nprod;
        c.tile = v & 4095;          /* Winograd form: tiles per workgroup (0 = the launcher's choice) */ c.kernel = (v & 4096) ? 1 : ((v & 16384) ? 2 : 0); c.abl = (v >> 16) & 31; c.opt = ((v >> 24) & 15) | ((v & (1 << 23)) ? 16 : 0) | ((v & (1 << 22)) ? 32 : 0);
        { const int gx = (v >> 28) & 7; c.xcd_gn = gx == 0 ? -1 : (gx == 1 ? 0 : 1 << (gx - 2)); }          // bits 28-30: 0 default, 1 linear, 2..5 grid with 1, 2, 4, 8 columns
        if (norm & 1) { c.alpha = al; c.beta = be; c.relu = 1; }
        if (norm & 2) {                                              // with the InstanceNorm statistics of the output, as the forward's layers run
            const size_t tpi = ((size_t)Ho * Wo + 63) / 64;
            c.stat_part = mem.alloc<double>(stat_part_doubles(N, tpi, Cout) * sizeof(double));
            c.fin_alpha = mem.alloc<float>((size_t)N * Cout * 4); c.fin_beta = mem.alloc<float>((size_t)N * Cout * 4);
            c.fin_counter = mem.alloc<int>(kFinCounterInts * sizeof(int));
            HIP_TRY(hipMemset(c.fin_counter, 0, kFinCounterInts * sizeof(int)));
        }
        for (int i = 0; i < 2; ++i) run_conv(ctx, op.L, c);
        hipEvent_t e0, e1;
        HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, s));
        for (int i = 0; i < iters; ++i) run_conv(ctx, cold.empty() ? op.L : cold[(size_t)i % cold.size()]->L, c);
        HIP_TRY(hipEventRecord(e1, s));
        HIP_TRY(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        *ms_out = ms / iters;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    OP_END
}

#ifdef TSNET_TOOLS
void tsnet_tools_set(int key, int value) { if (key >= 0 && key < 8) g_tools_knob[key] = value; }      // not in the product library / ABI header
#endif
void tsnet_debug_counters(int64_t out[4], int reset) {
    for (int i = 0; i < 4; ++i) { if (out) out[i] = g_launch_counters[i]; if (reset && i < 3) g_launch_counters[i] = 0; }
}

void tsnet_linspace(int n, float* out) { linspace_pm1(n, out); }
void tsnet_coord_table(int H, int W, float* out) { coord_table(H, W, out); }

}  // extern "C"
