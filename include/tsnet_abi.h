/*
 * tsnet_abi.h -- C ABI of the MI355X-native TS-Net generator forward path.
 *
 * The reference (nihaomiao/WACV23_TSNet) has NO plugin / operator / FFI boundary: callers
 * drive a Python object.  This header therefore defines the boundary
 * a binding for that object would use; every entry point cites the reference interface it
 * replaces.  Plain C: pointers and sizes only, no torch types.  All device pointers are
 * contiguous fp32 buffers in the caller's HIP context; `stream` is a hipStream_t passed as
 * void* (NULL = the null stream).  Work is enqueued stream-ordered and asynchronously; the
 * library never synchronises the device inside tsnet_forward*.
 *
 * Threading: one host thread per handle; handles are independent (one per GPU / process).
 * Errors: 0 = ok, negative = failure; tsnet_last_error(h) returns a message owned by the handle
 *         (tsnet_last_error(NULL) returns the message of the last failed tsnet_create).
 */
#ifndef TSNET_ABI_H
#define TSNET_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSNET_ABI_VERSION 5      /* 5: tsnet_op_warp_k (round 6); 4: tsnet_cfg.operand_mode = 2 (bf16 storage); tsnet_op_conv2d kernel = 3 (Winograd-along-x form); tsnet_op_flow_k, tsnet_flow_plan */
#define TSNET_MAX_SOURCES 8

enum {
    TSNET_OK = 0,
    TSNET_ERR_ARG = -1,     /* bad argument / shape / state */
    TSNET_ERR_WEIGHT = -2,  /* unknown, missing or mis-shaped parameter */
    TSNET_ERR_HIP = -3,     /* HIP runtime failure (message carries hipGetErrorString) */
    TSNET_ERR_NOMEM = -4
};

typedef struct tsnet_engine* tsnet_handle;

/* Constructor arguments of the reference model that shape the forward graph.
 * Replaces: TSNet.__init__ (model/TSNet.py:204-228), pose extras (model/TSNet_pose.py:214-215,276-280). */
typedef struct tsnet_cfg {
    int label_nc;        /* L: label channels (2 face, 25 pose) */
    int n_blocks;        /* decoder ResnetBlocks (0 quick-start, 4 demos) */
    int n_downsampling;  /* must be 3 with ngf=64 (FuseNet width is 2*ngf*2^n_down, TSNet.py:227) */
    int n_source;        /* K <= TSNET_MAX_SOURCES */
    int ngf;             /* 64 in every reference caller; a power of two >= 8 */
    int enc_blocks;      /* img_enc ResnetBlocks (Encoder default 9, TSNet.py:53) */
    int addcoords;       /* Encoder.coord_conv on/off (TSNet.py:89-90) */
    int pose_composite;  /* 1 = TSNet_pose use_mask epilogue (TSNet_pose.py:416-417); needs H=W=256 */
    float pose_mean[3];  /* BGR mean of the pose model (TSNet_pose.py:215) */
    int height, width;   /* input frame size (256x256 in the reference) */
    int max_batch;       /* largest B a forward will be called with (workspace is sized once) */
    int operand_mode;    /* 0 = fp32-class arithmetic (default: split operands, parity 1e-3 with the fp32 reference);
                          * 1 = bf16 operands (BASELINE.json configs[2] / [4]): every convolution input and weight is rounded to bf16,
                          *     one MFMA product, fp32 accumulate, fp32 InstanceNorm / softmax / RGB head.  Own tolerance.
                          * 2 = mode 1 + bf16 STORAGE of the large convolution-to-convolution activations (the encoders' 256^2 .. 64^2 maps,
                          *     the decoder's up-convolution outputs and upsampled inputs): statistics from the fp32 accumulators, the
                          *     stored tensor rounded once, widened exactly by its consumer. */
} tsnet_cfg;

/* ---- lifecycle ---------------------------------------------------------------------------
 * tsnet_create          <- TSNet(...) construction                       (model/TSNet.py:204-264)
 * tsnet_load_weights    <- net.load_state_dict(ckpt['img_enc'|...])      (demo/demo_face.py:126-129)
 *     `name` = "<net>.<state_dict key>", nets img_enc / lbl_enc / fuse_net / dec, e.g.
 *     "img_enc.model.13.conv_block.1.weight" (OIHW fp32) or "dec.map_conv.bias".
 *     `data` may be a host or a device pointer.  Unknown names -> TSNET_ERR_WEIGHT.
 * tsnet_finalize        <- networks.init_net's net.cuda()                (model/networks.py:116)
 *     checks that every parameter was loaded, re-packs weights into the kernels' K-major
 *     layout in HBM and allocates the workspace arena.  No allocation happens after it.
 */
int tsnet_abi_version(void);
int tsnet_create(const tsnet_cfg* cfg, tsnet_handle* out);
int tsnet_load_weights(tsnet_handle h, const char* name, const float* data, const int64_t* shape, int rank);
int tsnet_finalize(tsnet_handle h, void* stream);
void tsnet_destroy(tsnet_handle h);
const char* tsnet_last_error(tsnet_handle h);

/* Number of parameters the configuration expects and their names/shapes, so a binding can
 * validate a checkpoint before loading (key schema: SURVEY.md section 8-b). */
int tsnet_num_params(tsnet_handle h);
int tsnet_param_info(tsnet_handle h, int index, const char** name, int64_t shape_out[4], int* rank);

/* Packed-weight buffer (device) for replication across GPUs: rank 0 finalizes from a checkpoint,
 * other ranks finalize from zeros and receive this buffer by one RCCL broadcast
 * (SURVEY.md section 8-e).  The reference is single-GPU only; there is no reference call site. */
int tsnet_packed_weights(tsnet_handle h, void** dev_ptr, size_t* bytes);

/* ---- forward -----------------------------------------------------------------------------
 * tsnet_forward  <- TSNet.set_test_input(...) + TSNet.forward()  (model/TSNet.py:283-294, 309-407)
 *   src_img[i]  (B,3,H,W)  source frames exactly as the reference caller passes them, i.e. BEFORE
 *                          the /255 that set_test_input applies (TSNet.py:286)
 *   src_lbl[i]  (B,L,H,W)  src_bbox[i] (B,H,W)    i < n_source
 *   tar_lbl     (B,L,H,W)  tar_bbox    (B,H,W)
 *   out_rgb     (B,3,H,W)  = self.rec_tar_img     (TSNet.py:407; pose composite TSNet_pose.py:416-417)
 *   out_flow    NULL or (K,B,H/8,W/8,2) = self.warp_grid2d_list when return_flow (TSNet.py:369-370)
 */
int tsnet_forward(tsnet_handle h,
                  const float* const* src_img, const float* const* src_lbl, const float* const* src_bbox,
                  const float* tar_lbl, const float* tar_bbox,
                  float* out_rgb, float* out_flow, int B, void* stream);

/* Per-source divisor applied to the source IMAGES on load: 255 (default; the `/255.0` of set_test_input / set_train_input,
 * TSNet.py:267,286) or 1 for a source that is a previously generated frame already in [0,1] (`use_prev`, TSNet.py:269-276).
 * div: n host floats (n <= n_source; the rest reset to 255).  Drops the clip-mode source cache. */
int tsnet_set_source_divisors(tsnet_handle h, const float* div, int n);

/* Clip mode (caller pattern demo/demo_face.py:185-192: the same K
 * sources for every driving frame).  tsnet_set_sources runs img_enc once and caches its features;
 * tsnet_forward_target then costs only lbl_enc + both branches + decoder and returns the same
 * result as tsnet_forward on the same inputs. */
int tsnet_set_sources(tsnet_handle h, const float* const* src_img, const float* const* src_lbl,
                      const float* const* src_bbox, int B, void* stream);
int tsnet_forward_target(tsnet_handle h, const float* tar_lbl, const float* tar_bbox,
                         float* out_rgb, float* out_flow, int B, void* stream);

/* Training-mode extras of the forward (model/TSNet.py:327-331, 372-390, 402-405), computed
 * from the flows and features the LAST tsnet_forward / tsnet_forward_target left in the engine -- call it right after that
 * forward, same B, same source images.  src_img: n_source x (B,3,H,W) raw (the /255 of set_train_input is applied inside);
 * tar_img (B,3,H,W) raw.  warp_src_img (n_source,B,3,H,W): every source image warped patch-wise by its flow and
 * re-normalised to the target image's statistics (warp_src_img_list); losses: 2 device floats {loss_warp, loss_align}.
 * With cfg.pose_composite (the pose model, model/TSNet_pose.py:343-346, 386-404) the warped images get the fixed-background
 * composite before the L1 (:399-402) and losses[1] is 0: that model has no alignment loss. */
int tsnet_train_extras(tsnet_handle h, const float* const* src_img, const float* tar_img, int B,
                       float* warp_src_img, float* losses, void* stream);

/* Device copies of the stage tensors of the last forward, for stage-wise parity tests
 * (NHWC fp32).  name: "src_fea" (K*B,h,w,c; n = i*B+b), "tar_fea" (B,h,w,c), "pg", "sg" (B,h,w,c),
 * "dec_map" (B,h,w,c), "dec_up<i>" (B, h<<(i+1), w<<(i+1), c>>(i+1)): RAW output of the i-th decoder
 * up-convolution, before its InstanceNorm + ReLU.  Returns the element count through *count. */
int tsnet_stage_ptr(tsnet_handle h, const char* name, const float** dev_ptr, size_t* count);

/* Algorithmic work of one forward at batch B (multiply-accumulates, SURVEY.md section 8-d closed form). */
double tsnet_forward_macs(tsnet_handle h, int B);

/* Per-kernel-class timing of the next forward(s): when enabled the engine brackets every launch
 * class with hipEvents on the caller's stream (used by bench.py for the roofline object).
 * tsnet_timing_read returns accumulated milliseconds and launch counts per class. */
#define TSNET_TIMING_CLASSES 9
enum { TSNET_T_CONV = 0, TSNET_T_STATS = 1, TSNET_T_ELEMWISE = 2, TSNET_T_FLOW = 3, TSNET_T_WARP = 4,
       TSNET_T_PACK = 5, TSNET_T_UPSAMPLE = 6, TSNET_T_OTHER = 7,
       TSNET_T_CONV_RES = 8 /* the 3x3 convolutions of the ResnetBlocks: the dominant kernel, timed apart from TSNET_T_CONV */ };
int tsnet_timing_enable(tsnet_handle h, int on);
int tsnet_timing_read(tsnet_handle h, double ms_out[TSNET_TIMING_CLASSES], int64_t launches_out[TSNET_TIMING_CLASSES], int reset);

/* ---- single operators (NHWC fp32 device tensors), exported for op-level parity tests ------
 * These run the very kernels tsnet_forward uses.
 *
 * tsnet_op_conv2d <- nn.Conv2d (+ preceding nn.ReflectionPad2d / zero padding)  (TSNet.py:27,42,66,70,139,147,193)
 *   x (N,H,W,Cin) fp32, Cin = 8 or a multiple of 16 (pad channels with zeros); w OIHW (Cout,Cin,k,k) host or device; bias (Cout) or NULL;
 *   y (N,Ho,Wo,Cout).  ksize 1, 3 or 7; pad_mode 0 = zero, 1 = reflect.  in_alpha / in_beta (N*Cin each) or NULL: the producer's
 *   nn.InstanceNorm2d (+ nn.ReLU when in_relu) applied while the operand tile is staged, x' = max(alpha*x + beta, 0); zero padding pads x'.
 *   bound = an upper bound of |operand| after that transform (it fixes the power-of-two operand scale of the fp16 x 2 split).
 *   nprod = 3 (lo*hi, hi*lo, hi*hi), 4 (+ lo*lo; 3x3 / stride-1 patch kernel only) or 1 (bf16 operands, tsnet_cfg.operand_mode = 1).
 *   kernel: 0 = the kernel the forward runs this layer on at this frame size (patch kernels of conv_h2.hpp where the output splits into
 *   4 x 32 rectangles: 3x3 / stride 1, 3x3 / stride 2 / zero pad, 7x7 stem with 8 input channels; the general implicit GEMM of conv_h2r.hpp
 *   elsewhere), 1 = the general kernel, 2 = the patch kernel (error if the layer has none), 3 = the Winograd F(2,3)-along-x form of a
 *   3x3 / stride-1 / pad-1 layer on frames of whole 4 x 32 tiles (conv_w1.hpp: the kernel the forward runs its ResnetBlock / FuseNet / first
 *   up-convolution layers on; the op packs the transformed filters itself; `tile` = tiles per workgroup, 1, 2 or 3, 0 = the launcher's choice).
 *   tile: 0 = the launcher's choice; patch 3x3 / stride 1: 32, 64, 128 (4-row tiles), 2128 (2 rows x 128), 3128 (nprod = 1 only: 4 rows x 128
 *   with the four waves side by side, the bf16 modes' own tile), 20032 / 20064 (two-K-group tiles); patch 3x3 / stride 2: 64, 128 (4 rows)
 *   or 2128 (2 rows x 128: the forward's shape; 12128: its deep schedule, what a launch of at most two workgroups per CU runs); general kernel: 64 or 128 (16-deep steps, conv_h2r.hpp), 3064 / 3128 (64-deep steps,
 *   conv_g64.hpp, 64 / 128 rows: where the layer allows, the forward's choice).
 *   All one-group tiles of one kernel produce identical bits, and so do the two two-group tiles among themselves (tested); the two-group
 *   tiles, and patch vs general kernels, sum K in another association / order: agreement to fp32 rounding.
 * tsnet_op_conv2d_cat <- the same on torch.cat((x, x2), channel axis) formed on load (dec.map_conv on cat(pg, sg), TSNet.py:163):
 *   x (N,H,W,C1), x2 (x2_nmod,H,W,C2) read at image n % x2_nmod; C1 a multiple of 16.  No input transform.
 * tsnet_op_head <- the decoder's RGB head: ReflectionPad2d(3) + Conv2d(C -> 3, 7x7) + bias + Tanh (TSNet.py:151-152) on relu(alpha*x+beta),
 *   with the pose model's fixed-background composite (TSNet_pose.py:416-417: columns outside [64,192) <- bg) when composite != 0.
 *   x (N,H,W,C) NHWC, w (3,C,7,7), bias (3), bg 3 host floats or NULL; y (N,3,H,W) NCHW.  composite: bit 0 = the composite; bits 8.. = tile
 *   rows of the kernel to force (8, 16, 32; 0 = the launcher's choice by the number of workgroups: every choice gives the same bits).
 * tsnet_op_instnorm_stats <- nn.InstanceNorm2d statistics (TSNet.py:53; eps 1e-5, biased variance):
 *   alpha = 1/sqrt(var+eps), beta = -mean*alpha, each (N*C).
 * tsnet_op_norm_act   : y = alpha*x+beta (relu optional) ; if resid != NULL y += resid  (ResnetBlock tail, TSNet.py:48)
 * tsnet_op_upsample2x <- nn.Upsample(scale_factor=2, bilinear, align_corners=False) (TSNet.py:145), with the
 *   producer's InstanceNorm+ReLU fused on load when alpha/beta != NULL.
 * tsnet_op_flow       <- transformation branch up to the flow field (TSNet.py:319-323,339-365):
 *   tar_fea (B,h,w,C), src_fea (B,h,w,C) un-normalised NHWC; bboxes (B,H,W); flow (B,h,w,2).
 * tsnet_op_flow_k     <- the same for K sources per driving frame, as the model's loop over sources runs it (TSNet.py:336-366): tar_fea /
 *   tar_bbox hold B images, src_fea / src_bbox / flow K*B images (image k*B + b = source k of batch element b), 1 <= K <= 8.  Maps of
 *   >= 2048 positions (configs[4]: 64 x 64) run flow_kernel_p -- a workgroup keeps its 64 target positions for all K sources.  variant: 0 =
 *   as the forward runs it; 1 = flow_kernel whatever the plan says; 2 = flow_kernel_p without its exp pass (tools library only).  repeat > 1
 *   launches the flow kernel that many times (identical results); ms_out (nullable) receives the average time of launches 2 .. repeat by
 *   HIP events (tools/flow_bench.py), 0 when repeat <= 1.
 * tsnet_op_warp       <- F.grid_sample(bilinear, zeros, align_corners=False) (TSNet.py:366) on NHWC.
 * tsnet_op_warp_k     <- the same for K sources per driving frame fused with the mean over sources (TSNet.py:366, 392/400), as the forward
 *   runs it: src_fea / flow hold K*B images (image k*B + b), out (B,h,w,C).  repeat > 1 launches the kernel that many times; ms_out
 *   (nullable) receives the average time of launches 2 .. repeat by HIP events -- the kernel ALONE (tools/warp_bench.py).
 */
int tsnet_op_conv2d(const float* x, int N, int H, int W, int Cin,
                    const float* w_oihw, const float* bias, int Cout, int ksize, int stride, int pad, int pad_mode,
                    const float* in_alpha, const float* in_beta, int in_relu, float bound, int nprod, int kernel, int tile,
                    float* y, void* stream);
int tsnet_op_conv2d_cat(const float* x, const float* x2, int N, int H, int W, int C1, int C2, int x2_nmod,
                        const float* w_oihw, const float* bias, int Cout, int ksize, int stride, int pad, int pad_mode,
                        float bound, int nprod, float* y, void* stream);
int tsnet_op_head(const float* x, int N, int H, int W, int C, const float* in_alpha, const float* in_beta,
                  const float* w_oihw, const float* bias, int composite, const float* bg, float* y, void* stream);
int tsnet_op_instnorm_stats(const float* x, int N, int HW, int C, float* alpha, float* beta, void* stream);
int tsnet_op_norm_act(const float* x, const float* alpha, const float* beta, int relu, const float* resid,
                      int N, int HW, int C, float* y, void* stream);
int tsnet_op_upsample2x(const float* x, const float* alpha, const float* beta, int relu,
                        int N, int H, int W, int C, float* y, void* stream);
int tsnet_op_flow(const float* tar_fea, const float* src_fea, const float* tar_bbox, const float* src_bbox,
                  int B, int h, int w, int C, int H, int W, float* flow, void* stream);
int tsnet_op_flow_k(const float* tar_fea, const float* src_fea, const float* tar_bbox, const float* src_bbox,
                    int B, int K, int h, int w, int C, int H, int W, float* flow, int variant, int repeat, float* ms_out, void* stream);
/* Which kernel tsnet_op_flow / tsnet_op_flow_k / the forward run on B driving frames of h x w positions, C channels: 0 = flow_kernel (a
 * workgroup per (source, batch element, 64 targets)); G >= 1 = flow_kernel_p with G workgroups per target tile (csrc/flow_warp.hpp); -1 bad
 * arguments.  Host logic only. */
int tsnet_flow_plan(int B, int h, int w, int C);
int tsnet_op_warp(const float* src_fea, const float* flow, int B, int h, int w, int C, float* out, void* stream);
int tsnet_op_warp_k(const float* src_fea, const float* flow, int B, int K, int h, int w, int C, float* out, int repeat, float* ms_out, void* stream);
const char* tsnet_op_last_error(void);

/* ---- demo post-processing (demo/demo_face.py:96-105,180-198 = demo/demo_pose.py:98-107,
 * 186-203).  The reference does this per frame on the host with numpy + cv2; these entry points keep the frame on the
 * device and hand back packed uint8 RGB.
 * tsnet_frame_stats      <- x.view(B,C,-1) / div, then .mean(dim=2) and .std(dim=2) (unbiased): mean, std are (B*C) floats.
 *                           div = 255 for the reference image (:180), 1 for a generated frame (:195-196).
 * tsnet_demo_postprocess <- ((rec - gen_mean) / gen_std) * ref_std + ref_mean (:197-198), then sample_img (:96-105):
 *                           + img_mean_over_255, clip to [0,1], * 255, BGR -> RGB, and the .astype('uint8') of :222.
 *                           rec (B,3,H,W) fp32; gen_mean / gen_std (B*3); ref_mean / ref_std (3) device floats;
 *                           img_mean_over_255: 3 HOST floats (IMG_MEAN / 255); out_rgb (B,H,W,3) bytes. */
int tsnet_frame_stats(const float* x, int B, int C, int HW, float div, float* mean, float* std_unbiased, void* stream);
int tsnet_demo_postprocess(const float* rec, int B, int H, int W, const float* gen_mean, const float* gen_std,
                           const float* ref_mean, const float* ref_std, const float* img_mean_over_255,
                           unsigned char* out_rgb, void* stream);

/* Input rasterisation, one call per clip.
 * tsnet_fit_face_curves <- utils/keypoint2img.py interp_points (:319-354) up to the fitted curve, HOST code: for each of the 34 pieces of
 *                      FaceDatasetTest.part_list (dataset/dataset_video_face.py:271-280, 473-477) the axis choice and scipy.optimize.curve_fit's
 *                      Levenberg-Marquardt fit (MINPACK lmdif from p0 = ones), reproduced bit for bit (csrc/lmfit.hpp).  keypoints: (F,68,2)
 *                      HOST doubles, (x, y) already relative to the crop (read_keypoints, :497-505); curves: (F,34,8) HOST doubles
 *                      {kind, a, b, c, u_first, u_last, 0, 0}, kind 0 = nothing drawn (|a| > 1), bit 1 = curve is x(y), bit 2 = quadratic.
 * tsnet_raster_face <- FaceDatasetTest.get_face_image + get_bbox_image (:466-495; interp_points' sampling + draw_edge, keypoint2img.py:298-316,
 *                      335-354).  keypoints (F,68,2) and curves (F,34,8): DEVICE doubles; edges / bbox: (F,h,w) device bytes, 0 / 255,
 *                      either may be NULL (edges needs curves, bbox needs keypoints).
 * tsnet_vl2ch       <- utils/misc.py vl2ch (:50-67): labels (B,HW) class indices as floats -> out (B,num_classes,HW) one-hot floats. */
int tsnet_fit_face_curves(const double* keypoints, int F, double* curves);
int tsnet_raster_face(const double* keypoints, const double* curves, int F, int h, int w, int bw, unsigned char* edges, unsigned char* bbox, void* stream);
int tsnet_vl2ch(const float* labels, int B, int HW, int num_classes, float* out, void* stream);

/* Pose clips (dataset/dataset_video_pose.py PoseDatasetTestVideo.get_image / get_smooth_lbl :489-536, test mode).
 * tsnet_fit_pose_curves <- interp_points (utils/keypoint2img_posenorm.py:490-516) of every stroke of connect_keypoints (:265-311), HOST code as
 *                      tsnet_fit_face_curves: pts (F,137,2) HOST doubles, flags as below; curves (F,118,8) HOST doubles (24 limb strokes, 40 finger
 *                      segments, 54 face pieces; kind 0 = dropped by the flags or a missing point).
 * tsnet_raster_pose <- utils/keypoint2img_posenorm.py connect_keypoints (:265-311; draw_edge :469-487, interp_points' sampling) followed by
 *                      crop_person_region (:538-552) and utils/misc.py im2vl (:27-47).  pts: (F,137,2) device doubles in frame coordinates =
 *                      the four arrays connect_keypoints takes, concatenated (pose 25 | face 70 | left hand 21 | right hand 21), invalid points
 *                      zero (extract_valid_keypoints, :242-262); curves (F,118,8) device doubles from tsnet_fit_pose_curves.  The skeleton is drawn on the h x w frame; labels (F, win_y1-win_y0,
 *                      win_x1-win_x0) receives the CLASS INDEX (0..24) of every pixel of the window [win_x0,win_x1) x [win_y0,win_y1) -- what
 *                      im2vl makes of the cropped colour image.  labels must be 4-byte aligned and its allocation a multiple of 4 bytes.
 *                      flags: 1 = basic_point_only, 2 = remove_face_labels.  Stroke widths are the test-mode ones (isTrain = False).
 * tsnet_label_bbox  <- PoseDatasetTestVideo.get_bbox_image (:590-607): box of the non-zero label pixels grown by h/16, w/16; 0 / 255 bytes.
 * tsnet_resize_pad  <- Image.resize(size, NEAREST) + resize_square (:425-432, :471-477): out[f, pad_top+y, pad_left+x] = in[f, ytab[y], xtab[x]],
 *                      zero elsewhere; out (F,OH,OW) floats (binarise: != 0 -> 1, the `bbox != 0` of :441, :448).  ytab / xtab: device ints, the
 *                      source row / column of every output row / column (wacv23_tsnet_amd/raster.py computes them in PIL's order). */
/* tsnet_resize_label <- np.asarray(img_as_bool(skimage.transform.resize(map, (OH, OW)))) of the face loader (dataset/dataset_video_face.py:104-106,
 *                      316-317, 397-398; scikit-image 0.18.3): in (F,h,w) device bytes (0 / 255), out (F,OH,OW) device floats 0 / 1.  Restated from
 *                      the published algorithm of that version (oracle/skimage_resize.py) -- PARITY UNPINNED: scikit-image is not available to
 *                      check it against.  wts_rows / wts_cols: HOST doubles, the anti-aliasing Gaussian of an axis that shrinks, centre first:
 *                      w[j] = exp(-0.5 j^2 / sigma^2) / sum, j = 0..lw, sigma = (in / out - 1) / 2, lw = int(4 sigma + 0.5)
 *                      (scipy.ndimage.gaussian_filter1d); lw = -1 (or sigma = 0): no pass along that axis.  Synchronises the stream. */
int tsnet_resize_label(const unsigned char* in, int F, int h, int w, int OH, int OW, const double* wts_rows, int lw_rows,
                       const double* wts_cols, int lw_cols, float* out, void* stream);
int tsnet_fit_pose_curves(const double* pts, int F, int flags, double* curves);
int tsnet_raster_pose(const double* pts, const double* curves, int F, int h, int w, int win_x0, int win_y0, int win_x1, int win_y1, int flags,
                      unsigned char* labels, void* stream);
int tsnet_label_bbox(const unsigned char* labels, int F, int h, int w, unsigned char* bbox, void* stream);
int tsnet_resize_pad(const unsigned char* in, int F, int h, int w, const int* ytab, const int* xtab, int oh, int ow,
                     int pad_top, int pad_left, int OH, int OW, int binarise, float* out, void* stream);

/* Micro-benchmark of one convolution shape on synthetic (non-zero) data: average milliseconds per launch over `iters` back-to-back
 * launches, hipEvent-timed on `stream`.  variant: -1 = the layer's own kernel and tile; else bits 0-11 tile code (tsnet_op_conv2d),
 * bit 12 general kernel, bit 13 bf16 operands, bits 16-23 / 24-31 ablation / experiment masks (tools build).  Diagnostic only. */
int tsnet_bench_conv(int N, int H, int W, int Cin, int Cout, int ksize, int stride, int pad, int pad_mode, int norm,
                     int variant, int iters, float* ms_out, void* stream);

/* Launch counters since the last reset: out[0] = patch convolution kernels (conv_h2.hpp), out[1] = general convolution kernel
 * (conv_h2r.hpp), out[2] = RGB head; out[3] = tile code (rows * 1000 + width) of the last ResnetBlock-class patch convolution launched
 * (bench.py labels the roofline kernel with it).  Diagnostic. */
void tsnet_debug_counters(int64_t out[4], int reset);

/* Host-side constant tables, exported so CPU tests can pin them against torch:
 * tsnet_linspace <- torch.linspace(-1,1,n) as used by get_grid (TSNet.py:301-302);
 * tsnet_coord_table <- Encoder.coord_conv channels (xx,yy,rr) at (H,W), layout (H,W,3) (TSNet.py:107-122). */
void tsnet_linspace(int n, float* out);
void tsnet_coord_table(int H, int W, float* out);

#ifdef __cplusplus
}
#endif
#endif /* TSNET_ABI_H */
