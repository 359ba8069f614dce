// raster.hpp -- input rasterisation on the device: key points -> edge-map label, bounding-box mask,
// one-hot label.  Replaces the per-frame CPU work of the reference's data loader:
//   * FaceDatasetTest.get_face_image (dataset/dataset_video_face.py:466-481) with utils/keypoint2img.py interp_points (:319-354) and
//     draw_edge (:298-316)                                               -> face_edges_kernel
//   * FaceDatasetTest.get_bbox_image (:483-495)                          -> face_bbox_kernel
//   * utils/misc.py vl2ch (:50-67)                                       -> onehot_kernel
// interp_points fits every 2- / 3-point piece of a polyline with scipy.optimize.curve_fit (Levenberg-Marquardt), samples the curve at
// np.linspace(x0, xn, ceil(xn - x0)) and TRUNCATES to integer pixels.  The fit runs on the HOST (lmfit.hpp: MINPACK's lmdif reproduced to
// the last bit -- 34 fits of microseconds per face frame, 118 per pose frame) and hands the device one record per piece
// {kind, a, b, c, u_first, u_last}; the kernels sample a*x^2 + b*x + c in the reference's operation order and draw.  Edge maps, bounding
// boxes and one-hot labels are then EQUAL to the reference's on every frame of its demo clips (tests/test_raster.py, test_raster_pose.py).
//
// Pose clips (OpenPose key points -> colour-coded skeleton -> class-index label; dataset/dataset_video_pose.py:489-615):
//   * utils/keypoint2img_posenorm.py connect_keypoints (:265-311) + draw_edge (:469-487) + interp_points (:490-516, two-point pieces only:
//     a straight line) + utils/misc.py im2vl (:27-47, colour -> class index)                  -> pose_edges_kernel + pose_order_to_class_kernel
//   * PoseDatasetTestVideo.get_bbox_image (:590-607)                                          -> label_bbox_kernel
//   * Image.resize(.., NEAREST) + resize_square (:425-432, :471-477)                          -> gather_pad_kernel (index tables from the host)
//   * skimage.transform.resize + img_as_bool of the face loader (dataset_video_face.py:316-317)    -> gauss1d_u8 / u8_minmax / resize_label kernels
//     (restated from the published algorithm, PARITY UNPINNED: see there)
#pragma once
#include <hip/hip_runtime.h>

namespace tsnet {

constexpr int kFaceKeypoints = 68;
constexpr int kFaceSubEdges = 34;
constexpr int kCurveRec = 8;             // doubles per fitted piece: {kind, a, b, c, u_first, u_last, 0, 0} (lmfit.hpp fit_piece)
// FaceDatasetTest.part_list (:271-280) cut into pieces of three points sharing their end points (:473-477); -1 = two-point piece.
// (The host fits the pieces, the device draws them: one table, two storage classes.)
#define TSNET_FACE_SUB_EDGES                                                                                       \
    {0, 1, 2}, {2, 3, 4}, {4, 5, 6}, {6, 7, 8}, {8, 9, 10}, {10, 11, 12}, {12, 13, 14}, {14, 15, 16},                /* face contour */ \
    {17, 18, 19}, {19, 20, 21}, {22, 23, 24}, {24, 25, 26},                                                        /* eyebrows */ \
    {28, 31, -1}, {31, 32, 33}, {33, 34, 35}, {35, 28, -1},                                                        /* nose */ \
    {36, 37, 38}, {38, 39, -1}, {39, 40, 41}, {41, 36, -1}, {42, 43, 44}, {44, 45, -1}, {45, 46, 47}, {47, 42, -1}, /* eyes */ \
    {48, 49, 50}, {50, 51, 52}, {52, 53, 54}, {54, 55, 56}, {56, 57, 58}, {58, 59, 48},                            /* outer mouth */ \
    {60, 61, 62}, {62, 63, 64}, {64, 65, 66}, {66, 67, 60}                                                         /* inner mouth */
static const signed char hFaceSubEdgeTable[kFaceSubEdges][3] = {TSNET_FACE_SUB_EDGES};

// grid = (kFaceSubEdges, F); rec: (F, 34, 8) fitted pieces (tsnet_fit_face_curves); out: (F, h, w) bytes, zeroed by the caller
__global__ __launch_bounds__(64) void face_edges_kernel(const double* __restrict__ rec, unsigned char* __restrict__ out, int h, int w, int bw) {
    const int e = blockIdx.x, f = blockIdx.y;
    const double* r = rec + ((size_t)f * kFaceSubEdges + e) * kCurveRec;
    const int kind = (int)r[0];
    if (!(kind & 1)) return;                                      // |a| > 1 (:333-334) or no fit
    const bool swap = (kind & 2) != 0, quad = (kind & 4) != 0;
    const double a = r[1], b = r[2], c = r[3], u0 = r[4], u1 = r[5];
    const int num = (int)ceil(u1 - u0);
    const double step = num > 1 ? (u1 - u0) / (double)(num - 1) : 0.0;
    for (int i = threadIdx.x; i < num; i += blockDim.x) {
        const double cu = (i == num - 1 && num > 1) ? u1 : u0 + (double)i * step;     // np.linspace: arange * step + start, last = stop
        const double cv = quad ? (a * (cu * cu) + b * cu) + c : b * cu + c;
        const int iu = (int)cu, iv = (int)cv;                     // astype(int): truncation
        const int x = swap ? iv : iu, y = swap ? iu : iv;
        for (int di = -bw; di < bw; ++di)
            for (int dj = -bw; dj < bw; ++dj) {
                int yy = y + di, xx = x + dj;
                yy = yy < 0 ? 0 : (yy > h - 1 ? h - 1 : yy);
                xx = xx < 0 ? 0 : (xx > w - 1 ? w - 1 : xx);
                out[((size_t)f * h + yy) * w + xx] = 255;         // every writer stores the same value
            }
    }
}

// grid = F; out: (F, h, w) bytes, fully written
__global__ __launch_bounds__(256) void face_bbox_kernel(const double* __restrict__ kp, unsigned char* __restrict__ out, int h, int w) {
    const int f = blockIdx.x;
    const double* k = kp + (size_t)f * kFaceKeypoints * 2;
    double x0 = k[0], x1 = k[0], y0 = k[1], y1 = k[1];
    for (int i = 1; i < kFaceKeypoints; ++i) {
        x0 = fmin(x0, k[2 * i]); x1 = fmax(x1, k[2 * i]);
        y0 = fmin(y0, k[2 * i + 1]); y1 = fmax(y1, k[2 * i + 1]);
    }
    const int xm = w / 16, ym = h / 16;
    const int bx0 = (int)fmax(0.0, x0 - xm), bx1 = (int)fmin((double)w, x1 + xm);
    const int by0 = (int)fmax(0.0, y0 - ym), by1 = (int)fmin((double)h, y1 + ym);
    for (int i = threadIdx.x; i < h * w; i += blockDim.x) {
        const int y = i / w, x = i - y * w;
        out[(size_t)f * h * w + i] = (y >= by0 && y < by1 && x >= bx0 && x < bx1) ? 255 : 0;
    }
}

// lbl: (B, HW) class indices as float; out: (B, nc, HW) float one-hot
__global__ __launch_bounds__(256) void onehot_kernel(const float* __restrict__ lbl, float* __restrict__ out, int B, int HW, int nc) {
    const size_t total = (size_t)B * nc * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int p = (int)(i % HW);
        const int c = (int)((i / HW) % nc);
        const int b = (int)(i / ((size_t)HW * nc));
        out[i] = lbl[(size_t)b * HW + p] == (float)c ? 1.f : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------------- pose clips
constexpr int kPosePts = 137;            // pose 25 | face 70 | left hand 21 | right hand 21, the four arrays connect_keypoints takes
constexpr int kPoseEdges = 24, kHandSegs = 20, kFacePieces = 54;
constexpr int kPosePrims = kPoseEdges + 2 * kHandSegs + kFacePieces;
// pose_edge_list incl. the feet (define_edge_lists, keypoint2img_posenorm.py:396-421); with basic_point_only only the first 18
#define TSNET_POSE_EDGES \
    {17, 15}, {15, 0}, {0, 16}, {16, 18}, {0, 1}, {1, 8}, {1, 2}, {2, 3}, {3, 4}, {1, 5}, {5, 6}, {6, 7}, \
    {8, 9}, {9, 10}, {10, 11}, {8, 12}, {12, 13}, {13, 14}, {11, 24}, {11, 22}, {22, 23}, {14, 21}, {14, 19}, {19, 20}
static const unsigned char hPoseEdgeTable[kPoseEdges][2] = {TSNET_POSE_EDGES};
__device__ const unsigned char kPoseEdgeTable[kPoseEdges][2] = {TSNET_POSE_EDGES};
// face_list cut into consecutive pairs (edge_len = 2, :296-300)
#define TSNET_FACE_PIECES \
    {0, 1}, {1, 2}, {2, 3}, {3, 4}, {4, 5}, {5, 6}, {6, 7}, {7, 8}, {8, 9}, {9, 10}, {10, 11}, {11, 12}, {12, 13}, {13, 14}, {14, 15}, {15, 16}, \
    {17, 18}, {18, 19}, {19, 20}, {20, 21}, {22, 23}, {23, 24}, {24, 25}, {25, 26}, \
    {28, 31}, {31, 32}, {32, 33}, {33, 34}, {34, 35}, {35, 28}, \
    {36, 37}, {37, 38}, {38, 39}, {39, 40}, {40, 41}, {41, 36}, {42, 43}, {43, 44}, {44, 45}, {45, 46}, {46, 47}, {47, 42}, \
    {48, 49}, {49, 50}, {50, 51}, {51, 52}, {52, 53}, {53, 54}, {54, 55}, {55, 56}, {56, 57}, {57, 58}, {58, 59}, {59, 48}
static const unsigned char hFacePieceTable[kFacePieces][2] = {TSNET_FACE_PIECES};
__device__ const unsigned char kFacePieceTable[kFacePieces][2] = {TSNET_FACE_PIECES};


__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// Painting order decides which colour a pixel keeps (later primitives overwrite earlier ones, set_color :461-466).  The canvas holds
// (order + 1) per pixel, raised with a byte-wise atomic max -- order-independent, so the result does not depend on scheduling -- and
// pose_order_to_class_kernel maps it to the class index im2vl gives the colour of that primitive.
__device__ __forceinline__ void raise_byte(unsigned char* base, size_t idx, unsigned v) {
    unsigned* wp = reinterpret_cast<unsigned*>(base + (idx & ~(size_t)3));
    const unsigned sh = (unsigned)(idx & 3) * 8;
    unsigned old = *wp;
    while (((old >> sh) & 0xffu) < v) {
        const unsigned nw = (old & ~(0xffu << sh)) | (v << sh);
        const unsigned prev = atomicCAS(wp, old, nw);
        if (prev == old) break;
        old = prev;
    }
}

// HOST: the two point indices (into the 137-point array) of primitive e, or false when the flags drop it -- the index logic of
// pose_edges_kernel below, for tsnet_fit_pose_curves
inline bool pose_primitive_points(int e, int flags, int& ia, int& ib) {
    if (e < kPoseEdges) {
        if ((flags & 1) && e >= 18) return false;
        ia = hPoseEdgeTable[e][0]; ib = hPoseEdgeTable[e][1];
    } else if (e < kPoseEdges + 2 * kHandSegs) {
        if (flags & 1) return false;
        const int k = e - kPoseEdges, hand = k / kHandSegs, seg = k % kHandSegs, finger = seg / 4, j = seg % 4;
        const int base = 25 + 70 + hand * 21;
        ia = base + (j == 0 ? 0 : finger * 4 + j); ib = base + finger * 4 + j + 1;
    } else {
        if (flags & 3) return false;
        const int k = e - kPoseEdges - 2 * kHandSegs;
        ia = 25 + hFacePieceTable[k][0]; ib = 25 + hFacePieceTable[k][1];
    }
    return true;
}

// grid = (kPosePrims, F).  pts: (F, 137, 2) fp64 in frame coordinates, invalid points = 0 (extract_valid_keypoints); rec: (F, kPosePrims, 8)
// fitted pieces (tsnet_fit_pose_curves: straight lines through two points, fitted as the reference fits them).  The skeleton is drawn on
// the h x w frame (border clamping of draw_edge against THAT size), but only pixels inside the window [x0, x1) x [y0, y1) are stored:
// out is (F, y1 - y0, x1 - x0) -- crop_person_region (:538-552) of the drawn frame.  flags: 1 = basic_point_only, 2 = remove_face_labels.
__global__ __launch_bounds__(64) void pose_edges_kernel(const double* __restrict__ pts, const double* __restrict__ rec, unsigned char* __restrict__ out,
                                                        int h, int w, int x0, int y0, int x1, int y1, int flags) {
    const int e = blockIdx.x, f = blockIdx.y;
    const double* P = pts + (size_t)f * kPosePts * 2;
    int ia, ib, order, bw;
    bool ends = false;
    {   // stroke widths from the person's height in pixels (:272-273, 283, 295); the min runs over all 25 rows, zeros included
        double ymin = P[1], ymax = P[1];
        for (int i = 1; i < 25; ++i) { ymin = fmin(ymin, P[2 * i + 1]); ymax = fmax(ymax, P[2 * i + 1]); }
        const int ph = (int)(ymax - ymin);
        const int bw_pose = imin(imax(1, ph / 150), 5), bw_small = imin(imax(1, ph / 450), 3);
        if (e < kPoseEdges) {
            if ((flags & 1) && e >= 18) return;
            ia = kPoseEdgeTable[e][0]; ib = kPoseEdgeTable[e][1]; order = e; bw = bw_pose; ends = true;
        } else if (e < kPoseEdges + 2 * kHandSegs) {
            if (flags & 1) return;
            const int k = e - kPoseEdges, hand = k / kHandSegs, seg = k % kHandSegs, finger = seg / 4, j = seg % 4;
            const int base = 25 + 70 + hand * 21;
            ia = base + (j == 0 ? 0 : finger * 4 + j); ib = base + finger * 4 + j + 1;
            order = kPoseEdges + hand * 5 + finger; bw = bw_small;
        } else {
            if (flags & 3) return;
            const int k = e - kPoseEdges - 2 * kHandSegs;
            ia = 25 + kFacePieceTable[k][0]; ib = 25 + kFacePieceTable[k][1]; order = kPoseEdges + 10; bw = bw_small;
        }
    }
    (void)ia; (void)ib;
    const double* r = rec + ((size_t)f * kPosePrims + e) * kCurveRec;
    const int kind = (int)r[0];
    if (!(kind & 1)) return;                                      // a missing point (`0 not in x`), or no fit
    const bool swap = (kind & 2) != 0;                            // fitted along the axis with the larger extent (:491-492)
    const double b = r[2], c = r[3], u0 = r[4], u1 = r[5];        // the line curve_fit(linear, ..) converged to, sample range upwards
    const int num = (int)ceil(u1 - u0);
    if (num < 1) return;
    const double step = num > 1 ? (u1 - u0) / (double)(num - 1) : 0.0;
    const int cw = x1 - x0, ch = y1 - y0;
    const unsigned val = (unsigned)order + 1u;
    auto put = [&](int y, int x) {
        y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);
        if (y >= y0 && y < y1 && x >= x0 && x < x1) raise_byte(out, (size_t)f * ch * cw + (size_t)(y - y0) * cw + (x - x0), val);
    };
    auto sample = [&](int i, int& x, int& y) {
        const double cu = (i == num - 1 && num > 1) ? u1 : u0 + (double)i * step;       // np.linspace
        const double cv = b * cu + c;
        const int iu = (int)cu, iv = (int)cv;                     // astype(int)
        x = swap ? iv : iu; y = swap ? iu : iv;
    };
    for (int i = threadIdx.x; i < num; i += blockDim.x) {
        int x, y;
        sample(i, x, y);
        for (int di = -bw; di < bw; ++di)
            for (int dj = -bw; dj < bw; ++dj) put(y + di, x + dj);
    }
    if (ends) {                                                   // discs of radius 2 bw at the first and the last sample (:480-487)
        int ex[2], ey[2];
        sample(0, ex[0], ey[0]);
        sample(num - 1, ex[1], ey[1]);
        const int side = 4 * bw;
        for (int t = threadIdx.x; t < side * side * 2; t += blockDim.x) {
            const int which = t / (side * side), r = t % (side * side), di = r / side - 2 * bw, dj = r % side - 2 * bw;
            if (di * di + dj * dj < 4 * bw * bw) put(ey[which] + di, ex[which] + dj);
        }
    }
}

// (order + 1) -> class index of that primitive's colour (utils/misc.py global_pose_color_dict :10-25): pose edge i -> i + 1, the foot edges
// reuse the colours of edges 14 and 17, finger i -> 19 + i on both hands, face -> 24
__global__ __launch_bounds__(256) void pose_order_to_class_kernel(unsigned char* __restrict__ io, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int v = io[i];
        if (v == 0) continue;
        const int o = v - 1;
        int cls;
        if (o < 18) cls = o + 1;
        else if (o < 21) cls = 15;
        else if (o < 24) cls = 18;
        else if (o < 34) cls = 19 + (o - 24) % 5;
        else cls = 24;
        io[i] = (unsigned char)cls;
    }
}

// get_bbox_image of the pose dataset (:590-607): the box of the non-zero label pixels, grown by h/16, w/16.  grid = F, 256 threads.
// A frame without any label pixel (the reference raises there) gives an all-zero mask.
__global__ __launch_bounds__(256) void label_bbox_kernel(const unsigned char* __restrict__ lbl, unsigned char* __restrict__ out, int h, int w) {
    __shared__ int sx0[256], sx1[256], sy0[256], sy1[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    const unsigned char* L = lbl + (size_t)f * h * w;
    int x0 = w, x1 = -1, y0 = h, y1 = -1;
    for (int i = tid; i < h * w; i += 256)
        if (L[i]) { const int y = i / w, x = i - y * w; x0 = imin(x0, x); x1 = imax(x1, x); y0 = imin(y0, y); y1 = imax(y1, y); }
    sx0[tid] = x0; sx1[tid] = x1; sy0[tid] = y0; sy1[tid] = y1;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (tid < k) { sx0[tid] = imin(sx0[tid], sx0[tid + k]); sx1[tid] = imax(sx1[tid], sx1[tid + k]);
                       sy0[tid] = imin(sy0[tid], sy0[tid + k]); sy1[tid] = imax(sy1[tid], sy1[tid + k]); }
        __syncthreads();
    }
    x0 = sx0[0]; x1 = sx1[0]; y0 = sy0[0]; y1 = sy1[0];
    const bool any = x1 >= 0;
    const int bx0 = imax(0, x0 - w / 16), bx1 = imin(w, x1 + w / 16), by0 = imax(0, y0 - h / 16), by1 = imin(h, y1 + h / 16);
    for (int i = tid; i < h * w; i += 256) {
        const int y = i / w, x = i - y * w;
        out[(size_t)f * h * w + i] = (any && y >= by0 && y < by1 && x >= bx0 && x < bx1) ? 255 : 0;
    }
}

// out[f, py + y, px + x] = in[f, ytab[y], xtab[x]] for y < oh, x < ow; 0 elsewhere.  Nearest-neighbour resize with the host's index tables
// (PIL's accumulation order lives there) followed by the centred zero padding of resize_square.  out: (F, OH, OW) floats.
__global__ __launch_bounds__(256) void gather_pad_kernel(const unsigned char* __restrict__ in, int F, int h, int w, const int* __restrict__ ytab,
                                                         const int* __restrict__ xtab, int oh, int ow, int py, int px, int OH, int OW,
                                                         float* __restrict__ out, int binarise) {
    const size_t total = (size_t)F * OH * OW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % OW), Y = (int)((i / OW) % OH), f = (int)(i / ((size_t)OW * OH));
        const int y = Y - py, x = X - px;
        float v = 0.f;
        if (y >= 0 && y < oh && x >= 0 && x < ow) {
            const unsigned char s = in[((size_t)f * h + ytab[y]) * w + xtab[x]];
            v = binarise ? (s != 0 ? 1.f : 0.f) : (float)s;
        }
        out[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The face loader's 256 x 256 resize: np.asarray(img_as_bool(skimage.transform.resize(map, (256, 256)))) (dataset_video_face.py:104-106,
// 316-317, 397-398; scikit-image 0.18.3).  PARITY UNPINNED: scikit-image cannot be run in this image; the kernels below follow
// oracle/skimage_resize.py, the restatement of that version's published algorithm, operation for operation (fp64, no contraction).
//   1. anti-aliasing Gaussian on the uint8 image, one 1-D pass per axis with sigma = (in / out - 1) / 2 > 0, mirrored borders, scipy's
//      symmetric correlate order (centre, then pairs outwards), result truncated back to uint8 (the C cast of scipy's line buffer);
//   2. bilinear sampling at f (o + 0.5) - 0.5 between floor and ceil, mirrored borders, on the image / 255; clip to the image's range;
//   3. > 0.5.
__device__ __forceinline__ int mirror_index(int i, int n) {          // coord_map(mode 'R'): reflect about the edge pixel centres
    if (n == 1) return 0;
    const int c = n - 1;
    i = i < 0 ? -i : i;
    const int q = i / c, r = i - q * c;
    return (q & 1) ? c - r : r;
}

// one 1-D Gaussian pass along `axis` (0 = rows, 1 = columns) of F images (h, w); wts: lw + 1 doubles (centre first)
__global__ __launch_bounds__(256) void gauss1d_u8_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out, int F, int h, int w,
                                                         int axis, const double* __restrict__ wts, int lw) {
    const size_t total = (size_t)F * h * w;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)((i / w) % h);
        const unsigned char* img = in + (i / ((size_t)w * h)) * (size_t)w * h;
        const int n = axis ? w : h, pos = axis ? x : y;
        double acc = (double)in[i] * wts[0];
        for (int j = 1; j <= lw; ++j) {
            const int a = mirror_index(pos + j, n), b = mirror_index(pos - j, n);
            const double va = axis ? (double)img[(size_t)y * w + a] : (double)img[(size_t)a * w + x];
            const double vb = axis ? (double)img[(size_t)y * w + b] : (double)img[(size_t)b * w + x];
            acc = acc + (va + vb) * wts[j];
        }
        out[i] = (unsigned char)acc;                                  // truncation: acc is in [0, 255]
    }
}

// per-frame minimum / maximum of a byte image -> mm[f] = {min, max} (ints, initialised to {255, 0} by the caller)
__global__ __launch_bounds__(256) void u8_minmax_kernel(const unsigned char* __restrict__ in, int F, int hw, int* __restrict__ mm) {
    const int f = blockIdx.y;
    int lo = 255, hi = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += gridDim.x * blockDim.x) {
        const int v = in[(size_t)f * hw + i];
        lo = v < lo ? v : lo; hi = v > hi ? v : hi;
    }
    __hip_atomic_fetch_min(mm + 2 * f, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // order-independent: deterministic
    __hip_atomic_fetch_max(mm + 2 * f + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void resize_label_kernel(const unsigned char* __restrict__ in, int F, int h, int w, int OH, int OW,
                                                           const int* __restrict__ mm, float* __restrict__ out) {
    const size_t total = (size_t)F * OH * OW;
    const double fr = (double)h / (double)OH, fc = (double)w / (double)OW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int X = (int)(i % OW), Y = (int)((i / OW) % OH), f = (int)(i / ((size_t)OW * OH));
        const unsigned char* img = in + (size_t)f * h * w;
        const double r = fr * ((double)Y + 0.5) - 0.5, c = fc * ((double)X + 0.5) - 0.5;
        const double r0 = floor(r), c0 = floor(c), r1 = ceil(r), c1 = ceil(c);
        const double dr = r - r0, dc = c - c0;
        const int ir0 = mirror_index((int)r0, h), ir1 = mirror_index((int)r1, h), ic0 = mirror_index((int)c0, w), ic1 = mirror_index((int)c1, w);
        const double tl = (double)img[(size_t)ir0 * w + ic0] / 255.0, tr = (double)img[(size_t)ir0 * w + ic1] / 255.0;
        const double bl = (double)img[(size_t)ir1 * w + ic0] / 255.0, br = (double)img[(size_t)ir1 * w + ic1] / 255.0;
        const double top = (1.0 - dc) * tl + dc * tr, bot = (1.0 - dc) * bl + dc * br;
        double v = (1.0 - dr) * top + dr * bot;
        const double lo = (double)mm[2 * f] / 255.0, hi = (double)mm[2 * f + 1] / 255.0;
        v = v < lo ? lo : (v > hi ? hi : v);
        out[i] = v > 0.5 ? 1.f : 0.f;
    }
}

}  // namespace tsnet
