// raster.hpp -- input rasterisation on the device: key points -> edge-map label, bounding-box mask,
// one-hot label.  Replaces the per-frame CPU work of the reference's data loader:
//   * FaceDatasetTest.get_face_image (dataset/dataset_video_face.py:466-481) with utils/keypoint2img.py interp_points (:319-354) and
//     draw_edge (:298-316)                                               -> face_edges_kernel
//   * FaceDatasetTest.get_bbox_image (:483-495)                          -> face_bbox_kernel
//   * utils/misc.py vl2ch (:50-67)                                       -> onehot_kernel
// interp_points fits every 3-point piece of a face-part polyline with scipy.optimize.curve_fit (Levenberg-Marquardt) -- for three
// points and three parameters that is the interpolating parabola up to the optimiser's termination error (~1e-9 relative) -- samples it
// at np.linspace(x0, xn, ceil(xn - x0)) and TRUNCATES to integer pixels.  Here the parabola is the closed form (divided differences,
// fp64), evaluated in the reference's operation order a*x^2 + b*x + c.  The two differ only where a sample lands within the optimiser's
// error of an integer (the end points of a piece, whose true ordinates are the integer key points): tests/test_raster.py reports the
// Hamming distance to the reference's maps on every frame of the demo clips.  Bounding box and one-hot are integer work: bit-exact.
#pragma once
#include <hip/hip_runtime.h>

namespace tsnet {

constexpr int kFaceKeypoints = 68;
constexpr int kFaceSubEdges = 34;
// FaceDatasetTest.part_list (:271-280) cut into pieces of three points sharing their end points (:473-477); -1 = two-point piece
__device__ const signed char kFaceSubEdgeTable[kFaceSubEdges][3] = {
    {0, 1, 2}, {2, 3, 4}, {4, 5, 6}, {6, 7, 8}, {8, 9, 10}, {10, 11, 12}, {12, 13, 14}, {14, 15, 16},          // face contour
    {17, 18, 19}, {19, 20, 21}, {22, 23, 24}, {24, 25, 26},                                                  // eyebrows
    {28, 31, -1}, {31, 32, 33}, {33, 34, 35}, {35, 28, -1},                                                  // nose
    {36, 37, 38}, {38, 39, -1}, {39, 40, 41}, {41, 36, -1}, {42, 43, 44}, {44, 45, -1}, {45, 46, 47}, {47, 42, -1},   // eyes
    {48, 49, 50}, {50, 51, 52}, {52, 53, 54}, {54, 55, 56}, {56, 57, 58}, {58, 59, 48},                      // outer mouth
    {60, 61, 62}, {62, 63, 64}, {64, 65, 66}, {66, 67, 60}};                                                 // inner mouth

// grid = (kFaceSubEdges, F); kp: (F, 68, 2) cropped key points (x, y) fp64; out: (F, h, w) bytes, zeroed by the caller
__global__ __launch_bounds__(64) void face_edges_kernel(const double* __restrict__ kp, unsigned char* __restrict__ out, int h, int w, int bw) {
    const int e = blockIdx.x, f = blockIdx.y;
    const double* k = kp + (size_t)f * kFaceKeypoints * 2;
    const int n = kFaceSubEdgeTable[e][2] < 0 ? 2 : 3;
    double px[3], py[3];
    for (int i = 0; i < n; ++i) { px[i] = k[kFaceSubEdgeTable[e][i] * 2]; py[i] = k[kFaceSubEdgeTable[e][i] * 2 + 1]; }
    double mdx = 0, mdy = 0;
    for (int i = 0; i + 1 < n; ++i) { mdx = fmax(mdx, fabs(px[i] - px[i + 1])); mdy = fmax(mdy, fabs(py[i] - py[i + 1])); }
    const bool swap = mdx < mdy;                                  // fit along the axis with the larger extent (keypoint2img.py:320-321)
    double u[3], v[3];
    for (int i = 0; i < n; ++i) { u[i] = swap ? py[i] : px[i]; v[i] = swap ? px[i] : py[i]; }
    double a = 0, b, c;
    if (n == 3) {                                                 // interpolating parabola, divided differences
        const double d01 = (v[1] - v[0]) / (u[1] - u[0]), d12 = (v[2] - v[1]) / (u[2] - u[1]);
        a = (d12 - d01) / (u[2] - u[0]);
        b = d01 - a * (u[0] + u[1]);
        c = v[0] - (a * u[0] + b) * u[0];
        if (!(fabs(a) <= 1.0)) return;                            // curvature limit of the reference (:333-334); also drops degenerate pieces
    } else {
        b = (v[1] - v[0]) / (u[1] - u[0]);
        c = v[0] - b * u[0];
        if (!(fabs(b) <= 1.7e308)) return;                        // coincident abscissae: no line (NaN or infinite slope)
    }
    double u0 = u[0], u1 = u[n - 1];
    if (u0 > u1) { const double t = u0; u0 = u1; u1 = t; }       // the sample range runs upwards (:335-337)
    const int num = (int)ceil(u1 - u0);
    const double step = num > 1 ? (u1 - u0) / (double)(num - 1) : 0.0;
    for (int i = threadIdx.x; i < num; i += blockDim.x) {
        const double cu = (i == num - 1 && num > 1) ? u1 : u0 + (double)i * step;     // np.linspace: arange * step + start, last = stop
        const double cv = n == 3 ? (a * (cu * cu) + b * cu) + c : b * cu + c;
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

}  // namespace tsnet
