// Feasibility probe (round-2 planning): does a 3-way bf16 split of fp32 operands on the bf16 MFMA
// (v_mfma_f32_32x32x16_bf16, 6 products per k-group) reproduce fp32-class accuracy?
// Computes one 32x32 output tile over K with (a) the exact-fp32 MFMA, (b) bf16x3, (c) bf16x3 with
// the small cross terms accumulated separately, and compares all with an fp64 host reference.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float x, short& hi, short& mid, short& lo) {
    unsigned u = __float_as_uint(x);
    unsigned uh = u & 0xFFFF0000u; float fh = __uint_as_float(uh);
    float r1 = x - fh; unsigned um = __float_as_uint(r1) & 0xFFFF0000u; float fm = __uint_as_float(um);
    float r2 = r1 - fm; unsigned ul = __float_as_uint(r2) & 0xFFFF0000u;
    hi = (short)(uh >> 16); mid = (short)(um >> 16); lo = (short)(ul >> 16);
}

__global__ void probe(const float* A, const float* Bt, int K, float* out32, float* outx3, float* outx3s) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    f32x16 c32 = {0}, cx = {0}, cbig = {0}, csmall = {0};
    for (int kb = 0; kb < K; kb += 16) {
        // exact fp32 path: 8 MFMAs of K=2 cover 16 k
        for (int kk = 0; kk < 16; kk += 2)
            c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + kb + kk + h], Bt[i * K + kb + kk + h], c32, 0, 0, 0);
        s16x8 ah, am, al, bh, bm, bl;
        for (int j = 0; j < 8; ++j) {
            short x, y, z;
            split3(A[i * K + kb + 8 * h + j], x, y, z); ah[j] = x; am[j] = y; al[j] = z;
            split3(Bt[i * K + kb + 8 * h + j], x, y, z); bh[j] = x; bm[j] = y; bl[j] = z;
        }
#define MF(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)
        // small terms first, then large
        cx = MF(al, bh, cx); cx = MF(ah, bl, cx); cx = MF(am, bm, cx); cx = MF(am, bh, cx); cx = MF(ah, bm, cx); cx = MF(ah, bh, cx);
        csmall = MF(al, bh, csmall); csmall = MF(ah, bl, csmall); csmall = MF(am, bm, csmall); csmall = MF(am, bh, csmall); csmall = MF(ah, bm, csmall);
        cbig = MF(ah, bh, cbig);
    }
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        out32[row * 32 + i] = c32[r]; outx3[row * 32 + i] = cx[r]; outx3s[row * 32 + i] = cbig[r] + csmall[r];
    }
}

int main() {
    for (int K : {512, 4608, 9216}) {
        std::vector<float> A(32 * K), Bt(32 * K);
        unsigned st = 1234567u;
        auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)(st >> 8) / 16777216.0f; };
        for (auto& v : A) v = (rnd() < 0.5f ? 0.f : 1.f) * (rnd() * 2.f);                 // post-ReLU-like activations
        for (auto& v : Bt) v = 0.02f * 1.7f * (rnd() + rnd() + rnd() + rnd() - 2.f);      // ~N(0,0.02) weights
        float *dA, *dB, *d32, *dx, *dxs;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, Bt.size() * 4); hipMalloc(&d32, 4096); hipMalloc(&dx, 4096); hipMalloc(&dxs, 4096);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
        probe<<<1, 64>>>(dA, dB, K, d32, dx, dxs);
        std::vector<float> o32(1024), ox(1024), oxs(1024);
        hipMemcpy(o32.data(), d32, 4096, hipMemcpyDeviceToHost); hipMemcpy(ox.data(), dx, 4096, hipMemcpyDeviceToHost); hipMemcpy(oxs.data(), dxs, 4096, hipMemcpyDeviceToHost);
        double e32 = 0, ex = 0, exs = 0, m32 = 0, mx = 0, mxs = 0, ref_abs = 0;
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            double ref = 0; for (int k = 0; k < K; ++k) ref += (double)A[r * K + k] * (double)Bt[c * K + k];
            double a = fabs(o32[r * 32 + c] - ref), b = fabs(ox[r * 32 + c] - ref), d = fabs(oxs[r * 32 + c] - ref);
            e32 = fmax(e32, a); ex = fmax(ex, b); exs = fmax(exs, d); m32 += a; mx += b; mxs += d; ref_abs = fmax(ref_abs, fabs(ref));
        }
        printf("K=%5d |ref|max=%.3f  fp32-mfma: max %.3e mean %.3e | bf16x3: max %.3e mean %.3e | bf16x3(split acc): max %.3e mean %.3e\n",
               K, ref_abs, e32, m32 / 1024, ex, mx / 1024, exs, mxs / 1024);
    }
    return 0;
}
