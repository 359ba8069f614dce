// split3.hpp -- 3-way bf16 split of fp32 values (operand format of conv_x3.hpp).
// hi = rne_bf16(x), mid = rne_bf16(x - hi), lo = rne_bf16(x - hi - mid); both subtractions are exact, so
// x = hi + mid + lo up to 2^-27 |x|.  Round-to-nearest (not truncation) matters: truncated residuals all carry
// the sign of x, which biases the dropped cross terms mid*lo of a long dot product in one direction; rounded
// residuals have random signs and are one bit smaller.  Planes are separate bf16 tensors (plane stride = elements).
#pragma once
#include <hip/hip_runtime.h>

namespace tsnet {

__device__ __forceinline__ unsigned short bf16_rne(float f) {
    const unsigned u = __builtin_bit_cast(unsigned, f);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);     // finite inputs only (activations / weights)
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }

__device__ __forceinline__ void split3_scalar(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    hi = bf16_rne(x);
    const float r1 = x - bf16_to_f32(hi);
    mid = bf16_rne(r1);
    const float r2 = r1 - bf16_to_f32(mid);
    lo = bf16_rne(r2);
}

__device__ __forceinline__ void split3_store(float4 v, unsigned short* hi, unsigned short* mid, unsigned short* lo) {
    const float f[4] = {v.x, v.y, v.z, v.w};
    unsigned short h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3_scalar(f[e], h[e], m[e], l[e]);
    *reinterpret_cast<uint2*>(hi) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
    *reinterpret_cast<uint2*>(mid) = make_uint2(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16));
    *reinterpret_cast<uint2*>(lo) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
}

// max |v| of the workgroup's values -> ONE atomic per workgroup (atomics on one address serialise at ~12 ns each; non-negative
// floats order like their bit patterns).  Every thread of the workgroup must call it (two barriers).
__device__ __forceinline__ void tsnet_publish_amax(unsigned* slot, float m) {
    __shared__ float wave_max[16];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(m, off); m = o > m ? o : m; }
    __syncthreads();                                     // a previous use of wave_max in this workgroup is over
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (int)((blockDim.x + 63) >> 6);
        for (int i = 1; i < nw; ++i) m = wave_max[i] > m ? wave_max[i] : m;
        __hip_atomic_fetch_max(slot, __builtin_bit_cast(unsigned, m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// planes of a tensor with `elems` elements: store 4 consecutive elements starting at element index i
__device__ __forceinline__ void split3_store_at(float4 v, unsigned short* planes, size_t elems, size_t i) {
    split3_store(v, planes + i, planes + elems + i, planes + 2 * elems + i);
}

}  // namespace tsnet
