// split3.hpp -- exact 3-way bf16 split of fp32 values (operand format of conv_x3.hpp).
// x == hi + mid + lo exactly: hi = top 8 mantissa bits of x (truncation), mid = top 8 bits of x-hi,
// lo = top 8 bits of x-hi-mid.  Planes are stored as separate bf16 tensors (plane stride = element count).
#pragma once
#include <hip/hip_runtime.h>

namespace tsnet {

__device__ __forceinline__ void split3_store(float4 v, unsigned short* hi, unsigned short* mid, unsigned short* lo) {
    const float f[4] = {v.x, v.y, v.z, v.w};
    unsigned short h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned u = __builtin_bit_cast(unsigned, f[e]);
        const unsigned uh = u & 0xFFFF0000u;
        const float r1 = f[e] - __builtin_bit_cast(float, uh);
        const unsigned um = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
        const float r2 = r1 - __builtin_bit_cast(float, um);
        h[e] = (unsigned short)(uh >> 16); m[e] = (unsigned short)(um >> 16);
        l[e] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
    }
    *reinterpret_cast<uint2*>(hi) = make_uint2(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16));
    *reinterpret_cast<uint2*>(mid) = make_uint2(m[0] | ((unsigned)m[1] << 16), m[2] | ((unsigned)m[3] << 16));
    *reinterpret_cast<uint2*>(lo) = make_uint2(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16));
}

// planes of a tensor with `elems` elements: store 4 consecutive elements starting at element index i
__device__ __forceinline__ void split3_store_at(float4 v, unsigned short* planes, size_t elems, size_t i) {
    split3_store(v, planes + i, planes + elems + i, planes + 2 * elems + i);
}

}  // namespace tsnet
