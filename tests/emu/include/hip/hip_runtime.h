// TEST INFRASTRUCTURE ONLY -- a minimal CPU emulation of the HIP programming model.
//
// The product is built by hipcc for gfx950 and never sees this header.  tests/emu compiles the
// *unmodified* engine sources (wacv23_tsnet_amd/csrc/*.cpp, *.hpp) with a host compiler and this
// directory first on the include path, which turns every kernel launch into a sequential loop over
// workgroups whose threads run as cooperative fibers.  Wave-collective builtins (MFMA, shuffles)
// and __syncthreads are rendezvous points between fibers, and the MFMA emulation follows the gfx950
// fragment layout (cdna_hip_programming.md section 3), so an indexing or layout mistake in a kernel
// shows up on the CPU, against the oracle, before any GPU time is spent.
// It is slow (~1 GFLOP/s) and is used on tiny shapes only.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <stdexcept>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; } __attribute__((aligned(16)));
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
struct uint2 { unsigned x, y; } __attribute__((aligned(8)));
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r; r.x = x; r.y = y; return r; }

typedef int hipError_t;
typedef struct emu_stream* hipStream_t;
typedef struct emu_event* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace emu {
struct Idx { unsigned x, y, z; };
extern Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
extern unsigned char* g_dyn_smem;
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void syncthreads();
void wave_sync();
float shfl_xor(float v, int mask);
typedef float f32x16_t __attribute__((ext_vector_type(16)));
f32x16_t mfma_32x32x2(float a, float b, f32x16_t c);
void buf_dma16(const unsigned char* base, unsigned bytes, unsigned voff, unsigned soff, unsigned char* lds);
f32x16_t mfma_bf16_32x32x16(const void* a16, const void* b16, f32x16_t c);
f32x16_t mfma_f16_32x32x16(const void* a16, const void* b16, f32x16_t c);
}  // namespace emu

#define threadIdx (emu::g_threadIdx)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

// dynamic LDS: HIP's own HIP_DYNAMIC_SHARED macro (amd_device_functions.h) is what the kernels use
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<unsigned char*>(emu::g_dyn_smem);
static inline void __syncthreads() { emu::syncthreads(); }
#define TSNET_FAST_EXP(x) expf(x)     // device: __expf = v_exp_f32 of x * log2 e; the emulator evaluates it exactly
static inline float __shfl_xor(float v, int mask) { return emu::shfl_xor(v, mask); }
static inline double __shfl_xor(double v, int mask) {
    union { double d; float f[2]; } u, r; u.d = v;
    r.f[0] = emu::shfl_xor(u.f[0], mask); r.f[1] = emu::shfl_xor(u.f[1], mask);
    return r.d;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu::mfma_32x32x2((a), (b), (c))
// hooks of conv_dma.hpp (buffer_load ... lds): descriptor = (base, bytes), out-of-range lanes read zeros
struct tsnet_rsrc_t { const unsigned char* base; unsigned bytes; };
typedef unsigned char* tsnet_lds_t;
static inline tsnet_rsrc_t tsnet_make_rsrc(const void* p, unsigned bytes) { tsnet_rsrc_t r; r.base = (const unsigned char*)p; r.bytes = bytes; return r; }
#define TSNET_LDS_BASE(p) ((unsigned char*)(p))
#define TSNET_BUF_DMA16(rsrc, voff, soff, lds) emu::buf_dma16((rsrc).base, (rsrc).bytes, (voff), (soff), (lds))
#define TSNET_UNIFORM(x) (x)
// hook of conv_x3p.hpp (x3q): buffer_load_dwordx4 into registers, same descriptor and out-of-range rule as the DMA
typedef tsnet_rsrc_t tsnet_brsrc_t;
#define tsnet_make_brsrc tsnet_make_rsrc
template <class T> static inline T emu_buf_load16(const tsnet_rsrc_t& r, unsigned voff, unsigned soff) {
    T v; static_assert(sizeof(T) == 16, "16-byte load");
    const bool oob = soff > r.bytes || (unsigned long long)voff + 16 > (unsigned long long)(r.bytes - soff);
    if (oob) memset(&v, 0, 16); else memcpy(&v, r.base + (size_t)voff + soff, 16);
    return v;
}
#define TSNET_BUF_LOAD16(rsrc, voff, soff) emu_buf_load16<F4>((rsrc), (voff), (soff))
struct alignas(8) F2 { float v[2]; };
static inline F2 emu_buf_load8(const tsnet_rsrc_t& r, unsigned voff, unsigned soff) {
    F2 v;
    const bool oob = soff > r.bytes || (unsigned long long)voff + 8 > (unsigned long long)(r.bytes - soff);
    if (oob) memset(&v, 0, 8); else memcpy(&v, r.base + (size_t)voff + soff, 8);
    return v;
}
#define TSNET_BUF_LOAD8(rsrc, voff, soff) emu_buf_load8((rsrc), (voff), (soff))
// hook of conv_x3.hpp: v_mfma_f32_32x32x16_bf16 on raw 16-byte operands (8 bf16 per lane)
#define TSNET_MFMA_BF16(a, b, c) emu::mfma_bf16_32x32x16(&(a), &(b), (c))
// hook of conv_h2.hpp: v_mfma_f32_32x32x16_f16 on raw 16-byte operands (8 fp16 per lane)
#define TSNET_MFMA_F16(a, b, c) emu::mfma_f16_32x32x16(&(a), &(b), (c))
// hook of conv_common.hpp: the three-instruction fp16 x 2 split (v_cvt_pk_f16_f32 + v_fma_mix{lo,hi}_f16) in plain C
static inline void emu_split_pair(float a, float b, unsigned& hw, unsigned& lw) {
    const _Float16 h0 = (_Float16)a, h1 = (_Float16)b;
    const _Float16 l0 = (_Float16)(a - (float)h0), l1 = (_Float16)(b - (float)h1);
    unsigned short u0, u1, v0, v1;
    memcpy(&u0, &h0, 2); memcpy(&u1, &h1, 2); memcpy(&v0, &l0, 2); memcpy(&v1, &l1, 2);
    hw = (unsigned)u0 | ((unsigned)u1 << 16); lw = (unsigned)v0 | ((unsigned)v1 << 16);
}
#define TSNET_SPLIT_PAIR(a, b, hw, lw) emu_split_pair((a), (b), (hw), (lw))
#define TSNET_SPLIT_2PAIRS(a0, b0, a1, b1, h0, l0, h1, l1) do { emu_split_pair((a0), (b0), (h0), (l0)); emu_split_pair((a1), (b1), (h1), (l1)); } while (0)
#define TSNET_SPLIT_2PAIRS_SCALED(a0, b0, a1, b1, s, h0, l0, h1, l1) do { emu_split_pair((a0) * (s), (b0) * (s), (h0), (l0)); emu_split_pair((a1) * (s), (b1) * (s), (h1), (l1)); } while (0)
static inline unsigned emu_bf16_rne(float f) { unsigned u; __builtin_memcpy(&u, &f, 4); return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16; }
#define TSNET_CVT_PK_BF16(a, b) ((emu_bf16_rne(a) & 0xFFFFu) | (emu_bf16_rne(b) << 16))
#define TSNET_SETPRIO(n) ((void)0)
#define TSNET_OPAQUE_V(x) ((void)0)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define TSNET_DRAIN_VMEM() ((void)0)
#define TSNET_WAVE_SYNC() emu::wave_sync()        // lanes are independent fibers here: a rendezvous where the hardware's lock-step is relied on
// device-scope atomics of the statistics hand-off (conv_x3.hpp x3_epilogue): workgroups run one after another here
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_load(p, order, scope) (*(p))
template <class T> static inline T emu_fetch_add(T* p, T v) { T o = *p; *p = o + v; return o; }
#define __hip_atomic_fetch_add(p, v, order, scope) emu_fetch_add((p), (v))
template <class T> static inline T emu_fetch_max(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
#define __hip_atomic_fetch_max(p, v, order, scope) emu_fetch_max((p), (v))
template <class T> static inline T emu_fetch_min(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
#define __hip_atomic_fetch_min(p, v, order, scope) emu_fetch_min((p), (v))
static inline unsigned atomicCAS(unsigned* p, unsigned cmp, unsigned v) { const unsigned o = *p; if (o == cmp) *p = v; return o; }   // raster.hpp raise_byte
#define __builtin_amdgcn_s_barrier() emu::syncthreads()
#define __builtin_amdgcn_sched_barrier(x) ((void)0)

static inline const char* hipGetErrorString(hipError_t) { return "emu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(64, (n + 63) & ~(size_t)63); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }     // the MI355X's 256 CUs: the launch logic under test is the product's
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
#define hipStreamNonBlocking 1
#define hipEventDisableTiming 2
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }   // one queue: launches run in issue order
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); })
