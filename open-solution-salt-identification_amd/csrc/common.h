// common.h — shared device/host helpers for libsaltnet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "saltnet.h"

typedef unsigned short bf16_t;   // storage type of bf16 activations

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// fp32 -> bf16, round-to-nearest-even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two values per instruction)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ unsigned f2bf_pk(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VE = 4;                  // elements per 16-byte piece
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int VE = 8;
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// unpack / pack one 16-byte piece to floats
template <typename T> __device__ __forceinline__ void unpack16(const u32x4& v, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4& v, float* f) {
    f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const u32x4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ u32x4 pack16(const float* f);
template <> __device__ __forceinline__ u32x4 pack16<float>(const float* f) {
    u32x4 v; v.x = __float_as_uint(f[0]); v.y = __float_as_uint(f[1]); v.z = __float_as_uint(f[2]); v.w = __float_as_uint(f[3]);
    return v;
}
template <> __device__ __forceinline__ u32x4 pack16<bf16_t>(const float* f) {
    u32x4 v;
    v.x = f2bf_pk(f[0], f[1]); v.y = f2bf_pk(f[2], f[3]); v.z = f2bf_pk(f[4], f[5]); v.w = f2bf_pk(f[6], f[7]);
    return v;
}

// ---- host-side helpers
void salt_set_error(const char* fmt, ...);
// Fork hand-off (runtime.hip): when the two-stream executor is about to fork the side stream right after a main-stream
// entry it parks an event here; an entry whose LAST launch can carry a stop event (hipExtLaunchKernelGGL) takes it, so the
// kernel's own completion signal orders the side stream and no marker packet is queued behind the kernel.
hipEvent_t salt_take_fork_event();
#define SALT_FAIL(code, ...) do { salt_set_error(__VA_ARGS__); return (code); } while (0)
#define SALT_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    salt_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); return (int)e_; } } while (0)

static inline int ilog2_ceil(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline bool view_ok(const salt_view& v) { return v.p && v.B > 0 && v.H > 0 && v.W > 0 && v.C > 0 && v.cs >= v.C; }
static inline int64_t view_pixels(const salt_view& v) { return (int64_t)v.B * v.H * v.W; }

// generic dtype dispatch
#define SALT_DISPATCH_DTYPE(dtype, T, ...) \
    if ((dtype) == SALT_F32) { typedef float T; __VA_ARGS__; } \
    else if ((dtype) == SALT_BF16) { typedef bf16_t T; __VA_ARGS__; } \
    else SALT_FAIL(SALT_E_BADARG, "bad dtype %d", (int)(dtype));

// Deterministic block reduction of `v` over a 256-thread block (result valid in thread 0).
__device__ __forceinline__ float block_sum_256(float v, float* sm4) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm4[w] = v;
    __syncthreads();
    return sm4[0] + sm4[1] + sm4[2] + sm4[3];
}

// Pad ring of a replicate-padded H x W image (extended grid (H+top+bottom) x (W+left+right)): ring pixel (r, c) -> index in the
// strip buffer.  Rows above / below the interior are stored whole, the interior rows contribute their left+right columns.
__host__ __device__ inline int fold_ring_index(int r, int c, int H, int W, int top, int bottom, int left, int right) {
    const int Wp = W + left + right;
    if (r < top) return r * Wp + c;
    if (r >= top + H) return (top + (r - top - H)) * Wp + c;
    return (top + bottom) * Wp + (r - top) * (left + right) + (c < left ? c : c - W);
}
