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

// bilinear xR source coordinates of destination index d along an axis of n source elements (ac = 0: align_corners=False, 1: True;
// saltnet.h salt_bilinear_args): elementwise.hip's up-sampling kernels and hyper.hip's stencil must agree
__device__ __forceinline__ void bil_src(int d, int R, int n, int ac, int& i0, int& i1, float& lam) {
    float s;
    if (ac) s = n > 1 ? (float)d * ((float)(n - 1) / (float)(R * n - 1)) : 0.f;      // torch area_pixel_compute_scale, align_corners=True
    else { s = ((float)d + 0.5f) * (1.0f / (float)R) - 0.5f; s = s < 0.f ? 0.f : s; }
    i0 = (int)s;
    i0 = i0 < n - 1 ? i0 : n - 1;
    i1 = i0 + 1 < n ? i0 + 1 : n - 1;
    lam = s - (float)i0;
}

// The logit head's per-lane dot product (salt_head1x1's vector kernels and salt_hyper_stencil's fused head): the rounding sequence is
// PINNED - one multiply, then fused multiply-adds in channel order - so that every kernel that applies the head produces the same bits
// (left to the compiler, one kernel got v_pk_mul + v_pk_add and the other v_fmac for the same source line).
template <int N>
__device__ __forceinline__ float head_dot(const float* f, const float* w) {
    float a = __fmul_rn(f[0], w[0]);
#pragma unroll
    for (int j = 1; j < N; ++j) a = __fmaf_rn(f[j], w[j], a);
    return a;
}

// Adam + L2 on four consecutive parameters (models.py:74-75,289-297: optim.Adam with weight decay in the gradient).  ONE definition for
// adam_kernel (loss.hip) and adam_pack_kernel (conv_mfma.hip): the two must produce the same bits.
__device__ __forceinline__ void adam4(f32x4& pp, const f32x4& gg, f32x4& mm, f32x4& vv, float b1, float b2, float eps, float wd, float gs, float step_size, float rs) {
#define SALT_ADAM1V(X) { const float gr = __fmaf_rn(wd, pp.X, __fmul_rn(gg.X, gs)); mm.X = __fmaf_rn(1.f - b1, gr, __fmul_rn(b1, mm.X)); \
                         vv.X = __fmaf_rn(__fmul_rn(1.f - b2, gr), gr, __fmul_rn(b2, vv.X)); \
                         pp.X = __fsub_rn(pp.X, __fdiv_rn(__fmul_rn(step_size, mm.X), __fmaf_rn(sqrtf(vv.X), rs, eps))); }
    SALT_ADAM1V(x) SALT_ADAM1V(y) SALT_ADAM1V(z) SALT_ADAM1V(w)
#undef SALT_ADAM1V
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

// conv_ws.hip: the weight-stationary multi-tile kernel of the <= 64-channel 3x3 layers; salt_conv tries it first
bool conv_ws_eligible(const salt_conv_args* a);
int conv_ws_tiles(const salt_conv_args* a);
int conv_ws_launch(const salt_conv_args* a, hipStream_t st);
// conv_ws.hip: the loader-specialised streaming kernel of the deeper 3x3 layers (conv_ls_variant: 0 = not applicable, else channel blocks of 32 NI)
int conv_ls_variant(const salt_conv_args* a);
int conv_ls_launch(const salt_conv_args* a, hipStream_t st);
// conv_ws.hip: the streaming kernel of the 1x1 unit-step convolutions (eval-mode Bottleneck convs; 0 = not applicable, else channel blocks of 32 NI)
int conv1x1_ls_variant(const salt_conv_args* a);
int conv1x1_ls_launch(const salt_conv_args* a, hipStream_t st);

// conv_ws.hip: the ResNet stem after space-to-depth (bf16, 16 taps over 16 channels -> 64); 0 = not applicable
int conv_stem16_variant(const salt_conv_args* a);
int conv_stem16_launch(const salt_conv_args* a, hipStream_t st);
// conv_thin.hip: the persistent weight-stationary kernel of the fp32 3x3 layers with 16 / 32 channels on both sides (0 = not applicable)
int conv_thin_variant(const salt_conv_args* a);
int conv_thin_launch(const salt_conv_args* a, hipStream_t st);
// conv_thin.hip: weight gradient of the same layers; 0 = not one of its shapes, else the number of slabs
int conv_wgrad_thin(const salt_conv_wgrad_args* a, bool launch, hipStream_t st, int* rc);

// conv_wgrad_ls.hip: loader-specialised row-streaming weight gradient (bf16, 3x3, unit step); 0 = not one of its shapes, else nsplit
int conv_wgrad_ls(const salt_conv_wgrad_args* a, bool launch, hipStream_t st, int* rc);

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
// ---- in-launch BatchNorm finalize (saltnet.h: salt_conv_args.fin / bnb_fin, salt_bn_bwd_args.fin_acc) ------------------------------
struct BnFin {
    double* acc; unsigned* ticket;
    const float* gamma; const float* beta; float* running_mean; float* running_var; int64_t* nbt;
    float momentum, eps; float* mean; float* invstd; float* scale; float* shift;
};
struct BnbFin { double* acc; unsigned* ticket; float* dgamma; float* dbeta; float* coef; int accumulate; double M;
                const float* da_bias; unsigned hw; int hw_shift; };   // da_bias [B][C]: per-image, per-channel constant added to da on the fly (salt_bn_bwd_args.da_bias); image = pixel >> hw_shift (or / hw when hw_shift < 0)
// ---- in-launch BatchNorm finalize: sharded fp64 accumulators + arrival ticket ----------------------------------------------------
// Every workgroup adds its tile's sums to the shard of its XCD (workgroup id % 8: 64 arrivals per address instead of 512) with
// device-scope fp64 atomics, waits until they have been performed (vmcnt), and takes a ticket; the workgroup that draws the last ticket
// collects the shards with atomic exchanges (which also leave them zero for the next launch) and finalizes.  Atomics on both sides:
// no release / acquire fence, no L2 write-back of the output tile on the way (MI355X_MICROARCH.md, hand-off forms).
__device__ __forceinline__ void fin_add(double* p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ double fin_take(double* p) {
    return __longlong_as_double((long long)__hip_atomic_exchange(reinterpret_cast<unsigned long long*>(p), 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// true in every thread of the workgroup that arrived last; lds_flag: a free word of the kernel's one LDS array.  One ticket word for
// the whole grid: a two-level ticket (one word per shard + one over the shards) measured SLOWER (8.25 vs 8.18 ms per step) - the
// second dependent round trip costs more than 512 arrivals on one word.
__device__ __forceinline__ bool fin_arrive(unsigned* ticket, unsigned total, unsigned* lds_flag) {
#ifdef SALT_FIN_NOARRIVE                                            // timing experiment only: what the hand-off costs (results are wrong)
    return false;
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's atomics have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = t == total - 1u;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *lds_flag = last ? 1u : 0u;
    }
    __syncthreads();
    return *lds_flag != 0u;
}
// forward statistics: acc = [8][2 C + 1] (sum, sum of squares, count); same arithmetic as bn_finalize_kernel from there on
__device__ inline void fin_forward(const BnFin& f, int C, double* lds_n) {
    const int tid = threadIdx.x, nthr = blockDim.x, stride = 2 * C + 1;
    (void)lds_n;
    // every thread reads the eight shard counts itself (plain device-scope loads, in flight together with its channel exchanges:
    // one round trip); thread 0 zeroes them once everybody has read
    double nv[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
        nv[s] = __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(f.acc + s * stride + 2 * C), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    for (int c = tid; c < C; c += nthr) {
        double sv[8], qv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) { sv[s] = fin_take(f.acc + s * stride + c); qv[s] = fin_take(f.acc + s * stride + C + c); }
        double S = 0.0, Q = 0.0, N = 0.0;
#pragma unroll
        for (int s = 0; s < 8; ++s) { S += sv[s]; Q += qv[s]; N += nv[s]; }
        const double mean = N > 0 ? S / N : 0.0;
        double M2 = Q - S * mean;
        if (M2 < 0.0) M2 = 0.0;
        const double var = N > 0 ? M2 / N : 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
        const float sc = f.gamma[c] * invstd;
        f.mean[c] = (float)mean; f.invstd[c] = invstd; f.scale[c] = sc; f.shift[c] = f.beta[c] - (float)mean * sc;
        if (f.running_mean) {
            const double unb = N > 1 ? M2 / (N - 1) : var;
            f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)mean;
            f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unb;
        }
    }
    __syncthreads();                                               // every thread holds the counts
    if (tid < 8) __hip_atomic_store(reinterpret_cast<unsigned long long*>(f.acc + tid * stride + 2 * C), 0ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 0 && f.nbt) *f.nbt += 1;
}
// backward sums: acc = [8][2 C]; same arithmetic as bn_bwd_finalize_kernel
__device__ inline void fin_backward(const BnbFin& f, const float* gamma, const float* invstd, int C) {
    const int tid = threadIdx.x, nthr = blockDim.x, stride = 2 * C;
    for (int c = tid; c < C; c += nthr) {
        double v1[8], v2[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) { v1[s] = fin_take(f.acc + s * stride + c); v2[s] = fin_take(f.acc + s * stride + C + c); }
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int s = 0; s < 8; ++s) { s1 += v1[s]; s2 += v2[s]; }
        if (f.dgamma) {
            f.dgamma[c] = f.accumulate ? f.dgamma[c] + (float)s2 : (float)s2;
            f.dbeta[c] = f.accumulate ? f.dbeta[c] + (float)s1 : (float)s1;
        }
        f.coef[c] = gamma[c] * invstd[c];
        f.coef[C + c] = (float)(s1 / f.M);
        f.coef[2 * C + c] = (float)(s2 / f.M);
    }
}


// ---- consumer-side finalize: the producer launch only ADDS to the shards (ticket == nullptr: no wait, no ticket - the arrive +
// last-arriver tail measured 7.5 us per launch, more than half of what the fusion saved); the launch boundary orders the atomics
// before the consumer, whose every workgroup recomputes the per-channel coefficients from the 8 shards into LDS (thread per channel,
// all shard loads in flight together: one round trip); workgroup 0 also stores them for later readers.  The accumulators are
// zeroed by ONE salt_zero launch per program run (the shards of all layers are slices of one arena).
__device__ inline void fin_forward_consumer(const BnFin& f, int C, float* sm_scale, float* sm_shift, bool store) {
    const int stride = 2 * C + 1;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double sv[8], qv[8], nv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) { sv[s] = f.acc[s * stride + c]; qv[s] = f.acc[s * stride + C + c]; nv[s] = f.acc[s * stride + 2 * C]; }
        double S = 0.0, Q = 0.0, N = 0.0;
#pragma unroll
        for (int s = 0; s < 8; ++s) { S += sv[s]; Q += qv[s]; N += nv[s]; }
        const double mean = N > 0 ? S / N : 0.0;
        double M2 = Q - S * mean;
        if (M2 < 0.0) M2 = 0.0;
        const double var = N > 0 ? M2 / N : 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
        const float sc = f.gamma[c] * invstd, sh = f.beta[c] - (float)mean * sc;
        sm_scale[c] = sc; sm_shift[c] = sh;
        if (store) {
            f.mean[c] = (float)mean; f.invstd[c] = invstd; f.scale[c] = sc; f.shift[c] = sh;
            if (f.running_mean) {
                const double unb = N > 1 ? M2 / (N - 1) : var;
                f.running_mean[c] = (1.f - f.momentum) * f.running_mean[c] + f.momentum * (float)mean;
                f.running_var[c] = (1.f - f.momentum) * f.running_var[c] + f.momentum * (float)unb;
            }
        }
    }
    if (store && threadIdx.x == 0 && f.nbt) *f.nbt += 1;
}
// sm_k: [3][C] = k (gamma invstd), c1 (sum1 / M), c2 (sum2 / M)
__device__ inline void fin_backward_consumer(const BnbFin& f, const float* gamma, const float* invstd, int C, float* sm_k, bool store) {
    const int stride = 2 * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double v1[8], v2[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) { v1[s] = f.acc[s * stride + c]; v2[s] = f.acc[s * stride + C + c]; }
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int s = 0; s < 8; ++s) { s1 += v1[s]; s2 += v2[s]; }
        const float k0 = gamma[c] * invstd[c], k1 = (float)(s1 / f.M), k2 = (float)(s2 / f.M);
        sm_k[c] = k0; sm_k[C + c] = k1; sm_k[2 * C + c] = k2;
        if (store) {
            if (f.dgamma) {
                f.dgamma[c] = f.accumulate ? f.dgamma[c] + (float)s2 : (float)s2;
                f.dbeta[c] = f.accumulate ? f.dbeta[c] + (float)s1 : (float)s1;
            }
            if (f.coef) { f.coef[c] = k0; f.coef[C + c] = k1; f.coef[2 * C + c] = k2; }
        }
    }
}

__device__ __forceinline__ unsigned bnb_image_of(const BnbFin& f, int64_t pix) {
    return f.hw_shift >= 0 ? (unsigned)pix >> f.hw_shift : (unsigned)pix / f.hw;
}
