// elementwise.hip — the HBM-bound operators of the path: BatchNorm finalize/apply/backward, ReLU,
// residual add, 2x2 max/avg pooling, bilinear xR resize, replicate-pad adjoint, layout conversion.
// All are single-pass NHWC streaming kernels: 16-byte accesses per lane (8 bf16 / 4 f32 channels of one
// pixel, so a wave reads whole 128-B+ lines), fp32 math, deterministic two-level reductions (no atomics).
// Reference anchors are listed per entry point in include/saltnet.h.
#include "common.h"
#include <hip/hip_ext.h>
#include <stdlib.h>

namespace {

// A "unit" is one 16-byte piece (VEC) or one element (!VEC) of one pixel of a view.
template <typename T, bool VEC> struct Unit {
    static constexpr int N = VEC ? Elem<T>::VE : 1;
    static __device__ __forceinline__ void ld(const T* p, float* f) {
        if constexpr (VEC) { u32x4 v = *reinterpret_cast<const u32x4*>(p); unpack16<T>(v, f); }
        else f[0] = Elem<T>::ld(p);
    }
    static __device__ __forceinline__ void st(T* p, const float* f) {
        if constexpr (VEC) *reinterpret_cast<u32x4*>(p) = pack16<T>(f);
        else Elem<T>::st(p, f[0]);
    }
};

inline bool vec_ok(const salt_view& v, int ve) {
    return v.p == nullptr || ((v.C % ve) == 0 && (v.cs % ve) == 0 && (reinterpret_cast<uintptr_t>(v.p) & 15) == 0);
}
inline int ew_blocks(int64_t units) {
    static const int64_t cap = getenv("SALT_EW_BLOCKS") ? atoi(getenv("SALT_EW_BLOCKS")) : 512;      // round 6: 512 = two workgroups per CU (same box: 1024 5.27, 768 5.27, 512 5.20, 384 5.21, 256 5.23, 192 5.34 ms; round 3: 768 - since then every workgroup of the consumer-side finalize kernels pays the statistics prologue)
    // SALT_EW_MIN_UNITS (A/B, round 5): at least this many 16-byte units per thread - fewer workgroups on the small tensors, where every
    // workgroup's statistics prologue (8 shards x 2 C + 1 doubles from L2, fp64 division + square root per channel) outweighs its stream
    static const int64_t min_units = getenv("SALT_EW_MIN_UNITS") ? atoi(getenv("SALT_EW_MIN_UNITS")) : 1;
    int64_t b = (units + 256 * min_units - 1) / (256 * min_units); return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

#ifndef SALT_BNB_UNITS
#define SALT_BNB_UNITS 1          // units per thread in flight in bn_bwd_apply_kernel (2 and 4 measured slower)
#endif
#ifndef SALT_AFF_UNITS
#define SALT_AFF_UNITS 2          // ... in affine_act_kernel
#endif
#define EW_LAUNCH(KERN, T, allvec, units, stream, ...) \
    do { if (allvec) hipLaunchKernelGGL((KERN<T, true>), dim3(ew_blocks(units)), dim3(256), 0, stream, __VA_ARGS__); \
         else hipLaunchKernelGGL((KERN<T, false>), dim3(ew_blocks(units)), dim3(256), 0, stream, __VA_ARGS__); } while (0)

#define EW_LAUNCH_U(KERN, T, allvec, UU, units, stream, ...) \
    do { if (allvec) hipLaunchKernelGGL((KERN<T, true, UU>), dim3(ew_blocks(units)), dim3(256), 0, stream, __VA_ARGS__); \
         else hipLaunchKernelGGL((KERN<T, false, UU>), dim3(ew_blocks(units)), dim3(256), 0, stream, __VA_ARGS__); } while (0)

// same, with the launch's completion signal bound to ``ev`` when it is non-null (fork hand-off, see common.h)
#define EW_LAUNCH_EV(KERN, T, allvec, units, stream, ev, ...) \
    do { hipEvent_t ev_ = (ev); \
         if (!ev_) { EW_LAUNCH(KERN, T, allvec, units, stream, __VA_ARGS__); } \
         else if (allvec) hipExtLaunchKernelGGL((KERN<T, true>), dim3(ew_blocks(units)), dim3(256), 0, stream, nullptr, ev_, 0, __VA_ARGS__); \
         else hipExtLaunchKernelGGL((KERN<T, false>), dim3(ew_blocks(units)), dim3(256), 0, stream, nullptr, ev_, 0, __VA_ARGS__); } while (0)

// ---------------------------------------------------------------- affine + act (+ residual)
// FIN: scale / shift are not read from memory - every workgroup finalizes the producer's fp64 statistics shards itself
// (fin_forward_consumer, common.h) into LDS; workgroup 0 stores mean / invstd / scale / shift / running statistics.
template <typename T, bool VEC, int U = 2, bool FIN = false>
__global__ __launch_bounds__(256) void affine_act_kernel(salt_view y, const float* scale, const float* shift, salt_view res, int relu, salt_view a, BnFin fin) {
    constexpr int N = Unit<T, VEC>::N;
    const int cpv = y.C / N;
    const int64_t units = (int64_t)y.B * y.H * y.W * cpv;
    const int64_t stride = gridDim.x * 256LL;
    extern __shared__ float fin_sm[];
    auto prologue = [&]() {                                      // every thread of the workgroup calls it exactly once
        if constexpr (FIN) {
            fin_forward_consumer(fin, y.C, fin_sm, fin_sm + y.C, blockIdx.x == 0);
            __syncthreads();
            scale = fin_sm; shift = fin_sm + y.C;              // generic address space: the loads below become LDS reads
        }
    };
    if (stride % cpv == 0) {
        // the thread's channel piece is the same in every iteration: per-channel parameters live in registers, U units in flight
        const int64_t u0 = blockIdx.x * 256LL + threadIdx.x;
        const int c0 = (int)(u0 % cpv) * N;
        float f[U][N], r[U][N];
        int64_t pix[U];
        auto load_iter = [&](int64_t u) {
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int64_t ui = u + i * stride;
                pix[i] = (ui < units ? ui : u) / cpv;                 // past the end: re-read the first unit, never stored
                Unit<T, VEC>::ld((const T*)y.p + pix[i] * y.cs + c0, f[i]);
                if (res.p) Unit<T, VEC>::ld((const T*)res.p + pix[i] * res.cs + c0, r[i]);
            }
        };
        // the first iteration's loads go out BEFORE the statistics prologue (a chain of dependent shard loads + fp64 math + a barrier
        // that every one of the <= 1024 workgroups pays): its round trip hides under theirs
        if (u0 < units) load_iter(u0);
        prologue();
        float sc[N], sh[N];
#pragma unroll
        for (int j = 0; j < N; ++j) { sc[j] = scale ? scale[c0 + j] : 1.f; sh[j] = scale ? shift[c0 + j] : 0.f; }
        for (int64_t u = u0; u < units; u += U * stride) {
            if (u != u0) load_iter(u);
#pragma unroll
            for (int i = 0; i < U; ++i) {
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    float v = f[i][j] * sc[j] + sh[j];
                    if (res.p) v += r[i][j];
                    if (relu) v = fmaxf(v, 0.f);
                    f[i][j] = v;
                }
                if (i == 0 || u + i * stride < units) Unit<T, VEC>::st((T*)a.p + pix[i] * a.cs + c0, f[i]);
            }
        }
        return;
    }
    prologue();
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += stride) {
        const int64_t pix = u / cpv; const int c0 = (int)(u - pix * cpv) * N;
        float f[N], r[N];
        Unit<T, VEC>::ld((const T*)y.p + pix * y.cs + c0, f);
        if (res.p) Unit<T, VEC>::ld((const T*)res.p + pix * res.cs + c0, r);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float v = f[j];
            if (scale) v = v * scale[c0 + j] + shift[c0 + j];
            if (res.p) v += r[j];
            if (relu) v = fmaxf(v, 0.f);
            f[j] = v;
        }
        Unit<T, VEC>::st((T*)a.p + pix * a.cs + c0, f);
    }
}

// ---------------------------------------------------------------- BN finalize (one launch, fp64, fixed order, no atomics)
// Partials are (sum, M2 about the partial's own mean, count), one per convolution workgroup.  Merging uses the exact identity
//   mean = S/N,  M2 = sum_k [ M2_k + n_k (mean_k - mean)^2 ]
// The kernel is latency-bound (a few hundred KB at most), so the layout maximises loads in flight: 256 threads = ROWS part-rows x
// (256/ROWS) channels; a thread holds <= 16 partials in registers (one batch, no re-read in the second pass) up to ROWS*16
// partials, more loop over batches and re-read in pass two.  ROWS = 64 is used for every layer (bn_rows_for).
// cross-row sum of a double over the ROWS part-rows that share a channel: threads are laid out row * CPB + cl, so the rows of one
// channel sit CPB lanes apart inside a wave (64 / CPB rows per wave) - xor-shuffles over the lane strides CPB .. 32 add them in a
// fixed tree, and the 4 waves' results meet in LDS.  (Was: every thread walking all ROWS rows serially through LDS, twice.)
template <int CPB>
__device__ __forceinline__ double rows_sum(double v, double (*sm)[CPB], int cl) {
#pragma unroll
    for (int o = CPB; o < 64; o <<= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6;
    __syncthreads();                                              // previous use of sm
    if ((threadIdx.x & 63) < CPB) sm[wave][cl] = v;
    __syncthreads();
    return (sm[0][cl] + sm[1][cl]) + (sm[2][cl] + sm[3][cl]);
}

template <int ROWS>
__global__ __launch_bounds__(256) void bn_finalize_kernel(salt_bn_finalize_args a) {
    constexpr int CPB = 256 / ROWS, U = 16;
    static_assert(CPB <= 64 && 64 % CPB == 0, "rows of a channel are CPB lanes apart inside a wave");
    __shared__ double sm[4][CPB];
    const int cl = threadIdx.x % CPB, row = threadIdx.x / CPB;
    const int c = blockIdx.x * CPB + cl;
    const bool c_ok = c < a.C;
    const int nparts = a.nparts, C = a.C;
    const int nb = (nparts + ROWS * U - 1) / (ROWS * U);
    // the per-channel parameters are needed only at the very end: fetch them now so that their latency hides under the merge
    const bool writer = row == 0 && c_ok;
    float gam = 0.f, bet = 0.f, rmean = 0.f, rvar = 0.f;
    if (writer) {
        gam = a.gamma[c]; bet = a.beta[c];
        if (a.running_mean) { rmean = a.running_mean[c]; rvar = a.running_var[c]; }
    }
    float sv[U], mv[U], nv[U];
    auto load_batch = [&](int b) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = (b * U + u) * ROWS + row;
            const bool ok = c_ok && k < nparts;
            sv[u] = ok ? a.stats[((int64_t)k * 2 + 0) * C + c] : 0.f;
            mv[u] = ok ? a.stats[((int64_t)k * 2 + 1) * C + c] : 0.f;
            nv[u] = ok ? a.stats_cnt[k] : 0.f;
        }
    };
    double n = 0.0, sacc = 0.0;
    for (int b = 0; b < nb; ++b) {
        load_batch(b);
#pragma unroll
        for (int u = 0; u < U; ++u) { n += (double)nv[u]; sacc += (double)sv[u]; }
    }
    const double N = rows_sum<CPB>(n, sm, cl);
    const double S = rows_sum<CPB>(sacc, sm, cl);
    const double mean = N > 0 ? S / N : 0.0;
    // almost every partial covers a full tile: one reciprocal serves all of them (an fp64 division per partial was the longest
    // dependent chain of this kernel)
    const double nfull = (double)a.stats_cnt[0], rfull = 1.0 / nfull;
    double m2 = 0.0;
    for (int b = 0; b < nb; ++b) {
        if (nb > 1) load_batch(b);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (nv[u] > 0.f) {
                const double nk = (double)nv[u];
                const double d = (double)sv[u] * (nk == nfull ? rfull : 1.0 / nk) - mean;
                m2 += (double)mv[u] + nk * d * d;
            }
        }
    }
    const double M2 = rows_sum<CPB>(m2, sm, cl);
    if (threadIdx.x == 0 && blockIdx.x == 0 && a.num_batches_tracked) *a.num_batches_tracked += 1;
    if (!writer) return;
    const double var = N > 0 ? M2 / N : 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float sc = gam * invstd;
    a.mean[c] = (float)mean; a.invstd[c] = invstd; a.scale[c] = sc; a.shift[c] = bet - (float)mean * sc;
    if (a.running_mean) {
        const double unb = N > 1 ? M2 / (N - 1) : var;
        a.running_mean[c] = (1.f - a.momentum) * rmean + a.momentum * (float)mean;
        a.running_var[c] = (1.f - a.momentum) * rvar + a.momentum * (float)unb;
    }
}

// 64 part-rows x 4 channels for every layer: these kernels are pure latency (a few hundred KB), and the widest split - most
// workgroups, fewest loads per thread - measured fastest on the whole step (7.50 vs 7.72 ms against the size-dependent choice).
inline int bn_rows_for(int nparts) {
    static const int force = getenv("SALT_BN_ROWS") ? atoi(getenv("SALT_BN_ROWS")) : 64;
    (void)nparts;
    return force;
}

__global__ void bn_fold_kernel(salt_bn_fold_args a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.C) return;
    const float sc = a.gamma[c] / sqrtf(a.running_var[c] + a.eps);
    a.scale[c] = sc; a.shift[c] = a.beta[c] - a.running_mean[c] * sc;
}

// ---------------------------------------------------------------- BN backward
// pass 1: per-block partial sums of dyh = da*mask and dyh*xhat per channel.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(salt_view da, salt_view a, salt_view y, int relu,
                                                            const float* mean, const float* invstd, const float* gamma, const float* beta,
                                                            float* partials, int64_t pix_per_block, BnbFin fin) {
    constexpr int N = Unit<T, VEC>::N;
    extern __shared__ float sm[];
    const int C = y.C, cpv = C / N;
    const int64_t npix = (int64_t)y.B * y.H * y.W;
    const int64_t p0 = blockIdx.x * pix_per_block;
    const int64_t p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
    for (int cv0 = 0; cv0 < cpv; cv0 += 256) {
        const int cvn = cpv - cv0 < 256 ? cpv - cv0 : 256;
        const int R = 256 / cvn;
        const int row = threadIdx.x / cvn, cv = cv0 + threadIdx.x % cvn;
        float s1[N], s2[N], mu[N], is[N], sc[N], sh[N];
#pragma unroll
        for (int j = 0; j < N; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
        const bool mask_from_y = relu && a.p == nullptr;      // a = relu(y*scale + shift): the mask is recomputed, `a` is not read
        if (row < R) {
            const int c0 = cv * N;
#pragma unroll
            for (int j = 0; j < N; ++j) {
                mu[j] = mean[c0 + j]; is[j] = invstd[c0 + j];
                sc[j] = gamma[c0 + j] * is[j]; sh[j] = beta[c0 + j] - mu[j] * sc[j];
            }
            constexpr int U = 4;                               // U pixels in flight per thread: the loop is load-latency bound
            for (int64_t pixb = p0 + row; pixb < p1; pixb += (int64_t)U * R) {
                float g[U][N], yy[U][N], aa[U][N];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t pix = pixb + (int64_t)u * R;
#pragma unroll
                    for (int j = 0; j < N; ++j) aa[u][j] = 0.f;
                    if (pix < p1) {
                        Unit<T, VEC>::ld((const T*)da.p + pix * da.cs + c0, g[u]);
                        if (fin.da_bias) {
                            const float* bb = fin.da_bias + (size_t)bnb_image_of(fin, pix) * C + c0;
#pragma unroll
                            for (int j = 0; j < N; ++j) g[u][j] += bb[j];
                        }
                        Unit<T, VEC>::ld((const T*)y.p + pix * y.cs + c0, yy[u]);
                        if (relu && !mask_from_y) Unit<T, VEC>::ld((const T*)a.p + pix * a.cs + c0, aa[u]);
                    } else {
#pragma unroll
                        for (int j = 0; j < N; ++j) { g[u][j] = 0.f; yy[u][j] = mu[j]; }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        float m = 1.f;
                        if (relu) m = mask_from_y ? ((yy[u][j] * sc[j] + sh[j] > 0.f) ? 1.f : 0.f) : (aa[u][j] > 0.f ? 1.f : 0.f);
                        const float gg = g[u][j] * m;
                        s1[j] += gg; s2[j] += gg * (yy[u][j] - mu[j]) * is[j];
                    }
                }
            }
        }
        __syncthreads();
        if (row < R) {
#pragma unroll
            for (int j = 0; j < N; ++j) { sm[((row * cvn + (cv - cv0)) * N + j) * 2] = s1[j]; sm[((row * cvn + (cv - cv0)) * N + j) * 2 + 1] = s2[j]; }
        }
        __syncthreads();
        // cross-row sums: one thread per (statistic, channel) instead of row 0 walking all of them; rows are added in
        // ascending order exactly as before
        const int E = cvn * N;
        for (int e = threadIdx.x; e < 2 * E; e += 256) {
            const int st = e >= E ? 1 : 0, cl = e - st * E;
            float t = 0.f;
            for (int r = 0; r < R; ++r) t += sm[(r * E + cl) * 2 + st];
            if (fin.acc) fin_add(fin.acc + ((blockIdx.x & 7) * 2 + st) * C + cv0 * N + cl, (double)t);
            else partials[((int64_t)blockIdx.x * 2 + st) * C + cv0 * N + cl] = t;
        }
        __syncthreads();
    }
    if (fin.acc && fin.ticket) {                               // in-launch finalize (salt_bn_bwd_args.fin_acc): the last block writes coef / dgamma / dbeta
        const int cvn0 = cpv < 256 ? cpv : 256;
        unsigned* flag = reinterpret_cast<unsigned*>(sm + (256 / cvn0) * cvn0 * N * 2);
        if (fin_arrive(fin.ticket, gridDim.x, flag)) fin_backward(fin, gamma, invstd, C);
    }
}

// 256 threads = ROWS part-rows x (256/ROWS) channels (see bn_finalize_kernel); rows take parts k = row, row+ROWS, ... in fixed order.
template <int ROWS>
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* partials, int nparts, int C, double M, const float* gamma, const float* invstd,
                                       float* dgamma, float* dbeta, int accumulate, float* coef) {
    constexpr int CPB = 256 / ROWS, U = 16;
    __shared__ double sm[4][CPB];
    const int cl = threadIdx.x % CPB, row = threadIdx.x / CPB;
    const int c = blockIdx.x * CPB + cl;
    const bool writer = row == 0 && c < C;
    float gam = 0.f, inv = 0.f, og = 0.f, ob = 0.f;                // fetched up front: their latency hides under the partial loads
    if (writer) {
        gam = gamma[c]; inv = invstd[c];
        if (dgamma && accumulate) { og = dgamma[c]; ob = dbeta[c]; }
    }
    double s1 = 0.0, s2 = 0.0;
    for (int k0 = row; k0 < nparts; k0 += ROWS * U) {
        float v1[U], v2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * ROWS;
            const bool ok = c < C && k < nparts;
            v1[u] = ok ? partials[((int64_t)k * 2) * C + c] : 0.f;
            v2[u] = ok ? partials[((int64_t)k * 2 + 1) * C + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { s1 += (double)v1[u]; s2 += (double)v2[u]; }
    }
    s1 = rows_sum<CPB>(s1, sm, cl);
    s2 = rows_sum<CPB>(s2, sm, cl);
    if (!writer) return;
    if (dgamma) { dgamma[c] = accumulate ? og + (float)s2 : (float)s2; dbeta[c] = accumulate ? ob + (float)s1 : (float)s1; }
    coef[c] = gam * inv;
    coef[C + c] = (float)(s1 / M);
    coef[2 * C + c] = (float)(s2 / M);
}

// FIN: coef is not read from memory - every workgroup finalizes the fp64 shards of the BatchNorm-backward sums itself
// (fin_backward_consumer) into LDS; workgroup 0 stores dgamma / dbeta / coef.
// secondary sums of the apply pass (salt_bn_bwd_args.sec_*): the BatchNorm-backward sums of the layer that produced the residual
struct BnbSec { const void* y; int cs; const float* mean; const float* invstd; double* acc; };

template <typename T, bool VEC, bool FIN = false, bool SEC = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(salt_view da, salt_view a, salt_view y, int relu, const float* mean, const float* invstd,
                                    const float* gamma, const float* beta, const float* coef, salt_view dy, salt_view dres, int acc_dres, BnbFin fin,
                                    BnbSec sec) {
    constexpr int N = Unit<T, VEC>::N;
    const int C = y.C, cpv = C / N;
    const int64_t units = (int64_t)y.B * y.H * y.W * cpv;
    extern __shared__ float fin_sm[];
    auto prologue = [&]() {                                      // every thread of the workgroup calls it exactly once
        if constexpr (FIN) {
            fin_backward_consumer(fin, gamma, invstd, C, fin_sm, blockIdx.x == 0);
            __syncthreads();
            coef = fin_sm;
        }
    };
    {
        const int64_t stride_ = gridDim.x * 256LL;
        if (stride_ % cpv == 0) {
            // the thread's channel piece is loop invariant: the seven per-channel parameters are read once, not per unit
            const int64_t u0 = blockIdx.x * 256LL + threadIdx.x;
            const int c0 = (int)(u0 % cpv) * N;
            const bool from_y = relu && a.p == nullptr;
            constexpr int UB = SALT_BNB_UNITS;
            int64_t pix[UB];
            float g[UB][N], yy[UB][N], aa[UB][N], old[UB][N];
            auto load_iter = [&](int64_t u) {
#pragma unroll
                for (int i = 0; i < UB; ++i) {
                    const int64_t ui = u + i * stride_;
                    pix[i] = (ui < units ? ui : u) / cpv;                  // past the end: unit 0 again, never stored
                    Unit<T, VEC>::ld((const T*)da.p + pix[i] * da.cs + c0, g[i]);
                    Unit<T, VEC>::ld((const T*)y.p + pix[i] * y.cs + c0, yy[i]);
                    if (relu && !from_y) Unit<T, VEC>::ld((const T*)a.p + pix[i] * a.cs + c0, aa[i]);
                    if (dres.p && acc_dres) Unit<T, VEC>::ld((const T*)dres.p + pix[i] * dres.cs + c0, old[i]);
                }
            };
            // the first iteration's loads go out BEFORE the coefficient prologue (dependent shard loads + fp64 math + a barrier in
            // every workgroup): its round trip hides under theirs
            if (u0 < units) load_iter(u0);
            prologue();
            // Six per-channel vectors stay live in the loop: A, D, E with  dy = A gg + D (y - mean) + E  (= k (gg - c1 - xhat c2) with the
            // products folded), mean, and scale / shift of the forward pass for the ReLU mask (the SAME expression affine_act evaluated, so
            // the mask is the forward decision).  Round 2 kept nine (118 VGPRs, 4 waves per SIMD).
            float mu[N], A[N], D[N], E[N], sc[N], sh[N];
            float t1[N], t2[N], smu[N], sis[N];                    // SEC: this thread's part of (sum dres, sum dres xhat_sec) for its channel piece
#pragma unroll
            for (int j = 0; j < N; ++j) { t1[j] = 0.f; t2[j] = 0.f; smu[j] = SEC ? sec.mean[c0 + j] : 0.f; sis[j] = SEC ? sec.invstd[c0 + j] : 0.f; }
            const float* gptr = from_y ? gamma : mean;             // valid addresses either way: every load below is unconditional,
            const float* bptr = from_y ? beta : mean;              // so all of them are in flight together (one latency, not N)
            {
                float is[N], k0[N], k1[N], k2[N], ga[N], be[N];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    mu[j] = mean[c0 + j]; is[j] = invstd[c0 + j];
                    k0[j] = coef[c0 + j]; k1[j] = coef[C + c0 + j]; k2[j] = coef[2 * C + c0 + j];
                    ga[j] = gptr[c0 + j]; be[j] = bptr[c0 + j];
                }
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    sc[j] = ga[j] * is[j]; sh[j] = be[j] - mu[j] * sc[j];
                    A[j] = k0[j]; D[j] = -(k0[j] * k2[j]) * is[j]; E[j] = -(k0[j] * k1[j]);
                }
            }
            for (int64_t u = u0; u < units; u += UB * stride_) {
                if (u != u0) load_iter(u);
#pragma unroll
                for (int i = 0; i < UB; ++i) {
                    float o[N];
                    if (fin.da_bias) {
                        const float* bb = fin.da_bias + (size_t)bnb_image_of(fin, pix[i]) * C + c0;
#pragma unroll
                        for (int j = 0; j < N; ++j) g[i][j] += bb[j];
                    }
#pragma unroll
                    for (int j = 0; j < N; ++j) {
                        const float pre = from_y ? yy[i][j] * sc[j] + sh[j] : ((relu && !from_y) ? aa[i][j] : 1.f);
                        const float gg = (!relu || pre > 0.f) ? g[i][j] : 0.f;
                        g[i][j] = gg;
                        o[j] = A[j] * gg + (D[j] * (yy[i][j] - mu[j]) + E[j]);
                    }
                    if (i == 0 || u + i * stride_ < units) {
                        Unit<T, VEC>::st((T*)dy.p + pix[i] * dy.cs + c0, o);
                        if (dres.p) {
                            if (acc_dres) {
#pragma unroll
                                for (int j = 0; j < N; ++j) g[i][j] += old[i][j];
                            }
                            Unit<T, VEC>::st((T*)dres.p + pix[i] * dres.cs + c0, g[i]);
                            if constexpr (SEC) {
                                float ys[N];
                                Unit<T, VEC>::ld((const T*)sec.y + pix[i] * sec.cs + c0, ys);
                                if constexpr (sizeof(T) == 2 && VEC) { const u32x4 v = pack16<T>(g[i]); unpack16<T>(v, g[i]); }   // the stored value
#pragma unroll
                                for (int j = 0; j < N; ++j) { t1[j] += g[i][j]; t2[j] += g[i][j] * (ys[j] - smu[j]) * sis[j]; }
                            }
                        }
                    }
                }
            }
            if constexpr (SEC) {
                // threads tid, tid + cpv, .. share a channel piece (host: 256 % cpv == 0): rows through LDS in fixed order, then one set of
                // fp64 shard atomics per workgroup (the layout of the reduction pass: shard = workgroup id & 7)
                float* red = fin_sm + 3 * C;
                const int rows = 256 / cpv, row = threadIdx.x / cpv, pc = threadIdx.x % cpv;
                __syncthreads();
#pragma unroll
                for (int j = 0; j < N; ++j) { red[((row * cpv + pc) * N + j) * 2] = t1[j]; red[((row * cpv + pc) * N + j) * 2 + 1] = t2[j]; }
                __syncthreads();
                for (int e = threadIdx.x; e < 2 * C; e += 256) {
                    const int st = e >= C ? 1 : 0, cl = e - st * C;
                    float t = 0.f;
                    for (int r = 0; r < rows; ++r) t += red[(r * C + cl) * 2 + st];
                    fin_add(sec.acc + ((blockIdx.x & 7) * 2 + st) * C + cl, (double)t);
                }
            }
            return;
        }
    }
    prologue();
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
        const int64_t pix = u / cpv; const int c0 = (int)(u - pix * cpv) * N;
        float g[N], yy[N], o[N], msk[N];
        Unit<T, VEC>::ld((const T*)da.p + pix * da.cs + c0, g);
        if (fin.da_bias) {
            const float* bb = fin.da_bias + (size_t)bnb_image_of(fin, pix) * C + c0;
#pragma unroll
            for (int j = 0; j < N; ++j) g[j] += bb[j];
        }
        Unit<T, VEC>::ld((const T*)y.p + pix * y.cs + c0, yy);
#pragma unroll
        for (int j = 0; j < N; ++j) msk[j] = 1.f;
        if (relu) {
            if (a.p == nullptr) {                 // a = relu(y*scale + shift): recompute the mask, never read `a`
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float sc = gamma[c0 + j] * invstd[c0 + j];
                    msk[j] = (yy[j] * sc + (beta[c0 + j] - mean[c0 + j] * sc) > 0.f) ? 1.f : 0.f;
                }
            } else {
                float aa[N];
                Unit<T, VEC>::ld((const T*)a.p + pix * a.cs + c0, aa);
#pragma unroll
                for (int j = 0; j < N; ++j) msk[j] = aa[j] > 0.f ? 1.f : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float gg = g[j] * msk[j];
            g[j] = gg;
            const float xh = (yy[j] - mean[c0 + j]) * invstd[c0 + j];
            o[j] = coef[c0 + j] * (gg - coef[C + c0 + j] - xh * coef[2 * C + c0 + j]);
        }
        Unit<T, VEC>::st((T*)dy.p + pix * dy.cs + c0, o);
        if (dres.p) {
            if (acc_dres) {
                float old[N];
                Unit<T, VEC>::ld((const T*)dres.p + pix * dres.cs + c0, old);
#pragma unroll
                for (int j = 0; j < N; ++j) g[j] += old[j];
            }
            Unit<T, VEC>::st((T*)dres.p + pix * dres.cs + c0, g);
        }
    }
}

template <typename T, bool VEC>
__global__ void relu_bwd_kernel(salt_view da, salt_view a, salt_view dy, int accumulate) {
    constexpr int N = Unit<T, VEC>::N;
    const int cpv = da.C / N;
    const int64_t units = (int64_t)da.B * da.H * da.W * cpv;
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
        const int64_t pix = u / cpv; const int c0 = (int)(u - pix * cpv) * N;
        float g[N], aa[N], old[N];
        Unit<T, VEC>::ld((const T*)da.p + pix * da.cs + c0, g);
        if (a.p) Unit<T, VEC>::ld((const T*)a.p + pix * a.cs + c0, aa);
        if (accumulate) Unit<T, VEC>::ld((const T*)dy.p + pix * dy.cs + c0, old);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            float v = (a.p && !(aa[j] > 0.f)) ? 0.f : g[j];
            if (accumulate) v += old[j];
            g[j] = v;
        }
        Unit<T, VEC>::st((T*)dy.p + pix * dy.cs + c0, g);
    }
}

// ---------------------------------------------------------------- pooling
template <typename T, bool VEC>
__global__ void maxpool2_kernel(salt_view x, salt_view y) {
    constexpr int N = Unit<T, VEC>::N;
    const int cpv = x.C / N;
    const int64_t units = (int64_t)y.B * y.H * y.W * cpv;
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
        int64_t pix = u / cpv; const int c0 = (int)(u - pix * cpv) * N;
        const int ox = (int)(pix % y.W); int64_t r = pix / y.W; const int oy = (int)(r % y.H); const int b = (int)(r / y.H);
        const T* base = (const T*)x.p + (((int64_t)b * x.H + 2 * oy) * x.W + 2 * ox) * x.cs + c0;
        float m[N], f[N];
        Unit<T, VEC>::ld(base, m);
        Unit<T, VEC>::ld(base + x.cs, f);
#pragma unroll
        for (int j = 0; j < N; ++j) m[j] = fmaxf(m[j], f[j]);
        Unit<T, VEC>::ld(base + (int64_t)x.W * x.cs, f);
#pragma unroll
        for (int j = 0; j < N; ++j) m[j] = fmaxf(m[j], f[j]);
        Unit<T, VEC>::ld(base + (int64_t)x.W * x.cs + x.cs, f);
#pragma unroll
        for (int j = 0; j < N; ++j) m[j] = fmaxf(m[j], f[j]);
        Unit<T, VEC>::st((T*)y.p + pix * y.cs + c0, m);
    }
}

template <typename T, bool VEC>
__global__ void maxpool2_bwd_kernel(salt_view x, salt_view dy, salt_view dx, int accumulate) {
    constexpr int N = Unit<T, VEC>::N;
    const int cpv = x.C / N;
    const int64_t units = (int64_t)x.B * x.H * x.W * cpv;
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
        int64_t pix = u / cpv; const int c0 = (int)(u - pix * cpv) * N;
        const int ix = (int)(pix % x.W); int64_t r = pix / x.W; const int iy = (int)(r % x.H); const int b = (int)(r / x.H);
        const int oy = iy >> 1, ox = ix >> 1;
        float o[N];
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] = 0.f;
        if (oy < dy.H && ox < dy.W) {
            const T* base = (const T*)x.p + (((int64_t)b * x.H + 2 * oy) * x.W + 2 * ox) * x.cs + c0;
            float w[4][N], g[N];
            Unit<T, VEC>::ld(base, w[0]);
            Unit<T, VEC>::ld(base + x.cs, w[1]);
            Unit<T, VEC>::ld(base + (int64_t)x.W * x.cs, w[2]);
            Unit<T, VEC>::ld(base + (int64_t)x.W * x.cs + x.cs, w[3]);
            Unit<T, VEC>::ld((const T*)dy.p + (((int64_t)b * dy.H + oy) * dy.W + ox) * dy.cs + c0, g);
            const int me = ((iy & 1) << 1) | (ix & 1);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                int arg = 0; float m = w[0][j];
#pragma unroll
                for (int k = 1; k < 4; ++k) if (w[k][j] > m) { m = w[k][j]; arg = k; }     // first maximum wins
                o[j] = (arg == me) ? g[j] : 0.f;
            }
        }
        T* dst = (T*)dx.p + pix * dx.cs + c0;
        if (accumulate) { float old[N]; Unit<T, VEC>::ld(dst, old);
#pragma unroll
            for (int j = 0; j < N; ++j) o[j] += old[j]; }
        Unit<T, VEC>::st(dst, o);
    }
}

// nn.MaxPool2d(kernel 3, stride 2, padding 1): the ResNet stem pool the reference applies when pool0 is set (encoders.py:23-27).
// Padding is -inf (never selected).  Backward: torch routes an output's gradient to the FIRST maximum of its window in row-major
// window order; windows overlap, so an input pixel gathers from up to 4 outputs (deterministic, no atomics).
template <typename T, bool VEC>
__global__ void maxpool3s2_kernel(salt_view x, salt_view y) {
    constexpr int N = Unit<T, VEC>::N;
    const int cpv = x.C / N;
    const int64_t units = (int64_t)y.B * y.H * y.W * cpv;
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
        int64_t pix = u / cpv; const int c0 = (int)(u - pix * cpv) * N;
        const int ox = (int)(pix % y.W); int64_t r = pix / y.W; const int oy = (int)(r % y.H); const int b = (int)(r / y.H);
        float m[N], f[N];
#pragma unroll
        for (int j = 0; j < N; ++j) m[j] = -INFINITY;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
            if (iy < 0 || iy >= x.H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                if (ix < 0 || ix >= x.W) continue;
                Unit<T, VEC>::ld((const T*)x.p + (((int64_t)b * x.H + iy) * x.W + ix) * x.cs + c0, f);
#pragma unroll
                for (int j = 0; j < N; ++j) m[j] = fmaxf(m[j], f[j]);
            }
        }
        Unit<T, VEC>::st((T*)y.p + pix * y.cs + c0, m);
    }
}

template <typename T, bool VEC>
__global__ void maxpool3s2_bwd_kernel(salt_view x, salt_view dy, salt_view dx, int accumulate) {
    constexpr int N = Unit<T, VEC>::N;
    const int cpv = x.C / N;
    const int64_t units = (int64_t)x.B * x.H * x.W * cpv;
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
        int64_t pix = u / cpv; const int c0 = (int)(u - pix * cpv) * N;
        const int ix = (int)(pix % x.W); int64_t r = pix / x.W; const int iy = (int)(r % x.H); const int b = (int)(r / x.H);
        float o[N], me[N];
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] = 0.f;
        Unit<T, VEC>::ld((const T*)x.p + pix * x.cs + c0, me);
        // outputs whose window [2o-1, 2o+1] contains this pixel: o in {floor(i/2), floor((i+1)/2)}
        for (int oy = iy >> 1; oy <= (iy + 1) >> 1; ++oy) {
            if (oy >= dy.H) continue;
            for (int ox = ix >> 1; ox <= (ix + 1) >> 1; ++ox) {
                if (ox >= dy.W) continue;
                float g[N], f[N];
                bool first[N];                                   // is this pixel the first maximum of window (oy, ox)?
#pragma unroll
                for (int j = 0; j < N; ++j) first[j] = true;
                for (int ky = 0; ky < 3; ++ky) {
                    const int wy = 2 * oy - 1 + ky;
                    if (wy < 0 || wy >= x.H) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int wx = 2 * ox - 1 + kx;
                        if (wx < 0 || wx >= x.W || (wy == iy && wx == ix)) continue;
                        Unit<T, VEC>::ld((const T*)x.p + (((int64_t)b * x.H + wy) * x.W + wx) * x.cs + c0, f);
                        const bool before = wy < iy || (wy == iy && wx < ix);         // earlier in row-major window order
#pragma unroll
                        for (int j = 0; j < N; ++j) if (before ? f[j] >= me[j] : f[j] > me[j]) first[j] = false;
                    }
                }
                Unit<T, VEC>::ld((const T*)dy.p + (((int64_t)b * dy.H + oy) * dy.W + ox) * dy.cs + c0, g);
#pragma unroll
                for (int j = 0; j < N; ++j) if (first[j]) o[j] += g[j];
            }
        }
        T* dst = (T*)dx.p + pix * dx.cs + c0;
        if (accumulate) { float old[N]; Unit<T, VEC>::ld(dst, old);
#pragma unroll
            for (int j = 0; j < N; ++j) o[j] += old[j]; }
        Unit<T, VEC>::st(dst, o);
    }
}

template <typename T, bool VEC>
__global__ void avgpool2_kernel(salt_view x, salt_view y, int backward, int accumulate) {
    constexpr int N = Unit<T, VEC>::N;
    const int cpv = x.C / N;
    if (!backward) {
        const int64_t units = (int64_t)y.B * y.H * y.W * cpv;
        for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
            int64_t pix = u / cpv; const int c0 = (int)(u - pix * cpv) * N;
            const int ox = (int)(pix % y.W); int64_t r = pix / y.W; const int oy = (int)(r % y.H); const int b = (int)(r / y.H);
            const T* base = (const T*)x.p + (((int64_t)b * x.H + 2 * oy) * x.W + 2 * ox) * x.cs + c0;
            float s[N], f[N];
            Unit<T, VEC>::ld(base, s);
            Unit<T, VEC>::ld(base + x.cs, f);
#pragma unroll
            for (int j = 0; j < N; ++j) s[j] += f[j];
            Unit<T, VEC>::ld(base + (int64_t)x.W * x.cs, f);
#pragma unroll
            for (int j = 0; j < N; ++j) s[j] += f[j];
            Unit<T, VEC>::ld(base + (int64_t)x.W * x.cs + x.cs, f);
#pragma unroll
            for (int j = 0; j < N; ++j) s[j] = (s[j] + f[j]) * 0.25f;
            Unit<T, VEC>::st((T*)y.p + pix * y.cs + c0, s);
        }
    } else {
        const int64_t units = (int64_t)x.B * x.H * x.W * cpv;
        for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
            int64_t pix = u / cpv; const int c0 = (int)(u - pix * cpv) * N;
            const int ix = (int)(pix % x.W); int64_t r = pix / x.W; const int iy = (int)(r % x.H); const int b = (int)(r / x.H);
            float o[N];
#pragma unroll
            for (int j = 0; j < N; ++j) o[j] = 0.f;
            if ((iy >> 1) < y.H && (ix >> 1) < y.W) {
                Unit<T, VEC>::ld((const T*)y.p + (((int64_t)b * y.H + (iy >> 1)) * y.W + (ix >> 1)) * y.cs + c0, o);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] *= 0.25f;
            }
            T* dst = (T*)x.p + pix * x.cs + c0;
            if (accumulate) { float old[N]; Unit<T, VEC>::ld(dst, old);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] += old[j]; }
            Unit<T, VEC>::st(dst, o);
        }
    }
}

// ---------------------------------------------------------------- bilinear xR (AC = 0: align_corners=False, AC = 1: True; saltnet.h)
// (bil_src: common.h - shared with hyper.hip)

// U output units per thread in flight: the 4 U gathers of an iteration are issued before any of them is used.  (One unit per
// iteration was a chain of ~16 dependent memory round trips per thread: 33 us for a 67 MB level, whatever the write pitch - round 3.)
// the interpolation itself, with the contractions pinned (bilinear_fwd_kernel and hyper_rows_kernel must agree bit for bit)
__device__ __forceinline__ float bil_mix(float ly, float lx, float a, float b, float c, float d) {
    const float top = __fmaf_rn(lx, b, __fmul_rn(1.f - lx, a));
    const float bot = __fmaf_rn(lx, d, __fmul_rn(1.f - lx, c));
    return __fmaf_rn(ly, bot, __fmul_rn(1.f - ly, top));
}

template <typename T, bool VEC, int U = 4>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(salt_view x, salt_view y, int R, int ac) {
    constexpr int N = Unit<T, VEC>::N;
    const unsigned cpv = x.C / N;
    const int64_t units = (int64_t)y.B * y.H * y.W * cpv;
    const int64_t stride = gridDim.x * 256LL;
    for (int64_t u0 = blockIdx.x * 256LL + threadIdx.x; u0 < units; u0 += U * stride) {
        float a[U][N], bq[U][N], c[U][N], d[U][N], ly[U], lx[U];
        int64_t dsto[U];
#pragma unroll
        for (int i = 0; i < U; ++i) {
            const int64_t ui = u0 + i * stride;
            const int64_t uu = ui < units ? ui : u0;                   // past the end: recompute unit 0 of the thread, never stored
            int64_t pix; unsigned c0;
            int ox, oy, b;
            if (units < (1LL << 31)) {                                   // (wave-uniform) 32-bit index arithmetic
                const unsigned q = (unsigned)uu / cpv; c0 = ((unsigned)uu - q * cpv) * N;
                const unsigned r = q / (unsigned)y.W; ox = (int)(q - r * (unsigned)y.W);
                b = (int)(r / (unsigned)y.H); oy = (int)(r - (unsigned)b * (unsigned)y.H);
                pix = q;
            } else {
                pix = uu / cpv; c0 = (unsigned)(uu - pix * cpv) * N;
                ox = (int)(pix % y.W); const int64_t r = pix / y.W; oy = (int)(r % y.H); b = (int)(r / y.H);
            }
            int y0, y1, x0, x1;
            bil_src(oy, R, x.H, ac, y0, y1, ly[i]);
            bil_src(ox, R, x.W, ac, x0, x1, lx[i]);
            const T* base = (const T*)x.p + (int64_t)b * x.H * x.W * x.cs + c0;
            Unit<T, VEC>::ld(base + ((int64_t)y0 * x.W + x0) * x.cs, a[i]);
            Unit<T, VEC>::ld(base + ((int64_t)y0 * x.W + x1) * x.cs, bq[i]);
            Unit<T, VEC>::ld(base + ((int64_t)y1 * x.W + x0) * x.cs, c[i]);
            Unit<T, VEC>::ld(base + ((int64_t)y1 * x.W + x1) * x.cs, d[i]);
            dsto[i] = pix * y.cs + c0;
        }
#pragma unroll
        for (int i = 0; i < U; ++i) {
            float o[N];
#pragma unroll
            for (int j = 0; j < N; ++j)
                o[j] = bil_mix(ly[i], lx[i], a[i][j], bq[i][j], c[i][j], d[i][j]);
            if (i == 0 || u0 + i * stride < units) Unit<T, VEC>::st((T*)y.p + dsto[i], o);
        }
    }
}

// all levels of the hypercolumn in one pass: unit = (pixel, level, channel piece), so consecutive lanes write consecutive bytes of a pixel row
struct HyperKP { salt_view x[4]; int R[4]; salt_view y; int nlev, ac; };
template <typename T, bool VEC>
__global__ void hyper_rows_kernel(HyperKP p) {
    constexpr int N = Unit<T, VEC>::N;
    const int cpl = p.x[0].C / N, upp = cpl * p.nlev;
    const int64_t units = (int64_t)p.y.B * p.y.H * p.y.W * upp;
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
        int64_t pix = u / upp; const int r_ = (int)(u - pix * upp);
        const int lev = r_ / cpl, c0 = (r_ - lev * cpl) * N;
        const salt_view& x = p.x[lev];
        const int R = p.R[lev];
        const int ox = (int)(pix % p.y.W); int64_t r = pix / p.y.W; const int oy = (int)(r % p.y.H); const int b = (int)(r / p.y.H);
        int y0, y1, x0, x1; float ly, lx;
        bil_src(oy, R, x.H, p.ac, y0, y1, ly);
        bil_src(ox, R, x.W, p.ac, x0, x1, lx);
        const T* base = (const T*)x.p + (int64_t)b * x.H * x.W * x.cs + c0;
        float a[N], bq[N], c[N], d[N], o[N];
        Unit<T, VEC>::ld(base + ((int64_t)y0 * x.W + x0) * x.cs, a);
        Unit<T, VEC>::ld(base + ((int64_t)y0 * x.W + x1) * x.cs, bq);
        Unit<T, VEC>::ld(base + ((int64_t)y1 * x.W + x0) * x.cs, c);
        Unit<T, VEC>::ld(base + ((int64_t)y1 * x.W + x1) * x.cs, d);
#pragma unroll
        for (int j = 0; j < N; ++j)
            o[j] = bil_mix(ly, lx, a[j], bq[j], c[j], d[j]);
        Unit<T, VEC>::st((T*)p.y.p + pix * p.y.cs + lev * x.C + c0, o);
    }
}

// adjoint as a gather over the (<= 2R x 2R) outputs that reference each input pixel: deterministic.
// X_ONLY / Y_ONLY variants make it separable (two passes through a [B,OH,W,C] fp32-free temp of dtype T) for large R.
template <typename T, bool VEC, int MODE>      // MODE 0: full 2-D, 1: x only (y.H == x.H), 2: y only (y.W == x.W)
__global__ void bilinear_bwd_kernel(salt_view x, salt_view y, int R, int accumulate, int ac) {
    constexpr int N = Unit<T, VEC>::N;
    const int cpv = x.C / N;
    const int64_t units = (int64_t)x.B * x.H * x.W * cpv;
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
        int64_t pix = u / cpv; const int c0 = (int)(u - pix * cpv) * N;
        const int ix = (int)(pix % x.W); int64_t r = pix / x.W; const int iy = (int)(r % x.H); const int b = (int)(r / x.H);
        int oy_lo = max(0, R * iy - R / 2), oy_hi = min(y.H - 1, R * iy + (3 * R) / 2 - 1);
        int ox_lo = max(0, R * ix - R / 2), ox_hi = min(y.W - 1, R * ix + (3 * R) / 2 - 1);
        if (ac) {                                     // outputs with src = d (n - 1) / (R n - 1) in (i - 1, i + 1): d in ((i - 1) q, (i + 1) q),
            // q = (R n - 1) / (n - 1) > R; integer floor / ceil of the ends, one more on each side for the fp32 rounding of src
            // (a superset: zero weights are skipped below)
            const int qy = max(x.H - 1, 1), qx = max(x.W - 1, 1);
            oy_lo = max(0, (max(iy - 1, 0) * (y.H - 1)) / qy - 1); oy_hi = min(y.H - 1, ((iy + 1) * (y.H - 1) + qy - 1) / qy + 1);
            ox_lo = max(0, (max(ix - 1, 0) * (y.W - 1)) / qx - 1); ox_hi = min(y.W - 1, ((ix + 1) * (y.W - 1) + qx - 1) / qx + 1);
        }
        if (MODE == 1) { oy_lo = iy; oy_hi = iy; }
        if (MODE == 2) { ox_lo = ix; ox_hi = ix; }
        float o[N];
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] = 0.f;
        const T* base = (const T*)y.p + (int64_t)b * y.H * y.W * y.cs + c0;
        // one-dimensional weight of output o on input i (0 when o does not reference i)
        auto wgt = [&](int o_, int n, int i_) -> float {
            int i0, i1; float l;
            bil_src(o_, R, n, ac, i0, i1, l);
            return (i0 == i_ ? 1.f - l : 0.f) + (i1 == i_ ? l : 0.f);
        };
        // The gathers are issued in batches of G before any is used (same summation order and the same skipped zero-weight terms as the
        // one-at-a-time loop this replaces, which was a chain of up to (2R)^2 dependent memory round trips per thread).
        constexpr int G = 4;
        if (MODE == 0) {
            for (int oy = oy_lo; oy <= oy_hi; ++oy) {
                const float wy = wgt(oy, x.H, iy);
                if (wy == 0.f) continue;
                for (int ox = ox_lo; ox <= ox_hi; ox += G) {
                    float g[G][N], w[G];
#pragma unroll
                    for (int k = 0; k < G; ++k) {
                        const int oxk = min(ox + k, ox_hi);
                        w[k] = ox + k <= ox_hi ? wy * wgt(oxk, x.W, ix) : 0.f;
                        Unit<T, VEC>::ld(base + ((int64_t)oy * y.W + oxk) * y.cs, g[k]);
                    }
#pragma unroll
                    for (int k = 0; k < G; ++k)
                        if (w[k] != 0.f) {
#pragma unroll
                            for (int j = 0; j < N; ++j) o[j] += w[k] * g[k][j];
                        }
                }
            }
        } else {
            // separable passes: one line of outputs along x (MODE 1) or y (MODE 2)
            const int lo = MODE == 1 ? ox_lo : oy_lo, hi = MODE == 1 ? ox_hi : oy_hi;
            const int n = MODE == 1 ? x.W : x.H, ii = MODE == 1 ? ix : iy;
            const int64_t line = MODE == 1 ? (int64_t)iy * y.W * y.cs : (int64_t)ix * y.cs;     // MODE 1: oy == iy;  MODE 2: ox == ix
            const int64_t step = MODE == 1 ? (int64_t)y.cs : (int64_t)y.W * y.cs;
            for (int q = lo; q <= hi; q += G) {
                float g[G][N], w[G];
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const int qk = min(q + k, hi);
                    w[k] = q + k <= hi ? wgt(qk, n, ii) : 0.f;
                    Unit<T, VEC>::ld(base + line + qk * step, g[k]);
                }
#pragma unroll
                for (int k = 0; k < G; ++k)
                    if (w[k] != 0.f) {
#pragma unroll
                        for (int j = 0; j < N; ++j) o[j] += w[k] * g[k][j];
                    }
            }
        }
        T* dst = (T*)x.p + pix * x.cs + c0;
        if (accumulate) { float old[N]; Unit<T, VEC>::ld(dst, old);
#pragma unroll
            for (int j = 0; j < N; ++j) o[j] += old[j]; }
        Unit<T, VEC>::st(dst, o);
    }
}

#ifndef SALT_BIL_G
#define SALT_BIL_G 8
#endif
// ---------------------------------------------------------------- bilinear, row-structured (round 4; vectorised views, align_corners = False)
// The unit-per-thread kernels above decode every unit with three 64-bit divisions and evaluate the interpolation weights per gather; the
// x2 adjoint issues its 4 x 4 window as four dependent batches.  Measured in the C2 step: 27 us to WRITE a 16.8 MB level, 35 us to
// reduce one - 0.6 / 0.5 TB/s.  Here a workgroup owns one row of the result: image and row (and the row weights) are wave-uniform,
// a thread keeps its channel piece and walks the pixels of the row with a constant stride (256 / pieces-per-pixel), and the x2
// adjoint requests its whole window before the first use.  Same operations in the same order as the kernels above: bit-identical.
template <typename T, int U>
__global__ __launch_bounds__(256) void bilinear_fwd_rows_kernel(salt_view x, salt_view y, int R) {
    constexpr int N = Elem<T>::VE;
    const int cpv = x.C / N, ppb = 256 / cpv;                              // host: 256 % cpv == 0
    const int c0 = (int)(threadIdx.x % cpv) * N, px0 = (int)(threadIdx.x / cpv);
    const int rows = y.B * y.H;
    // workgroup -> row: XCD x (= blockIdx % 8) owns a contiguous range of rows, so that neighbouring rows - which read the same
    // source rows - meet in ONE L2 (round-robin placement sent them to 8 different ones: PMC FETCH_SIZE 3.4x the tensor for the x2 adjoint)
    const int per_xcd = (rows + 7) >> 3, r_end = min(rows, ((int)(blockIdx.x & 7) + 1) * per_xcd);
    for (int row = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3); row < r_end; row += (int)(gridDim.x >> 3)) {
        const int b = row / y.H, oy = row - b * y.H;
        int y0, y1; float ly;
        bil_src(oy, R, x.H, 0, y0, y1, ly);
        const T* r0 = (const T*)x.p + ((int64_t)b * x.H + y0) * x.W * x.cs + c0;
        const T* r1 = (const T*)x.p + ((int64_t)b * x.H + y1) * x.W * x.cs + c0;
        T* drow = (T*)y.p + (int64_t)row * y.W * y.cs + c0;
        for (int ox = px0; ox < y.W; ox += U * ppb) {
            u32x4 a[U], bq[U], c[U], d[U]; float lx[U];
#pragma unroll
            for (int i = 0; i < U; ++i) {
                const int oxi = min(ox + i * ppb, y.W - 1);                 // past the end: a valid pixel, never stored
                int x0, x1;
                bil_src(oxi, R, x.W, 0, x0, x1, lx[i]);
                a[i] = *reinterpret_cast<const u32x4*>(r0 + x0 * x.cs); bq[i] = *reinterpret_cast<const u32x4*>(r0 + x1 * x.cs);
                c[i] = *reinterpret_cast<const u32x4*>(r1 + x0 * x.cs); d[i] = *reinterpret_cast<const u32x4*>(r1 + x1 * x.cs);
            }
#pragma unroll
            for (int i = 0; i < U; ++i) {
                if (ox + i * ppb >= y.W) continue;
                float fa[N], fb[N], fc[N], fd[N], o[N];
                unpack16<T>(a[i], fa); unpack16<T>(bq[i], fb); unpack16<T>(c[i], fc); unpack16<T>(d[i], fd);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = bil_mix(ly, lx[i], fa[j], fb[j], fc[j], fd[j]);
                *reinterpret_cast<u32x4*>(drow + (ox + i * ppb) * y.cs) = pack16<T>(o);
            }
        }
    }
}

// xR (R = 2, 4, 8, 16): a thread owns an R x R block of OUTPUT pixels for one channel piece - rows [gy R, (gy + 1) R) x columns [gx R, (gx + 1) R).
// Its source pixels are the 3 x 3 neighbourhood of (gy, gx) (the first half of the rows interpolates between source rows gy - 1 and gy,
// the second between gy and gy + 1; columns likewise): 9 gathers, then R^2 outputs from registers - the horizontal interpolation of a
// column serves the R / 2 rows of its quadrant.  Same bil_src / bil_mix operations as the kernels above: bit-identical.  The stores are
// what is left: lanes = channel pieces of one pixel, so a wave writes whole pixel rows (C4: 2.1 GB per level).
template <typename T, int R>
__global__ __launch_bounds__(256) void bilinear_fwd_cells_kernel(salt_view x, salt_view y) {
    constexpr int N = Elem<T>::VE, H2 = R / 2;
    const int cpv = x.C / N;
    const int64_t units = (int64_t)x.B * x.H * x.W * cpv;
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
        const unsigned q = (unsigned)(u / cpv); const int c0 = (int)(u - (int64_t)q * cpv) * N;       // (host: B H W < 2^31)
        const unsigned r = q / (unsigned)x.W; const int gx = (int)(q - r * (unsigned)x.W);
        const int b = (int)(r / (unsigned)x.H), gy = (int)(r - (unsigned)b * (unsigned)x.H);
        const T* base = (const T*)x.p + (int64_t)b * x.H * x.W * x.cs + c0;
        int ry[3], rx[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { ry[k] = min(max(gy - 1 + k, 0), x.H - 1); rx[k] = min(max(gx - 1 + k, 0), x.W - 1); }
        u32x4 cr[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) cr[i][j] = *reinterpret_cast<const u32x4*>(base + ((int64_t)ry[i] * x.W + rx[j]) * x.cs);
        T* obase = (T*)y.p + (((int64_t)b * y.H + (int64_t)gy * R) * y.W + (int64_t)gx * R) * y.cs + c0;
#pragma unroll
        for (int qx = 0; qx < 2; ++qx) {
#pragma unroll
            for (int qy = 0; qy < 2; ++qy) {
                float fa[N], fb[N], fc[N], fd[N];
                unpack16<T>(cr[qy][qx], fa); unpack16<T>(cr[qy][qx + 1], fb); unpack16<T>(cr[qy + 1][qx], fc); unpack16<T>(cr[qy + 1][qx + 1], fd);
                for (int ox = 0; ox < H2; ++ox) {
                    const int oxl = qx * H2 + ox;
                    int x0, x1; float lx;
                    bil_src(gx * R + oxl, R, x.W, 0, x0, x1, lx);
                    float top[N], bot[N];
#pragma unroll
                    for (int j = 0; j < N; ++j) { top[j] = __fmaf_rn(lx, fb[j], __fmul_rn(1.f - lx, fa[j])); bot[j] = __fmaf_rn(lx, fd[j], __fmul_rn(1.f - lx, fc[j])); }
#pragma unroll
                    for (int oy = 0; oy < H2; ++oy) {
                        const int oyl = qy * H2 + oy;
                        int y0, y1; float ly;
                        bil_src(gy * R + oyl, R, x.H, 0, y0, y1, ly);
                        float o[N];
#pragma unroll
                        for (int j = 0; j < N; ++j) o[j] = __fmaf_rn(ly, bot[j], __fmul_rn(1.f - ly, top[j]));
                        *reinterpret_cast<u32x4*>(obase + ((int64_t)oyl * y.W + oxl) * y.cs) = pack16<T>(o);
                    }
                }
            }
        }
    }
}

__device__ __forceinline__ float bil_wgt(int o_, int R, int n, int i_) {      // weight of output o on input i along one axis (0: not referenced)
    int i0, i1; float l;
    bil_src(o_, R, n, 0, i0, i1, l);
    return (i0 == i_ ? 1.f - l : 0.f) + (i1 == i_ ? l : 0.f);
}

// MODE as in bilinear_bwd_kernel (0 direct, 1 x-pass, 2 y-pass).  R2: the x2 direct adjoint, whole 4 x 4 window in flight as raw pieces.
template <typename T, int MODE, bool R2>
__global__ __launch_bounds__(256) void bilinear_bwd_rows_kernel(salt_view x, salt_view y, int R, int accumulate) {
    constexpr int N = Elem<T>::VE;
    const int cpv = x.C / N, ppb = 256 / cpv;
    const int c0 = (int)(threadIdx.x % cpv) * N, px0 = (int)(threadIdx.x / cpv);
    const int rows = x.B * x.H;
    const int per_xcd = (rows + 7) >> 3, r_end = min(rows, ((int)(blockIdx.x & 7) + 1) * per_xcd);      // see bilinear_fwd_rows_kernel
    for (int row = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3); row < r_end; row += (int)(gridDim.x >> 3)) {
        const int b = row / x.H, iy = row - b * x.H;
        const T* base = (const T*)y.p + (int64_t)b * y.H * y.W * y.cs + c0;
        T* drow = (T*)x.p + (int64_t)row * x.W * x.cs + c0;
        int oy_lo = max(0, R * iy - R / 2), oy_hi = min(y.H - 1, R * iy + (3 * R) / 2 - 1);
        if (MODE == 1) { oy_lo = iy; oy_hi = iy; }
        if constexpr (R2) {
            float wy[4]; int oyc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int oy = 2 * iy - 1 + r;
                const bool in = oy >= 0 && oy < y.H;
                oyc[r] = min(max(oy, 0), y.H - 1);
                wy[r] = in ? bil_wgt(oy, 2, x.H, iy) : 0.f;
            }
            for (int ix = px0; ix < x.W; ix += ppb) {
                float wx[4]; int oxc[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ox = 2 * ix - 1 + q;
                    const bool in = ox >= 0 && ox < y.W;
                    oxc[q] = min(max(ox, 0), y.W - 1);
                    wx[q] = in ? bil_wgt(ox, 2, x.W, ix) : 0.f;
                }
                u32x4 g[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q) g[r][q] = *reinterpret_cast<const u32x4*>(base + ((int64_t)oyc[r] * y.W + oxc[q]) * y.cs);
                float o[N];
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (wy[r] == 0.f) continue;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float w = wy[r] * wx[q];
                        if (w != 0.f) {
                            float f[N];
                            unpack16<T>(g[r][q], f);
#pragma unroll
                            for (int j = 0; j < N; ++j) o[j] += w * f[j];
                        }
                    }
                }
                T* dst = drow + ix * x.cs;
                if (accumulate) { float old[N]; unpack16<T>(*reinterpret_cast<const u32x4*>(dst), old);
#pragma unroll
                    for (int j = 0; j < N; ++j) o[j] += old[j]; }
                *reinterpret_cast<u32x4*>(dst) = pack16<T>(o);
            }
        } else {
            constexpr int G = SALT_BIL_G;                                       // gathers in flight per batch (raw 16-byte pieces: 4 registers each)
            for (int ix = px0; ix < x.W; ix += ppb) {
                int ox_lo = max(0, R * ix - R / 2), ox_hi = min(y.W - 1, R * ix + (3 * R) / 2 - 1);
                if (MODE == 2) { ox_lo = ix; ox_hi = ix; }
                float o[N];
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] = 0.f;
                if (MODE == 0) {
                    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
                        const float wy = bil_wgt(oy, R, x.H, iy);
                        if (wy == 0.f) continue;
                        for (int ox = ox_lo; ox <= ox_hi; ox += G) {
                            u32x4 g[G]; float w[G];
#pragma unroll
                            for (int k = 0; k < G; ++k) {
                                const int oxk = min(ox + k, ox_hi);
                                w[k] = ox + k <= ox_hi ? wy * bil_wgt(oxk, R, x.W, ix) : 0.f;
                                g[k] = *reinterpret_cast<const u32x4*>(base + ((int64_t)oy * y.W + oxk) * y.cs);
                            }
#pragma unroll
                            for (int k = 0; k < G; ++k)
                                if (w[k] != 0.f) {
                                    float f[N];
                                    unpack16<T>(g[k], f);
#pragma unroll
                                    for (int j = 0; j < N; ++j) o[j] += w[k] * f[j];
                                }
                        }
                    }
                } else {
                    const int lo = MODE == 1 ? ox_lo : oy_lo, hi = MODE == 1 ? ox_hi : oy_hi;
                    const int n = MODE == 1 ? x.W : x.H, ii = MODE == 1 ? ix : iy;
                    const int64_t line = MODE == 1 ? (int64_t)iy * y.W * y.cs : (int64_t)ix * y.cs;
                    const int64_t step = MODE == 1 ? (int64_t)y.cs : (int64_t)y.W * y.cs;
                    for (int q = lo; q <= hi; q += G) {
                        u32x4 g[G]; float w[G];
#pragma unroll
                        for (int k = 0; k < G; ++k) {
                            const int qk = min(q + k, hi);
                            w[k] = q + k <= hi ? bil_wgt(qk, R, n, ii) : 0.f;
                            g[k] = *reinterpret_cast<const u32x4*>(base + line + qk * step);
                        }
#pragma unroll
                        for (int k = 0; k < G; ++k)
                            if (w[k] != 0.f) {
                                float f[N];
                                unpack16<T>(g[k], f);
#pragma unroll
                                for (int j = 0; j < N; ++j) o[j] += w[k] * f[j];
                            }
                    }
                }
                T* dst = drow + ix * x.cs;
                if (accumulate) { float old[N]; unpack16<T>(*reinterpret_cast<const u32x4*>(dst), old);
#pragma unroll
                    for (int j = 0; j < N; ++j) o[j] += old[j]; }
                *reinterpret_cast<u32x4*>(dst) = pack16<T>(o);
            }
        }
    }
}

// ---------------------------------------------------------------- replicate-pad adjoint
template <typename T, bool VEC>
__global__ void pad_fold_kernel(salt_view xp, int top, int bottom, int left, int right, salt_view x, int accumulate) {
    constexpr int N = Unit<T, VEC>::N;
    const int cpv = x.C / N;
    const int64_t units = (int64_t)x.B * x.H * x.W * cpv;
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
        int64_t pix = u / cpv; const int c0 = (int)(u - pix * cpv) * N;
        const int ix = (int)(pix % x.W); int64_t r = pix / x.W; const int iy = (int)(r % x.H); const int b = (int)(r / x.H);
        const int py0 = iy == 0 ? 0 : iy + top, py1 = iy == x.H - 1 ? iy + top + bottom : iy + top;
        const int px0 = ix == 0 ? 0 : ix + left, px1 = ix == x.W - 1 ? ix + left + right : ix + left;
        float o[N];
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] = 0.f;
        for (int py = py0; py <= py1; ++py)
            for (int px = px0; px <= px1; ++px) {
                float g[N];
                Unit<T, VEC>::ld((const T*)xp.p + (((int64_t)b * xp.H + py) * xp.W + px) * xp.cs + c0, g);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] += g[j];
            }
        T* dst = (T*)x.p + pix * x.cs + c0;
        if (accumulate) { float old[N]; Unit<T, VEC>::ld(dst, old);
#pragma unroll
            for (int j = 0; j < N; ++j) o[j] += old[j]; }
        Unit<T, VEC>::st(dst, o);
    }
}

// second half of the fused fold: one unit per (image, perimeter pixel, channel piece); sums the pad-ring pixels that clamp onto it
template <typename T, bool VEC>
__global__ void pad_fold_strip_kernel(const T* strip, int strip_cs, int top, int bottom, int left, int right, salt_view x, int nperim) {
    constexpr int N = Unit<T, VEC>::N;
    const int cpv = x.C / N, H = x.H, W = x.W;
    const int64_t ring = (int64_t)(top + bottom) * (W + left + right) + (int64_t)H * (left + right);
    const int64_t units = (int64_t)x.B * nperim * cpv;
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
        int64_t r = u / cpv; const int c0 = (int)(u - r * cpv) * N;
        const int e = (int)(r % nperim); const int b = (int)(r / nperim);
        int iy, ix;
        if (nperim == H * W) { iy = e / W; ix = e - iy * W; }                 // tiny maps: every pixel
        else if (e < W) { iy = 0; ix = e; }
        else if (e < 2 * W) { iy = H - 1; ix = e - W; }
        else { const int k = e - 2 * W; iy = 1 + (k >> 1); ix = (k & 1) ? W - 1 : 0; }
        const int py0 = iy == 0 ? 0 : iy + top, py1 = iy == H - 1 ? iy + top + bottom : iy + top;
        const int px0 = ix == 0 ? 0 : ix + left, px1 = ix == W - 1 ? ix + left + right : ix + left;
        if (py0 == py1 && px0 == px1) continue;                               // interior pixel of a tiny map
        float o[N];
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] = 0.f;
        for (int py = py0; py <= py1; ++py)
            for (int px = px0; px <= px1; ++px) {
                if (py == iy + top && px == ix + left) continue;              // the interior value is already in x
                float g[N];
                Unit<T, VEC>::ld(strip + ((int64_t)b * ring + fold_ring_index(py, px, H, W, top, bottom, left, right)) * strip_cs + c0, g);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] += g[j];
            }
        T* dst = (T*)x.p + (((int64_t)b * H + iy) * W + ix) * x.cs + c0;
        float old[N];
        Unit<T, VEC>::ld(dst, old);
#pragma unroll
        for (int j = 0; j < N; ++j) o[j] += old[j];
        Unit<T, VEC>::st(dst, o);
    }
}

template <typename T, bool VEC>
__global__ void add_kernel(salt_view a, salt_view b, salt_view y, int accumulate) {
    constexpr int N = Unit<T, VEC>::N;
    const int cpv = a.C / N;
    const int64_t units = (int64_t)a.B * a.H * a.W * cpv;
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
        const int64_t pix = u / cpv; const int c0 = (int)(u - pix * cpv) * N;
        float f[N], g[N];
        Unit<T, VEC>::ld((const T*)a.p + pix * a.cs + c0, f);
        if (b.p) { Unit<T, VEC>::ld((const T*)b.p + pix * b.cs + c0, g);
#pragma unroll
            for (int j = 0; j < N; ++j) f[j] += g[j]; }
        if (accumulate) { Unit<T, VEC>::ld((const T*)y.p + pix * y.cs + c0, g);
#pragma unroll
            for (int j = 0; j < N; ++j) f[j] += g[j]; }
        Unit<T, VEC>::st((T*)y.p + pix * y.cs + c0, f);
    }
}

template <typename T>
__global__ void layout_kernel(float* nchw, salt_view v, int to_nhwc) {
    const int64_t hw = (int64_t)v.H * v.W;
    const int64_t npix = (int64_t)v.B * hw;
    for (int64_t pix = blockIdx.x * 256LL + threadIdx.x; pix < npix; pix += gridDim.x * 256LL) {
        const int64_t b = pix / hw, s = pix - b * hw;
        T* row = (T*)v.p + pix * v.cs;
        for (int c = 0; c < v.C; ++c) {
            float* q = nchw + (b * v.C + c) * hw + s;
            if (to_nhwc) Elem<T>::st(row + c, *q); else *q = Elem<T>::ld(row + c);
        }
    }
}

inline bool same_shape(const salt_view& a, const salt_view& b) { return a.B == b.B && a.H == b.H && a.W == b.W && a.C == b.C; }

}  // namespace

extern "C" int salt_affine_act(const salt_affine_act_args* a, void* stream) {
    if (!a || !view_ok(a->y) || !view_ok(a->a) || !same_shape(a->y, a->a)) SALT_FAIL(SALT_E_BADARG, "affine_act: bad views");
    if (a->res.p && !same_shape(a->y, a->res)) SALT_FAIL(SALT_E_BADARG, "affine_act: residual shape");
    if ((a->scale == nullptr) != (a->shift == nullptr)) SALT_FAIL(SALT_E_BADARG, "affine_act: scale/shift");
    if (a->fin_acc && a->y.C > 4096) SALT_FAIL(SALT_E_BADARG, "affine_act: fin_acc supports <= 4096 channels");
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        const int ve = Elem<T>::VE;
        const bool v = vec_ok(a->y, ve) && vec_ok(a->a, ve) && vec_ok(a->res, ve);
        const int64_t units = view_pixels(a->y) * (a->y.C / (v ? ve : 1));
        if (a->fin_acc) {
            const salt_bn_finalize_args* f = static_cast<const salt_bn_finalize_args*>(a->fin);
            if (!f || f->C != a->y.C || !f->gamma || !f->beta || !f->mean || !f->invstd || !f->scale || !f->shift)
                SALT_FAIL(SALT_E_BADARG, "affine_act: fin_acc needs the complete salt_bn_finalize arguments");
            const BnFin fin{const_cast<double*>(a->fin_acc), nullptr, f->gamma, f->beta, f->running_mean, f->running_var, f->num_batches_tracked,
                            f->momentum, f->eps, f->mean, f->invstd, f->scale, f->shift};
            const size_t lds = (size_t)a->y.C * 2 * sizeof(float);
            if (v) hipLaunchKernelGGL((affine_act_kernel<T, true, SALT_AFF_UNITS, true>), dim3(ew_blocks(units)), dim3(256), lds, (hipStream_t)stream, a->y, nullptr, nullptr, a->res, a->relu, a->a, fin);
            else hipLaunchKernelGGL((affine_act_kernel<T, false, SALT_AFF_UNITS, true>), dim3(ew_blocks(units)), dim3(256), lds, (hipStream_t)stream, a->y, nullptr, nullptr, a->res, a->relu, a->a, fin);
        } else {
            EW_LAUNCH_U(affine_act_kernel, T, v, SALT_AFF_UNITS, units, (hipStream_t)stream, a->y, a->scale, a->shift, a->res, a->relu, a->a, BnFin{});
        }
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int64_t salt_bn_stats_floats(int nparts, int C) {
    return (int64_t)nparts * 2 * C;
}

extern "C" int salt_bn_finalize(const salt_bn_finalize_args* a, void* stream) {
    if (!a || !a->stats || !a->stats_cnt || a->C < 1 || a->nparts < 1 || !a->gamma || !a->beta || !a->mean || !a->invstd || !a->scale || !a->shift)
        SALT_FAIL(SALT_E_BADARG, "bn_finalize: bad args");
    const int rows = bn_rows_for(a->nparts);
    hipStream_t st = (hipStream_t)stream;
    if (rows == 4) hipLaunchKernelGGL(bn_finalize_kernel<4>, dim3(cdiv(a->C, 64)), dim3(256), 0, st, *a);
    else if (rows == 16) hipLaunchKernelGGL(bn_finalize_kernel<16>, dim3(cdiv(a->C, 16)), dim3(256), 0, st, *a);
    else hipLaunchKernelGGL(bn_finalize_kernel<64>, dim3(cdiv(a->C, 4)), dim3(256), 0, st, *a);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_bn_fold(const salt_bn_fold_args* a, void* stream) {
    if (!a || a->C < 1 || !a->gamma || !a->beta || !a->running_mean || !a->running_var || !a->scale || !a->shift) SALT_FAIL(SALT_E_BADARG, "bn_fold: bad args");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(a->C, 64)), dim3(64), 0, (hipStream_t)stream, *a);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

static int bn_bwd_nparts(const salt_bn_bwd_args* a, int64_t* ppb) {
    const int64_t npix = view_pixels(a->y);
    static const int64_t max_parts = getenv("SALT_BNB_PARTS") ? atoi(getenv("SALT_BNB_PARTS")) : 512;
    int64_t parts = (npix + 31) / 32;                       // >= 32 pixels per block (small maps need the blocks), <= 512 blocks
    if (parts > max_parts) parts = max_parts;
    if (parts < 1) parts = 1;
    const int64_t per = (npix + parts - 1) / parts;
    if (ppb) *ppb = per;
    return (int)((npix + per - 1) / per);
}

extern "C" int salt_bn_bwd_parts(const salt_bn_bwd_args* a) {
    if (!a || !view_ok(a->y)) return -1;
    return bn_bwd_nparts(a, nullptr);
}

extern "C" int salt_bn_bwd(const salt_bn_bwd_args* a, void* stream) {
    if (!a || !view_ok(a->da) || !view_ok(a->y) || !view_ok(a->dy) || !same_shape(a->da, a->y) || !same_shape(a->dy, a->y))
        SALT_FAIL(SALT_E_BADARG, "bn_bwd: bad views");
    if (a->relu && a->a.p && (!view_ok(a->a) || !same_shape(a->a, a->y))) SALT_FAIL(SALT_E_BADARG, "bn_bwd: forward output shape");
    if (a->relu && !a->a.p && (a->dres.p || !a->beta)) SALT_FAIL(SALT_E_BADARG, "bn_bwd: the ReLU mask can be recomputed from y only without a residual (and needs beta)");
    if (a->dres.p && !same_shape(a->dres, a->y)) SALT_FAIL(SALT_E_BADARG, "bn_bwd: dres shape");
    if (!a->mean || !a->invstd || !a->gamma || (!a->partials && a->partials_ready != 2 && a->partials_ready != 3 && !(a->fin_acc && !a->partials_ready)) || !a->coef) SALT_FAIL(SALT_E_BADARG, "bn_bwd: missing buffers");
    int64_t per = 0;
    int nparts = bn_bwd_nparts(a, &per);
    if (a->partials_ready == 2 || a->partials_ready == 3) {  // ... and finalized (salt_conv_args.bnb_fin): coef / dgamma / dbeta are ready; 3: see below
    } else if (a->partials_ready) {                         // the producer of da reduced already (salt_conv_args.bnb_*)
        if (a->nparts < 1) SALT_FAIL(SALT_E_BADARG, "bn_bwd: partials_ready needs nparts >= 1");
        nparts = a->nparts;
    } else if (a->nparts != nparts) SALT_FAIL(SALT_E_BADARG, "bn_bwd: nparts %d, expected %d", a->nparts, nparts);
    hipStream_t st = (hipStream_t)stream;
    const int C = a->y.C;
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        const int ve = Elem<T>::VE;
        const bool v = vec_ok(a->da, ve) && vec_ok(a->y, ve) && vec_ok(a->dy, ve) && vec_ok(a->dres, ve) && vec_ok(a->a, ve);
        const int N = v ? ve : 1;
        const int cpv = C / N;
        const int cvn = cpv < 256 ? cpv : 256;
        const size_t lds = (size_t)(256 / cvn) * cvn * N * 2 * sizeof(float) + 16;
        // fin_acc: the sums go through fp64 shards instead of partials.  With fin_ticket the reduce launch's last block finalizes
        // (coef ready for the apply pass); without, the apply pass finalizes the shards itself (partials_ready 3: they were filled
        // by the producer of da, salt_conv_args.bnb_acc)
        const bool fin_here = !a->partials_ready && a->fin_acc != nullptr;
        const bool fin_apply = a->fin_acc != nullptr && !a->fin_ticket && (a->partials_ready == 0 || a->partials_ready == 3);
        if (a->partials_ready == 3 && !fin_apply) SALT_FAIL(SALT_E_BADARG, "bn_bwd: partials_ready 3 needs fin_acc and no fin_ticket");
        if (fin_apply && C > 4096) SALT_FAIL(SALT_E_BADARG, "bn_bwd: fin_acc supports <= 4096 channels");
        // (partials_ready 3 + da_bias: the sums in fin_acc already include the bias term - salt_scse_bwd_args.bnb_acc, round 6)
        if (a->da_bias && ((a->partials_ready && a->partials_ready != 3) || view_pixels(a->y) >= ((int64_t)1 << 31)))
            SALT_FAIL(SALT_E_BADARG, "bn_bwd: da_bias needs the reduction pass of this call (partials_ready 0) or sums that include it (partials_ready 3)");
        const unsigned hw = (unsigned)(a->y.H * a->y.W);
        const int hw_shift = (hw & (hw - 1)) == 0 ? ilog2_ceil((int)hw) : -1;
        const BnbFin fin{fin_here ? a->fin_acc : nullptr, a->fin_ticket, a->dgamma, a->dbeta, a->coef, a->accumulate_param_grads, (double)view_pixels(a->y), a->da_bias, hw, hw_shift};
        if (!a->partials_ready) {
            if (v) hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, true>), dim3(nparts), dim3(256), lds, st, a->da, a->a, a->y, a->relu, a->mean, a->invstd, a->gamma, a->beta, a->partials, per, fin);
            else hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, false>), dim3(nparts), dim3(256), lds, st, a->da, a->a, a->y, a->relu, a->mean, a->invstd, a->gamma, a->beta, a->partials, per, fin);
            SALT_CHECK_LAUNCH();
        }
        if (a->partials_ready != 2 && a->partials_ready != 3 && !fin_here) {
            const int rows = bn_rows_for(nparts);
            const double M = (double)view_pixels(a->y);
            if (rows == 4) hipLaunchKernelGGL(bn_bwd_finalize_kernel<4>, dim3(cdiv(C, 64)), dim3(256), 0, st, a->partials, nparts, C, M, a->gamma, a->invstd, a->dgamma, a->dbeta, a->accumulate_param_grads, a->coef);
            else if (rows == 16) hipLaunchKernelGGL(bn_bwd_finalize_kernel<16>, dim3(cdiv(C, 16)), dim3(256), 0, st, a->partials, nparts, C, M, a->gamma, a->invstd, a->dgamma, a->dbeta, a->accumulate_param_grads, a->coef);
            else hipLaunchKernelGGL(bn_bwd_finalize_kernel<64>, dim3(cdiv(C, 4)), dim3(256), 0, st, a->partials, nparts, C, M, a->gamma, a->invstd, a->dgamma, a->dbeta, a->accumulate_param_grads, a->coef);
            SALT_CHECK_LAUNCH();
        }
        const int64_t units = view_pixels(a->y) * cpv;
        if (a->sec_acc) {
            // secondary sums: the vectorised fixed-piece path of the consumer-finalize apply pass only
            const int64_t grid_stride = (int64_t)ew_blocks(view_pixels(a->y) * cpv) * 256;
            if (!fin_apply || !v || !a->dres.p || a->accumulate_dres || !a->sec_mean || !a->sec_invstd || !view_ok(a->sec_y) || !same_shape(a->sec_y, a->y) ||
                !vec_ok(a->sec_y, ve) || 256 % cpv || grid_stride % cpv || a->da_bias)
                SALT_FAIL(SALT_E_BADARG, "bn_bwd: secondary sums need the consumer-finalize apply pass (fin_acc, no ticket), a freshly written dres, aligned views and a channel-piece count that divides 256");
            const BnbFin fa{a->fin_acc, nullptr, a->dgamma, a->dbeta, a->coef, a->accumulate_param_grads, (double)view_pixels(a->y), nullptr, hw, hw_shift};
            const BnbSec sc{a->sec_y.p, a->sec_y.cs, a->sec_mean, a->sec_invstd, a->sec_acc};
            const size_t ldss = (size_t)C * 3 * sizeof(float) + (size_t)256 * ve * 2 * sizeof(float);
            hipEvent_t ev_ = salt_take_fork_event();
            hipExtLaunchKernelGGL((bn_bwd_apply_kernel<T, true, true, true>), dim3(ew_blocks(view_pixels(a->y) * cpv)), dim3(256), ldss, st, nullptr, ev_, 0, a->da, a->a, a->y, a->relu,
                                  a->mean, a->invstd, a->gamma, a->beta, nullptr, a->dy, a->dres, a->accumulate_dres, fa, sc);
        } else if (fin_apply) {
            const BnbFin fa{a->fin_acc, nullptr, a->dgamma, a->dbeta, a->coef, a->accumulate_param_grads, (double)view_pixels(a->y), a->da_bias, hw, hw_shift};
            const size_t lds3 = (size_t)C * 3 * sizeof(float);
            hipEvent_t ev_ = salt_take_fork_event();
            if (v) hipExtLaunchKernelGGL((bn_bwd_apply_kernel<T, true, true>), dim3(ew_blocks(units)), dim3(256), lds3, st, nullptr, ev_, 0, a->da, a->a, a->y, a->relu, a->mean, a->invstd, a->gamma, a->beta, nullptr, a->dy, a->dres, a->accumulate_dres, fa, BnbSec{});
            else hipExtLaunchKernelGGL((bn_bwd_apply_kernel<T, false, true>), dim3(ew_blocks(units)), dim3(256), lds3, st, nullptr, ev_, 0, a->da, a->a, a->y, a->relu, a->mean, a->invstd, a->gamma, a->beta, nullptr, a->dy, a->dres, a->accumulate_dres, fa, BnbSec{});
        } else {
            EW_LAUNCH_EV(bn_bwd_apply_kernel, T, v, units, st, salt_take_fork_event(), a->da, a->a, a->y, a->relu, a->mean, a->invstd, a->gamma, a->beta, a->coef, a->dy, a->dres, a->accumulate_dres, BnbFin{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.0, a->da_bias, hw, hw_shift}, BnbSec{});
        }
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_relu_bwd(const salt_relu_bwd_args* a, void* stream) {
    if (!a || !view_ok(a->da) || !view_ok(a->dy) || !same_shape(a->da, a->dy)) SALT_FAIL(SALT_E_BADARG, "relu_bwd: bad views");
    if (a->a.p && !same_shape(a->a, a->da)) SALT_FAIL(SALT_E_BADARG, "relu_bwd: mask shape");
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        const int ve = Elem<T>::VE;
        const bool v = vec_ok(a->da, ve) && vec_ok(a->dy, ve) && vec_ok(a->a, ve);
        const int64_t units = view_pixels(a->da) * (a->da.C / (v ? ve : 1));
        EW_LAUNCH(relu_bwd_kernel, T, v, units, (hipStream_t)stream, a->da, a->a, a->dy, a->accumulate);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_maxpool2(const salt_maxpool2_args* a, void* stream) {
    if (!a || !view_ok(a->x) || !view_ok(a->y) || a->y.H != a->x.H / 2 || a->y.W != a->x.W / 2 || a->y.C != a->x.C || a->y.B != a->x.B)
        SALT_FAIL(SALT_E_BADARG, "maxpool2: bad views");
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        const int ve = Elem<T>::VE;
        const bool v = vec_ok(a->x, ve) && vec_ok(a->y, ve);
        const int64_t units = view_pixels(a->y) * (a->y.C / (v ? ve : 1));
        EW_LAUNCH(maxpool2_kernel, T, v, units, (hipStream_t)stream, a->x, a->y);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_maxpool2_bwd(const salt_maxpool2_bwd_args* a, void* stream) {
    if (!a || !view_ok(a->x) || !view_ok(a->dy) || !view_ok(a->dx) || !same_shape(a->x, a->dx) || a->dy.H != a->x.H / 2 || a->dy.W != a->x.W / 2 || a->dy.C != a->x.C)
        SALT_FAIL(SALT_E_BADARG, "maxpool2_bwd: bad views");
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        const int ve = Elem<T>::VE;
        const bool v = vec_ok(a->x, ve) && vec_ok(a->dy, ve) && vec_ok(a->dx, ve);
        const int64_t units = view_pixels(a->x) * (a->x.C / (v ? ve : 1));
        EW_LAUNCH(maxpool2_bwd_kernel, T, v, units, (hipStream_t)stream, a->x, a->dy, a->dx, a->accumulate);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_maxpool3s2(const salt_maxpool2_args* a, void* stream) {
    if (!a || !view_ok(a->x) || !view_ok(a->y) || a->y.H != (a->x.H + 1) / 2 || a->y.W != (a->x.W + 1) / 2 || a->y.C != a->x.C || a->y.B != a->x.B)
        SALT_FAIL(SALT_E_BADARG, "maxpool3s2: bad views (output is ceil(H / 2) x ceil(W / 2))");
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        const int ve = Elem<T>::VE;
        const bool v = vec_ok(a->x, ve) && vec_ok(a->y, ve);
        const int64_t units = view_pixels(a->y) * (a->y.C / (v ? ve : 1));
        EW_LAUNCH(maxpool3s2_kernel, T, v, units, (hipStream_t)stream, a->x, a->y);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_maxpool3s2_bwd(const salt_maxpool2_bwd_args* a, void* stream) {
    if (!a || !view_ok(a->x) || !view_ok(a->dy) || !view_ok(a->dx) || !same_shape(a->x, a->dx) || a->dy.H != (a->x.H + 1) / 2 ||
        a->dy.W != (a->x.W + 1) / 2 || a->dy.C != a->x.C)
        SALT_FAIL(SALT_E_BADARG, "maxpool3s2_bwd: bad views");
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        const int ve = Elem<T>::VE;
        const bool v = vec_ok(a->x, ve) && vec_ok(a->dy, ve) && vec_ok(a->dx, ve);
        const int64_t units = view_pixels(a->x) * (a->x.C / (v ? ve : 1));
        EW_LAUNCH(maxpool3s2_bwd_kernel, T, v, units, (hipStream_t)stream, a->x, a->dy, a->dx, a->accumulate);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_avgpool2(const salt_avgpool2_args* a, void* stream) {
    if (!a || !view_ok(a->x) || !view_ok(a->y) || a->y.H != a->x.H / 2 || a->y.W != a->x.W / 2 || a->y.C != a->x.C || a->y.B != a->x.B)
        SALT_FAIL(SALT_E_BADARG, "avgpool2: bad views");
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        const int ve = Elem<T>::VE;
        const bool v = vec_ok(a->x, ve) && vec_ok(a->y, ve);
        const int64_t units = (a->backward ? view_pixels(a->x) : view_pixels(a->y)) * (a->y.C / (v ? ve : 1));
        EW_LAUNCH(avgpool2_kernel, T, v, units, (hipStream_t)stream, a->x, a->y, a->backward, a->accumulate);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_bilinear(const salt_bilinear_args* a, void* stream) {
    if (!a || !view_ok(a->x) || !view_ok(a->y) || a->R < 1 || a->y.H != a->x.H * a->R || a->y.W != a->x.W * a->R || a->y.C != a->x.C || a->y.B != a->x.B)
        SALT_FAIL(SALT_E_BADARG, "bilinear: bad views");
    static const bool rows_off = getenv("SALT_BILINEAR_ROWS") && atoi(getenv("SALT_BILINEAR_ROWS")) == 0;      // A/B: the unit-per-thread kernels
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        const int ve = Elem<T>::VE;
        const bool v = vec_ok(a->x, ve) && vec_ok(a->y, ve);
        // row-structured kernels: vectorised views, align_corners = False, a whole number of pixels per 256-thread sweep, 32-bit row offsets
        const int cpv = a->x.C / ve;
        const bool rows_ok = !rows_off && v && !a->align_corners && cpv >= 1 && cpv <= 256 && 256 % cpv == 0 &&
                             (int64_t)a->y.W * a->y.cs < (1ll << 30) && (int64_t)a->y.H * a->y.W * a->y.cs < (1ll << 31) && view_pixels(a->y) / a->y.W < (1ll << 31);
        auto row_grid = [](int64_t rows) { const int64_t g = (rows + 7) / 8 * 8; return dim3((unsigned)(g < 16384 ? g : 16384)); };      // a multiple of 8: one row range per XCD
        static const bool cells_off = getenv("SALT_BILINEAR_CELLS") && atoi(getenv("SALT_BILINEAR_CELLS")) == 0;
        const bool cells_ok = rows_ok && !cells_off && !a->backward && view_pixels(a->x) < (1ll << 31) && (a->R == 2 || a->R == 4 || a->R == 8 || a->R == 16);
        if (cells_ok) {
            const int64_t units = view_pixels(a->x) * cpv;
            const dim3 grid((unsigned)((units + 255) / 256 < 65536 ? (units + 255) / 256 : 65536));
            if (a->R == 2) hipLaunchKernelGGL((bilinear_fwd_cells_kernel<T, 2>), grid, dim3(256), 0, (hipStream_t)stream, a->x, a->y);
            else if (a->R == 4) hipLaunchKernelGGL((bilinear_fwd_cells_kernel<T, 4>), grid, dim3(256), 0, (hipStream_t)stream, a->x, a->y);
            else if (a->R == 8) hipLaunchKernelGGL((bilinear_fwd_cells_kernel<T, 8>), grid, dim3(256), 0, (hipStream_t)stream, a->x, a->y);
            else hipLaunchKernelGGL((bilinear_fwd_cells_kernel<T, 16>), grid, dim3(256), 0, (hipStream_t)stream, a->x, a->y);
        } else if (!a->backward && rows_ok) {
            hipLaunchKernelGGL((bilinear_fwd_rows_kernel<T, 4>), row_grid((int64_t)a->y.B * a->y.H), dim3(256), 0, (hipStream_t)stream, a->x, a->y, a->R);
        } else if (a->backward && rows_ok) {
            if (a->R >= 4 && a->tmp) {
                salt_view t = a->x; t.p = a->tmp; t.H = a->y.H; t.cs = ((a->x.C + ve - 1) / ve) * ve;
                hipLaunchKernelGGL((bilinear_bwd_rows_kernel<T, 1, false>), row_grid((int64_t)t.B * t.H), dim3(256), 0, (hipStream_t)stream, t, a->y, a->R, 0);
                SALT_CHECK_LAUNCH();
                hipLaunchKernelGGL((bilinear_bwd_rows_kernel<T, 2, false>), row_grid((int64_t)a->x.B * a->x.H), dim3(256), 0, (hipStream_t)stream, a->x, t, a->R, a->accumulate);
            } else if (a->R == 2) {
                hipLaunchKernelGGL((bilinear_bwd_rows_kernel<T, 0, true>), row_grid((int64_t)a->x.B * a->x.H), dim3(256), 0, (hipStream_t)stream, a->x, a->y, a->R, a->accumulate);
            } else {
                hipLaunchKernelGGL((bilinear_bwd_rows_kernel<T, 0, false>), row_grid((int64_t)a->x.B * a->x.H), dim3(256), 0, (hipStream_t)stream, a->x, a->y, a->R, a->accumulate);
            }
        } else if (!a->backward) {
            const int64_t units = view_pixels(a->y) * (a->y.C / (v ? ve : 1));
            EW_LAUNCH(bilinear_fwd_kernel, T, v, units, (hipStream_t)stream, a->x, a->y, a->R, a->align_corners);
        } else {
            const int64_t units = view_pixels(a->x) * (a->x.C / (v ? ve : 1));
            if (a->R >= 4 && a->tmp) {
                // separable: x-pass into tmp [B, OH, W, C] (same dtype, contiguous), then y-pass into x
                salt_view t = a->x; t.p = a->tmp; t.H = a->y.H; t.cs = ((a->x.C + ve - 1) / ve) * ve;
                const int64_t tunits = view_pixels(t) * (a->x.C / (v ? ve : 1));
                if (v) hipLaunchKernelGGL((bilinear_bwd_kernel<T, true, 1>), dim3(ew_blocks(tunits)), dim3(256), 0, (hipStream_t)stream, t, a->y, a->R, 0, a->align_corners);
                else hipLaunchKernelGGL((bilinear_bwd_kernel<T, false, 1>), dim3(ew_blocks(tunits)), dim3(256), 0, (hipStream_t)stream, t, a->y, a->R, 0, a->align_corners);
                SALT_CHECK_LAUNCH();
                if (v) hipLaunchKernelGGL((bilinear_bwd_kernel<T, true, 2>), dim3(ew_blocks(units)), dim3(256), 0, (hipStream_t)stream, a->x, t, a->R, a->accumulate, a->align_corners);
                else hipLaunchKernelGGL((bilinear_bwd_kernel<T, false, 2>), dim3(ew_blocks(units)), dim3(256), 0, (hipStream_t)stream, a->x, t, a->R, a->accumulate, a->align_corners);
            } else {
                if (v) hipLaunchKernelGGL((bilinear_bwd_kernel<T, true, 0>), dim3(ew_blocks(units)), dim3(256), 0, (hipStream_t)stream, a->x, a->y, a->R, a->accumulate, a->align_corners);
                else hipLaunchKernelGGL((bilinear_bwd_kernel<T, false, 0>), dim3(ew_blocks(units)), dim3(256), 0, (hipStream_t)stream, a->x, a->y, a->R, a->accumulate, a->align_corners);
            }
        }
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_hyper_rows(const salt_hyper_rows_args* a, void* stream) {
    if (!a || a->nlev < 1 || a->nlev > 4 || !view_ok(a->y)) SALT_FAIL(SALT_E_BADARG, "hyper_rows: bad args");
    HyperKP k;
    k.y = a->y; k.nlev = a->nlev; k.ac = a->align_corners;
    for (int i = 0; i < 4; ++i) { k.x[i] = a->x[i < a->nlev ? i : 0]; k.R[i] = a->R[i < a->nlev ? i : 0]; }
    for (int i = 0; i < a->nlev; ++i) {
        const salt_view& x = a->x[i];
        if (!view_ok(x) || a->R[i] < 1 || a->y.H != x.H * a->R[i] || a->y.W != x.W * a->R[i] || x.B != a->y.B || x.C != a->x[0].C)
            SALT_FAIL(SALT_E_BADARG, "hyper_rows: level %d does not match the output grid", i);
    }
    if (a->y.C != a->nlev * a->x[0].C) SALT_FAIL(SALT_E_BADARG, "hyper_rows: y.C must be nlev * C");
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        const int ve = Elem<T>::VE;
        bool v = vec_ok(a->y, ve) && a->x[0].C % ve == 0;
        for (int i = 0; i < a->nlev; ++i) v = v && vec_ok(a->x[i], ve);
        const int64_t units = view_pixels(a->y) * (a->y.C / (v ? ve : 1));
        EW_LAUNCH(hyper_rows_kernel, T, v, units, (hipStream_t)stream, k);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_pad_fold(const salt_pad_fold_args* a, void* stream) {
    if (!a || !view_ok(a->x) || !view_ok(a->xp) || a->xp.H != a->x.H + a->top + a->bottom || a->xp.W != a->x.W + a->left + a->right || a->xp.C != a->x.C || a->xp.B != a->x.B)
        SALT_FAIL(SALT_E_BADARG, "pad_fold: bad views");
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        const int ve = Elem<T>::VE;
        const bool v = vec_ok(a->x, ve) && vec_ok(a->xp, ve);
        const int64_t units = view_pixels(a->x) * (a->x.C / (v ? ve : 1));
        EW_LAUNCH(pad_fold_kernel, T, v, units, (hipStream_t)stream, a->xp, a->top, a->bottom, a->left, a->right, a->x, a->accumulate);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int64_t salt_fold_strip_pixels(int H, int W, int top, int bottom, int left, int right) {
    return (int64_t)(top + bottom) * (W + left + right) + (int64_t)H * (left + right);
}

extern "C" int salt_pad_fold_strip(const salt_pad_fold_strip_args* a, void* stream) {
    if (!a || !view_ok(a->x) || !a->strip || a->strip_cs < a->x.C || a->top < 0 || a->bottom < 0 || a->left < 0 || a->right < 0)
        SALT_FAIL(SALT_E_BADARG, "pad_fold_strip: bad args");
    const int H = a->x.H, W = a->x.W;
    const int nperim = (H < 3 || W < 3) ? H * W : 2 * W + 2 * (H - 2);
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        const int ve = Elem<T>::VE;
        const bool v = vec_ok(a->x, ve) && (a->strip_cs % ve) == 0 && (reinterpret_cast<uintptr_t>(a->strip) & 15) == 0;
        const int64_t units = (int64_t)a->x.B * nperim * (a->x.C / (v ? ve : 1));
        EW_LAUNCH(pad_fold_strip_kernel, T, v, units, (hipStream_t)stream, (const T*)a->strip, a->strip_cs, a->top, a->bottom, a->left, a->right, a->x, nperim);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_add(const salt_add_args* a, void* stream) {
    if (!a || !view_ok(a->a) || !view_ok(a->y) || !same_shape(a->a, a->y)) SALT_FAIL(SALT_E_BADARG, "add: bad views");
    if (a->b.p && !same_shape(a->a, a->b)) SALT_FAIL(SALT_E_BADARG, "add: b shape");
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        const int ve = Elem<T>::VE;
        const bool v = vec_ok(a->a, ve) && vec_ok(a->b, ve) && vec_ok(a->y, ve);
        const int64_t units = view_pixels(a->a) * (a->a.C / (v ? ve : 1));
        EW_LAUNCH(add_kernel, T, v, units, (hipStream_t)stream, a->a, a->b, a->y, a->accumulate);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_layout(const salt_layout_args* a, void* stream) {
    if (!a || !a->nchw || !view_ok(a->nhwc)) SALT_FAIL(SALT_E_BADARG, "layout: bad args");
    const int64_t npix = view_pixels(a->nhwc);
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        hipLaunchKernelGGL(layout_kernel<T>, dim3(ew_blocks(npix)), dim3(256), 0, (hipStream_t)stream, a->nchw, a->nhwc, a->to_nhwc);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}
