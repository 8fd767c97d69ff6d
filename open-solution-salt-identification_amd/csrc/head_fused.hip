// head_fused.hip — train-mode BatchNorm + ReLU + 1x1 logit head without the two tensors between them (round 6).
//
// Reference: the `final` Sequential of architectures/unet.py:84-87 (base.Conv2dBnRelu -> nn.Conv2d(C, num_classes, 1)) inside the
// training step of common_blocks/models.py:105-136.  In the step the final block's activation a = relu(bn(y)) [B,H,W,C] is read by
// the head only, and dL/da = W^T dlogits (rank num_classes per pixel) by the BatchNorm backward only.  The unfused operators move
// a: write + 2 reads, da: write + 2 reads, y: 3 reads, dy: write = 9 passes over a 67 MB tensor at the C2 shape (0.6 GB with the
// logits); here y is read three times and dy written once (0.27 GB) and two launches disappear from the critical queue.
//
//   salt_head_bn       y --(finalize shards, scale/shift/ReLU, round to storage dtype)--> head dot product --> fp32 NCHW logits
//   salt_head_bn_bwd   pass 1: per pixel da = round(W^T dl), mask from the SAME pinned pre-activation expression as forward;
//                              head gw / gb partials (fixed-order, like salt_head1x1_bwd) + BatchNorm-backward sums -> fp64 shards
//                      pass 2: every workgroup finalizes the shards (fin_backward_consumer), dy = A gg + D (y - mean) + E
//
// Rounding points are the unfused path's: `a` and `da` are rounded to the storage dtype where salt_affine_act / salt_head1x1_bwd would
// have stored them, so the two paths differ only in summation order (tests/test_gpu_head_fused.py bounds it).
#include "common.h"
#include <hip/hip_ext.h>
#include <stdlib.h>

namespace {

template <typename T, int N> __device__ __forceinline__ void round_storage(float* f) {
    if constexpr (sizeof(T) == 2) { const u32x4 v = pack16<T>(f); unpack16<T>(v, f); }
}

// the pre-activation of the fused layer: ONE pinned expression (fused multiply-add) in all three kernels, so the backward mask is the
// forward decision bit for bit
__device__ __forceinline__ float pre_act(float y, float sc, float sh) { return __fmaf_rn(y, sc, sh); }

// da_j = sum_o dl_o w[o][j]: pinned multiply, then fused multiply-adds in output order (both backward passes must agree on its bits)
template <int CO, int N>
__device__ __forceinline__ void head_da(const float* dl, const float (*wv)[N], float* da) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
        float t = __fmul_rn(dl[0], wv[0][j]);
#pragma unroll
        for (int o = 1; o < CO; ++o) t = __fmaf_rn(dl[o], wv[o][j], t);
        da[j] = t;
    }
}

// ---------------------------------------------------------------- forward
template <typename T, int CO>
__global__ __launch_bounds__(256) void head_bn_fwd_kernel(salt_view x, BnFin fin, int relu, const float* w, const float* bias, float* y_nchw,
                                                          int cpv_log2, int hw_shift) {
    constexpr int VE = Elem<T>::VE;
    extern __shared__ float fin_sm[];                    // [C] scale, [C] shift
    const int cpv = 1 << cpv_log2;
    const int64_t hw = (int64_t)x.H * x.W, npix = (int64_t)x.B * hw;
    const int64_t units = npix << cpv_log2;
    const int64_t units_pad = (units + 255) & ~255LL;
    const int cv = threadIdx.x & (cpv - 1);
    const int64_t stride = gridDim.x * 256LL;
    const int64_t first = blockIdx.x * 256LL + threadIdx.x;
    u32x4 raw[2]; int64_t pixs[2];
    auto load_iter = [&](int64_t u0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t u = u0 + q * stride;
            pixs[q] = u >> cpv_log2;
            const int64_t pp = pixs[q] < npix ? pixs[q] : 0;                // past the end: a valid pixel, never stored
            raw[q] = *reinterpret_cast<const u32x4*>((const T*)x.p + pp * x.cs + cv * VE);
        }
    };
    // the first iteration's loads go out before the statistics prologue (dependent shard loads + fp64 math + a barrier)
    if (first < units_pad) load_iter(first);
    fin_forward_consumer(fin, x.C, fin_sm, fin_sm + x.C, blockIdx.x == 0);
    __syncthreads();
    float wr[CO][VE], sc[VE], sh[VE], bs[CO];
#pragma unroll
    for (int j = 0; j < VE; ++j) { sc[j] = fin_sm[cv * VE + j]; sh[j] = fin_sm[x.C + cv * VE + j]; }
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        bs[o] = bias ? bias[o] : 0.f;
#pragma unroll
        for (int j = 0; j < VE; ++j) wr[o][j] = w[o * x.C + cv * VE + j];
    }
    for (int64_t u0 = first; u0 < units_pad; u0 += 2 * stride) {
        if (u0 != first) load_iter(u0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (u0 + q * stride >= units_pad) continue;                     // (wave-uniform: units_pad and the stride are multiples of 256)
            float f[VE], acc[CO];
            unpack16<T>(raw[q], f);
#pragma unroll
            for (int j = 0; j < VE; ++j) { const float v = pre_act(f[j], sc[j], sh[j]); f[j] = relu ? fmaxf(v, 0.f) : v; }
            round_storage<T, VE>(f);                                         // where salt_affine_act would have stored `a`
#pragma unroll
            for (int o = 0; o < CO; ++o) acc[o] = head_dot<VE>(f, wr[o]);    // the pinned sequence of salt_head1x1's kernels
#pragma unroll
            for (int o = 0; o < CO; ++o)
                for (int s = 1; s < cpv; s <<= 1) acc[o] += __shfl_xor(acc[o], s);
            const int64_t pix = pixs[q];
            if (pix < npix && cv == 0) {
                const int64_t b = hw_shift >= 0 ? (pix >> hw_shift) : pix / hw, sp = pix - b * hw;
#pragma unroll
                for (int o = 0; o < CO; ++o) y_nchw[(b * CO + o) * hw + sp] = acc[o] + bs[o];
            }
        }
    }
}

// ---------------------------------------------------------------- backward, pass 1: head parameter gradients + BatchNorm-backward sums
// thread = (pixel row, 16-byte channel piece); U pixels in flight; per-thread register sums, rows combined through LDS in fixed order
template <typename T, int CO>
__global__ __launch_bounds__(256) void head_bn_bwd_reduce_kernel(salt_view y, int relu, const float* mean, const float* invstd, const float* gamma,
                                                                 const float* beta, const float* w, const float* dy_nchw, float* partials,
                                                                 int64_t pix_per_block, double* acc, int hw_shift) {
    constexpr int N = Elem<T>::VE;
    constexpr int U = 4;
    extern __shared__ float sm[];                         // [R][cpv][N][CO] head sums, then [R][C][2] BatchNorm sums
    const int C = y.C, cpv = C / N;
    const int64_t hw = (int64_t)y.H * y.W, npix = (int64_t)y.B * hw;
    const int64_t p0 = blockIdx.x * pix_per_block;
    const int64_t p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
    const int PW = CO * (C + 1);
    const int R = 256 / cpv;
    const int row = threadIdx.x / cpv, cv = threadIdx.x % cpv, c0 = cv * N;
    float gw[CO][N], gb[CO], s1[N], s2[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        gb[o] = 0.f;
#pragma unroll
        for (int j = 0; j < N; ++j) gw[o][j] = 0.f;
    }
    if (row < R) {
        float wv[CO][N], mu[N], is[N], sc[N], sh[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            mu[j] = mean[c0 + j]; is[j] = invstd[c0 + j];
            sc[j] = gamma[c0 + j] * is[j]; sh[j] = beta[c0 + j] - mu[j] * sc[j];       // = what the forward finalize stored as scale / shift
        }
#pragma unroll
        for (int o = 0; o < CO; ++o)
#pragma unroll
            for (int j = 0; j < N; ++j) wv[o][j] = w[o * C + c0 + j];
        for (int64_t pixb = p0 + row; pixb < p1; pixb += (int64_t)U * R) {
            float yy[U][N], g[U][CO];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t pix = pixb + (int64_t)u * R;
#pragma unroll
                for (int o = 0; o < CO; ++o) g[u][o] = 0.f;
                if (pix < p1) {
                    const int64_t b = hw_shift >= 0 ? (pix >> hw_shift) : pix / hw, sp = pix - b * hw;
                    unpack16<T>(*reinterpret_cast<const u32x4*>((const T*)y.p + pix * y.cs + c0), yy[u]);
#pragma unroll
                    for (int o = 0; o < CO; ++o) g[u][o] = dy_nchw[(b * CO + o) * hw + sp];
                } else {
#pragma unroll
                    for (int j = 0; j < N; ++j) yy[u][j] = mu[j];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float a[N], da[N], pre[N];
#pragma unroll
                for (int j = 0; j < N; ++j) { pre[j] = pre_act(yy[u][j], sc[j], sh[j]); a[j] = relu ? fmaxf(pre[j], 0.f) : pre[j]; }
                round_storage<T, N>(a);
                head_da<CO, N>(g[u], wv, da);
                round_storage<T, N>(da);                                     // where salt_head1x1_bwd would have stored dL/da
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float gg = (!relu || pre[j] > 0.f) ? da[j] : 0.f;
                    s1[j] += gg; s2[j] += gg * (yy[u][j] - mu[j]) * is[j];
#pragma unroll
                    for (int o = 0; o < CO; ++o) gw[o][j] += g[u][o] * a[j];
                }
#pragma unroll
                for (int o = 0; o < CO; ++o) gb[o] += g[u][o];
            }
        }
#pragma unroll
        for (int o = 0; o < CO; ++o)
#pragma unroll
            for (int j = 0; j < N; ++j) sm[((row * cpv + cv) * N + j) * CO + o] = gw[o][j];
    }
    __syncthreads();
    // cross-row sums, one thread per (channel, output) pair (rows in ascending order)
    for (int e = threadIdx.x; e < C * CO; e += 256) {
        const int o = e % CO, c = e / CO;
        float t = 0.f;
        for (int r = 0; r < R; ++r) t += sm[r * C * CO + e];
        partials[(int64_t)blockIdx.x * PW + o * (C + 1) + c] = t;
    }
    __syncthreads();
    if (row < R && cv == 0) {
#pragma unroll
        for (int o = 0; o < CO; ++o) sm[row * CO + o] = gb[o];
    }
    __syncthreads();
    if (threadIdx.x < CO) {
        float t = 0.f;
        for (int r = 0; r < R; ++r) t += sm[r * CO + threadIdx.x];
        partials[(int64_t)blockIdx.x * PW + threadIdx.x * (C + 1) + C] = t;
    }
    __syncthreads();
    if (row < R) {
#pragma unroll
        for (int j = 0; j < N; ++j) { sm[((row * cpv + cv) * N + j) * 2] = s1[j]; sm[((row * cpv + cv) * N + j) * 2 + 1] = s2[j]; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 2 * C; e += 256) {
        const int st = e >= C ? 1 : 0, cl = e - st * C;
        float t = 0.f;
        for (int r = 0; r < R; ++r) t += sm[(r * C + cl) * 2 + st];
        fin_add(acc + ((blockIdx.x & 7) * 2 + st) * C + cl, (double)t);       // the shard layout of salt_bn_bwd's reduction pass
    }
}

// 256 threads = 16 part-rows x 16 outputs, 8 loads in flight (salt_head1x1_bwd's finalize, restated for this file)
__global__ __launch_bounds__(256) void head_bn_gw_finalize(const float* partials, int nparts, int Cout, int C, float* gw, float* gb) {
    __shared__ float sm[16][16];
    const int il = threadIdx.x & 15, row = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + il;
    const int PW = Cout * (C + 1);
    float s = 0.f;
    if (i < PW)
        for (int k = row; k < nparts; k += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int kk = k + 16 * u; v[u] = kk < nparts ? partials[(int64_t)kk * PW + i] : 0.f; }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
    sm[row][il] = s;
    __syncthreads();
    if (row != 0 || i >= PW) return;
    s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += sm[r][il];
    const int o = i / (C + 1), c = i - o * (C + 1);
    if (c < C) gw[o * C + c] = s; else if (gb) gb[o] = s;
}

// ---------------------------------------------------------------- backward, pass 2: dy
template <typename T, int CO>
__global__ __launch_bounds__(256) void head_bn_bwd_apply_kernel(salt_view y, int relu, const float* mean, const float* invstd, const float* gamma,
                                                                const float* beta, const float* w, const float* dy_nchw, salt_view dy, BnbFin fin,
                                                                int cpv_log2, int hw_shift) {
    constexpr int N = Elem<T>::VE;
    extern __shared__ float fin_sm[];                    // [3][C]: k, c1, c2
    const int C = y.C, cpv = 1 << cpv_log2;
    const int64_t hw = (int64_t)y.H * y.W, npix = (int64_t)y.B * hw;
    const int64_t units = npix << cpv_log2;
    const int64_t stride = gridDim.x * 256LL;
    const int64_t u0 = blockIdx.x * 256LL + threadIdx.x;
    const int c0 = (int)(u0 & (cpv - 1)) * N;
    constexpr int UB = 2;
    int64_t pix[UB];
    float yy[UB][N], g[UB][CO];
    auto load_iter = [&](int64_t u) {
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            const int64_t ui = u + i * stride;
            pix[i] = (ui < units ? ui : u) >> cpv_log2;                     // past the end: the first unit again, never stored
            unpack16<T>(*reinterpret_cast<const u32x4*>((const T*)y.p + pix[i] * y.cs + c0), yy[i]);
            const int64_t b = hw_shift >= 0 ? (pix[i] >> hw_shift) : pix[i] / hw, sp = pix[i] - b * hw;
#pragma unroll
            for (int o = 0; o < CO; ++o) g[i][o] = dy_nchw[(b * CO + o) * hw + sp];
        }
    };
    if (u0 < units) load_iter(u0);
    fin_backward_consumer(fin, gamma, invstd, C, fin_sm, blockIdx.x == 0);
    __syncthreads();
    float mu[N], A[N], D[N], E[N], sc[N], sh[N], wv[CO][N];
    {
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const float is = invstd[c0 + j], k0 = fin_sm[c0 + j], k1 = fin_sm[C + c0 + j], k2 = fin_sm[2 * C + c0 + j];
            mu[j] = mean[c0 + j];
            sc[j] = gamma[c0 + j] * is; sh[j] = beta[c0 + j] - mu[j] * sc[j];
            A[j] = k0; D[j] = -(k0 * k2) * is; E[j] = -(k0 * k1);
        }
#pragma unroll
        for (int o = 0; o < CO; ++o)
#pragma unroll
            for (int j = 0; j < N; ++j) wv[o][j] = w[o * C + c0 + j];
    }
    for (int64_t u = u0; u < units; u += UB * stride) {
        if (u != u0) load_iter(u);
#pragma unroll
        for (int i = 0; i < UB; ++i) {
            float da[N], o_[N];
            head_da<CO, N>(g[i], wv, da);
            round_storage<T, N>(da);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const float pre = pre_act(yy[i][j], sc[j], sh[j]);
                const float gg = (!relu || pre > 0.f) ? da[j] : 0.f;
                o_[j] = A[j] * gg + (D[j] * (yy[i][j] - mu[j]) + E[j]);
            }
            if (i == 0 || u + i * stride < units) *reinterpret_cast<u32x4*>((T*)dy.p + pix[i] * dy.cs + c0) = pack16<T>(o_);
        }
    }
}

template <typename T> bool head_bn_ok(const salt_view& v) {
    constexpr int VE = Elem<T>::VE;
    const int cpv = v.C / VE;
    return (v.C % VE) == 0 && (v.cs % VE) == 0 && ((reinterpret_cast<uintptr_t>(v.p) & 15) == 0) && cpv >= 1 && cpv <= 64 && (cpv & (cpv - 1)) == 0;
}

int head_bn_parts(const salt_view& x, int64_t* per) {
    const int64_t npix = view_pixels(x);
    int64_t parts = (npix + 63) / 64;
    if (parts > 1024) parts = 1024;
    if (parts < 1) parts = 1;
    const int64_t pp = (npix + parts - 1) / parts;
    if (per) *per = pp;
    return (int)((npix + pp - 1) / pp);
}

inline int head_bn_blocks(int64_t units) {
    static const int64_t cap = getenv("SALT_HEAD_BLOCKS") ? atoi(getenv("SALT_HEAD_BLOCKS")) : 1024;
    int64_t b = (units + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

#define SALT_HB(CO) hipLaunchKernelGGL((head_bn_fwd_kernel<T, CO>), dim3(blocks), dim3(256), lds, st, a->y, fin, a->relu, a->w, a->bias, a->y_nchw, ilog2_ceil(cpv), hw_shift)
#define SALT_HBR(CO) hipLaunchKernelGGL((head_bn_bwd_reduce_kernel<T, CO>), dim3(nparts), dim3(256), lds1, st, a->y, a->relu, a->mean, a->invstd, a->gamma, a->beta, \
                                        a->w, a->dy_nchw, a->partials, per, a->fin_acc, hw_shift)
#define SALT_HBA(CO) hipExtLaunchKernelGGL((head_bn_bwd_apply_kernel<T, CO>), dim3(blocks), dim3(256), lds2, st, nullptr, ev_, 0, a->y, a->relu, a->mean, a->invstd, \
                                           a->gamma, a->beta, a->w, a->dy_nchw, a->dy, fa, ilog2_ceil(cpv), hw_shift)

extern "C" int salt_head_bn(const salt_head_bn_args* a, void* stream) {
    if (!a || !view_ok(a->y) || !a->w || !a->y_nchw || !a->fin || !a->fin_acc || a->Cout < 1 || a->Cout > 4) SALT_FAIL(SALT_E_BADARG, "head_bn: bad args");
    const salt_bn_finalize_args* f = static_cast<const salt_bn_finalize_args*>(a->fin);
    if (f->C != a->y.C || !f->gamma || !f->beta || !f->mean || !f->invstd || !f->scale || !f->shift || a->y.C > 4096)
        SALT_FAIL(SALT_E_BADARG, "head_bn: needs the complete salt_bn_finalize arguments of a layer with %d channels", a->y.C);
    const BnFin fin{const_cast<double*>(a->fin_acc), nullptr, f->gamma, f->beta, f->running_mean, f->running_var, f->num_batches_tracked,
                    f->momentum, f->eps, f->mean, f->invstd, f->scale, f->shift};
    const int64_t hw = (int64_t)a->y.H * a->y.W;
    if (hw >= (1ll << 30)) SALT_FAIL(SALT_E_UNSUPPORTED, "head_bn: image too large");
    const int hw_shift = (hw & (hw - 1)) == 0 ? ilog2_ceil((int)hw) : -1;
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        constexpr int VE = Elem<T>::VE;
        if (!head_bn_ok<T>(a->y)) SALT_FAIL(SALT_E_UNSUPPORTED, "head_bn: C=%d must be a power-of-two number of aligned 16-byte pieces (<= 64)", a->y.C);
        const int cpv = a->y.C / VE;
        const int blocks = head_bn_blocks(view_pixels(a->y) * cpv);
        const size_t lds = (size_t)a->y.C * 2 * sizeof(float);
        hipStream_t st = (hipStream_t)stream;
        if (a->Cout == 1) SALT_HB(1); else if (a->Cout == 2) SALT_HB(2); else if (a->Cout == 3) SALT_HB(3); else SALT_HB(4);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_head_bn_bwd_parts(const salt_head_bn_bwd_args* a) {
    if (!a || !view_ok(a->y)) return -1;
    return head_bn_parts(a->y, nullptr);
}

extern "C" int salt_head_bn_bwd(const salt_head_bn_bwd_args* a, void* stream) {
    if (!a || !view_ok(a->y) || !view_ok(a->dy) || a->dy.B != a->y.B || a->dy.H != a->y.H || a->dy.W != a->y.W || a->dy.C != a->y.C || !a->mean || !a->invstd ||
        !a->gamma || !a->beta || !a->w || !a->dy_nchw || !a->partials || !a->gw || !a->fin_acc || !a->dgamma || !a->dbeta || a->Cout < 1 || a->Cout > 4 || a->y.C > 4096)
        SALT_FAIL(SALT_E_BADARG, "head_bn_bwd: bad args");
    int64_t per = 0;
    const int nparts = head_bn_parts(a->y, &per);
    if (a->nparts != nparts) SALT_FAIL(SALT_E_BADARG, "head_bn_bwd: nparts %d, expected %d", a->nparts, nparts);
    const int64_t hw = (int64_t)a->y.H * a->y.W;
    if (hw >= (1ll << 30)) SALT_FAIL(SALT_E_UNSUPPORTED, "head_bn_bwd: image too large");
    const int hw_shift = (hw & (hw - 1)) == 0 ? ilog2_ceil((int)hw) : -1;
    const int C = a->y.C, PW = a->Cout * (C + 1);
    hipStream_t st = (hipStream_t)stream;
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        constexpr int VE = Elem<T>::VE;
        if (!head_bn_ok<T>(a->y) || !head_bn_ok<T>(a->dy)) SALT_FAIL(SALT_E_UNSUPPORTED, "head_bn_bwd: C=%d must be a power-of-two number of aligned 16-byte pieces (<= 64)", C);
        const int cpv = C / VE;
        const size_t lds1 = (size_t)256 * VE * (a->Cout > 2 ? a->Cout : 2) * sizeof(float);
        if (a->Cout == 1) SALT_HBR(1); else if (a->Cout == 2) SALT_HBR(2); else if (a->Cout == 3) SALT_HBR(3); else SALT_HBR(4);
        SALT_CHECK_LAUNCH();
        hipLaunchKernelGGL(head_bn_gw_finalize, dim3(cdiv(PW, 16)), dim3(256), 0, st, a->partials, nparts, a->Cout, C, a->gw, a->gb);
        SALT_CHECK_LAUNCH();
        const BnbFin fa{a->fin_acc, nullptr, a->dgamma, a->dbeta, a->coef, 0, (double)view_pixels(a->y), nullptr, (unsigned)hw, hw_shift};
        const int blocks = head_bn_blocks(view_pixels(a->y) * cpv);
        const size_t lds2 = (size_t)C * 3 * sizeof(float);
        hipEvent_t ev_ = salt_take_fork_event();       // the launch that completes dL/dy carries the weight-gradient queue's fork (common.h)
        if (a->Cout == 1) SALT_HBA(1); else if (a->Cout == 2) SALT_HBA(2); else if (a->Cout == 3) SALT_HBA(3); else SALT_HBA(4);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}
