// se.hip — concurrent spatial + channel squeeze-excitation of the reference decoder
// (architectures/base.py:82-117): out = relu(x*cSE(x) + x*sSE(x)).
// The reference makes 3 extra full-tensor passes per SE branch; here forward is: one reduction pass
// (global average pool partials), a tiny per-image FC kernel, one fused apply pass.  Backward is one fused
// pass (dx direct terms + per-image partial sums), the FC backward, and one broadcast-add pass.
// Channel count must be a power of two (64 / 256 in the reference's U-Nets); C/VE lanes share a pixel.
#include "common.h"
#include <stdlib.h>

namespace {

template <typename T> struct P16 {
    static constexpr int N = Elem<T>::VE;
    static __device__ __forceinline__ void ld(const T* p, float* f) { unpack16<T>(*reinterpret_cast<const u32x4*>(p), f); }
    static __device__ __forceinline__ void st(T* p, const float* f) { *reinterpret_cast<u32x4*>(p) = pack16<T>(f); }
};

// Input transform of the scSE kernels (salt_scse_args.in_fin): x is a raw convolution output and the element the operator sees is
// a = round_to_storage(relu?(y scale + shift)) - ONE pinned expression in the three kernels that read x, so they agree bit for bit
// with each other (and with what salt_affine_act would have stored, up to the contraction of its own multiply-add).
template <typename T, int N>
__device__ __forceinline__ void in_transform(float* f, const float* sc, const float* sh, int relu) {
#pragma unroll
    for (int j = 0; j < N; ++j) { const float v = __fmaf_rn(f[j], sc[j], sh[j]); f[j] = relu ? fmaxf(v, 0.f) : v; }
    if constexpr (sizeof(T) == 2) { const u32x4 v = pack16<T>(f); unpack16<T>(v, f); }
}

// per-image partial channel sums: partials[b][part][C]
// FIN: x is raw - every workgroup finalizes the producer's statistics shards into LDS (fin_forward_consumer), workgroup 0 stores them
template <typename T, bool FIN = false>
__global__ __launch_bounds__(256) void gap_partial_kernel(salt_view x, float* partials, int nparts, int pix_per_part, double* acc, BnFin fin, int in_relu) {
    constexpr int N = P16<T>::N;
    extern __shared__ float sm[];
    const int C = x.C, cpv = C / N, R = 256 / cpv;
    const int b = blockIdx.x / nparts, part = blockIdx.x % nparts;
    const int hw = x.H * x.W;
    const int p0 = part * pix_per_part, p1 = min(p0 + pix_per_part, hw);
    const int row = threadIdx.x / cpv, cv = threadIdx.x % cpv;
    float s[N], sc[N], sh[N];
#pragma unroll
    for (int j = 0; j < N; ++j) s[j] = 0.f;
    if constexpr (FIN) {
        float* fs = sm + 256 * N;                          // [C] scale, [C] shift behind the row sums
        fin_forward_consumer(fin, C, fs, fs + C, blockIdx.x == 0);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < N; ++j) { sc[j] = fs[cv * N + j]; sh[j] = fs[C + cv * N + j]; }
    }
    for (int pix = p0 + row; pix < p1; pix += R) {
        float f[N];
        P16<T>::ld((const T*)x.p + ((int64_t)b * hw + pix) * x.cs + cv * N, f);
        if constexpr (FIN) in_transform<T, N>(f, sc, sh, in_relu);
#pragma unroll
        for (int j = 0; j < N; ++j) s[j] += f[j];
    }
#pragma unroll
    for (int j = 0; j < N; ++j) sm[(row * cpv + cv) * N + j] = s[j];
    __syncthreads();
    // cross-row sums, one thread per channel (rows in ascending order)
    for (int c = threadIdx.x; c < C; c += 256) {
        float t = 0.f;
        for (int r = 0; r < R; ++r) t += sm[r * C + c];
        if (acc) unsafeAtomicAdd(acc + (int64_t)b * C + c, (double)t);      // <= 64 parts meet per address; the consumer's prologue reads the sum
        else partials[((int64_t)b * nparts + part) * C + c] = t;
    }
}

// Apply pass with the per-image FC in its prologue (salt_scse_args.gap_acc): one workgroup per (image, pixel part) recomputes
// gap -> hidden -> gate_c of ITS image from the fp64 channel sums the launch before left in gap_acc (2 C R MACs: nothing), part 0
// stores them for backward.  Replaces se_fc_kernel + the grid-stride apply kernel.
template <typename T>
__global__ __launch_bounds__(256) void scse_apply_fc_kernel(salt_view x, const double* acc, int R, float inv_hw, const float* w1, const float* b1,
                                                            const float* w2, const float* b2, float* gap, float* hidden, float* gate_c,
                                                            const float* ws, const float* bs, float* gate_s, salt_view y, int cpv_log2,
                                                            int nparts, int pix_per_part, const float* in_scale, const float* in_shift, int in_relu) {
    constexpr int N = P16<T>::N;
    extern __shared__ float sm[];       // [C] gap, [R] hidden, [C] gate
    const int C = x.C, cpv = 1 << cpv_log2, RW = 256 >> cpv_log2;
    const int b = blockIdx.x / nparts, part = blockIdx.x % nparts;
    const int hw = x.H * x.W;
    float* sg = sm; float* shid = sm + C; float* sgate = sm + C + R;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float t = (float)acc[(int64_t)b * C + c] * inv_hw;
        sg[c] = t;
        if (part == 0) gap[b * C + c] = t;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += 256) {
        float t = b1[r];
        for (int c = 0; c < C; ++c) t += w1[r * C + c] * sg[c];
        t = fmaxf(t, 0.f);
        shid[r] = t;
        if (part == 0) hidden[b * R + r] = t;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float t = b2[c];
        for (int r = 0; r < R; ++r) t += w2[c * R + r] * shid[r];
        const float g = 1.f / (1.f + __expf(-t));
        sgate[c] = g;
        if (part == 0) gate_c[b * C + c] = g;
    }
    __syncthreads();
    const int p0 = part * pix_per_part, p1 = min(p0 + pix_per_part, hw);
    const int row = threadIdx.x >> cpv_log2, cv = threadIdx.x & (cpv - 1);
    const float bsv = bs[0];
    float wsv[N], gc[N], isc[N], ish[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { wsv[j] = ws[cv * N + j]; gc[j] = sgate[cv * N + j]; isc[j] = in_scale ? in_scale[cv * N + j] : 1.f; ish[j] = in_scale ? in_shift[cv * N + j] : 0.f; }
    const int iters = (p1 - p0 + RW - 1) / RW;
    for (int it = 0; it < iters; ++it) {
        const int pix = p0 + it * RW + row;
        const bool ok = pix < p1;
        const int64_t gp = (int64_t)b * hw + pix;
        float f[N];
        float dot = 0.f;
        if (ok) {
            P16<T>::ld((const T*)x.p + gp * x.cs + cv * N, f);
            if (in_scale) in_transform<T, N>(f, isc, ish, in_relu);
#pragma unroll
            for (int j = 0; j < N; ++j) dot += f[j] * wsv[j];
        }
        for (int s2 = 1; s2 < cpv; s2 <<= 1) dot += __shfl_xor(dot, s2);
        if (ok) {
            const float gs = 1.f / (1.f + __expf(-(dot + bsv)));
#pragma unroll
            for (int j = 0; j < N; ++j) f[j] = fmaxf(f[j] * (gc[j] + gs), 0.f);
            P16<T>::st((T*)y.p + gp * y.cs + cv * N, f);
            if (cv == 0) gate_s[gp] = gs;
        }
    }
}

// one block per image: gap -> hidden -> gate_c
__global__ void se_fc_kernel(const float* partials, int nparts, int C, int R, float inv_hw, const float* w1, const float* b1,
                             const float* w2, const float* b2, float* gap, float* hidden, float* gate_c) {
    extern __shared__ float sm[];       // [C] gap, [R] hidden
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float t = 0.f;
        for (int k = 0; k < nparts; ++k) t += partials[((int64_t)b * nparts + k) * C + c];
        t *= inv_hw;
        sm[c] = t; gap[b * C + c] = t;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        float t = b1[r];
        for (int c = 0; c < C; ++c) t += w1[r * C + c] * sm[c];
        t = fmaxf(t, 0.f);
        sm[C + r] = t; hidden[b * R + r] = t;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float t = b2[c];
        for (int r = 0; r < R; ++r) t += w2[c * R + r] * sm[C + r];
        gate_c[b * C + c] = 1.f / (1.f + __expf(-t));
    }
}

template <typename T>
__global__ void scse_apply_kernel(salt_view x, const float* gate_c, const float* ws, const float* bs, float* gate_s, salt_view y, int cpv_log2) {
    constexpr int N = P16<T>::N;
    const int cpv = 1 << cpv_log2;
    const int hw = x.H * x.W;
    const int64_t npix = (int64_t)x.B * hw;
    const int64_t units_pad = ((npix << cpv_log2) + 255) & ~255LL;
    const float bsv = bs[0];
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units_pad; u += gridDim.x * 256LL) {
        const int64_t pix = u >> cpv_log2; const int cv = (int)(u & (cpv - 1));
        float f[N];
        float dot = 0.f;
        const bool ok = pix < npix;
        if (ok) {
            P16<T>::ld((const T*)x.p + pix * x.cs + cv * N, f);
#pragma unroll
            for (int j = 0; j < N; ++j) dot += f[j] * ws[cv * N + j];
        }
        for (int s = 1; s < cpv; s <<= 1) dot += __shfl_xor(dot, s);
        if (ok) {
            const float gs = 1.f / (1.f + __expf(-(dot + bsv)));
            const int b = (int)(pix / hw);
#pragma unroll
            for (int j = 0; j < N; ++j) f[j] = fmaxf(f[j] * (gate_c[b * x.C + cv * N + j] + gs), 0.f);
            P16<T>::st((T*)y.p + pix * y.cs + cv * N, f);
            if (cv == 0) gate_s[pix] = gs;
        }
    }
}

// backward pass 1: dx = g*(gc+gs) + ds*ws ; per-(image,part) partial sums of g*x (-> d gate_c), ds*x (-> g_ws), ds (-> g_bs)
// BNB (round 6; needs in_scale and acc): x is the raw output of the Conv-BN-ReLU layer in front of the block and dx is that layer's dL/da
// (minus the per-image channel-SE constant dgap, which the layer's salt_bn_bwd adds on the fly) - so this pass also takes the four
// per-image sums that layer's BatchNorm backward needs, and the FC backward kernel turns them into the layer's (sum, sum xhat) shards:
//   A1 = sum m o, A2 = sum m o xhat (o = the stored dx), M0 = sum m, M1 = sum m xhat  =>  s1 = sum_b A1 + dgap M0, s2 = sum_b A2 + dgap M1
// salt_bn_bwd then runs WITHOUT its reduction pass (partials_ready 3 + da_bias).  acc rows are [6 C + 1] wide in this mode.
template <typename T, bool BNB = false>
__global__ __launch_bounds__(256) void scse_bwd1_kernel(salt_view x, salt_view y, salt_view dy, const float* gate_c, const float* gate_s,
                                                        const float* ws, salt_view dx, int accumulate, float* partials, int nparts,
                                                        int pix_per_part, int cpv_log2, double* acc, const float* in_scale, const float* in_shift, int in_relu,
                                                        const float* bn_mean, const float* bn_invstd) {
    constexpr int N = P16<T>::N;
    extern __shared__ float sm[];
    const int C = x.C, cpv = 1 << cpv_log2, R = 256 >> cpv_log2;
    const int b = blockIdx.x / nparts, part = blockIdx.x % nparts;
    const int hw = x.H * x.W;
    const int p0 = part * pix_per_part, p1 = min(p0 + pix_per_part, hw);
    const int row = threadIdx.x >> cpv_log2, cv = threadIdx.x & (cpv - 1);
    float s_gc[N], s_ws[N], s_bs = 0.f, gc[N], wsv[N], isc[N], ish[N];
    float a1[N], a2[N], m0[N], m1[N], bmu[N], bis[N];
#pragma unroll
    for (int j = 0; j < N; ++j) { s_gc[j] = 0.f; s_ws[j] = 0.f; gc[j] = gate_c[b * C + cv * N + j]; wsv[j] = ws[cv * N + j];
                                  isc[j] = in_scale ? in_scale[cv * N + j] : 1.f; ish[j] = in_scale ? in_shift[cv * N + j] : 0.f;
                                  a1[j] = 0.f; a2[j] = 0.f; m0[j] = 0.f; m1[j] = 0.f;
                                  bmu[j] = BNB ? bn_mean[cv * N + j] : 0.f; bis[j] = BNB ? bn_invstd[cv * N + j] : 0.f; }
    const int iters = (p1 - p0 + R - 1) / R;
    // the next iteration's four 16-byte loads are requested before this iteration's arithmetic (round 6: one pixel in flight per
    // thread left the 128 x 128 layer at 2.6 TB/s); same operations in the same order per pixel
    struct Ld { u32x4 x, y, g, old; float gs; };
    auto load = [&](int it, Ld& L) {
        const int pix = p0 + it * R + row;
        const int64_t gp = (int64_t)b * hw + (pix < p1 ? pix : p0);          // past the part: a valid pixel, never used
        L.x = *reinterpret_cast<const u32x4*>((const T*)x.p + gp * x.cs + cv * N);
        L.y = *reinterpret_cast<const u32x4*>((const T*)y.p + gp * y.cs + cv * N);
        L.g = *reinterpret_cast<const u32x4*>((const T*)dy.p + gp * dy.cs + cv * N);
        if (accumulate) L.old = *reinterpret_cast<const u32x4*>((const T*)dx.p + gp * dx.cs + cv * N);
        L.gs = gate_s[gp];
    };
    Ld cur, nxt;
    if (iters > 0) load(0, cur);
    for (int it = 0; it < iters; ++it) {
        if (it + 1 < iters) load(it + 1, nxt);
        const int pix = p0 + it * R + row;
        const bool ok = pix < p1;
        float xv[N], g[N], xraw[N];
        float dgs = 0.f, gs = 0.f;
        const int64_t gp = (int64_t)b * hw + pix;
        if (ok) {
            float yv[N];
            unpack16<T>(cur.x, xv);
            if constexpr (BNB) {
#pragma unroll
                for (int j = 0; j < N; ++j) xraw[j] = xv[j];
            }
            if (in_scale) in_transform<T, N>(xv, isc, ish, in_relu);
            unpack16<T>(cur.y, yv);
            unpack16<T>(cur.g, g);
            gs = cur.gs;
#pragma unroll
            for (int j = 0; j < N; ++j) { g[j] = yv[j] > 0.f ? g[j] : 0.f; dgs += g[j] * xv[j]; }
        }
        for (int s = 1; s < cpv; s <<= 1) dgs += __shfl_xor(dgs, s);
        if (ok) {
            const float ds = dgs * gs * (1.f - gs);
            float o[N];
#pragma unroll
            for (int j = 0; j < N; ++j) {
                o[j] = g[j] * (gc[j] + gs) + ds * wsv[j];
                s_gc[j] += g[j] * xv[j];
                s_ws[j] += ds * xv[j];
            }
            if (cv == 0) s_bs += ds;
            T* dst = (T*)dx.p + gp * dx.cs + cv * N;
            if (accumulate) { float old[N]; unpack16<T>(cur.old, old);
#pragma unroll
                for (int j = 0; j < N; ++j) o[j] += old[j]; }
            P16<T>::st(dst, o);
            if constexpr (BNB) {
                if constexpr (sizeof(T) == 2) { const u32x4 v = pack16<T>(o); unpack16<T>(v, o); }     // the value salt_bn_bwd will read
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    const float xh = (xraw[j] - bmu[j]) * bis[j];
                    const bool on = !in_relu || __fmaf_rn(xraw[j], isc[j], ish[j]) > 0.f;       // the forward decision (in_transform's pre-activation)
                    const float gg = on ? o[j] : 0.f, mm = on ? 1.f : 0.f;
                    a1[j] += gg; a2[j] += gg * xh; m0[j] += mm; m1[j] += mm * xh;
                }
            }
        }
        cur = nxt;
    }
    // block combine (fixed order)
    float* sg = sm; float* sw = sm + 256 * N; float* sb = sm + 512 * N;
#pragma unroll
    for (int j = 0; j < N; ++j) { sg[(row * cpv + cv) * N + j] = s_gc[j]; sw[(row * cpv + cv) * N + j] = s_ws[j]; }
    if (cv == 0) sb[row] = s_bs;
    __syncthreads();
    float* out = partials + ((int64_t)b * nparts + part) * (2 * C + 1);
    const int AW = BNB ? 6 * C + 1 : 2 * C + 1;           // width of an image's row of `acc`
    // cross-row sums, one thread per output (rows in ascending order)
    for (int e = threadIdx.x; e < 2 * C + 1; e += 256) {
        float t = 0.f;
        if (e < C) { for (int r = 0; r < R; ++r) t += sg[r * C + e]; }
        else if (e < 2 * C) { for (int r = 0; r < R; ++r) t += sw[r * C + e - C]; }
        else { for (int r = 0; r < R; ++r) t += sb[r]; }
        if (acc) unsafeAtomicAdd(acc + (int64_t)b * AW + e, (double)t);   // per-image sums over the parts: no se_parts_reduce launch
        else out[e] = t;
    }
    if constexpr (BNB) {
        __syncthreads();
        float* q0 = sm; float* q1 = sm + 256 * N; float* q2 = sm + 512 * N; float* q3 = sm + 768 * N;
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const int i = (row * cpv + cv) * N + j;
            q0[i] = a1[j]; q1[i] = a2[j]; q2[i] = m0[j]; q3[i] = m1[j];
        }
        __syncthreads();
        for (int e = threadIdx.x; e < 4 * C; e += 256) {
            const int k = e / C, c = e - k * C;
            const float* q = sm + k * 256 * N;
            float t = 0.f;
            for (int r = 0; r < R; ++r) t += q[r * C + c];
            unsafeAtomicAdd(acc + (int64_t)b * AW + 2 * C + 1 + e, (double)t);
        }
    }
}

// per image: sum the per-part partial rows into part 0's row in place (each thread owns one column)
__global__ void se_parts_reduce_kernel(float* partials, int nparts, int width) {
    float* base = partials + (int64_t)blockIdx.x * nparts * width;
    for (int i = threadIdx.x; i < width; i += blockDim.x) {
        float t = 0.f;
        for (int k = 0; k < nparts; ++k) t += base[(int64_t)k * width + i];
        base[i] = t;
    }
}

// FC backward: single block (sizes are B x C x R = tiny).  dgap[b][c] out; parameter grads written (not accumulated).
// `partials` rows have already been reduced over parts (row 0 of every image holds the sums).
// Round 6: the batch is walked in tiles of Bt images (Bt = B whenever the per-image vectors of the whole batch fit the 160 KB of LDS -
// every case that ran before, bit for bit); a later tile ADDS its share of the parameter gradients to what the first tile stored (the
// same thread owns the same output in every tile, so no atomics), which lifts the old limit of 51 images at C = 256 (R101 / R152
// decoders at BASELINE C3's batch 64).
__global__ __launch_bounds__(1024) void se_fc_bwd_kernel(const float* partials, int nparts, int Ball, int C, int R, const float* w1, const float* w2,
                                 const float* gap, const float* hidden, const float* gate_c,
                                 float* g_w1, float* g_b1, float* g_w2, float* g_b2, float* g_ws, float* g_bs, float* dgap, float inv_hw,
                                 const double* acc, int stage_w, int Bt, int AW, double* bnb_out, int write_dgap) {
    // AW: doubles per image row of `acc` (2 C + 1, or 6 C + 1 when scse_bwd1_kernel<.., BNB> also left A1, A2, M0, M1 there);
    // bnb_out != NULL: shard 0 of the producer layer's BatchNorm-backward sums [2][C] (the other shards stay zero)
    // everything the loops touch repeatedly is staged in LDS first (one coalesced sweep); the batch loops then run out of LDS
    extern __shared__ float sm[];       // du [Bt][C], dh [Bt][R], gp [Bt][C], hd [Bt][R], ps [Bt][C+1] (spatial-SE sums)
    float* du = sm; float* dh = du + Bt * C; float* gp = dh + Bt * R; float* hd = gp + Bt * C; float* ps = hd + Bt * R;
    // the FC weights: the batch loops below read them C / R times per output - staged too when the batch leaves room (stage_w), else read
    // from global as before round 4 (R101 / R152 decoders, C = 256, R = 16: batches 41 - 51 fit only without them)
    float* lw1 = ps + Bt * (C + 1); float* lw2 = lw1 + R * C;
    const float* sw1 = stage_w ? lw1 : w1; const float* sw2 = stage_w ? lw2 : w2;
    const int tid = threadIdx.x, nt = blockDim.x;
    if (stage_w) for (int i = tid; i < R * C; i += nt) { lw1[i] = w1[i]; lw2[i] = w2[i]; }
    for (int b0 = 0; b0 < Ball; b0 += Bt) {
        const int B = min(Bt, Ball - b0);
        const bool first = b0 == 0;
        if (!first) __syncthreads();                      // the previous tile's readers are done with the staged vectors
        for (int i = tid; i < B * C; i += nt) {
            const int b = i / C, c = i - b * C;
            const float* row = partials + ((int64_t)(b0 + b) * nparts) * (2 * C + 1);
            const double* arow = acc + (int64_t)(b0 + b) * AW;
            const float g = gate_c[(int64_t)b0 * C + i];
            du[i] = (acc ? (float)arow[c] : row[c]) * g * (1.f - g);
            gp[i] = gap[(int64_t)b0 * C + i];
            ps[b * (C + 1) + c] = acc ? (float)arow[C + c] : row[C + c];
            if (c == 0) ps[b * (C + 1) + C] = acc ? (float)arow[2 * C] : row[2 * C];
        }
        for (int i = tid; i < B * R; i += nt) hd[i] = hidden[(int64_t)b0 * R + i];
        __syncthreads();
        for (int c = tid; c < C + 1; c += nt) {
            float t = 0.f;
            for (int b = 0; b < B; ++b) t += ps[b * (C + 1) + c];
            float* o = c < C ? g_ws + c : g_bs;
            *o = first ? t : *o + t;
        }
        for (int i = tid; i < B * R; i += nt) {
            const int b = i / R, r = i - b * R;
            float t = 0.f;
            for (int c = 0; c < C; ++c) t += du[b * C + c] * sw2[c * R + r];
            dh[i] = hd[i] > 0.f ? t : 0.f;
        }
        for (int i = tid; i < C * R; i += nt) {
            const int c = i / R, r = i - c * R;
            float t = 0.f;
            for (int b = 0; b < B; ++b) t += du[b * C + c] * hd[b * R + r];
            g_w2[i] = first ? t : g_w2[i] + t;
        }
        for (int c = tid; c < C; c += nt) { float t = 0.f; for (int b = 0; b < B; ++b) t += du[b * C + c]; g_b2[c] = first ? t : g_b2[c] + t; }
        __syncthreads();
        for (int i = tid; i < R * C; i += nt) {
            const int r = i / C, c = i - r * C;
            float t = 0.f;
            for (int b = 0; b < B; ++b) t += dh[b * R + r] * gp[b * C + c];
            g_w1[i] = first ? t : g_w1[i] + t;
        }
        for (int r = tid; r < R; r += nt) { float t = 0.f; for (int b = 0; b < B; ++b) t += dh[b * R + r]; g_b1[r] = first ? t : g_b1[r] + t; }
        for (int i = tid; i < B * C; i += nt) {
            const int b = i / C, c = i - b * C;
            float t = 0.f;
            for (int r = 0; r < R; ++r) t += dh[b * R + r] * sw1[r * C + c];
            if (write_dgap) dgap[(int64_t)b0 * C + i] = t * inv_hw;
        }
        if (bnb_out) {
            __syncthreads();                              // this tile's dgap rows are visible to the whole workgroup
            // 16 adjacent lanes share a channel, each takes every 16th image of the tile (five independent loads per image, all in flight
            // together), then a 4-step butterfly: the first version - a thread per channel walking the whole tile - left 960 of the 1024
            // threads idle behind a chain of 32 dependent round trips and cost 14 us per launch
            constexpr int G = 16;
            for (int i = tid; i < C * G; i += nt) {
                const int c = i / G, gl = i % G;
                double t1 = 0.0, t2 = 0.0;
                for (int b = gl; b < B; b += G) {
                    const double* arow = acc + (int64_t)(b0 + b) * AW + 2 * C + 1;
                    const double dg = (double)dgap[(int64_t)(b0 + b) * C + c];
                    t1 += arow[c] + dg * arow[2 * C + c];
                    t2 += arow[C + c] + dg * arow[3 * C + c];
                }
#pragma unroll
                for (int sft = 1; sft < G; sft <<= 1) { t1 += __shfl_xor(t1, sft); t2 += __shfl_xor(t2, sft); }
                if (gl == 0) {
                    bnb_out[c] = first ? t1 : bnb_out[c] + t1;
                    bnb_out[C + c] = first ? t2 : bnb_out[C + c] + t2;
                }
            }
        }
    }
}

// Round 6 - the CRITICAL half of the FC backward, one workgroup per image: du -> dh -> dgap[b][:] (what the producer layer's
// salt_bn_bwd needs through da_bias) and, with bnb_out, that layer's BatchNorm-backward sums (fp64 atomics into shard b & 7).
// se_fc_bwd_kernel is ONE workgroup walking the whole batch (18 - 24 us of latency chain on the queue the data gradients wait on);
// the parameter gradients it also produces are nobody's input before the optimizer, so with salt_scse_bwd_args.defer_param_grads
// they move to salt_scse_fc_grads (the weight-gradient queue) and only this kernel stays on the critical one.  Same arithmetic per
// element as se_fc_bwd_kernel (du, dh, dgap in fp32, in the same order).
__global__ __launch_bounds__(256) void se_dgap_kernel(int C, int R, const float* w1, const float* w2, const float* hidden, const float* gate_c,
                                                      float* dgap, float inv_hw, const double* acc, int AW, double* bnb_out) {
    extern __shared__ float sm[];                        // du [C], dh [R]
    float* du = sm; float* dh = sm + C;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const double* arow = acc + (int64_t)b * AW;
    for (int c = tid; c < C; c += nt) {
        const float g = gate_c[(int64_t)b * C + c];
        du[c] = (float)arow[c] * g * (1.f - g);
    }
    __syncthreads();
    for (int r = tid; r < R; r += nt) {
        float t = 0.f;
        for (int c = 0; c < C; ++c) t += du[c] * w2[c * R + r];
        dh[r] = hidden[(int64_t)b * R + r] > 0.f ? t : 0.f;
    }
    __syncthreads();
    for (int c = tid; c < C; c += nt) {
        float t = 0.f;
        for (int r = 0; r < R; ++r) t += dh[r] * w1[r * C + c];
        const float dg = t * inv_hw;
        dgap[(int64_t)b * C + c] = dg;
        if (bnb_out) {
            const double* q = arow + 2 * C + 1;
            double* sh = bnb_out + (int64_t)(b & 7) * 2 * C;
            unsafeAtomicAdd(sh + c, q[c] + (double)dg * q[2 * C + c]);
            unsafeAtomicAdd(sh + C + c, q[C + c] + (double)dg * q[3 * C + c]);
        }
    }
}

template <typename T>
__global__ void bcast_add_kernel(salt_view dx, const float* dgap) {
    constexpr int N = P16<T>::N;
    const int cpv = dx.C / N;
    const int hw = dx.H * dx.W;
    const int64_t units = (int64_t)dx.B * hw * cpv;
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units; u += gridDim.x * 256LL) {
        const int64_t pix = u / cpv; const int cv = (int)(u - pix * cpv);
        const int b = (int)(pix / hw);
        float f[N];
        T* p = (T*)dx.p + pix * dx.cs + cv * N;
        P16<T>::ld(p, f);
#pragma unroll
        for (int j = 0; j < N; ++j) f[j] += dgap[b * dx.C + cv * N + j];
        P16<T>::st(p, f);
    }
}

int scse_nparts(const salt_view& x, int* per) {
    const int hw = x.H * x.W;
    static const int max_parts = getenv("SALT_SE_PARTS") ? atoi(getenv("SALT_SE_PARTS")) : 16;      // workgroups per image (every one pays the FC / statistics prologue); round 6: 16 (x 32 images = two per CU; 64: +0.02 ms, 8: +0.06, 4: +0.24)
    int parts = cdiv(hw, 256);
    if (parts > max_parts) parts = max_parts;
    if (parts < 1) parts = 1;
    const int pp = cdiv(hw, parts);
    if (per) *per = pp;
    return cdiv(hw, pp);
}

template <typename T> bool se_ok(const salt_view& v) {
    constexpr int VE = Elem<T>::VE;
    const int cpv = v.C / VE;
    return (v.C % VE) == 0 && (v.cs % VE) == 0 && ((reinterpret_cast<uintptr_t>(v.p) & 15) == 0) && cpv >= 1 && cpv <= 64 && (cpv & (cpv - 1)) == 0;
}

}  // namespace

extern "C" int salt_scse_parts(const salt_scse_args* a) {
    if (!a || !view_ok(a->x)) return -1;
    return scse_nparts(a->x, nullptr);
}

extern "C" int salt_scse(const salt_scse_args* a, void* stream) {
    if (!a || !view_ok(a->x) || !view_ok(a->y) || a->x.C != a->y.C || a->R < 1 || !a->w1 || !a->b1 || !a->w2 || !a->b2 || !a->ws || !a->bs ||
        (!a->gap_partials && !a->gap_acc) || !a->gap || !a->hidden || !a->gate_c || !a->gate_s) SALT_FAIL(SALT_E_BADARG, "scse: bad args");
    int per = 0;
    const int nparts = scse_nparts(a->x, &per);
    if (a->nparts != nparts) SALT_FAIL(SALT_E_BADARG, "scse: nparts %d, expected %d", a->nparts, nparts);
    hipStream_t st = (hipStream_t)stream;
    const int C = a->x.C;
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        if (!se_ok<T>(a->x) || !se_ok<T>(a->y)) SALT_FAIL(SALT_E_UNSUPPORTED, "scse: C=%d must be a power of two with 16-byte aligned rows", C);
        constexpr int VE = Elem<T>::VE;
        const salt_bn_finalize_args* f = static_cast<const salt_bn_finalize_args*>(a->in_fin);
        if (f) {
            if (!a->gap_acc || !a->in_fin_acc || f->C != C || !f->gamma || !f->beta || !f->mean || !f->invstd || !f->scale || !f->shift || C > 4096)
                SALT_FAIL(SALT_E_BADARG, "scse: in_fin needs gap_acc, in_fin_acc and the complete salt_bn_finalize arguments of a %d-channel layer", C);
            const BnFin fin{const_cast<double*>(a->in_fin_acc), nullptr, f->gamma, f->beta, f->running_mean, f->running_var, f->num_batches_tracked,
                            f->momentum, f->eps, f->mean, f->invstd, f->scale, f->shift};
            hipLaunchKernelGGL((gap_partial_kernel<T, true>), dim3(a->x.B * nparts), dim3(256), (256 * VE + 2 * C) * sizeof(float), st, a->x, a->gap_partials, nparts, per,
                               a->gap_acc, fin, a->in_relu);
        } else {
            hipLaunchKernelGGL((gap_partial_kernel<T, false>), dim3(a->x.B * nparts), dim3(256), 256 * VE * sizeof(float), st, a->x, a->gap_partials, nparts, per, a->gap_acc,
                               BnFin{}, 0);
        }
        SALT_CHECK_LAUNCH();
        if (a->gap_acc) {
            hipLaunchKernelGGL(scse_apply_fc_kernel<T>, dim3(a->x.B * nparts), dim3(256), (2 * C + a->R) * sizeof(float), st, a->x, a->gap_acc, a->R,
                               1.0f / (float)(a->x.H * a->x.W), a->w1, a->b1, a->w2, a->b2, a->gap, a->hidden, a->gate_c, a->ws, a->bs, a->gate_s, a->y,
                               ilog2_ceil(C / VE), nparts, per, f ? f->scale : nullptr, f ? f->shift : nullptr, a->in_relu);
            SALT_CHECK_LAUNCH();
            return SALT_OK;
        }
        hipLaunchKernelGGL(se_fc_kernel, dim3(a->x.B), dim3(256), (C + a->R) * sizeof(float), st, a->gap_partials, nparts, C, a->R,
                           1.0f / (float)(a->x.H * a->x.W), a->w1, a->b1, a->w2, a->b2, a->gap, a->hidden, a->gate_c);
        SALT_CHECK_LAUNCH();
        const int cpv_log2 = ilog2_ceil(C / VE);
        const int64_t units = view_pixels(a->x) << cpv_log2;
        const int blocks = (int)((units + 255) / 256 < 4096 ? (units + 255) / 256 : 4096);
        hipLaunchKernelGGL(scse_apply_kernel<T>, dim3(blocks), dim3(256), 0, st, a->x, a->gate_c, a->ws, a->bs, a->gate_s, a->y, cpv_log2);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

// the single-workgroup FC backward over the whole batch (parameter gradients; with write_dgap also dgap and, with bnb_out, the sums)
static int launch_se_fc(const salt_scse_bwd_args* a, int nparts, hipStream_t st, int write_dgap, double* bnb_out) {
    const int C = a->x.C, B = a->x.B;
    const bool bnb = a->bnb_acc != nullptr;
    const size_t fc_img = (size_t)(3 * C + 2 * a->R + 1) * sizeof(float), fc_w = (size_t)2 * a->R * C * sizeof(float);
    int stage_w = fc_img * B + fc_w <= 160 * 1024;                // the FC weights ride along only when they fit beside the per-image vectors
    int Bt = B;
    if (fc_img * B > 160 * 1024) {
        // the batch does not fit: tiles of Bt images with the weights staged (round 6; C = 256 above 51 images per GPU failed before)
        if (fc_w + fc_img > 160 * 1024) SALT_FAIL(SALT_E_LDS, "scse_bwd: %d channels x %d hidden units do not fit the FC backward", C, a->R);
        stage_w = 1;
        Bt = (int)((160 * 1024 - fc_w) / fc_img);
    }
    const size_t fc_lds = fc_img * Bt + (stage_w ? fc_w : 0);
    if (fc_lds > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(se_fc_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_set = true;
        }
    }
    hipLaunchKernelGGL(se_fc_bwd_kernel, dim3(1), dim3(1024), fc_lds, st, a->partials, nparts, B, C, a->R, a->w1, a->w2, a->gap, a->hidden,
                       a->gate_c, a->g_w1, a->g_b1, a->g_w2, a->g_b2, a->g_ws, a->g_bs, a->dgap, 1.0f / (float)(a->x.H * a->x.W), a->acc, stage_w, Bt,
                       bnb ? 6 * C + 1 : 2 * C + 1, bnb_out, write_dgap);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_scse_fc_grads(const salt_scse_bwd_args* a, void* stream) {
    if (!a || !view_ok(a->x) || !a->w1 || !a->w2 || !a->gap || !a->hidden || !a->gate_c || !a->acc || !a->g_w1 || !a->g_b1 || !a->g_w2 || !a->g_b2 || !a->g_ws ||
        !a->g_bs || !a->dgap) SALT_FAIL(SALT_E_BADARG, "scse_fc_grads: bad args");
    return launch_se_fc(a, scse_nparts(a->x, nullptr), (hipStream_t)stream, 0, nullptr);
}

extern "C" int salt_scse_bwd(const salt_scse_bwd_args* a, void* stream) {
    if (!a || !view_ok(a->x) || !view_ok(a->y) || !view_ok(a->dy) || !view_ok(a->dx) || !a->w1 || !a->w2 || !a->ws || !a->gap || !a->hidden ||
        !a->gate_c || !a->gate_s || (!a->partials && !a->acc) || ((a->in_scale == nullptr) != (a->in_shift == nullptr)) || !a->g_w1 || !a->g_b1 || !a->g_w2 || !a->g_b2 || !a->g_ws || !a->g_bs || !a->dgap)
        SALT_FAIL(SALT_E_BADARG, "scse_bwd: bad args");
    int per = 0;
    const int nparts = scse_nparts(a->x, &per);
    if (a->nparts != nparts) SALT_FAIL(SALT_E_BADARG, "scse_bwd: nparts %d, expected %d", a->nparts, nparts);
    hipStream_t st = (hipStream_t)stream;
    const int C = a->x.C, B = a->x.B;
    const bool bnb = a->bnb_acc != nullptr;
    if (bnb && (!a->acc || !a->in_scale || !a->bn_mean || !a->bn_invstd || !a->skip_bcast))
        SALT_FAIL(SALT_E_BADARG, "scse_bwd: bnb_acc needs acc, the input transform (in_scale / in_shift), bn_mean / bn_invstd and skip_bcast");
    if (a->defer_param_grads && !a->acc) SALT_FAIL(SALT_E_BADARG, "scse_bwd: defer_param_grads needs the fp64 sums (acc)");
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        if (!se_ok<T>(a->x) || !se_ok<T>(a->y) || !se_ok<T>(a->dy) || !se_ok<T>(a->dx)) SALT_FAIL(SALT_E_UNSUPPORTED, "scse_bwd: layout");
        constexpr int VE = Elem<T>::VE;
        const int cpv_log2 = ilog2_ceil(C / VE);
        if (bnb) hipLaunchKernelGGL((scse_bwd1_kernel<T, true>), dim3(B * nparts), dim3(256), 1024 * VE * sizeof(float), st, a->x, a->y, a->dy, a->gate_c,
                                    a->gate_s, a->ws, a->dx, a->accumulate, a->partials, nparts, per, cpv_log2, a->acc, a->in_scale, a->in_shift, a->in_relu,
                                    a->bn_mean, a->bn_invstd);
        else hipLaunchKernelGGL((scse_bwd1_kernel<T, false>), dim3(B * nparts), dim3(256), (512 * VE + 256) * sizeof(float), st, a->x, a->y, a->dy, a->gate_c,
                                a->gate_s, a->ws, a->dx, a->accumulate, a->partials, nparts, per, cpv_log2, a->acc, a->in_scale, a->in_shift, a->in_relu,
                                nullptr, nullptr);
        SALT_CHECK_LAUNCH();
        if (!a->acc) {
            hipLaunchKernelGGL(se_parts_reduce_kernel, dim3(B), dim3(256), 0, st, a->partials, nparts, 2 * C + 1);
            SALT_CHECK_LAUNCH();
        }
        if (a->defer_param_grads) {
            // only what the critical queue needs: dgap (+ the producer layer's BatchNorm-backward sums), one workgroup per image
            hipLaunchKernelGGL(se_dgap_kernel, dim3(B), dim3(256), (size_t)(C + a->R) * sizeof(float), st, C, a->R, a->w1, a->w2, a->hidden, a->gate_c, a->dgap,
                               1.0f / (float)(a->x.H * a->x.W), a->acc, bnb ? 6 * C + 1 : 2 * C + 1, bnb ? a->bnb_acc : nullptr);
        } else {
            const int rc = launch_se_fc(a, nparts, st, 1, bnb ? a->bnb_acc : nullptr);
            if (rc) return rc;
        }
        SALT_CHECK_LAUNCH();
        if (!a->skip_bcast) {                      // else: the consumer adds dgap[b][c] on the fly (salt_bn_bwd_args.da_bias)
            const int64_t units = view_pixels(a->dx) * (C / VE);
            const int blocks = (int)((units + 255) / 256 < 4096 ? (units + 255) / 256 : 4096);
            hipLaunchKernelGGL(bcast_add_kernel<T>, dim3(blocks), dim3(256), 0, st, a->dx, a->dgap);
        }
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}
