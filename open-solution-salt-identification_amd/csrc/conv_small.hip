// conv_small.hip — thin-channel convolutions that are NOT dense contractions and therefore run on the
// vector ALUs instead of MFMA:
//   salt_conv_first      : first layer, Cin <= 4 straight from the fp32 NCHW batch the reference's loader
//                          hands over (ResNet stem conv7x7 s2 p3: encoders.py:23-31 / torchvision resnet.py;
//                          vanilla U-Net's first 3x3).  K*K*Cin <= 147 taps: input halo tile + weights in LDS.
//   salt_conv_first_wgrad: its weight gradient (the input image needs no data gradient).
//   salt_head1x1(_bwd)   : 1x1 convolution to <= 4 output channels (logit head: unet.py:84-87,
//                          unet_models.py:138), writing the fp32 NCHW logits the reference's loss and
//                          SegmentationModel.transform expect (models.py:121-126,164-167).
#include "common.h"

namespace {

constexpr int FT = 16;            // output tile edge (16x16 pixels per 256-thread workgroup)

struct FirstKP {
    const float* x; const float* w; void* y;
    const float* bias; const float* scale; const float* shift; float* stats; float* stats_cnt;
    int B, Cin, H, W, K, stride, pad, OH, OW, Cout, y_cs, relu, tiles_y, tiles_x, halo;
};

template <typename T>
__global__ __launch_bounds__(256) void conv_first_kernel(FirstKP p) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const int KKC = p.K * p.K * p.Cin;
    const int CoutP = (p.Cout + 15) & ~15;
    float* s_in = smf;                                   // [Cin][halo][halo]
    float* s_w = smf + p.Cin * p.halo * p.halo;          // [KKC][CoutP]
    float* s_red = s_w + KKC * CoutP;                    // [4][16]
    const int tid = threadIdx.x;
    int tile = blockIdx.x;
    const int txi = tile % p.tiles_x; tile /= p.tiles_x;
    const int tyi = tile % p.tiles_y; const int b = tile / p.tiles_y;
    const int oy0 = tyi * FT, ox0 = txi * FT;
    const int iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;
    for (int i = tid; i < p.Cin * p.halo * p.halo; i += 256) {
        const int hx = i % p.halo; int r = i / p.halo; const int hy = r % p.halo; const int ci = r / p.halo;
        const int iy = iy0 + hy, ix = ix0 + hx;
        s_in[i] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? p.x[(((int64_t)b * p.Cin + ci) * p.H + iy) * p.W + ix] : 0.f;
    }
    for (int i = tid; i < KKC * CoutP; i += 256) {
        const int co = i % CoutP, kk = i / CoutP;         // kk = (ci*K + kh)*K + kw
        s_w[i] = co < p.Cout ? p.w[(int64_t)co * KKC + kk] : 0.f;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    const int oy = oy0 + ty, ox = ox0 + tx;
    const bool valid = oy < p.OH && ox < p.OW;
    float cnt = valid ? 1.f : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((tid & 63) == 0) s_red[tid >> 6] = cnt;
    __syncthreads();
    cnt = s_red[0] + s_red[1] + s_red[2] + s_red[3];
    __syncthreads();
    T* yrow = (T*)p.y + (((int64_t)b * p.OH + (valid ? oy : 0)) * p.OW + (valid ? ox : 0)) * p.y_cs;
    for (int co0 = 0; co0 < p.Cout; co0 += 16) {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        for (int ci = 0; ci < p.Cin; ++ci)
            for (int kh = 0; kh < p.K; ++kh)
                for (int kw = 0; kw < p.K; ++kw) {
                    const float xv = s_in[(ci * p.halo + ty * p.stride + kh) * p.halo + tx * p.stride + kw];
                    const float4* wr = reinterpret_cast<const float4*>(s_w + ((ci * p.K + kh) * p.K + kw) * CoutP + co0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 w4 = wr[q];
                        acc[q * 4 + 0] += xv * w4.x; acc[q * 4 + 1] += xv * w4.y; acc[q * 4 + 2] += xv * w4.z; acc[q * 4 + 3] += xv * w4.w;
                    }
                }
        // whole 16-byte pieces where the layout allows it (a lane's 16 channels are 64 / 32 contiguous bytes: 4 / 2 stores instead of 16
        // scalar ones at a 64-byte lane stride - the kernel ran at 0.5 TB/s of stores)
        constexpr int VE = Elem<T>::VE;
        const bool vec = co0 + 16 <= p.Cout && (p.y_cs % VE) == 0 && (reinterpret_cast<uintptr_t>(p.y) & 15) == 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int co = co0 + j;
            if (co < p.Cout) {
                float v = acc[j] + (p.bias ? p.bias[co] : 0.f);
                if (p.scale) v = v * p.scale[co] + p.shift[co];
                if (p.relu) v = fmaxf(v, 0.f);
                acc[j] = v;
                if (valid && !vec) Elem<T>::st(yrow + co, v);
            }
        }
        if (valid && vec) {
#pragma unroll
            for (int q = 0; q < 16 / VE; ++q) *reinterpret_cast<u32x4*>(yrow + co0 + q * VE) = pack16<T>(acc + q * VE);
        }
        if (p.stats) {
            // tile statistics for 16 channels: sum, then M2 about the tile mean (two block reductions)
            float mean[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float s = valid ? acc[j] : 0.f;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                if ((tid & 63) == 0) s_red[(tid >> 6) * 16 + j] = s;
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; ++j) mean[j] = (s_red[j] + s_red[16 + j] + s_red[32 + j] + s_red[48 + j]);   // tile SUM
            if (tid < 16 && co0 + tid < p.Cout)
                p.stats[((int64_t)blockIdx.x * 2) * p.Cout + co0 + tid] = s_red[tid] + s_red[16 + tid] + s_red[32 + tid] + s_red[48 + tid];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float d = valid ? acc[j] - mean[j] / cnt : 0.f;
                float s = d * d;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                if ((tid & 63) == 0) s_red[(tid >> 6) * 16 + j] = s;
            }
            __syncthreads();
            if (tid < 16 && co0 + tid < p.Cout)
                p.stats[((int64_t)blockIdx.x * 2 + 1) * p.Cout + co0 + tid] = s_red[tid] + s_red[16 + tid] + s_red[32 + tid] + s_red[48 + tid];
            __syncthreads();
        }
    }
    if (p.stats && tid == 0) p.stats_cnt[blockIdx.x] = cnt;
}

int first_geom(int H, int W, int K, int stride, int pad, int* OH, int* OW) {
    *OH = (H + 2 * pad - K) / stride + 1;
    *OW = (W + 2 * pad - K) / stride + 1;
    return (FT - 1) * stride + K;
}

constexpr int MAXKK = 40;
struct FirstWgKP {
    const float* x; const void* dy; float* partials;
    int B, Cin, H, W, K, stride, pad, OH, OW, Cout, dy_cs, tiles_y, tiles_x, halo, groups, per;
    int mfma;                        // 1: the MFMA path of conv_first_wgrad_kernel (Cout % 16 == 0, K K Cin <= 16)
};

// PER: compile-time bound of the taps a thread accumulates (p.per <= PER): with the runtime bound alone the pixel loop carried
// MAXKK predicated multiply-adds per pixel even for a 3x3 single-channel layer that needs one
template <typename T, int PER>
__global__ __launch_bounds__(256) void conv_first_wgrad_kernel(FirstWgKP p) {
    extern __shared__ __attribute__((aligned(16))) float smf[];
    const int KKC = p.K * p.K * p.Cin;
    float* s_in = smf;                                   // [Cin][halo][halo]
    float* s_dy = smf + p.Cin * p.halo * p.halo;         // [256][Cout]
    const int tid = threadIdx.x;
    int tile = blockIdx.x;
    const int txi = tile % p.tiles_x; tile /= p.tiles_x;
    const int tyi = tile % p.tiles_y; const int b = tile / p.tiles_y;
    const int oy0 = tyi * FT, ox0 = txi * FT;
    const int iy0 = oy0 * p.stride - p.pad, ix0 = ox0 * p.stride - p.pad;
    for (int i = tid; i < p.Cin * p.halo * p.halo; i += 256) {
        const int hx = i % p.halo; int r = i / p.halo; const int hy = r % p.halo; const int ci = r / p.halo;
        const int iy = iy0 + hy, ix = ix0 + hx;
        s_in[i] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? p.x[(((int64_t)b * p.Cin + ci) * p.H + iy) * p.W + ix] : 0.f;
    }
    for (int i = tid; i < 256 * p.Cout; i += 256) {
        const int co = i % p.Cout, px = i / p.Cout;
        const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
        s_dy[i] = (oy < p.OH && ox < p.OW) ? Elem<T>::ld((const T*)p.dy + (((int64_t)b * p.OH + oy) * p.OW + ox) * p.dy_cs + co) : 0.f;
    }
    __syncthreads();
    if (p.mfma) {
        // Cout a multiple of 16, <= 16 taps (host): dW[co][tap] = sum_px dy[px][co] x[px + tap] on v_mfma_f32_16x16x4_f32 - rows = 16
        // output channels, columns = taps (padded to 16), contraction = 4 pixels per instruction; a wave takes 64 of the tile's 256
        // pixels, the four partial results meet in LDS.  (The scalar loop below walks the 256 pixels in ONE thread per (co, tap).)
        const int lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kg = lane >> 4;
        float* s_red = s_dy + 256 * p.Cout;                                   // [4][16][16]
        int koff = 0;
        const bool tap_on = l15 < KKC;
        if (tap_on) { const int kw = l15 % p.K; int r = l15 / p.K; const int kh = r % p.K; const int ci = r / p.K; koff = (ci * p.halo + kh) * p.halo + kw; }
        for (int cb = 0; cb < p.Cout; cb += 16) {
            f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
            for (int q = 0; q < 16; ++q) {
                const int px = wave * 64 + q * 4 + kg;
                const float a = s_dy[px * p.Cout + cb + l15];
                const float bq = tap_on ? s_in[((px >> 4) * p.stride) * p.halo + (px & 15) * p.stride + koff] : 0.f;
                acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, acc4, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) s_red[(wave * 16 + 4 * kg + r) * 16 + l15] = acc4[r];      // D[co = 4 kg + r][tap = l15]
            __syncthreads();
            {
                const int c16 = tid >> 4, t16 = tid & 15;
                const float v = ((s_red[c16 * 16 + t16] + s_red[(16 + c16) * 16 + t16]) + s_red[(32 + c16) * 16 + t16]) + s_red[(48 + c16) * 16 + t16];
                if (t16 < KKC) p.partials[(int64_t)blockIdx.x * p.Cout * KKC + (int64_t)(cb + c16) * KKC + t16] = v;
            }
            __syncthreads();
        }
        return;
    }
    const int co = tid % p.Cout, grp = tid / p.Cout;
    if (grp >= p.groups) return;
    float acc[PER];
    int koff[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        acc[j] = 0.f;
        const int kk = grp + j * p.groups;
        int off = 0;
        if (j < p.per && kk < KKC) { const int kw = kk % p.K; int r = kk / p.K; const int kh = r % p.K; const int ci = r / p.K; off = (ci * p.halo + kh) * p.halo + kw; }
        koff[j] = off;
    }
    for (int px = 0; px < 256; ++px) {
        const float g = s_dy[px * p.Cout + co];
        const int base = ((px >> 4) * p.stride) * p.halo + (px & 15) * p.stride;
#pragma unroll
        for (int j = 0; j < PER; ++j)
            if (j < p.per) acc[j] += g * s_in[base + koff[j]];
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int kk = grp + j * p.groups;
        if (j < p.per && kk < KKC) p.partials[(int64_t)blockIdx.x * p.Cout * KKC + (int64_t)co * KKC + kk] = acc[j];
    }
}

__global__ void partial_sum_kernel(const float* partials, int nparts, int64_t n, float* out, int accumulate) {
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
        float s = 0.f;
        for (int k = 0; k < nparts; ++k) s += partials[(int64_t)k * n + i];
        out[i] = accumulate ? out[i] + s : s;
    }
}

// many partials, few outputs (the first-layer weight gradient: 2048 partials of 144 values): one workgroup per output, thread t
// adds parts t, t + 256, ... in ascending order, then a fixed-order tree over the 256 threads
__global__ __launch_bounds__(256) void partial_sum_wide_kernel(const float* partials, int nparts, int64_t n, float* out, int accumulate) {
    __shared__ float sm[256];
    const int64_t i = blockIdx.x;
    float s = 0.f;
    for (int k = threadIdx.x; k < nparts; k += 256) s += partials[(int64_t)k * n + i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[i] = accumulate ? out[i] + sm[0] : sm[0];
}

// ---------------------------------------------------------------- 1x1 head, Cout <= 4
template <typename T>
__global__ void head1x1_kernel(salt_view x, const float* w, const float* bias, int Cout, float* y_nchw, salt_view y) {
    const int64_t hw = (int64_t)x.H * x.W, npix = (int64_t)x.B * hw;
    for (int64_t pix = blockIdx.x * 256LL + threadIdx.x; pix < npix; pix += gridDim.x * 256LL) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const T* row = (const T*)x.p + pix * x.cs;
        for (int c = 0; c < x.C; ++c) {
            const float v = Elem<T>::ld(row + c);
#pragma unroll
            for (int o = 0; o < 4; ++o) if (o < Cout) acc[o] += v * w[o * x.C + c];
        }
        const int64_t b = pix / hw, s = pix - b * hw;
#pragma unroll
        for (int o = 0; o < 4; ++o) if (o < Cout) {
            const float v = acc[o] + (bias ? bias[o] : 0.f);
            if (y_nchw) y_nchw[(b * Cout + o) * hw + s] = v; else Elem<T>::st((T*)y.p + pix * y.cs + o, v);
        }
    }
}

// vectorised: C/VE lanes cooperate on one pixel (each reads one 16-byte piece), butterfly-reduce the dots.
// CO > 0: the output count as a compile-time constant (2 for the logit head: half the FMAs and shuffles of the 4-wide loop, and no
// 64-bit division per pixel when H W is a power of two: hw_shift >= 0) with two pixels in flight per thread.  C4 (256 channels at
// 256 x 256 x 64 images: 2.1 GB) 0.91 ms -> see DESIGN 7.
template <typename T, int CO>
__global__ __launch_bounds__(256) void head1x1_vec_co_kernel(salt_view x, const float* w, const float* bias, float* y_nchw, salt_view y, int cpv_log2, int hw_shift) {
    constexpr int VE = Elem<T>::VE;
    const int cpv = 1 << cpv_log2;
    const int64_t hw = (int64_t)x.H * x.W, npix = (int64_t)x.B * hw;
    const int64_t units = npix << cpv_log2;
    const int64_t units_pad = (units + 255) & ~255LL;
    const int cv = threadIdx.x & (cpv - 1);
    float wr[CO][VE];
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int j = 0; j < VE; ++j) wr[o][j] = w[o * x.C + cv * VE + j];
    float bs[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) bs[o] = bias ? bias[o] : 0.f;
    const int64_t stride = gridDim.x * 256LL;
    for (int64_t u0 = blockIdx.x * 256LL + threadIdx.x; u0 < units_pad; u0 += 2 * stride) {
        u32x4 raw[2]; int64_t pixs[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t u = u0 + q * stride;
            pixs[q] = u >> cpv_log2;
            const int64_t pp = pixs[q] < npix ? pixs[q] : 0;                // past the end: a valid pixel, never stored
            raw[q] = *reinterpret_cast<const u32x4*>((const T*)x.p + pp * x.cs + cv * VE);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (u0 + q * stride >= units_pad) continue;                     // (wave-uniform: units_pad and the stride are multiples of 256)
            float f[VE], acc[CO];
            unpack16<T>(raw[q], f);
#pragma unroll
            for (int o = 0; o < CO; ++o) {
                acc[o] = head_dot<VE>(f, wr[o]);                             // same pinned sequence as head1x1_vec_kernel: bit-identical
            }
#pragma unroll
            for (int o = 0; o < CO; ++o)
                for (int s = 1; s < cpv; s <<= 1) acc[o] += __shfl_xor(acc[o], s);
            const int64_t pix = pixs[q];
            if (pix < npix && cv == 0) {
                const int64_t b = hw_shift >= 0 ? (pix >> hw_shift) : pix / hw, sp = pix - b * hw;
#pragma unroll
                for (int o = 0; o < CO; ++o) {
                    const float v = acc[o] + bs[o];
                    if (y_nchw) y_nchw[(b * CO + o) * hw + sp] = v; else Elem<T>::st((T*)y.p + pix * y.cs + o, v);
                }
            }
        }
    }
}

template <typename T>
__global__ void head1x1_vec_kernel(salt_view x, const float* w, const float* bias, int Cout, float* y_nchw, salt_view y, int cpv_log2) {
    constexpr int VE = Elem<T>::VE;
    const int cpv = 1 << cpv_log2;
    const int64_t hw = (int64_t)x.H * x.W, npix = (int64_t)x.B * hw;
    const int64_t units = npix << cpv_log2;
    const int64_t units_pad = (units + 255) & ~255LL;
    // 256 and the grid stride are multiples of cpv (a power of two): the thread's channel piece is loop invariant and its weights
    // live in registers (they were re-read from memory for every pixel: 16 loads per 16-byte piece of data)
    const int cv = threadIdx.x & (cpv - 1);
    float wr[4][VE];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < VE; ++j) wr[o][j] = o < Cout ? w[o * x.C + cv * VE + j] : 0.f;
    for (int64_t u = blockIdx.x * 256LL + threadIdx.x; u < units_pad; u += gridDim.x * 256LL) {
        const int64_t pix = u >> cpv_log2;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        if (pix < npix) {
            float f[VE];
            unpack16<T>(*reinterpret_cast<const u32x4*>((const T*)x.p + pix * x.cs + cv * VE), f);
#pragma unroll
            for (int o = 0; o < 4; ++o) if (o < Cout) acc[o] = head_dot<VE>(f, wr[o]);
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
            for (int s = 1; s < cpv; s <<= 1) acc[o] += __shfl_xor(acc[o], s);
        if (pix < npix && cv == 0) {
            const int64_t b = pix / hw, s = pix - b * hw;
#pragma unroll
            for (int o = 0; o < 4; ++o) if (o < Cout) {
                const float v = acc[o] + (bias ? bias[o] : 0.f);
                if (y_nchw) y_nchw[(b * Cout + o) * hw + s] = v; else Elem<T>::st((T*)y.p + pix * y.cs + o, v);
            }
        }
    }
}

// backward: dx[pix][c] = sum_o dy[o][pix] w[o][c];  gw[o][c] = sum_pix dy[o][pix] x[pix][c];  gb[o] = sum_pix dy[o][pix]
// threads = (pixel row, channel); per-thread register sums, rows combined through LDS in fixed order.
template <typename T>
__global__ __launch_bounds__(256) void head1x1_bwd_kernel(salt_view x, const float* w, int Cout, const float* dy_nchw, salt_view dx,
                                                          int accumulate, float* partials, int64_t pix_per_block) {
    extern __shared__ float sm[];                         // [R][cn][5]
    const int C = x.C;
    const int64_t hw = (int64_t)x.H * x.W, npix = (int64_t)x.B * hw;
    const int64_t p0 = blockIdx.x * pix_per_block;
    const int64_t p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
    const int PW = Cout * (C + 1);
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int cn = C - c0 < 256 ? C - c0 : 256;
        const int R = 256 / cn;
        const int row = threadIdx.x / cn, cl = threadIdx.x % cn, c = c0 + cl;
        float gw[4] = {0.f, 0.f, 0.f, 0.f}, gb[4] = {0.f, 0.f, 0.f, 0.f};
        if (row < R) {
            float wv[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) wv[o] = o < Cout ? w[o * C + c] : 0.f;
            for (int64_t pix = p0 + row; pix < p1; pix += R) {
                const int64_t b = pix / hw, s = pix - b * hw;
                const float xv = Elem<T>::ld((const T*)x.p + pix * x.cs + c);
                float d = 0.f;
#pragma unroll
                for (int o = 0; o < 4; ++o) if (o < Cout) {
                    const float g = dy_nchw[(b * Cout + o) * hw + s];
                    d += g * wv[o]; gw[o] += g * xv; gb[o] += g;
                }
                T* dst = (T*)dx.p + pix * dx.cs + c;
                if (accumulate) d += Elem<T>::ld(dst);
                Elem<T>::st(dst, d);
            }
#pragma unroll
            for (int o = 0; o < 4; ++o) sm[(row * cn + cl) * 5 + o] = gw[o];
            sm[(row * cn + cl) * 5 + 4] = 0.f;
            if (cl == 0 && c0 == 0) { /* bias sums ride in slot 4 of channel 0, one per output via a second pass below */ }
        }
        __syncthreads();
        if (row == 0) {
#pragma unroll
            for (int o = 0; o < 4; ++o) if (o < Cout) {
                float t = 0.f;
                for (int r = 0; r < R; ++r) t += sm[(r * cn + cl) * 5 + o];
                partials[(int64_t)blockIdx.x * PW + o * (C + 1) + c] = t;
            }
        }
        __syncthreads();
        if (c0 == 0) {                                     // bias: combine gb of the channel-0 threads of every row
            if (row < R && cl == 0) {
#pragma unroll
                for (int o = 0; o < 4; ++o) sm[row * 4 + o] = gb[o];
            }
            __syncthreads();
            if (threadIdx.x < Cout) {
                float t = 0.f;
                for (int r = 0; r < R; ++r) t += sm[r * 4 + threadIdx.x];
                partials[(int64_t)blockIdx.x * PW + threadIdx.x * (C + 1) + C] = t;
            }
            __syncthreads();
        }
    }
}

// 16-byte variant: thread = (pixel row, 16-byte channel group); U pixels in flight per thread.
template <typename T>
__global__ __launch_bounds__(256) void head1x1_bwd_vec_kernel(salt_view x, const float* w, int Cout, const float* dy_nchw, salt_view dx,
                                                              int accumulate, float* partials, int64_t pix_per_block) {
    constexpr int N = Elem<T>::VE;
    constexpr int U = 4;
    extern __shared__ float sm[];                         // [R][cpv][N][4]
    const int C = x.C, cpv = C / N;                       // host guarantees cpv <= 256
    const int64_t hw = (int64_t)x.H * x.W, npix = (int64_t)x.B * hw;
    const int64_t p0 = blockIdx.x * pix_per_block;
    const int64_t p1 = p0 + pix_per_block < npix ? p0 + pix_per_block : npix;
    const int PW = Cout * (C + 1);
    const int R = 256 / cpv;
    const int row = threadIdx.x / cpv, cv = threadIdx.x % cpv, c0 = cv * N;
    float gw[4][N], gb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int j = 0; j < N; ++j) gw[o][j] = 0.f;
    if (row < R) {
        float wv[4][N];
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int j = 0; j < N; ++j) wv[o][j] = o < Cout ? w[o * C + c0 + j] : 0.f;
        for (int64_t pixb = p0 + row; pixb < p1; pixb += (int64_t)U * R) {
            float xv[U][N], old[U][N], g[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t pix = pixb + (int64_t)u * R;
#pragma unroll
                for (int j = 0; j < N; ++j) { xv[u][j] = 0.f; old[u][j] = 0.f; }
#pragma unroll
                for (int o = 0; o < 4; ++o) g[u][o] = 0.f;
                if (pix < p1) {
                    const int64_t b = pix / hw, sp = pix - b * hw;
                    unpack16<T>(*reinterpret_cast<const u32x4*>((const T*)x.p + pix * x.cs + c0), xv[u]);
                    if (accumulate) unpack16<T>(*reinterpret_cast<const u32x4*>((const T*)dx.p + pix * dx.cs + c0), old[u]);
#pragma unroll
                    for (int o = 0; o < 4; ++o) if (o < Cout) g[u][o] = dy_nchw[(b * Cout + o) * hw + sp];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t pix = pixb + (int64_t)u * R;
                float d[N];
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    float t = 0.f;
#pragma unroll
                    for (int o = 0; o < 4; ++o) { t += g[u][o] * wv[o][j]; gw[o][j] += g[u][o] * xv[u][j]; }
                    d[j] = t + old[u][j];
                }
#pragma unroll
                for (int o = 0; o < 4; ++o) gb[o] += g[u][o];
                if (pix < p1) *reinterpret_cast<u32x4*>((T*)dx.p + pix * dx.cs + c0) = pack16<T>(d);
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int j = 0; j < N; ++j) sm[((row * cpv + cv) * N + j) * 4 + o] = gw[o][j];
    }
    __syncthreads();
    // cross-row sums, one thread per (channel, output) pair (rows in ascending order)
    for (int e = threadIdx.x; e < C * 4; e += 256) {
        const int o = e & 3, c = e >> 2;
        if (o < Cout) {
            float t = 0.f;
            for (int r = 0; r < R; ++r) t += sm[r * C * 4 + e];
            partials[(int64_t)blockIdx.x * PW + o * (C + 1) + c] = t;
        }
    }
    __syncthreads();
    if (row < R && cv == 0) {
#pragma unroll
        for (int o = 0; o < 4; ++o) sm[row * 4 + o] = gb[o];
    }
    __syncthreads();
    if (threadIdx.x < Cout) {
        float t = 0.f;
        for (int r = 0; r < R; ++r) t += sm[r * 4 + threadIdx.x];
        partials[(int64_t)blockIdx.x * PW + threadIdx.x * (C + 1) + C] = t;
    }
}

// 256 threads = 16 part-rows x 16 outputs, 8 loads in flight
__global__ __launch_bounds__(256) void head1x1_bwd_finalize(const float* partials, int nparts, int Cout, int C, float* gw, float* gb) {
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

// ---------------------------------------------------------------- stem via space-to-depth (see saltnet.h)
template <typename T>
__global__ void s2d_kernel(const float* x, int B, int Cin, int H, int W, salt_view z) {
    const int64_t npix = (int64_t)z.B * z.H * z.W;
    for (int64_t pix = blockIdx.x * 256LL + threadIdx.x; pix < npix; pix += gridDim.x * 256LL) {
        const int X = (int)(pix % z.W); int64_t r = pix / z.W; const int Y = (int)(r % z.H); const int b = (int)(r / z.H);
        float f[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) f[j] = 0.f;
        for (int c = 0; c < Cin; ++c) {
            const float* src = x + (((int64_t)b * Cin + c) * H + 2 * Y) * W + 2 * X;
            const float2 r0 = *reinterpret_cast<const float2*>(src), r1 = *reinterpret_cast<const float2*>(src + W);
            f[0 * Cin + c] = r0.x; f[1 * Cin + c] = r0.y; f[2 * Cin + c] = r1.x; f[3 * Cin + c] = r1.y;
        }
        T* dst = (T*)z.p + pix * z.cs;
        if (sizeof(T) == 2) { *reinterpret_cast<u32x4*>(dst) = pack16<T>(f); *reinterpret_cast<u32x4*>(dst + 8) = pack16<T>(f + 8); }
        else {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<u32x4*>(dst + 4 * q) = pack16<T>(f + 4 * q);
        }
    }
}

template <typename T>
__global__ void pack_stem_weight_kernel(const float* w, int Cout, int Cin, int K, T* wp) {
    constexpr int KCE = 64 / (int)sizeof(T);
    const int TT = (K + 1) / 2, half = TT / 2, pad = K / 2;
    const int total = TT * TT * Cout * KCE;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int kc = i % KCE; int r = i / KCE; const int n = r % Cout; const int t = r / Cout;
        const int dh = t / TT - half, dw = t % TT - half;
        float v = 0.f;
        if (kc < 4 * Cin) {
            const int pp = kc / Cin, c = kc - pp * Cin, ph = pp >> 1, pw = pp & 1;
            const int kh = 2 * dh + ph + pad, kw = 2 * dw + pw + pad;
            if (kh >= 0 && kh < K && kw >= 0 && kw < K) v = w[(((int64_t)n * Cin + c) * K + kh) * K + kw];
        }
        Elem<T>::st(wp + i, v);
    }
}

__global__ void stem_grad_unfold_kernel(const float* g16, int Cout, int Cin, int K, float* grad, int accumulate) {
    const int TT = (K + 1) / 2, half = TT / 2, pad = K / 2;
    const int total = Cout * Cin * K * K;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int kw = i % K; int r = i / K; const int kh = r % K; r /= K; const int c = r % Cin; const int n = r / Cin;
        // kh - pad = 2*dh + ph with ph in {0,1}
        const int eh = kh - pad, ew = kw - pad;
        const int ph = eh & 1, pw = ew & 1;
        const int dh = (eh - ph) / 2, dw = (ew - pw) / 2;
        const float v = g16[(((int64_t)n * 16 + (ph * 2 + pw) * Cin + c) * TT + dh + half) * TT + dw + half];
        grad[i] = accumulate ? grad[i] + v : v;
    }
}

int head_parts(const salt_view& x, int64_t* per) {
    const int64_t npix = view_pixels(x);
    int64_t parts = (npix + 63) / 64;
    if (parts > 1024) parts = 1024;
    if (parts < 1) parts = 1;
    const int64_t pp = (npix + parts - 1) / parts;
    if (per) *per = pp;
    return (int)((npix + pp - 1) / pp);
}

}  // namespace

extern "C" int salt_conv_first_stats_parts(const salt_conv_first_args* a) {
    if (!a) return -1;
    int OH, OW; first_geom(a->H, a->W, a->K, a->stride, a->pad, &OH, &OW);
    return a->B * cdiv(OH, FT) * cdiv(OW, FT);
}

extern "C" int salt_conv_first(const salt_conv_first_args* a, void* stream) {
    if (!a || !a->x || !a->w || !view_ok(a->y) || a->Cin < 1 || a->Cin > 4 || a->K < 1 || a->K > 7 || a->stride < 1 || a->stride > 2)
        SALT_FAIL(SALT_E_BADARG, "conv_first: bad args");
    FirstKP p;
    int OH, OW;
    p.halo = first_geom(a->H, a->W, a->K, a->stride, a->pad, &OH, &OW);
    if (a->y.H != OH || a->y.W != OW || a->y.B != a->B) SALT_FAIL(SALT_E_BADARG, "conv_first: output view is %dx%d, expected %dx%d", a->y.H, a->y.W, OH, OW);
    if ((a->scale == nullptr) != (a->shift == nullptr)) SALT_FAIL(SALT_E_BADARG, "conv_first: scale/shift");
    p.x = a->x; p.w = a->w; p.y = a->y.p; p.bias = a->bias; p.scale = a->scale; p.shift = a->shift; p.stats = a->stats; p.stats_cnt = a->stats_cnt;
    p.B = a->B; p.Cin = a->Cin; p.H = a->H; p.W = a->W; p.K = a->K; p.stride = a->stride; p.pad = a->pad; p.OH = OH; p.OW = OW;
    p.Cout = a->y.C; p.y_cs = a->y.cs; p.relu = a->relu; p.tiles_y = cdiv(OH, FT); p.tiles_x = cdiv(OW, FT);
    const int CoutP = (p.Cout + 15) & ~15;
    const size_t lds = sizeof(float) * ((size_t)p.Cin * p.halo * p.halo + (size_t)p.K * p.K * p.Cin * CoutP + 64);
    if (lds > 160 * 1024) SALT_FAIL(SALT_E_LDS, "conv_first: needs %zu bytes of LDS", lds);
    const dim3 grid((unsigned)(p.B * p.tiles_y * p.tiles_x));
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        auto kern = conv_first_kernel<T>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)stream, p);
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_conv_first_wgrad_parts(const salt_conv_first_wgrad_args* a) {
    if (!a) return -1;
    int OH, OW; first_geom(a->H, a->W, a->K, a->stride, a->pad, &OH, &OW);
    return a->B * cdiv(OH, FT) * cdiv(OW, FT);
}

extern "C" int salt_conv_first_wgrad(const salt_conv_first_wgrad_args* a, void* stream) {
    if (!a || !a->x || !view_ok(a->dy) || !a->partials || !a->grad || a->Cin < 1 || a->Cin > 4 || a->K < 1 || a->K > 7)
        SALT_FAIL(SALT_E_BADARG, "conv_first_wgrad: bad args");
    FirstWgKP p;
    int OH, OW;
    p.halo = first_geom(a->H, a->W, a->K, a->stride, a->pad, &OH, &OW);
    if (a->dy.H != OH || a->dy.W != OW || a->dy.B != a->B) SALT_FAIL(SALT_E_BADARG, "conv_first_wgrad: dy view shape");
    const int Cout = a->dy.C;
    if (Cout > 256) SALT_FAIL(SALT_E_UNSUPPORTED, "conv_first_wgrad: Cout %d > 256", Cout);
    p.x = a->x; p.dy = a->dy.p; p.partials = a->partials;
    p.B = a->B; p.Cin = a->Cin; p.H = a->H; p.W = a->W; p.K = a->K; p.stride = a->stride; p.pad = a->pad; p.OH = OH; p.OW = OW;
    p.Cout = Cout; p.dy_cs = a->dy.cs; p.tiles_y = cdiv(OH, FT); p.tiles_x = cdiv(OW, FT);
    const int KKC = a->K * a->K * a->Cin;
    p.groups = 256 / Cout;
    p.per = cdiv(KKC, p.groups);
    if (p.per > MAXKK) SALT_FAIL(SALT_E_UNSUPPORTED, "conv_first_wgrad: %d taps per thread > %d", p.per, MAXKK);
    const int nparts = a->B * p.tiles_y * p.tiles_x;
    if (a->nparts != nparts) SALT_FAIL(SALT_E_BADARG, "conv_first_wgrad: nparts %d, expected %d", a->nparts, nparts);
    static const bool mfma_off = getenv("SALT_FIRST_WGRAD_MFMA") && atoi(getenv("SALT_FIRST_WGRAD_MFMA")) == 0;
    p.mfma = (!mfma_off && Cout % 16 == 0 && KKC <= 16) ? 1 : 0;
    const size_t lds = sizeof(float) * ((size_t)p.Cin * p.halo * p.halo + (size_t)256 * Cout + (p.mfma ? 1024 : 0));
    if (lds > 160 * 1024) SALT_FAIL(SALT_E_LDS, "conv_first_wgrad: needs %zu bytes of LDS", lds);
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        auto kern = p.per <= 1 ? conv_first_wgrad_kernel<T, 1> : p.per <= 4 ? conv_first_wgrad_kernel<T, 4> : p.per <= 12 ? conv_first_wgrad_kernel<T, 12>
                                                                                                                 : conv_first_wgrad_kernel<T, MAXKK>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipLaunchKernelGGL(kern, dim3(nparts), dim3(256), lds, (hipStream_t)stream, p);
    })
    SALT_CHECK_LAUNCH();
    const int64_t n = (int64_t)Cout * KKC;
    if (nparts >= 64 && n <= 4096) hipLaunchKernelGGL(partial_sum_wide_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, a->partials, nparts, n, a->grad, a->accumulate);
    else hipLaunchKernelGGL(partial_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a->partials, nparts, n, a->grad, a->accumulate);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_s2d(const salt_s2d_args* a, void* stream) {
    if (!a || !a->x || !view_ok(a->z) || a->Cin < 1 || a->Cin > 4 || (a->H & 1) || (a->W & 1) || a->z.H != a->H / 2 || a->z.W != a->W / 2 ||
        a->z.C != 16 || a->z.B != a->B || (a->z.cs % 16) != 0 || (reinterpret_cast<uintptr_t>(a->z.p) & 15) || (reinterpret_cast<uintptr_t>(a->x) & 7))
        SALT_FAIL(SALT_E_BADARG, "s2d: bad args");
    const int64_t npix = view_pixels(a->z);
    const int blocks = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
    SALT_DISPATCH_DTYPE(a->dtype, T, { hipLaunchKernelGGL(s2d_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a->x, a->B, a->Cin, a->H, a->W, a->z); })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_pack_stem_weight(const salt_pack_stem_weight_args* a, void* stream) {
    if (!a || !a->w || !a->wp || a->Cin < 1 || a->Cin > 4 || !(a->K & 1) || a->K < 3 || a->K > 7) SALT_FAIL(SALT_E_BADARG, "pack_stem_weight: bad args");
    const int TT = (a->K + 1) / 2;
    const int total = TT * TT * a->Cout * (a->dtype == SALT_F32 ? 16 : 32);
    SALT_DISPATCH_DTYPE(a->dtype, T, { hipLaunchKernelGGL(pack_stem_weight_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, a->w, a->Cout, a->Cin, a->K, (T*)a->wp); })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_stem_grad_unfold(const salt_stem_grad_unfold_args* a, void* stream) {
    if (!a || !a->g16 || !a->grad || a->Cin < 1 || a->Cin > 4 || !(a->K & 1) || a->K < 3 || a->K > 7) SALT_FAIL(SALT_E_BADARG, "stem_grad_unfold: bad args");
    const int total = a->Cout * a->Cin * a->K * a->K;
    hipLaunchKernelGGL(stem_grad_unfold_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, a->g16, a->Cout, a->Cin, a->K, a->grad, a->accumulate);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_head1x1(const salt_head1x1_args* a, void* stream) {
    if (!a || !view_ok(a->x) || !a->w || a->Cout < 1 || a->Cout > 4) SALT_FAIL(SALT_E_BADARG, "head1x1: bad args");
    if (!a->y_nchw && (!view_ok(a->y) || a->y.C != a->Cout)) SALT_FAIL(SALT_E_BADARG, "head1x1: no output");
    const int64_t npix = view_pixels(a->x);
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        constexpr int VE = Elem<T>::VE;
        const int cpv = a->x.C / VE;
        const bool vec = (a->x.C % VE) == 0 && (a->x.cs % VE) == 0 && ((reinterpret_cast<uintptr_t>(a->x.p) & 15) == 0) && cpv >= 1 && cpv <= 64 && (cpv & (cpv - 1)) == 0;
        if (vec) {
            const int64_t units = npix * cpv;
            const int blocks = (int)((units + 255) / 256 < 4096 ? (units + 255) / 256 : 4096);
            const int64_t hw = (int64_t)a->x.H * a->x.W;
            const int hw_shift = (hw & (hw - 1)) == 0 ? ilog2_ceil((int)hw) : -1;
            if (a->Cout == 2 && hw < (1ll << 30))
                hipLaunchKernelGGL((head1x1_vec_co_kernel<T, 2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a->x, a->w, a->bias, a->y_nchw, a->y, ilog2_ceil(cpv), hw_shift);
            else
                hipLaunchKernelGGL(head1x1_vec_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a->x, a->w, a->bias, a->Cout, a->y_nchw, a->y, ilog2_ceil(cpv));
        } else {
            const int blocks = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
            hipLaunchKernelGGL(head1x1_kernel<T>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a->x, a->w, a->bias, a->Cout, a->y_nchw, a->y);
        }
    })
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_head1x1_bwd_parts(const salt_head1x1_bwd_args* a) {
    if (!a || !view_ok(a->x)) return -1;
    return head_parts(a->x, nullptr);
}

extern "C" int salt_head1x1_bwd(const salt_head1x1_bwd_args* a, void* stream) {
    if (!a || !view_ok(a->x) || !view_ok(a->dx) || !a->w || !a->dy_nchw || !a->partials || !a->gw || a->Cout < 1 || a->Cout > 4)
        SALT_FAIL(SALT_E_BADARG, "head1x1_bwd: bad args");
    int64_t per = 0;
    const int nparts = head_parts(a->x, &per);
    if (a->nparts != nparts) SALT_FAIL(SALT_E_BADARG, "head1x1_bwd: nparts %d, expected %d", a->nparts, nparts);
    const int PW = a->Cout * (a->x.C + 1);
    SALT_DISPATCH_DTYPE(a->dtype, T, {
        constexpr int VE = Elem<T>::VE;
        const bool v = a->x.C % VE == 0 && a->x.cs % VE == 0 && a->dx.cs % VE == 0 && a->x.C / VE <= 256 &&
                       ((reinterpret_cast<uintptr_t>(a->x.p) | reinterpret_cast<uintptr_t>(a->dx.p)) & 15) == 0;
        if (v) hipLaunchKernelGGL(head1x1_bwd_vec_kernel<T>, dim3(nparts), dim3(256), 256 * VE * 4 * sizeof(float), (hipStream_t)stream,
                                  a->x, a->w, a->Cout, a->dy_nchw, a->dx, a->accumulate, a->partials, per);
        else hipLaunchKernelGGL(head1x1_bwd_kernel<T>, dim3(nparts), dim3(256), 256 * 5 * sizeof(float), (hipStream_t)stream,
                                a->x, a->w, a->Cout, a->dy_nchw, a->dx, a->accumulate, a->partials, per);
    })
    SALT_CHECK_LAUNCH();
    hipLaunchKernelGGL(head1x1_bwd_finalize, dim3(cdiv(PW, 16)), dim3(256), 0, (hipStream_t)stream, a->partials, nparts, a->Cout, a->x.C, a->gw, a->gb);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}
