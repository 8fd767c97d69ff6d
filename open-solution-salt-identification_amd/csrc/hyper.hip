// hyper.hip — the factored hypercolumn (saltnet.h: salt_hyper_stencil).
//
// Reference: architectures/unet.py:101-109 builds cat([dec1, up2(dec2), up4(dec3), up8(dec4), up16(dec5)]) and runs
// Conv2dBnRelu(5 C, C) over it (architectures/base.py:21-37: replicate pad top 2 / right 2, 3x3, BatchNorm, ReLU).  For an
// up-sampled level the 1x1 contraction over channels commutes with the bilinear interpolation and with the tap shift:
//     conv3x3(pad(up_R(x)))[o, Y, X] = sum_{kh, kw} up_R(z[kh, kw])[o, max(Y + kh - 2, 0), min(X + kw, W - 1)],   z[kh, kw] = W[:, :, kh, kw] x
// z is computed at LOW resolution by a 1x1 convolution (Cin -> 9 Cout); what is left is a 2-D stencil with separable bilinear
// weights - no channel contraction, so it is vector-ALU / LDS work bounded by the stream of y:
//   forward   y = y_in + sum_levels sum_taps shift_tap(up_R(z[tap]))       hyper_stencil_fwd_kernel
//   backward  dz[tap] = up_R^T(shift_tap^T(dy))                            hyper_stencil_bwd_kernel
// Both are evaluated separably.  Forward, per 16 x 16 pixel tile and 64-channel block, per level and kernel row kh: the z patch of the
// three taps of that row (<= 7 x 7 low-resolution pixels for R = 4) is staged in LDS; stage H interpolates horizontally at the patch's
// low-resolution rows for the tile's 16 columns, summing the three column taps (6 FMAs per value); stage V interpolates vertically
// into per-thread fp32 accumulators (a thread owns 8 rows of one column x 8 channels and slides a two-row register window down the
// patch).  36 FMAs per output element for R = 4 / 8 / 16 together instead of 3 x 9 x Cin MACs, and the three up-sampled planes are
// never written or read.  Backward, per low-resolution row: stage V^T streams the full-resolution rows that touch it (thread = column
// x 8 channels, wave-uniform row weights), stage H^T gathers along the row out of LDS.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int HS_TH = 32, HS_TW = 16, HS_CB = 64, HS_NT = 1024;     // forward: pixel tile, channel block and threads of a workgroup
#ifndef HS_ABLATE
#define HS_ABLATE 0          // timing ablations of the forward kernel (tools/build_variant.sh hyper ... -DHS_ABLATE=n): 1 no stage H, 2 no stage V, 4 no z loads
#endif
constexpr int HS_TWL = 4;                                           // log2(HS_TW).  (32 x 8 tiles on 512-thread workgroups, two per CU, measured 3 % SLOWER:
                                                                    //  the rounds are VALU / LDS bound, not latency bound - DESIGN 7)
constexpr int HS_VR = HS_TH * HS_TW * (HS_CB / 8) / HS_NT;          // rows of a column a thread accumulates in stage V (4)

struct HsCoef { int i0, i1; float lam; int pad_; };    // bil_src of one full-resolution coordinate

struct HsLevel { void* z; int h, w, cs, R; unsigned blk0, nblk; int blk_lg, xlen; };   // adjoint: first workgroup id (a multiple of 8) / workgroups of the level, log2 of the X-range split of its gather items, columns per gather range
struct HsKP {
    HsLevel lev[4]; int nlev;
    const void* yin; int yin_cs;
    void* y; int y_cs;
    int B, H, W, C, ac;
    const float* scale; const float* shift; int relu;
    double* fin_acc;
    int tiles_x, tiles_y, cblocks, nr_max, nc_max;
    unsigned nblocks, per_xcd;
    const float* head_w; const float* head_b; float* head_y; float* head_ws; int head_co;     // eval: the 1x1 logit head on the stored values (per channel block; head_sum_kernel joins the blocks)
};

template <typename T> __device__ __forceinline__ void ld8(const T* p, float* f);
template <> __device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float* f) { unpack16<bf16_t>(*reinterpret_cast<const u32x4*>(p), f); }
template <> __device__ __forceinline__ void ld8<float>(const float* p, float* f) {
    unpack16<float>(*reinterpret_cast<const u32x4*>(p), f); unpack16<float>(*reinterpret_cast<const u32x4*>(p + 4), f + 4);
}
template <typename T> __device__ __forceinline__ void st8(T* p, const float* f);
template <> __device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float* f) { *reinterpret_cast<u32x4*>(p) = pack16<bf16_t>(f); }
template <> __device__ __forceinline__ void st8<float>(float* p, const float* f) {
    *reinterpret_cast<u32x4*>(p) = pack16<float>(f); *reinterpret_cast<u32x4*>(p + 4) = pack16<float>(f + 4);
}
// four consecutive channels from LDS as two packed pairs
template <typename T> __device__ __forceinline__ void ld4p(const T* p, f32x2_t& lo, f32x2_t& hi);
template <> __device__ __forceinline__ void ld4p<bf16_t>(const bf16_t* p, f32x2_t& lo, f32x2_t& hi) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    lo.x = __uint_as_float(v.x << 16); lo.y = __uint_as_float(v.x & 0xffff0000u); hi.x = __uint_as_float(v.y << 16); hi.y = __uint_as_float(v.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void ld4p<float>(const float* p, f32x2_t& lo, f32x2_t& hi) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p);
    lo.x = v.x; lo.y = v.y; hi.x = v.z; hi.y = v.w;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// hbuf[i][x][c]: the 16-byte slot of a channel quad is XOR-ed with bit 1 of the column, so that the four 16-lane groups of a
// ds_read_b128 in stage V (lane = channel piece + 8 x column) meet 16 different slots of the 256-byte bank row
__device__ __forceinline__ int hbuf_off(int i, int x, int quad) { return ((i * HS_TW + x) * (HS_CB / 4) + (quad ^ ((x >> 1) & 1))) * 4; }

struct HsGeo { int ib, nr, jb, nc; };
__device__ __forceinline__ HsGeo hs_geo(const HsKP& p, const HsLevel& L, int Y0, int X0) {
    HsGeo g; int a0, a1, b0, b1; float t;
    bil_src(clampi(Y0 - 2, 0, p.H - 1), L.R, L.h, p.ac, a0, a1, t);
    bil_src(clampi(Y0 + HS_TH - 1, 0, p.H - 1), L.R, L.h, p.ac, b0, b1, t);
    g.ib = a0; g.nr = b1 - a0 + 1;
    bil_src(clampi(X0, 0, p.W - 1), L.R, L.w, p.ac, a0, a1, t);
    bil_src(clampi(X0 + HS_TW + 1, 0, p.W - 1), L.R, L.w, p.ac, b0, b1, t);
    g.jb = a0; g.nc = b1 - a0 + 1;
    return g;
}

__device__ __forceinline__ HsLevel hs_level(const HsKP& p, int l) {          // (uniform selects: indexing the kernel argument with a run-time l puts it in scratch)
    HsLevel L = p.lev[0];
    if (l == 1) L = p.lev[1];
    if (l == 2) L = p.lev[2];
    if (l == 3) L = p.lev[3];
    return L;
}

// One workgroup (16 waves: the CU's 4 waves per SIMD at 128 registers) = one 32 x 16 pixel tile x 64 channels.  A ROUND is (level,
// kernel row dy): the z patch of the row's three taps goes global -> registers one round ahead (issued before stage H, stored to LDS
// after it), stage H interpolates horizontally at the patch rows (thread = column, channel quad, every fourth row), stage V vertically
// into the accumulators (thread = 4 rows of a column x 8 channels, a two-row register window sliding down the patch).  Two barriers
// per round; no global latency inside a round.
template <typename T, int NPF, bool FH>
__global__ __launch_bounds__(HS_NT) void hyper_stencil_fwd_kernel(HsKP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hs_smem[];
    constexpr int VE = 16 / (int)sizeof(T), PPC = HS_CB / VE;
    // consecutive workgroup ids go round-robin over the 8 XCDs: every XCD takes a contiguous range of tiles (a level's z of the images
    // it works on stays in ITS L2)
    const unsigned lid = (blockIdx.x & 7u) * p.per_xcd + (blockIdx.x >> 3);
    if (lid >= p.nblocks) return;
    unsigned r = lid;
    constexpr bool fh = FH;                                      // fused logit head (eval)
    const int cb = (int)(r % (unsigned)p.cblocks); r /= (unsigned)p.cblocks;
    const int tx = (int)(r % (unsigned)p.tiles_x); r /= (unsigned)p.tiles_x;
    const int ty = (int)(r % (unsigned)p.tiles_y);
    const int b = (int)(r / (unsigned)p.tiles_y);
    const int Y0 = ty * HS_TH, X0 = tx * HS_TW;

    const size_t hbuf_bytes = (size_t)p.nr_max * HS_TW * HS_CB * 4, zp_bytes = (size_t)p.nr_max * p.nc_max * 3 * HS_CB * 4;
    float* hbuf = reinterpret_cast<float*>(hs_smem);                                            // [nr_max][HS_TW][HS_CB] fp32
    float* zp = reinterpret_cast<float*>(hs_smem + hbuf_bytes);                                 // [nr][nc][3][HS_CB] fp32 (converted once, at the store: stage H reads every element ~5 times)
    HsCoef* tabs = reinterpret_cast<HsCoef*>(hs_smem + hbuf_bytes + zp_bytes);                  // [2 parities][rows HS_TH + 2 | columns HS_TW + 2]
    constexpr int TABN = HS_TH + 2 + HS_TW + 2;
    float* hwl = reinterpret_cast<float*>(tabs + 2 * TABN);                                     // fused head: the block's weights [2][HS_CB] (global latency off the epilogue)
    const int tid = threadIdx.x;
    // stage V / epilogue role: HS_VR rows (HS_VR vq ..) of column vx, channels 8 cp8 .. 8 cp8 + 7
    const int cp8 = tid & 7, vx = (tid >> 3) & (HS_TW - 1), vq = tid >> (3 + HS_TWL);
    // stage H role: column hx, channel quad cq, patch rows hi, hi + HS_NT / (16 HS_TW), ..
    const int hx = (tid >> 4) & (HS_TW - 1), cq = tid & 15, hi = tid >> (4 + HS_TWL);
    const int X = X0 + vx;
    const int c0 = cb * HS_CB;
    const int cbn = min(HS_CB, p.C - c0);
    const bool cok = cp8 * 8 < cbn;
    float acc[HS_VR][8];
#pragma unroll
    for (int k = 0; k < HS_VR; ++k) {
        const int Y = Y0 + HS_VR * vq + k;
        if (p.yin && cok && Y < p.H && X < p.W)
            ld8<T>(reinterpret_cast<const T*>(p.yin) + (((int64_t)b * p.H + Y) * p.W + X) * p.yin_cs + c0 + cp8 * 8, acc[k]);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[k][e] = 0.f;
        }
    }

    // ---- per-level state of the prefetcher: this thread's pieces of a round (global element offset without the dy term, LDS offset)
    int goff[NPF], loff[NPF];
    u32x4 pf[NPF];
    const T* zbase = nullptr;
    HsGeo G;
    auto level_setup = [&](int l) {
        const HsLevel L = hs_level(p, l);
        G = hs_geo(p, L, Y0, X0);
        zbase = reinterpret_cast<const T*>(L.z) + (((int64_t)b * L.h + G.ib) * L.w + G.jb) * L.cs + c0;
        const int npieces = G.nr * G.nc * 3 * PPC;
#pragma unroll
        for (int s = 0; s < NPF; ++s) {
            const int e = tid + HS_NT * s;
            const int cp = e % PPC; int q = e / PPC;
            const int dx = q % 3; q /= 3;
            const int j = q % G.nc, i = q / G.nc;
            const bool ok = e < npieces && cp * VE < cbn;
            goff[s] = ok ? (i * L.w + j) * L.cs + dx * p.C + cp * VE : 0;       // (a dropped piece loads the patch's first piece: no branch around the load)
            loff[s] = ok ? ((i * G.nc + j) * 3 + dx) * HS_CB + cp * VE : -1;
        }
    };
    auto tables_write = [&](int l) {          // (threads 0 .. TABN - 1) bil_src of the tile's rows / columns, relative to the patch origin
        const HsLevel L = hs_level(p, l);
        HsCoef* t = tabs + (l & 1) * TABN;
        if (tid < HS_TH + 2) {                                 // row yy <-> full-resolution row max(Y0 + yy - 2, 0)
            HsCoef c; bil_src(clampi(Y0 + tid - 2, 0, p.H - 1), L.R, L.h, p.ac, c.i0, c.i1, c.lam);
            c.i0 -= G.ib; c.i1 -= G.ib; c.pad_ = 0; t[tid] = c;
        } else if (tid >= 64 && tid < 64 + HS_TW + 2) {         // column xx <-> min(X0 + xx, W - 1)
            HsCoef c; bil_src(clampi(X0 + tid - 64, 0, p.W - 1), L.R, L.w, p.ac, c.i0, c.i1, c.lam);
            c.i0 -= G.jb; c.i1 -= G.jb; c.pad_ = 0; t[HS_TH + 2 + tid - 64] = c;
        }
    };
    auto pf_issue = [&](int dy) {
        const T* src = zbase + dy * 3 * p.C;
#pragma unroll
        for (int s = 0; s < NPF; ++s) pf[s] = *reinterpret_cast<const u32x4*>(src + goff[s]);
    };
    auto pf_store = [&]() {
#pragma unroll
        for (int s = 0; s < NPF; ++s)
            if (loff[s] >= 0) {
                if constexpr (sizeof(T) == 2) {
                    float f[8];
                    unpack16<bf16_t>(pf[s], f);
                    *reinterpret_cast<f32x4*>(zp + loff[s]) = *reinterpret_cast<f32x4*>(f);
                    *reinterpret_cast<f32x4*>(zp + loff[s] + 4) = *reinterpret_cast<f32x4*>(f + 4);
                } else *reinterpret_cast<u32x4*>(zp + loff[s]) = pf[s];
            }
    };

    float hwv = 0.f;                                           // fused head: this thread's weight, parked in LDS behind the first patch (one global latency, not two)
    if constexpr (FH) {
        if (tid >= 256 && tid < 256 + 2 * HS_CB) {
            const int o = (tid - 256) / HS_CB, n = (tid - 256) % HS_CB;
            if (o < p.head_co && n < cbn) hwv = p.head_w[o * p.C + c0 + n];
        }
    }
    const int nrounds = 3 * p.nlev;
    level_setup(0);
    int nr = G.nr, nc = G.nc;                                  // geometry of the round being COMPUTED (G runs one round ahead)
    tables_write(0);
    pf_issue(0);
    pf_store();
    if constexpr (FH) { if (tid >= 256 && tid < 256 + 2 * HS_CB) hwl[tid - 256] = hwv; }
    __syncthreads();
#pragma unroll 1
    for (int rr = 0; rr < nrounds; ++rr) {
        const int l = rr / 3, dy = rr - 3 * l;
        const bool more = rr + 1 < nrounds, newlev = more && dy == 2;
        if (newlev) level_setup(l + 1);
#if !(HS_ABLATE & 4)
        if (more) pf_issue(newlev ? 0 : dy + 1);
#endif
        const HsCoef* rowt = tabs + (l & 1) * TABN;
        const HsCoef* colt = rowt + HS_TH + 2;
#if !(HS_ABLATE & 1)
        {   // stage H: hbuf[i][hx][4 cq ..] = sum_dx (1 - lam) z[i][j0(hx + dx)][dx] + lam z[i][j1(hx + dx)][dx]
            int o0[3], o1[3]; f32x2_t w0[3], w1[3];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const HsCoef c = colt[hx + dx];
                o0[dx] = (c.i0 * 3 + dx) * HS_CB + cq * 4; o1[dx] = (c.i1 * 3 + dx) * HS_CB + cq * 4;
                w1[dx].x = c.lam; w1[dx].y = c.lam; w0[dx].x = 1.f - c.lam; w0[dx].y = w0[dx].x;
            }
            for (int i = hi; i < nr; i += HS_NT / (16 * HS_TW)) {
                const float* row = zp + i * nc * 3 * HS_CB;
                f32x2_t vlo = {0.f, 0.f}, vhi = {0.f, 0.f};
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    f32x2_t alo, ahi, blo, bhi;
                    ld4p<float>(row + o0[dx], alo, ahi); ld4p<float>(row + o1[dx], blo, bhi);
                    vlo = w0[dx] * alo + vlo; vhi = w0[dx] * ahi + vhi;
                    vlo = w1[dx] * blo + vlo; vhi = w1[dx] * bhi + vhi;
                }
                f32x4 v; v.x = vlo.x; v.y = vlo.y; v.z = vhi.x; v.w = vhi.y;
                *reinterpret_cast<f32x4*>(hbuf + hbuf_off(i, hx, cq)) = v;
            }
        }
#endif
        __syncthreads();                                       // B: hbuf complete, zp free
        if (more) {
            pf_store();
            if (newlev) tables_write(l + 1);                   // (the other parity: stage V below still reads this level's rows)
        }
#if !(HS_ABLATE & 2)
        {   // stage V: acc[k] += (1 - lam) hbuf[i0(row k + dy)] + lam hbuf[i1(..)] (branch-free: both patch rows are read for every step)
#pragma unroll
            for (int k = 0; k < HS_VR; ++k) {
                const HsCoef c = rowt[HS_VR * vq + k + dy];
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(hbuf + hbuf_off(c.i0, vx, 2 * cp8));
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(hbuf + hbuf_off(c.i0, vx, 2 * cp8 + 1));
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(hbuf + hbuf_off(c.i1, vx, 2 * cp8));
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(hbuf + hbuf_off(c.i1, vx, 2 * cp8 + 1));
                const f32x2_t w1 = {c.lam, c.lam}, w0 = {1.f - c.lam, 1.f - c.lam};
                const f32x2_t r0[4] = {{a0.x, a0.y}, {a0.z, a0.w}, {a1.x, a1.y}, {a1.z, a1.w}};
                const f32x2_t r1[4] = {{b0.x, b0.y}, {b0.z, b0.w}, {b1.x, b1.y}, {b1.z, b1.w}};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f32x2_t a = {acc[k][2 * e], acc[k][2 * e + 1]};
                    a = w0 * r0[e] + a; a = w1 * r1[e] + a;
                    acc[k][2 * e] = a.x; acc[k][2 * e + 1] = a.y;
                }
            }
        }
#endif
        nr = G.nr; nc = G.nc;
        __syncthreads();                                       // A: zp (+ tables) of the next round visible, hbuf free
    }

    // epilogue: eval affine + ReLU, store, train-mode statistics of the fp32 values (as the convolution kernels' epilogues take them)
    float s8[8], q8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s8[e] = 0.f; q8[e] = 0.f; }
    float sc[8], sh[8];
    if (p.scale && cok) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = p.scale[c0 + cp8 * 8 + e]; sh[e] = p.shift[c0 + cp8 * 8 + e]; }
    }
#pragma unroll
    for (int k = 0; k < HS_VR; ++k) {
        const int Y = Y0 + HS_VR * vq + k;
        if (!(cok && Y < p.H && X < p.W)) continue;
        if (p.scale) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[k][e] = acc[k][e] * sc[e] + sh[e];
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[k][e] = fmaxf(acc[k][e], 0.f);
        }
        if (p.y) st8<T>(reinterpret_cast<T*>(p.y) + (((int64_t)b * p.H + Y) * p.W + X) * p.y_cs + c0 + cp8 * 8, acc[k]);
        if constexpr (!FH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { s8[e] += acc[k][e]; q8[e] += acc[k][e] * acc[k][e]; }
        }
    }
    if constexpr (FH) {
        // salt_head1x1 on the values as stored (head1x1_vec_co_kernel: a lane's channels in order, then a butterfly over the lanes of a
        // pixel - here its steps inside the 64-channel block; the steps across channel blocks are head_sum_kernel's, in the same tree order).
        // The loop's last barrier has passed: hbuf is free and stages the block's [class][HS_TH x HS_TW] values for row-wise stores.
#pragma unroll 1
        for (int o = 0; o < p.head_co; ++o) {
            float hw[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) hw[e] = hwl[o * HS_CB + cp8 * 8 + e];
#pragma unroll
            for (int k = 0; k < HS_VR; ++k) {
                float fv[8];
                if constexpr (sizeof(T) == 2) { const u32x4 pk = pack16<bf16_t>(acc[k]); unpack16<bf16_t>(pk, fv); }
                else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) fv[e] = acc[k][e];
                }
                float pl;
                if constexpr (sizeof(T) == 2) pl = head_dot<8>(fv, hw);                          // head1x1's lane = 8 bf16 channels ..
                else pl = __fadd_rn(head_dot<4>(fv, hw), head_dot<4>(fv + 4, hw + 4));            // .. or 4 fp32 channels, two lanes joined by its first butterfly step
                if (!cok) pl = 0.f;
#pragma unroll
                for (int sft = 1; sft < 8; sft <<= 1) pl += __shfl_xor(pl, sft);
                if (cp8 == 0) hbuf[o * (HS_TH * HS_TW) + (HS_VR * vq + k) * HS_TW + vx] = pl;
            }
        }
        __syncthreads();
        const int o = tid >> 9, pix = tid & (HS_TH * HS_TW - 1);
        const int Y = Y0 + (pix >> HS_TWL), Xo = X0 + (pix & (HS_TW - 1));
        if (o < p.head_co && Y < p.H && Xo < p.W) {
            const float v = hbuf[o * (HS_TH * HS_TW) + pix];
            if (p.cblocks == 1) p.head_y[(((int64_t)b * p.head_co + o) * p.H + Y) * p.W + Xo] = v + (p.head_b ? p.head_b[o] : 0.f);
            else p.head_ws[((((int64_t)b * p.cblocks + cb) * p.head_co + o) * p.H + Y) * p.W + Xo] = v;
        }
    }
    if (!FH && p.fin_acc) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) { s8[e] += __shfl_xor(s8[e], o); q8[e] += __shfl_xor(q8[e], o); }
        }
        // (the loop's last barrier: everybody is done with hbuf) reuse it as [waves][2][64]
        const int lane = tid & 63, wv = tid >> 6;
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { hbuf[(wv * 2 + 0) * HS_CB + lane * 8 + e] = s8[e]; hbuf[(wv * 2 + 1) * HS_CB + lane * 8 + e] = q8[e]; }
        }
        __syncthreads();
        if (tid < 2 * HS_CB) {
            const int st = tid / HS_CB, n = tid - st * HS_CB;
            if (n < cbn) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < HS_NT / 64; ++w) t += (double)hbuf[(w * 2 + st) * HS_CB + n];
                fin_add(p.fin_acc + (size_t)(blockIdx.x & 7) * (2 * p.C + 1) + st * p.C + c0 + n, t);
            }
            if (tid == 0 && cb == 0) {
                const int ny = min(HS_TH, p.H - Y0), nx = min(HS_TW, p.W - X0);
                fin_add(p.fin_acc + (size_t)(blockIdx.x & 7) * (2 * p.C + 1) + 2 * p.C, (double)ny * (double)nx);
            }
        }
    }
}

// The fused head's butterfly steps ACROSS channel blocks (head1x1's lane distance 8, 16, 32: block c takes block c ^ 1, the pair takes
// the pair c ^ 2, ..) + bias: logits [B, co, H, W] from the per-block values [B, blocks, co, H, W] hyper_stencil_fwd_kernel<.., true> left.
__global__ __launch_bounds__(256) void head_sum_kernel(const float* __restrict__ ws, const float* __restrict__ bias, float* __restrict__ y, int nblk, int co, int64_t hw, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (b, o, pixel)
    if (i >= total) return;
    const int64_t bo = i / hw, px = i - bo * hw, b = bo / co;
    const int o = (int)(bo - b * co);
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = c < nblk ? ws[((b * nblk + c) * co + o) * hw + px] : 0.f;
    if (nblk > 1) { v[0] += v[1]; v[2] += v[3]; v[4] += v[5]; v[6] += v[7]; }
    if (nblk > 2) { v[0] += v[2]; v[4] += v[6]; }
    if (nblk > 4) v[0] += v[4];
    y[i] = v[0] + (bias ? bias[o] : 0.f);
}

// ------------------------------------------------------------------------------------------ adjoint
constexpr int HB_CB = 64, HB_NT = 1024, HB_UY = 8;      // backward: channel block, threads, padding of the row-weight table (rows in flight per thread: 8 bf16 / 4 fp32)      // backward: channel block, threads, full-resolution rows in flight per thread

struct HsbKP {
    HsLevel lev[4]; int nlev;
    const void* g; int g_cs;
    int B, H, W, C, ac, cblocks, nrows_pad, xt_floats;      // xt_floats: LDS floats reserved for the column-weight table (widest level)
    unsigned nblocks, per_xcd;
};

// weight with which low-resolution index `i` enters full-resolution coordinate d (0: not referenced)
__device__ __forceinline__ float hs_wgt(int d, int R, int n, int ac, int i) {
    int i0, i1; float lam;
    bil_src(d, R, n, ac, i0, i1, lam);
    return (i0 == i ? 1.f - lam : 0.f) + (i1 == i ? lam : 0.f);
}
// conservative range of full-resolution coordinates d (before the tap shift) that reference low-resolution index i
__host__ __device__ __forceinline__ void hs_range(int i, int R, int n, int N, int ac, int& lo, int& hi) {
    float flo, fhi;
    if (ac) {       // src = d (n - 1) / (N - 1) in (i - 1, i + 1); one coordinate of slack for the rounding of the fp32 scale
        const float s = n > 1 ? (float)(N - 1) / (float)(n - 1) : (float)N;
        flo = ((float)i - 1.f) * s - 1.f; fhi = ((float)i + 1.f) * s + 1.f;
    } else { flo = (float)R * ((float)i - 0.5f) - 0.5f; fhi = (float)R * ((float)i + 1.5f) - 0.5f; }      // src = (d + 0.5) / R - 0.5, exact
    lo = (int)floorf(flo); hi = (int)ceilf(fhi);
}

// One workgroup (16 waves) = TI low-resolution rows of one image and level x 64 channels.
// Stage V^T: G[ti][dy][X][c] = sum_Y wy(i0 + ti | max(Y + dy - 2, 0)) g[Y][X][c] - thread = column X (+ 128 per slot) x 8 channels, streaming
// the (TI + 1) R + 2 full-resolution rows that touch the block, HB_UY row loads in flight, wave-uniform weights from an LDS table.
// Stage H^T, per (ti, dy): the G row goes through LDS ([X][64] fp32: a ds_read_b128 of 16 consecutive channel quads is conflict-free
// for every X) and dz[(dy, dx)][i][j][c] = sum_X wx(j | min(X + dx, W - 1)) G[X][c] is gathered by (dx, j, channel quad) items, the X
// range of an item split over 1 / 2 / 4 lanes where a level has few columns.
template <typename T, int NSLOT, int TI, int NDY>
__global__ __launch_bounds__(HB_NT) void hyper_stencil_bwd_kernel(HsbKP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hs_smem[];
    constexpr int NK = TI * 3, UY = (sizeof(T) == 2 ? 8 : 4) / NSLOT;      // full-resolution rows in flight per thread
    const int tid = threadIdx.x;
    // Every level owns a range of workgroup ids that is a multiple of 8, and inside it the ids go round-robin over the XCDs: an XCD
    // takes a contiguous eighth of EVERY level (neighbouring row blocks share their full-resolution rows in its L2, and the long
    // workgroups of the coarse levels - listed first by the host - are spread over all XCDs instead of filling the last ones)
    int l = 0;
    if (p.nlev > 1 && blockIdx.x >= p.lev[1].blk0) l = 1;
    if (p.nlev > 2 && blockIdx.x >= p.lev[2].blk0) l = 2;
    if (p.nlev > 3 && blockIdx.x >= p.lev[3].blk0) l = 3;
    HsLevel L = p.lev[0];
    if (l == 1) L = p.lev[1];
    if (l == 2) L = p.lev[2];
    if (l == 3) L = p.lev[3];
    const unsigned bl = blockIdx.x - L.blk0, per = (L.nblk + 7u) >> 3;
    unsigned r = (bl & 7u) * per + (bl >> 3);
    if (r >= L.nblk) return;
    const int cb = (int)(r % (unsigned)p.cblocks); r /= (unsigned)p.cblocks;
    const int nib = (L.h + TI - 1) / TI;
    const int i0 = (int)(r % (unsigned)nib) * TI;
    const int b = (int)(r / (unsigned)nib);
    const int c0 = cb * HB_CB, cbn = min(HB_CB, p.C - c0);
    float* Gs = reinterpret_cast<float*>(hs_smem);                                              // [NDY][W][64]
    float* xt = Gs + (size_t)NDY * p.W * HB_CB;                                                 // [3 dx][w][xlen] column weights, then [3 dx][w] first columns
    float* wtab = xt + p.xt_floats;                                                             // [nrows_pad][8] row weights
    int* xs0 = reinterpret_cast<int*>(xt + 3 * L.w * L.xlen);

    int ylo, yhi, t0, t1;
    hs_range(i0, L.R, L.h, p.H, p.ac, ylo, t0);
    hs_range(min(i0 + TI - 1, L.h - 1), L.R, L.h, p.H, p.ac, t1, yhi);
    yhi += 2;                                                  // Y = Y' + 2 - dy
    ylo = max(ylo, 0); yhi = min(yhi, p.H - 1);
    const int nrows = min(yhi - ylo + 1, p.nrows_pad);
    // column weights of every gather item (dx, j): wx(j | min(X + dx, W - 1)) for X = xs0 .. xs0 + xlen - 1 (0 past the item's range)
    for (int e = tid; e < 3 * L.w * L.xlen; e += HB_NT) {
        const int n = e % L.xlen, pj = e / L.xlen, dx = pj / L.w, j = pj - dx * L.w;
        int xlo, xhi;
        hs_range(j, L.R, L.w, p.W, p.ac, xlo, xhi);
        xlo = max(xlo - dx, 0); xhi = min(xhi, p.W - 1);
        const int X = xlo + n;
        xt[e] = X <= xhi ? hs_wgt(min(X + dx, p.W - 1), L.R, L.w, p.ac, j) : 0.f;
        if (n == 0) xs0[pj] = xlo;
    }
    for (int e = tid; e < p.nrows_pad * 8; e += HB_NT) {
        const int yr = e >> 3, k = e & 7, ti = k / 3, dy = k - 3 * ti;
        float w = 0.f;
        if (k < NK && yr < nrows && i0 + ti < L.h) w = hs_wgt(max(ylo + yr + dy - 2, 0), L.R, L.h, p.ac, i0 + ti);
        wtab[e] = w;
    }
    __syncthreads();

    // ---- stage V^T
    const int cp8 = tid & 7, x = tid >> 3;
    const bool cok = cp8 * 8 < cbn;
    f32x2_t G[NSLOT][NK][4];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s)
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) G[s][k][e] = f32x2_t{0.f, 0.f};
    // uniform base of the block's first row + a 32-bit per-thread byte offset; a lane without work re-reads the block's first piece
    const char* gb = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.g) + ((int64_t)b * p.H + ylo) * p.W * p.g_cs + c0);
    const unsigned rowb = (unsigned)p.W * (unsigned)p.g_cs * (unsigned)sizeof(T);
    unsigned voff[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) voff[s] = (cok && x + 128 * s < p.W) ? ((unsigned)(x + 128 * s) * (unsigned)p.g_cs + cp8 * 8) * (unsigned)sizeof(T) : 0u;
#pragma unroll 1
    for (int yb = 0; yb < nrows; yb += UY) {
        u32x4 raw[UY][NSLOT][sizeof(T) == 2 ? 1 : 2];
#pragma unroll
        for (int u = 0; u < UY; ++u) {
            const char* rp = gb + (size_t)min(yb + u, nrows - 1) * rowb;      // (uniform)
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                raw[u][s][0] = *reinterpret_cast<const u32x4*>(rp + voff[s]);
                if (sizeof(T) == 4) raw[u][s][sizeof(T) == 2 ? 0 : 1] = *reinterpret_cast<const u32x4*>(rp + voff[s] + 16);
            }
        }
#pragma unroll
        for (int u = 0; u < UY; ++u) {
            const f32x4 wa = *reinterpret_cast<const f32x4*>(wtab + (yb + u) * 8);      // (rows past nrows: zero weights)
            const f32x4 wb = *reinterpret_cast<const f32x4*>(wtab + (yb + u) * 8 + 4);
            const float w[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
            for (int s = 0; s < NSLOT; ++s) {
                f32x2_t v[4];
                if (sizeof(T) == 2) {
                    const u32x4 q = raw[u][s][0];
                    v[0] = f32x2_t{__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u)};
                    v[1] = f32x2_t{__uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
                    v[2] = f32x2_t{__uint_as_float(q.z << 16), __uint_as_float(q.z & 0xffff0000u)};
                    v[3] = f32x2_t{__uint_as_float(q.w << 16), __uint_as_float(q.w & 0xffff0000u)};
                } else {
                    const u32x4 q = raw[u][s][0], q2 = raw[u][s][sizeof(T) == 2 ? 0 : 1];
                    v[0] = f32x2_t{__uint_as_float(q.x), __uint_as_float(q.y)}; v[1] = f32x2_t{__uint_as_float(q.z), __uint_as_float(q.w)};
                    v[2] = f32x2_t{__uint_as_float(q2.x), __uint_as_float(q2.y)}; v[3] = f32x2_t{__uint_as_float(q2.z), __uint_as_float(q2.w)};
                }
#pragma unroll
                for (int k = 0; k < NK; ++k) {
                    const f32x2_t wk = {w[k], w[k]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) G[s][k][e] = __builtin_elementwise_fma(wk, v[e], G[s][k][e]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);                 // (keep the rows' unpacked values from being live all at once: 128 registers)
        }
    }

    // ---- stage H^T: NDY rows of G (all kernel rows dy of one low-resolution row ti, or one at a time for wide images) go through LDS
    const int lg = (int)L.blk_lg, nsp = 1 << lg;
    const int q16 = tid & 15, part = (tid >> 4) & (nsp - 1), pair0 = tid >> (4 + lg), ipp = HB_NT >> (4 + lg);
    const int npairs = 3 * L.w;                                // items (j, dx) x channel quad, each for all NDY kernel rows: one weight read feeds NDY G reads
    const int plen = (L.xlen + nsp - 1) >> lg, n0 = part * plen, n1 = min(n0 + plen, L.xlen);
#pragma unroll
    for (int kc = 0; kc < NK / NDY; ++kc) {
        const int ti = (kc * NDY) / 3, dy0 = kc * NDY - 3 * ti;
        if (i0 + ti < L.h) {                                   // (uniform)
            if (kc) __syncthreads();                           // the previous chunk's gathers are done
#pragma unroll
            for (int d = 0; d < NDY; ++d)
#pragma unroll
                for (int s = 0; s < NSLOT; ++s) {
                    const int X = x + 128 * s;
                    if (X < p.W) {
                        float* dst = Gs + ((size_t)d * p.W + X) * HB_CB + cp8 * 8;
                        const f32x2_t* g = G[s][kc * NDY + d];
                        *reinterpret_cast<f32x4*>(dst) = f32x4{g[0].x, g[0].y, g[1].x, g[1].y};
                        *reinterpret_cast<f32x4*>(dst + 4) = f32x4{g[2].x, g[2].y, g[3].x, g[3].y};
                    }
                }
            __syncthreads();
            T* zb = reinterpret_cast<T*>(L.z) + (((int64_t)b * L.h + i0 + ti) * L.w) * L.cs + c0 + q16 * 4;
            for (int pr = pair0; pr - pair0 < npairs; pr += ipp) {     // (every lane runs the same number of passes: the shuffles below need all of them)
                const bool valid = pr < npairs;
                const int prv = valid ? pr : 0;
                const int j = prv / 3, dx = prv - 3 * j;
                const int pj = dx * L.w + j;
                const float* wrow = xt + pj * L.xlen;
                const int xs = xs0[pj], nmax = p.W - 1 - xs;       // (weights past the item's range are 0; the row address is clamped)
                const float* gs = Gs + (size_t)xs * HB_CB + q16 * 4;
                f32x2_t alo[NDY], ahi[NDY];
#pragma unroll
                for (int d = 0; d < NDY; ++d) { alo[d] = f32x2_t{0.f, 0.f}; ahi[d] = f32x2_t{0.f, 0.f}; }
                for (int n = n0; n < n1; ++n) {
                    const float wx = wrow[n];
                    const f32x2_t w2 = {wx, wx};
                    const float* gp = gs + (size_t)min(n, nmax) * HB_CB;
#pragma unroll
                    for (int d = 0; d < NDY; ++d) {
                        const f32x4 gv = *reinterpret_cast<const f32x4*>(gp + (size_t)d * p.W * HB_CB);
                        alo[d] = __builtin_elementwise_fma(w2, f32x2_t{gv.x, gv.y}, alo[d]);
                        ahi[d] = __builtin_elementwise_fma(w2, f32x2_t{gv.z, gv.w}, ahi[d]);
                    }
                }
#pragma unroll
                for (int d = 0; d < NDY; ++d) {
                    for (int o = 16; o < (16 << lg); o <<= 1) {
                        alo[d].x += __shfl_xor(alo[d].x, o); alo[d].y += __shfl_xor(alo[d].y, o);
                        ahi[d].x += __shfl_xor(ahi[d].x, o); ahi[d].y += __shfl_xor(ahi[d].y, o);
                    }
                    if (valid && part == 0 && q16 * 4 < cbn) {
                        T* dst = zb + (int64_t)j * L.cs + ((dy0 + d) * 3 + dx) * p.C;
                        if (sizeof(T) == 2) {
                            uint2 o; o.x = f2bf_pk(alo[d].x, alo[d].y); o.y = f2bf_pk(ahi[d].x, ahi[d].y);
                            *reinterpret_cast<uint2*>(dst) = o;
                        } else {
                            *reinterpret_cast<f32x4*>(dst) = f32x4{alo[d].x, alo[d].y, ahi[d].x, ahi[d].y};
                        }
                    }
                }
            }
        }
    }
}

static int hs_check(const salt_hyper_stencil_args* a) {
    if (!a || a->nlev < 1 || a->nlev > 4) SALT_FAIL(SALT_E_BADARG, "hyper_stencil: bad args");
    {
        salt_view yv = a->y;
        if (!yv.p && a->head_y_nchw && !a->backward) yv.p = const_cast<float*>(a->head_w);      // (the fused head may drop y: only its shape is used)
        if (!view_ok(yv)) SALT_FAIL(SALT_E_BADARG, "hyper_stencil: bad y view");
    }
    if (a->head_y_nchw) {
        const int ncb = (a->y.C + 63) / 64;
        // (the shapes salt_head1x1's vector kernels take: C a power of two, at most 64 lanes of 16 bytes per pixel - their summation tree is the one reproduced)
        if (a->backward || a->fin_acc || !a->head_w || a->head_cout < 1 || a->head_cout > 2 || a->y.C % 64 || (ncb != 1 && ncb != 2 && ncb != 4 && ncb != 8) ||
            a->y.C > (a->dtype == SALT_F32 ? 256 : 512) || (ncb > 1 && !a->head_ws))
            SALT_FAIL(SALT_E_UNSUPPORTED, "hyper_stencil: the fused head is an eval-mode epilogue for head_cout 1..2 and C = 64, 128, 256 (bf16: 512)");
    }
    const int es = a->dtype == SALT_F32 ? 4 : 2;
    if (a->dtype != SALT_F32 && a->dtype != SALT_BF16) SALT_FAIL(SALT_E_BADARG, "hyper_stencil: dtype %d", a->dtype);
    const salt_view& y = a->y;
    if (y.C % 8 || y.cs % 8 || (reinterpret_cast<uintptr_t>(y.p) & 15)) SALT_FAIL(SALT_E_UNSUPPORTED, "hyper_stencil: y needs C and cs multiples of 8 and a 16-byte aligned pointer");
    if (a->y_in.p) {
        const salt_view& v = a->y_in;
        if (v.B != y.B || v.H != y.H || v.W != y.W || v.C != y.C || v.cs % 8 || (reinterpret_cast<uintptr_t>(v.p) & 15)) SALT_FAIL(SALT_E_BADARG, "hyper_stencil: y_in does not match y");
    }
    for (int k = 0; k < a->nlev; ++k) {
        const salt_view& z = a->z[k];
        const int R = a->R[k];
        if (!view_ok(z) || R < 4 || R > 32 || (R & (R - 1)) || z.B != y.B || z.H * R != y.H || z.W * R != y.W || z.C != 9 * y.C || z.cs % 8 ||
            (reinterpret_cast<uintptr_t>(z.p) & 15))
            SALT_FAIL(SALT_E_BADARG, "hyper_stencil: level %d: z must be [B, H / R, W / R, 9 C] with R a power of two in 4..32, cs %% 8 == 0, 16-byte aligned", k);
        if ((int64_t)z.B * z.H * z.W * z.cs * es >= (1LL << 40)) SALT_FAIL(SALT_E_UNSUPPORTED, "hyper_stencil: level too large");
    }
    return SALT_OK;
}

}  // namespace

extern "C" int salt_hyper_stencil(const salt_hyper_stencil_args* a, void* stream) {
    int rc = hs_check(a);
    if (rc) return rc;
    const salt_view& y = a->y;
    const hipStream_t st = (hipStream_t)stream;
    const size_t es = a->dtype == SALT_F32 ? 4 : 2;
    if (!a->backward) {
        HsKP p;
        p.nlev = a->nlev;
        int nr_max = 1, nc_max = 1;
        for (int k = 0; k < a->nlev; ++k) {
            p.lev[k].z = a->z[k].p; p.lev[k].h = a->z[k].H; p.lev[k].w = a->z[k].W; p.lev[k].cs = a->z[k].cs; p.lev[k].R = a->R[k]; p.lev[k].blk0 = 0; p.lev[k].nblk = 0; p.lev[k].blk_lg = 0; p.lev[k].xlen = 0;
            const int nr = (HS_TH + 1) / a->R[k] + 3, nc = (HS_TW + 1) / a->R[k] + 3;      // low-resolution rows / columns a (TH + 2) / (TW + 2)-wide window can touch
            nr_max = nr > nr_max ? nr : nr_max; nc_max = nc > nc_max ? nc : nc_max;
            if ((int64_t)a->z[k].H * a->z[k].W * a->z[k].cs >= (1LL << 31)) SALT_FAIL(SALT_E_UNSUPPORTED, "hyper_stencil: level image too large for 32-bit offsets");
        }
        p.yin = a->y_in.p; p.yin_cs = a->y_in.cs; p.y = y.p; p.y_cs = y.cs;
        p.B = y.B; p.H = y.H; p.W = y.W; p.C = y.C; p.ac = a->align_corners;
        p.scale = a->scale; p.shift = a->shift; p.relu = a->relu; p.fin_acc = a->fin_acc;
        if ((a->scale == nullptr) != (a->shift == nullptr)) SALT_FAIL(SALT_E_BADARG, "hyper_stencil: scale and shift come together");
        p.tiles_x = cdiv(y.W, HS_TW); p.tiles_y = cdiv(y.H, HS_TH); p.cblocks = cdiv(y.C, HS_CB); p.nr_max = nr_max; p.nc_max = nc_max;
        p.head_w = a->head_w; p.head_b = a->head_b; p.head_y = a->head_y_nchw; p.head_co = a->head_cout;
        p.head_ws = a->head_ws;
        const int64_t nb = (int64_t)y.B * p.tiles_x * p.tiles_y * p.cblocks;
        if (nb >= (1LL << 30)) SALT_FAIL(SALT_E_UNSUPPORTED, "hyper_stencil: too many tiles");
        p.nblocks = (unsigned)nb; p.per_xcd = (unsigned)((nb + 7) / 8);
        const size_t lds = (size_t)nr_max * HS_TW * HS_CB * 4 + (size_t)nr_max * nc_max * 3 * HS_CB * 4 + 2 * (HS_TH + 2 + HS_TW + 2) * sizeof(HsCoef) + 2 * HS_CB * 4;
        if (lds > 160 * 1024) SALT_FAIL(SALT_E_LDS, "hyper_stencil: needs %zu bytes of LDS", lds);
        const int npf = cdiv(nr_max * nc_max * 3 * (HS_CB * (int)es / 16), HS_NT);      // 16-byte pieces of a round per thread
        void (*kern)(HsKP) = nullptr;
        if (a->dtype == SALT_F32) kern = npf <= 4 ? (p.head_y ? hyper_stencil_fwd_kernel<float, 4, true> : hyper_stencil_fwd_kernel<float, 4, false>) : nullptr;
        else kern = npf <= 2 ? (p.head_y ? hyper_stencil_fwd_kernel<bf16_t, 2, true> : hyper_stencil_fwd_kernel<bf16_t, 2, false>) : nullptr;
        if (!kern) SALT_FAIL(SALT_E_UNSUPPORTED, "hyper_stencil: %d pieces per thread", npf);
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(kern, dim3(p.per_xcd * 8), dim3(HS_NT), lds, st, p);
        SALT_CHECK_LAUNCH();
        if (p.head_y && p.cblocks > 1) {
            const int64_t hw = (int64_t)y.H * y.W, total = (int64_t)y.B * p.head_co * hw;
            hipLaunchKernelGGL(head_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p.head_ws, p.head_b, p.head_y, p.cblocks, p.head_co, hw, total);
            SALT_CHECK_LAUNCH();
        }
        return SALT_OK;
    }
    if (a->scale || a->fin_acc) SALT_FAIL(SALT_E_BADARG, "hyper_stencil: the adjoint has no epilogue");
    if (y.W > 256) SALT_FAIL(SALT_E_UNSUPPORTED, "hyper_stencil: the adjoint handles W <= 256 (got %d)", y.W);
    if ((int64_t)y.H * y.W * y.cs >= (1LL << 31)) SALT_FAIL(SALT_E_UNSUPPORTED, "hyper_stencil: image too large for 32-bit offsets");
    HsbKP p;
    p.nlev = a->nlev;
    p.g = y.p; p.g_cs = y.cs; p.B = y.B; p.H = y.H; p.W = y.W; p.C = y.C; p.ac = a->align_corners; p.cblocks = cdiv(y.C, HB_CB);
    const int nslot = cdiv(y.W, 128), ti = nslot == 1 ? 2 : 1;
    int64_t nb = 0;
    int nrows = 1, xt_floats = 0;
    int order[4] = {0, 1, 2, 3};                                // coarsest level first: its workgroups stream the most rows
    for (int i = 0; i < a->nlev; ++i)
        for (int j = i + 1; j < a->nlev; ++j)
            if (a->R[order[j]] > a->R[order[i]]) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    for (int q = 0; q < a->nlev; ++q) {
        const int k = order[q];
        HsLevel& L = p.lev[q];
        L.z = a->z[k].p; L.h = a->z[k].H; L.w = a->z[k].W; L.cs = a->z[k].cs; L.R = a->R[k]; L.blk0 = (unsigned)nb;
        int lg = 0;
        while (lg < 2 && 16 * 3 * a->z[k].W * (2 << lg) <= HB_NT) ++lg;
        L.blk_lg = lg;
        const int64_t nl = (int64_t)y.B * cdiv(a->z[k].H, ti) * p.cblocks;
        if (nl >= (1LL << 28)) SALT_FAIL(SALT_E_UNSUPPORTED, "hyper_stencil: too many rows");
        L.nblk = (unsigned)nl;
        nb += (nl + 7) / 8 * 8;
        // rows a block of ti low-resolution rows can reference (hs_range; align_corners: (H - 1) / (h - 1) > R rows per step), + 2 for the tap shift
        const int step = a->align_corners ? (a->z[k].H > 1 ? cdiv(y.H - 1, a->z[k].H - 1) : y.H) : a->R[k];
        int nr = (ti + 1) * step + 8;
        // ... and the EXACT span of every row block with the kernel's own hs_range (if the bound above ever fell short of it the
        // kernel's min(span, nrows_pad) would drop rows silently) - the larger of the two sizes the row-weight table
        for (int i0 = 0; i0 < a->z[k].H; i0 += ti) {
            int ylo, yhi, t0, t1;
            hs_range(i0, a->R[k], a->z[k].H, y.H, a->align_corners, ylo, t0);
            hs_range(i0 + ti - 1 < a->z[k].H - 1 ? i0 + ti - 1 : a->z[k].H - 1, a->R[k], a->z[k].H, y.H, a->align_corners, t1, yhi);
            yhi += 2;
            ylo = ylo > 0 ? ylo : 0; yhi = yhi < y.H - 1 ? yhi : y.H - 1;
            if (yhi - ylo + 1 > nr) nr = yhi - ylo + 1;
        }
        const int stepx = a->align_corners ? (a->z[k].W > 1 ? cdiv(y.W - 1, a->z[k].W - 1) : y.W) : a->R[k];
        L.xlen = 2 * stepx + 8 < y.W ? 2 * stepx + 8 : y.W;      // columns a gather range can span (hs_range + the tap shift)
        const int xtf = 3 * a->z[k].W * (L.xlen + 1);
        xt_floats = xtf > xt_floats ? xtf : xt_floats;
        nrows = nr > nrows ? nr : nrows;
        if ((int64_t)a->z[k].H * a->z[k].W * a->z[k].cs >= (1LL << 31)) SALT_FAIL(SALT_E_UNSUPPORTED, "hyper_stencil: level image too large for 32-bit offsets");
    }
    p.nrows_pad = cdiv(nrows, HB_UY) * HB_UY;
    p.xt_floats = (xt_floats + 3) & ~3;
    if (nb >= (1LL << 30)) SALT_FAIL(SALT_E_UNSUPPORTED, "hyper_stencil: too many rows");
    p.nblocks = (unsigned)nb; p.per_xcd = (unsigned)(nb / 8);
    const size_t lds = (size_t)(nslot == 1 ? 3 : 1) * y.W * HB_CB * 4 + (size_t)p.xt_floats * 4 + (size_t)p.nrows_pad * 8 * 4;
    if (lds > 160 * 1024) SALT_FAIL(SALT_E_LDS, "hyper_stencil: the adjoint needs %zu bytes of LDS", lds);
    void (*kern)(HsbKP) = nullptr;
    if (a->dtype == SALT_F32) kern = nslot == 1 ? hyper_stencil_bwd_kernel<float, 1, 2, 3> : hyper_stencil_bwd_kernel<float, 2, 1, 1>;
    else kern = nslot == 1 ? hyper_stencil_bwd_kernel<bf16_t, 1, 2, 3> : hyper_stencil_bwd_kernel<bf16_t, 2, 1, 1>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(p.per_xcd * 8), dim3(HB_NT), lds, st, p);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}
