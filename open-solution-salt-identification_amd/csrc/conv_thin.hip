// conv_thin_kernel: fp32 3x3 convolutions with 16 or 32 channels on both sides - the first / last levels of the vanilla U-Net
// (BASELINE C0 / C1: ConvBnRelu 16 -> 16, 32 -> 16, 16 -> 32 at 128 x 128, 32 -> 32 at 64 x 64, unet_models.py:21-30) and their data
// gradients.  On conv_mfma_kernel these launches are 4096 workgroups of one 128 x 32 tile with a SINGLE channel chunk each: nothing
// inside a workgroup overlaps (load -> barrier -> 72 MFMAs -> staged epilogue), half of the 32-wide MFMA tile computes padding
// channels, and every tile stages its own copy of the weights: 85 us for 2.4 GFLOP / 67 MB (16 -> 16 at 128 x 128, B = 32).
//   * weights of all taps stay in LDS for the life of a PERSISTENT workgroup (9 - 37 KB), 16 x 16-pixel output tiles stream through a
//     double-buffered halo (18 x 18 rows of 64 bytes per 16-channel chunk) filled by LDS-DMA (global_load_lds_dwordx4, XOR slot swizzle
//     on the source address as in conv_ws_kernel) one tile ahead;
//   * v_mfma_f32_16x16x4_f32 with the operands swapped (D = W X^T): a 16-channel block is ONE MFMA row block - no padding channels -
//     and a lane ends up with 4 consecutive channels of one pixel, i.e. one 16-byte NHWC piece: stores go straight to HBM, no LDS
//     staging.  A ds_read_b128 feeds four MFMAs (lane group g holds channels 4 g .. 4 g + 3 of the chunk; the contraction order inside
//     a chunk is permuted the same way on both operands);
//   * epilogue modes as conv_ws_kernel: 0 bias / folded BN / ReLU / (+)=, 1 train-mode BatchNorm statistics (fp64 shards, ONE set of
//     atomics per workgroup), 2 (+)= and the BatchNorm-backward sums of the stored gradient.
#include "common.h"

#include <cstdlib>

namespace {

struct ThKP {
    const float* x; const float* w; float* y;
    const float* bias; const float* scale; const float* shift;
    int B, H, W, x_cs, y_cs, OH, OW;
    int yH, yW, out_step;                // y is [B,yH,yW,*]; grid pixel (oy, ox) of phase (py, px) -> y pixel (oy out_step + py, ox out_step + px)
    int tiles_x, tiles_y, ntiles;
    int min_dy, min_dx, pad_mode;
    int tap_off[9];
    int relu, accumulate;
    const float* bnb_y; const float* bnb_a; int bnb_cs, bnb_acs, bnb_relu;
    const float* bnb_mean; const float* bnb_invstd; const float* bnb_gamma; const float* bnb_beta;
    double* fin_acc; double* bnb_acc;
};

__device__ __attribute__((aligned(16))) unsigned int g_thin_zero[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ int th_swz(int row, int slot) { return row * 64 + (((slot ^ (row >> 2)) & 3) << 4); }

// CI / CO: 16-channel chunks of the input / blocks of the output; NT taps; PH = 4: the phase-fused launch of a stride-2 transposed
// convolution (salt_conv_args.nphase: four output-parity phases with a packed weight block each, out_step 2) - a tile's halo is staged
// once and serves its four phases
template <int CI, int CO, int MODE, int NT = 9, int PH = 1>
__global__ __launch_bounds__(256) void conv_thin_kernel(ThKP p) {
    constexpr int BN = 16 * CO;
    constexpr int WP = PH * CI * NT * CO;                // weight DMA pieces (1 KB = 16 rows of 64 bytes): rows (phase, chunk, tap, n)
    constexpr int HPC = 21;                              // halo pieces per chunk: 18 x 18 = 324 rows, padded to 336
    constexpr int HP = CI * HPC;
    constexpr int NSW = (WP + 3) / 4, NSH = (HP + 3) / 4;
    constexpr int W_BYTES = WP * 1024, HC_BYTES = HPC * 1024, H_BYTES = HP * 1024;
    constexpr int OFF_H = W_BYTES, OFF_DUMMY = OFF_H + 2 * H_BYTES, OFF_CONST = OFF_DUMMY + 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int n_my = ((int)blockIdx.x < p.ntiles) ? (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (n_my <= 0) return;
    struct TC { int b, oy0, ox0; };
    auto coords = [&](int k) {
        const int t = (int)blockIdx.x + k * (int)gridDim.x;
        TC c; const int tx = t % p.tiles_x; const int r = t / p.tiles_x;
        c.ox0 = tx << 4; c.oy0 = (r % p.tiles_y) << 4; c.b = r / p.tiles_y; return c;
    };
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_thin_zero);
    auto dma = [&](const void* src, int dst) {
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(smem + dst), 16, 0, 0);
    };
    // ---- weights: the packed [chunk][tap][Cout][16] rows are the LDS rows (Cout == BN)
#pragma unroll
    for (int i = 0; i < NSW; ++i) {
        const int q = wave + 4 * i;
        const bool real = q < WP;
        const int R = q * 16 + (lane >> 2);
        const int slot = (lane ^ (R >> 2)) & 3;
        dma(real ? reinterpret_cast<const unsigned char*>(p.w + (R * 16 + slot * 4)) : zp, real ? q * 1024 : OFF_DUMMY);
    }
    auto issue_halo = [&](const TC& c, int buf) {
        int ln = lane;
        asm volatile("" : "+v"(ln));                                        // keeps the index math inside the tile loop
        const float* xb = p.x + (int64_t)c.b * p.H * p.W * p.x_cs;
        const int iy0 = c.oy0 + p.min_dy, ix0 = c.ox0 + p.min_dx;
        const bool clamp = p.pad_mode != 0;
#pragma unroll
        for (int i = 0; i < NSH; ++i) {
            const int pidx = wave + 4 * i;
            const bool real = pidx < HP;
            const int ch = pidx / HPC;                                      // (constant-folded for CI == 1)
            const int row = (pidx - ch * HPC) * 16 + (ln >> 2);
            const int hy = (int)__umulhi((unsigned)row, 238609295u);        // row / 18
            const int hx = row - hy * 18;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
            const bool inside = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            const bool valid = real & (row < 324) & (clamp | inside);
            const int slot = (ln ^ (row >> 2)) & 3;
            const unsigned char* src = reinterpret_cast<const unsigned char*>(xb + ((iyc * p.W + ixc) * p.x_cs + ch * 16 + slot * 4));
            dma(valid ? src : zp, real ? OFF_H + buf * H_BYTES + pidx * 1024 : OFF_DUMMY);
        }
    };
    issue_halo(coords(0), 0);
    if (tid < BN) {
        float* sc = reinterpret_cast<float*>(smem + OFF_CONST);
        if (MODE == 2) {
            if (p.bnb_acc) {
                const float mu = p.bnb_mean[tid], is = p.bnb_invstd[tid], k = p.bnb_gamma[tid] * is;
                sc[tid] = mu; sc[BN + tid] = is; sc[2 * BN + tid] = k; sc[3 * BN + tid] = p.bnb_beta[tid] - mu * k;
            }
        } else {
            sc[tid] = p.bias ? p.bias[tid] : 0.f; sc[BN + tid] = p.scale ? p.scale[tid] : 1.f; sc[2 * BN + tid] = p.shift ? p.shift[tid] : 0.f;
        }
    }
    const bool has_affine = p.bias || p.scale || p.shift || p.relu;
    const bool sums = (MODE == 1 && p.fin_acc) || (MODE == 2 && p.bnb_acc);
    float rs0[CO][4], rs1[CO][4];
#pragma unroll
    for (int cb = 0; cb < CO; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) { rs0[cb][e] = 0.f; rs1[cb][e] = 0.f; }
    const float* cst = reinterpret_cast<const float*>(smem + OFF_CONST);

#pragma unroll 1
    for (int k = 0; k < n_my; ++k) {
        if (k == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the weights, this wave's pieces of tile 0, the constants
        __builtin_amdgcn_s_barrier();                                        // tile k landed for everybody (each wave waited for its own pieces
        asm volatile("" ::: "memory");                                       // before its last epilogue); everybody is done reading the other buffer
        const TC cur = coords(k);
        if (k + 1 < n_my) issue_halo(coords(k + 1), (k + 1) & 1);
        const unsigned char* hb = smem + OFF_H + (k & 1) * H_BYTES;
#pragma unroll 1
        for (int ph = 0; ph < PH; ++ph) {
        // ---- MFMA phase: this wave's 4 tile rows (one 16-pixel block each) x CO channel blocks
        f32x4 acc[CO][4];
#pragma unroll
        for (int cb = 0; cb < CO; ++cb)
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};
        int brow[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) brow[pb] = (wave * 4 + pb) * 18 + l15;
        struct Frag { f32x4 a[CO], b[4]; };
        const int wrow0 = ph * (CI * NT * BN);                               // this phase's weight rows
        auto load_frag = [&](int s, Frag& f) {                               // s = (chunk, tap), a constant after unrolling
            const int c = s / NT, t = s - c * NT;
#pragma unroll
            for (int cb = 0; cb < CO; ++cb) f.a[cb] = *reinterpret_cast<const f32x4*>(smem + th_swz(wrow0 + s * BN + cb * 16 + l15, kg));
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) f.b[pb] = *reinterpret_cast<const f32x4*>(hb + c * HC_BYTES + th_swz(brow[pb] + p.tap_off[t], kg));
        };
        auto mma_frag = [&](const Frag& f) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int cb = 0; cb < CO; ++cb)
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb) acc[cb][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[cb][e], f.b[pb][e], acc[cb][pb], 0, 0, 0);
        };
        constexpr int NST = CI * NT;
        Frag f[2];
        load_frag(0, f[0]);
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            if (s + 1 < NST) load_frag(s + 1, f[(s + 1) & 1]);
            mma_frag(f[s & 1]);
        }
        // the next tile's pieces of this wave had the whole MFMA phase to land; waiting HERE keeps this tile's stores out of the wait
        if (ph == 0 && k + 1 < n_my) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int py = PH > 1 ? ph >> 1 : 0, px = PH > 1 ? ph & 1 : 0;
        // ---- epilogue: lane = pixel l15 of tile row 4 wave + pb, channels 16 cb + 4 kg .. + 3: one 16-byte piece
#pragma unroll
        for (int cb = 0; cb < CO; ++cb) {
            const int ch0 = 16 * cb + 4 * kg;
            f32x4 c0 = *reinterpret_cast<const f32x4*>(cst + ch0), c1 = *reinterpret_cast<const f32x4*>(cst + BN + ch0),
                  c2 = *reinterpret_cast<const f32x4*>(cst + 2 * BN + ch0), c3 = *reinterpret_cast<const f32x4*>(cst + 3 * BN + ch0);
            f32x4 oldv[4], yv[4], av[4];
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const unsigned pix = (unsigned)((cur.b * p.yH + (cur.oy0 + wave * 4 + pb) * p.out_step + py) * p.yW + (cur.ox0 + l15) * p.out_step + px);
                if (MODE != 1 && p.accumulate) oldv[pb] = *reinterpret_cast<const f32x4*>(p.y + (pix * (unsigned)p.y_cs + ch0));
                if (MODE == 2 && sums) {
                    yv[pb] = *reinterpret_cast<const f32x4*>(p.bnb_y + (pix * (unsigned)p.bnb_cs + ch0));
                    if (p.bnb_a) av[pb] = *reinterpret_cast<const f32x4*>(p.bnb_a + (pix * (unsigned)p.bnb_acs + ch0));
                }
            }
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const unsigned pix = (unsigned)((cur.b * p.yH + (cur.oy0 + wave * 4 + pb) * p.out_step + py) * p.yW + (cur.ox0 + l15) * p.out_step + px);
                f32x4 v = acc[cb][pb];
                if (MODE != 2 && has_affine) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = (v[e] + c0[e]) * c1[e] + c2[e];
                        if (p.relu) t = fmaxf(t, 0.f);
                        v[e] = t;
                    }
                }
                if (MODE != 1 && p.accumulate) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += oldv[pb][e];
                }
                *reinterpret_cast<f32x4*>(p.y + (pix * (unsigned)p.y_cs + ch0)) = v;
                if (MODE == 1 && sums) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { rs0[cb][e] += v[e]; rs1[cb][e] += v[e] * v[e]; }
                }
                if (MODE == 2 && sums) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool on = !p.bnb_relu || (p.bnb_a ? av[pb][e] > 0.f : yv[pb][e] * c2[e] + c3[e] > 0.f);
                        const float gg = on ? v[e] : 0.f;
                        rs0[cb][e] += gg; rs1[cb][e] += gg * (yv[pb][e] - c0[e]) * c1[e];
                    }
                }
            }
        }
        }                                                                     // phases
    }
    // ---- per-workgroup sums -> fp64 shard atomics: the 16 lanes of a channel group, then the 4 waves through LDS (fixed order)
    if (MODE != 0 && sums) {
#pragma unroll
        for (int cb = 0; cb < CO; ++cb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { rs0[cb][e] += __shfl_xor(rs0[cb][e], o); rs1[cb][e] += __shfl_xor(rs1[cb][e], o); }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                      // every wave is done with the LDS
        float* red = reinterpret_cast<float*>(smem);                          // [4 waves][2][BN]
        if (l15 == 0) {
#pragma unroll
            for (int cb = 0; cb < CO; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    red[(wave * 2 + 0) * BN + 16 * cb + 4 * kg + e] = rs0[cb][e];
                    red[(wave * 2 + 1) * BN + 16 * cb + 4 * kg + e] = rs1[cb][e];
                }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int st = tid / BN, n = tid - st * BN;
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) t += (double)red[(w * 2 + st) * BN + n];
            if (MODE == 1) {
                double* a = p.fin_acc + (blockIdx.x & 7) * (2 * BN + 1);
                fin_add(a + st * BN + n, t);
                if (tid == 0) fin_add(a + 2 * BN, (double)n_my * 256.0 * PH);
            } else {
                fin_add(p.bnb_acc + ((blockIdx.x & 7) * 2 + st) * BN + n, t);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ conv_wgrad_thin_kernel
// Weight gradient of the same layers: dW[t][a][b] = sum_pixels P[pixel, a] Q[pixel + tap_t, b] with 16 / 32 channels on both sides -
// a 9 x 16 x 16 result (9 KB) contracted over B H W = 524 288 pixels.  The 64 x 64 (a x b) blocks of the general kernels compute
// 16 x the MACs there (105 us per launch).  Here the contraction dimension of v_mfma_f32_16x16x4_f32 is FOUR CONSECUTIVE PIXELS of a
// tile row: lane (channel l15, pixel kg) reads one float of P and, per tap, one float of Q's halo tile (ds_read_b32: 64 lanes = 256
// contiguous bytes, conflict free without a swizzle), 9 AB BB MFMAs per pixel quad.  A persistent workgroup streams 16 x 16-pixel
// tiles (P tile + 18 x 18 Q halo, double buffered by LDS-DMA), its four waves take a quarter of each tile's pixels, their accumulators
// meet in LDS at the end, and the workgroup writes ONE slab [taps][Ca][Cb] (salt_wgrad_reduce sums the slabs as for every other kernel).
struct TwKP {
    const float* P; const float* Q; float* partials;
    int B, PH, PW, p_cs, QH, QW, q_cs;
    int tiles_x, tiles_y, ntiles;
    int min_dy, min_dx, pad_mode;
    int tap_off[9];
};

// QS = 2: the stride-2 operand of a transposed convolution's weight gradient (Q pixel = 2 p + tap): 8 x 16-pixel P tiles, 17 x 33 Q halo
template <int AB, int BB, int QS = 1>
__global__ __launch_bounds__(256) void conv_wgrad_thin_kernel(TwKP p) {
    constexpr int Ca = 16 * AB, Cb = 16 * BB;
    constexpr int TR = QS == 1 ? 16 : 8;                                  // tile rows
    constexpr int HH = QS * (TR - 1) + 3, HW = QS * 15 + 3;               // Q halo: 18 x 18 | 17 x 33
    constexpr int QPC = (HH * HW + 15) / 16, PPC = TR;                    // DMA pieces (1 KB) per 16-channel chunk
    constexpr unsigned HW_MAGIC = (unsigned)((1ull << 32) / HW) + 1u;
    constexpr int PP = AB * PPC, QP = BB * QPC, NP = PP + QP;             // DMA pieces per tile: P rows, then Q halo rows
    constexpr int NS = (NP + 3) / 4;
    constexpr int T_BYTES = NP * 1024, OFF_DUMMY = 2 * T_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int n_my = ((int)blockIdx.x < p.ntiles) ? (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    struct TC { int b, oy0, ox0; };
    auto coords = [&](int k) {
        const int t = (int)blockIdx.x + k * (int)gridDim.x;
        TC c; const int tx = t % p.tiles_x; const int r = t / p.tiles_x;
        c.ox0 = tx << 4; c.oy0 = (r % p.tiles_y) * TR; c.b = r / p.tiles_y; return c;
    };
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_thin_zero);
    auto dma = [&](const void* src, int dst) {
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(smem + dst), 16, 0, 0);
    };
    auto issue_tile = [&](const TC& c, int buf) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const float* pb = p.P + (int64_t)c.b * p.PH * p.PW * p.p_cs;
        const float* qb = p.Q + (int64_t)c.b * p.QH * p.QW * p.q_cs;
        const int iy0 = c.oy0 * QS + p.min_dy, ix0 = c.ox0 * QS + p.min_dx;
        const bool clamp = p.pad_mode != 0;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int pidx = wave + 4 * i;
            const unsigned char* src = zp;
            int dst = OFF_DUMMY;
            if (pidx < PP) {
                const int ch = pidx / PPC, row = (pidx - ch * PPC) * 16 + (ln >> 2);      // row = pixel (ty, tx) of the tile
                src = reinterpret_cast<const unsigned char*>(pb + (((c.oy0 + (row >> 4)) * p.PW + c.ox0 + (row & 15)) * p.p_cs + ch * 16 + (ln & 3) * 4));
                dst = buf * T_BYTES + pidx * 1024;
            } else if (pidx < NP) {
                const int q = pidx - PP;
                const int ch = q / QPC, row = (q - ch * QPC) * 16 + (ln >> 2);
                const int hy = (int)__umulhi((unsigned)row, HW_MAGIC);                    // row / HW
                const int hx = row - hy * HW;
                const int iy = iy0 + hy, ix = ix0 + hx;
                const int iyc = min(max(iy, 0), p.QH - 1), ixc = min(max(ix, 0), p.QW - 1);
                const bool inside = ((unsigned)iy < (unsigned)p.QH) & ((unsigned)ix < (unsigned)p.QW);
                if ((row < HH * HW) & (clamp | inside))
                    src = reinterpret_cast<const unsigned char*>(qb + ((iyc * p.QW + ixc) * p.q_cs + ch * 16 + (ln & 3) * 4));
                dst = buf * T_BYTES + pidx * 1024;
            }
            dma(src, dst);
        }
    };
    f32x4 acc[AB][BB][9];
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int b = 0; b < BB; ++b)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[a][b][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (n_my > 0) issue_tile(coords(0), 0);
#pragma unroll 1
    for (int k = 0; k < n_my; ++k) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // this wave's pieces of tile k landed
        __builtin_amdgcn_s_barrier();                                        // ... everybody's; everybody is done reading the other buffer
        asm volatile("" ::: "memory");
        if (k + 1 < n_my) issue_tile(coords(k + 1), (k + 1) & 1);
        const unsigned char* tb = smem + (k & 1) * T_BYTES;
        const unsigned char* qh = tb + PP * 1024;
#pragma unroll 2
        for (int qd = 0; qd < TR; ++qd) {                                    // pixel quads of this wave's TR / 4 tile rows
            const int r = wave * (TR / 4) + (qd >> 2), x0 = (qd & 3) * 4 + kg;
            float av[AB], bv[BB][9];
#pragma unroll
            for (int a = 0; a < AB; ++a) av[a] = *reinterpret_cast<const float*>(tb + a * (PPC * 1024) + (r * 16 + x0) * 64 + l15 * 4);
#pragma unroll
            for (int b = 0; b < BB; ++b)
#pragma unroll
                for (int t = 0; t < 9; ++t) bv[b][t] = *reinterpret_cast<const float*>(qh + b * (QPC * 1024) + (QS * (r * HW + x0) + p.tap_off[t]) * 64 + l15 * 4);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int a = 0; a < AB; ++a)
#pragma unroll
                    for (int b = 0; b < BB; ++b) acc[a][b][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b][t], acc[a][b][t], 0, 0, 0);
        }
    }
    // ---- the four waves' accumulators meet in LDS (the tile buffers are free), fixed order; one slab per workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                             // [4][9][Ca][Cb]
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int b = 0; b < BB; ++b)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    red[((wave * 9 + t) * Ca + a * 16 + 4 * kg + e) * Cb + b * 16 + l15] = acc[a][b][t][e];
    __syncthreads();
    float* slab = p.partials + (int64_t)blockIdx.x * 9 * Ca * Cb;
    for (int i = tid; i < 9 * Ca * Cb; i += 256)
        slab[i] = ((red[i] + red[9 * Ca * Cb + i]) + red[2 * 9 * Ca * Cb + i]) + red[3 * 9 * Ca * Cb + i];
}

int thin_cus() {
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t pr; int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
        if (cus < 8) cus = 256;
    }
    return cus;
}

template <int CI, int CO, int MODE, int NT = 9, int PH = 1>
int thin_launch_mode(const ThKP& k, hipStream_t st) {
    constexpr int LDS = PH * CI * NT * CO * 1024 + 2 * CI * 21 * 1024 + 1024 + 4 * 16 * CO * 4;
    static_assert(LDS <= 160 * 1024 && 4 * 2 * 16 * CO * 4 <= PH * CI * NT * CO * 1024, "LDS budget");
    auto kern = conv_thin_kernel<CI, CO, MODE, NT, PH>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    int per_cu = (160 * 1024) / LDS;
    static const int cap_env = getenv("SALT_THIN_WGS_PER_CU") ? atoi(getenv("SALT_THIN_WGS_PER_CU")) : 2;
    if (per_cu > cap_env) per_cu = cap_env;
    if (per_cu < 1) per_cu = 1;
    int wgs = thin_cus() * per_cu;
    if (wgs > k.ntiles) wgs = k.ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256), LDS, st, k);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

template <int CI, int CO>
int thin_launch(const ThKP& k, hipStream_t st) {
    if (k.out_step == 2) {                                                  // phase-fused transposed convolution: 4 taps x 4 phases
        if (k.fin_acc) return thin_launch_mode<CI, CO, 1, 4, 4>(k, st);
        return thin_launch_mode<CI, CO, 0, 4, 4>(k, st);
    }
    if (k.fin_acc) return thin_launch_mode<CI, CO, 1>(k, st);
    if (k.bnb_acc) return thin_launch_mode<CI, CO, 2>(k, st);
    return thin_launch_mode<CI, CO, 0>(k, st);
}

}  // namespace

// ---- host interface (conv_mfma.hip: salt_conv / salt_conv_kernel_id).  SALT_CONV_THIN = 0: off unless asked for per launch
// (cfg & 0xff == 12).  Returns 0 (not applicable) or 4 CI + CO.
int conv_thin_variant(const salt_conv_args* a) {
    static const int env = getenv("SALT_CONV_THIN") ? atoi(getenv("SALT_CONV_THIN")) : 1;
    if (!a || a->dtype != SALT_F32) return 0;
    const bool phased = a->nphase > 1;           // the phase-fused launch of a stride-2 transposed convolution (4 taps, 4 phases, out_step 2)
    if (a->ntaps != (phased ? 4 : 9)) return 0;
    const bool asked = (a->cfg & 0xff) == 12;
    if ((a->cfg & 0xff) != 0 && !asked) return 0;
    if (!asked && !env) return 0;
    if (a->in_step != 1 || a->out_oy || a->out_ox) return 0;
    if (phased ? (a->nphase != 4 || a->out_step != 2 || a->bnb_acc || a->accumulate ||
                  a->w_phase_elems != (int64_t)(a->x.C / 16) * 4 * a->y.C * 16 || a->y.H != 2 * a->OH || a->y.W != 2 * a->OW)
               : (a->out_step != 1 || a->OH != a->y.H || a->OW != a->y.W)) return 0;
    if (a->strip || a->fold_top || a->fold_bottom || a->fold_left || a->fold_right || a->x_plane || a->y_plane || a->res.p) return 0;
    if (a->stats || a->fin_ticket || a->bnb_partials || a->bnb_ticket || a->in_scale || a->in_fin_acc) return 0;
    if ((a->fin_acc && (a->accumulate || a->bnb_acc)) || (a->bnb_acc && (a->bias || a->scale || a->shift || a->relu))) return 0;     // MODE dispatch
    const int Cin = a->x.C, Cout = a->y.C;
    if ((Cin != 16 && Cin != 32) || (Cout != 16 && Cout != 32)) return 0;
    if (a->x.B != a->y.B || a->OH % 16 || a->OW % 16) return 0;
    int min_dy = 1 << 30, max_dy = -(1 << 30), min_dx = 1 << 30, max_dx = -(1 << 30);
    for (int t = 0; t < a->ntaps; ++t) {
        min_dy = a->tap_dy[t] < min_dy ? a->tap_dy[t] : min_dy; max_dy = a->tap_dy[t] > max_dy ? a->tap_dy[t] : max_dy;
        min_dx = a->tap_dx[t] < min_dx ? a->tap_dx[t] : min_dx; max_dx = a->tap_dx[t] > max_dx ? a->tap_dx[t] : max_dx;
    }
    if (max_dy - min_dy > 2 || max_dx - min_dx > 2) return 0;
    if (a->x.cs % 4 || a->y.cs % 4 || ((reinterpret_cast<uintptr_t>(a->x.p) | reinterpret_cast<uintptr_t>(a->y.p) | reinterpret_cast<uintptr_t>(a->w)) & 15)) return 0;
    auto small = [&](const salt_view& v) { return !v.p || (int64_t)v.B * v.H * v.W * v.cs < (int64_t)1 << 31; };
    if (!small(a->x) || !small(a->y) || !small(a->bnb_y) || !small(a->bnb_a)) return 0;
    if (a->bnb_acc) {
        if (!view_ok(a->bnb_y) || a->bnb_y.B != a->y.B || a->bnb_y.H != a->y.H || a->bnb_y.W != a->y.W || a->bnb_y.C != Cout || a->bnb_y.cs % 4 ||
            (reinterpret_cast<uintptr_t>(a->bnb_y.p) & 15) || !a->bnb_mean || !a->bnb_invstd || !a->bnb_gamma || !a->bnb_beta) return 0;
        if (a->bnb_a.p && (a->bnb_a.B != a->y.B || a->bnb_a.H != a->y.H || a->bnb_a.W != a->y.W || a->bnb_a.C != Cout || a->bnb_a.cs % 4 ||
                           (reinterpret_cast<uintptr_t>(a->bnb_a.p) & 15))) return 0;
    }
    const int64_t ntiles = (int64_t)a->y.B * (a->OH / 16) * (a->OW / 16);
    if (!asked && ntiles < thin_cus() / 2) return 0;                       // too few tiles to fill the chip
    return 4 * (Cin / 16) + Cout / 16;
}

int conv_thin_launch(const salt_conv_args* a, hipStream_t st) {
    const int v = conv_thin_variant(a);
    if (!v) SALT_FAIL(SALT_E_UNSUPPORTED, "conv_thin: not applicable");
    ThKP k;
    k.x = reinterpret_cast<const float*>(a->x.p); k.w = reinterpret_cast<const float*>(a->w); k.y = reinterpret_cast<float*>(a->y.p);
    k.bias = a->bias; k.scale = a->scale; k.shift = a->shift;
    k.B = a->x.B; k.H = a->x.H; k.W = a->x.W; k.x_cs = a->x.cs; k.y_cs = a->y.cs; k.OH = a->OH; k.OW = a->OW;
    k.yH = a->y.H; k.yW = a->y.W; k.out_step = a->out_step;
    k.tiles_x = a->OW / 16; k.tiles_y = a->OH / 16; k.ntiles = a->y.B * k.tiles_x * k.tiles_y;
    int min_dy = 1 << 30, min_dx = 1 << 30;
    for (int t = 0; t < a->ntaps; ++t) { min_dy = a->tap_dy[t] < min_dy ? a->tap_dy[t] : min_dy; min_dx = a->tap_dx[t] < min_dx ? a->tap_dx[t] : min_dx; }
    k.min_dy = min_dy; k.min_dx = min_dx; k.pad_mode = a->pad_mode;
    for (int t = 0; t < 9; ++t) k.tap_off[t] = t < a->ntaps ? (a->tap_dy[t] - min_dy) * 18 + (a->tap_dx[t] - min_dx) : 0;
    k.relu = a->relu; k.accumulate = a->accumulate;
    k.bnb_y = reinterpret_cast<const float*>(a->bnb_y.p); k.bnb_a = reinterpret_cast<const float*>(a->bnb_a.p);
    k.bnb_cs = a->bnb_y.cs; k.bnb_acs = a->bnb_a.cs; k.bnb_relu = a->bnb_relu;
    k.bnb_mean = a->bnb_mean; k.bnb_invstd = a->bnb_invstd; k.bnb_gamma = a->bnb_gamma; k.bnb_beta = a->bnb_beta;
    k.fin_acc = a->fin_acc; k.bnb_acc = a->bnb_acc;
    if (!a->bnb_acc) { k.bnb_y = nullptr; k.bnb_a = nullptr; }
    switch (v) {
        case 5: return thin_launch<1, 1>(k, st);
        case 6: return thin_launch<1, 2>(k, st);
        case 9: return thin_launch<2, 1>(k, st);
        case 10: return thin_launch<2, 2>(k, st);
    }
    SALT_FAIL(SALT_E_UNSUPPORTED, "conv_thin: variant %d", v);
}

// ---- weight gradient of the same layers (conv_mfma.hip: salt_conv_wgrad_nsplit / salt_conv_wgrad).  Returns 0 (not one of its shapes)
// or the number of slabs it writes; launch: also enqueues it (*rc = status).  SALT_WGRAD_THIN = 0: off.
namespace {
template <int AB, int BB, int QS>
int tw_launch(const TwKP& k, int wgs, hipStream_t st) {
    constexpr int LDS = 2 * (AB * (QS == 1 ? 16 : 8) + BB * (QS == 1 ? 21 : 36)) * 1024 + 1024;
    static_assert(LDS <= 160 * 1024 && 4 * 9 * 16 * AB * 16 * BB * 4 <= LDS, "LDS budget");
    auto kern = conv_wgrad_thin_kernel<AB, BB, QS>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256), LDS, st, k);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}
}  // namespace

int conv_wgrad_thin(const salt_conv_wgrad_args* a, bool launch, hipStream_t st, int* rc) {
    static const int env = getenv("SALT_WGRAD_THIN") ? atoi(getenv("SALT_WGRAD_THIN")) : 1;
    if (!a || !env || a->dtype != SALT_F32 || a->ntaps != 9 || (a->q_step != 1 && a->q_step != 2) || a->q_plane) return 0;
    if (!view_ok(a->p) || !view_ok(a->q) || a->p.B != a->q.B) return 0;
    const int Ca = a->p.C, Cb = a->q.C, qs = a->q_step, tr = qs == 1 ? 16 : 8;
    if ((Ca != 16 && Ca != 32) || (Cb != 16 && Cb != 32) || a->p.H % tr || a->p.W % 16 || (qs == 2 && Cb != 16)) return 0;
    int min_dy = 1 << 30, max_dy = -(1 << 30), min_dx = 1 << 30, max_dx = -(1 << 30);
    for (int t = 0; t < 9; ++t) {
        min_dy = a->tap_dy[t] < min_dy ? a->tap_dy[t] : min_dy; max_dy = a->tap_dy[t] > max_dy ? a->tap_dy[t] : max_dy;
        min_dx = a->tap_dx[t] < min_dx ? a->tap_dx[t] : min_dx; max_dx = a->tap_dx[t] > max_dx ? a->tap_dx[t] : max_dx;
    }
    if (max_dy - min_dy > 2 || max_dx - min_dx > 2) return 0;
    if (a->p.cs % 4 || a->q.cs % 4 || ((reinterpret_cast<uintptr_t>(a->p.p) | reinterpret_cast<uintptr_t>(a->q.p)) & 15)) return 0;
    if ((int64_t)a->p.B * a->p.H * a->p.W * a->p.cs >= (int64_t)1 << 31 || (int64_t)a->q.B * a->q.H * a->q.W * a->q.cs >= (int64_t)1 << 31) return 0;
    const int ntiles = a->p.B * (a->p.H / tr) * (a->p.W / 16);
    if (ntiles < thin_cus() / 2) return 0;
    const int lds = 2 * ((Ca / 16) * tr + (Cb / 16) * (qs == 1 ? 21 : 36)) * 1024 + 1024;
    int per_cu = (160 * 1024) / lds;
    if (per_cu > 2) per_cu = 2;
    int wgs = thin_cus() * per_cu;
    if (wgs > ntiles) wgs = ntiles;
    if (!launch) return wgs;
    if (!a->partials || a->nsplit != wgs) { salt_set_error("wgrad_thin: nsplit %d, expected %d", a->nsplit, wgs); *rc = SALT_E_BADARG; return wgs; }
    TwKP k;
    k.P = reinterpret_cast<const float*>(a->p.p); k.Q = reinterpret_cast<const float*>(a->q.p); k.partials = a->partials;
    k.B = a->p.B; k.PH = a->p.H; k.PW = a->p.W; k.p_cs = a->p.cs; k.QH = a->q.H; k.QW = a->q.W; k.q_cs = a->q.cs;
    k.tiles_x = a->p.W / 16; k.tiles_y = a->p.H / tr; k.ntiles = ntiles;
    k.min_dy = min_dy; k.min_dx = min_dx; k.pad_mode = a->pad_mode;
    for (int t = 0; t < 9; ++t) k.tap_off[t] = (a->tap_dy[t] - min_dy) * (qs == 1 ? 18 : 33) + (a->tap_dx[t] - min_dx);
    const int v = 4 * (Ca / 16) + Cb / 16;
    if (qs == 1) *rc = v == 5 ? tw_launch<1, 1, 1>(k, wgs, st) : v == 6 ? tw_launch<1, 2, 1>(k, wgs, st) : v == 9 ? tw_launch<2, 1, 1>(k, wgs, st) : tw_launch<2, 2, 1>(k, wgs, st);
    else *rc = v == 5 ? tw_launch<1, 1, 2>(k, wgs, st) : tw_launch<2, 1, 2>(k, wgs, st);      // (Cb == 16: the 17 x 33 halo of 32 channels would not fit twice)
    return wgs;
}
