// conv_ws.hip — weight-stationary, multi-tile 3x3 convolution for the <= 64-channel layers (bf16, gfx950).
//
// Why a second structure beside conv_mfma_kernel.  The 64-channel layers of the path (torchvision BasicBlock conv1 / conv2 of
// ResNet34 layer1 via architectures/encoders.py:6-45, Conv2dBnRelu of the two shallowest DecoderBlocks, architectures/base.py:7-37,
// and their data gradients) have K = 9 * Cin = 288..576: a conv_mfma_kernel workgroup stages its weights (36.9 KB per 32-channel
// chunk) for every 256-pixel tile it computes, twice per CU, and pays prologue + epilogue per 1.9 us of MFMAs.  In-kernel clocks and
// the prologue-burst figure of the guide (~11 B/clk/CU with every CU loading at once) say the per-CU load path, not the matrix pipe,
// sets those launches' time.  This kernel loads every byte ONCE per CU:
//
//   * ONE 512-thread workgroup per CU owns the whole LDS (160 KB): the packed weights of ALL taps and ALL input channels stay
//     resident for the lifetime of the workgroup (73.7 KB for 64 -> 64), next to two halo buffers (a 16 x 16 pixel tile + its
//     3 x 3 halo over all input channels, 42 KB for 64 channels).  Everything arrives by LDS-DMA (global_load_lds_dwordx4: no
//     staging registers, no ds_write pass); the XOR slot swizzle of the 64-byte rows is applied to the per-lane SOURCE address.
//   * The 8 waves are TWO groups of 4.  A group alternates between two roles, one phase each, and the groups run in antiphase:
//       MMA(tile k): 4 waves x (64 pixels x all output channels), MI x NI = 2 x 2 register blocking (one ds_read_b128 per MFMA),
//                    all 9 taps x all chunks straight through - no barrier inside: operands are fully resident;
//       EPI(tile k): the same waves, one phase later, with the tile still in their accumulators - and NO trip through LDS: the MFMA
//                    operands are swapped (A = weights, B = pixels), so a lane holds, for ONE pixel, 4 x 4 consecutive output
//                    channels per 32 x 32 block; v_cvt_pk_bf16_f32 + v_permlane32_swap pair the two half-waves into whole 16-byte
//                    pieces of the NHWC pixel row, which go straight to HBM ((+)= and the BatchNorm-backward operand tiles are
//                    16-byte loads at the same addresses); then the LDS-DMA of the halo of tile k+2 into the buffer tile k released.
//     So the epilogue + next-tile load of one group hide under the MFMAs of the other; each SIMD hosts one wave of either group.
//     ONE raw s_barrier per phase (never __syncthreads inside the loop: DMA stays in flight across a phase).
//   * Per-channel sums (BatchNorm forward statistics, BatchNorm-backward sums) are per-LANE fp32 accumulators over all tiles of the
//     workgroup (a lane's pixels differ, its channel set does not); ONE halving butterfly over the 32 lanes at kernel end, an 8-wave
//     merge through LDS, and ONE set of fp64 shard atomics per workgroup (salt_conv_args.fin_acc / bnb_acc without ticket).
//   * First version of this kernel (kept in git history, profiles/r03_ws_clocks_v1.txt): lane = channel, transposition of the tile
//     through wave-private LDS slices.  In-kernel clocks: epilogue 16.7 k cycles per 256 x 64 tile against 7.2 k for its MFMAs
//     (register spills around 64 ds_write_b16 + the store loop) - 25 us per 64 -> 64 @64x64 launch against 17.7 for conv_mfma_kernel.
//
// Scope (host: conv_ws_eligible): bf16, 9 taps inside a 3 x 3 window, unit steps, Cin in {32, 64}, Cout in {32, 64}, output grid a
// multiple of 16 x 16 and equal to y, zero or replicate (clamp) padding, bias / folded BN / ReLU / accumulate / statistics shards /
// BatchNorm-backward shards.  Everything else stays on conv_mfma_kernel.
#include <cstdlib>
#include <type_traits>
#include "common.h"

#ifndef SALT_WS_CLK
#define SALT_WS_CLK 0            // 1: per-workgroup s_memtime stamps into g_ws_clk (tools/ws_clocks.py; timing build only)
#endif

namespace {

struct WsKP {
    const bf16_t* x; const bf16_t* w; bf16_t* y;
    const float* bias; const float* scale; const float* shift;
    int B, H, W, x_cs, y_cs, OH, OW;
    int tiles_x, tiles_y, ntiles, per_xcd, wg_per_xcd;
    int min_dy, min_dx, pad_mode;
    int tap_off[9];
    int relu, accumulate;
    const bf16_t* bnb_y; const bf16_t* bnb_a; int bnb_cs, bnb_acs, bnb_relu;
    const float* bnb_mean; const float* bnb_invstd; const float* bnb_gamma; const float* bnb_beta;
    double* fin_acc;             // [8][2 Cout + 1] forward statistics shards (nullptr: off)
    double* bnb_acc;             // [8][2][Cout] BatchNorm-backward shards (nullptr: off)
};

__device__ __attribute__((aligned(16))) unsigned int g_ws_zero[4] = {0u, 0u, 0u, 0u};
#if SALT_WS_CLK
__device__ unsigned long long g_ws_clk[256 * 48];
#endif

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

__device__ __forceinline__ int ws_swz(int row, int slot) { return row * 64 + (((slot ^ (row >> 2)) & 3) << 4); }
// MFMA row -> pixel of a 32-pixel (2 image rows x 16) block such that every 16-lane group of a ds_read_b128 covers 16 CONSECUTIVE
// pixels of one image row (conflict-free for the 18-pixel halo pitch; see conv_glds_kernel)
__device__ __forceinline__ int ws_perm(int m) { return (int)((0x73261540u >> ((m >> 2) * 4)) & 0xfu) * 4 + (m & 3); }

template <int N> __device__ __forceinline__ void ws_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// One step of the halving butterfly that sums per-lane values over the 32 lanes of a half-wave: exchange with lane ^ (1 << S); a lane
// whose bit S is clear keeps the lower half of its N values, else the upper half, and adds the partner's copy of the half it keeps.
// After steps 0 .. 3 on 16 values (and a plain xor-16 add) v[0] of lane l holds the total of value index
// 8 b0 + 4 b1 + 2 b2 + b3 (b = bits of l & 31).
template <int N, int S> __device__ __forceinline__ void ws_halve(float* v, int l31) {
    const bool up = (l31 >> S) & 1;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const float keep = up ? v[i + N / 2] : v[i], give = up ? v[i] : v[i + N / 2];
        v[i] = keep + __shfl_xor(give, 1 << S);
    }
}

// ---- epilogue of one 64-pixel x (32 NI)-channel wave tile held in swapped-operand accumulators (shared by conv_ws_kernel and
// conv_ls_kernel).  acc[i][j]: lane = pixel ws_perm(lane & 31) of sub-tile i, register r = channel 32 j + (r & 3) + 8 (r >> 2) + 4 khalf
// of the tile's channel block.  MODE 0: bias / folded BN / ReLU / (+)=;  1: + train-mode BatchNorm statistics of the result;
// 2: (+)= and the BatchNorm-backward sums of the stored gradient (salt_conv_args.bnb_*).  cst = LDS [4][BN] floats (MODE 0 / 1 bias,
// scale, shift; MODE 2 mean, invstd, gamma invstd, beta - mean gamma invstd) of the tile's channel block; n0 = its first channel in y.
// Per-channel sums: per channel block j the lane gathers 16 values per statistic over its two pixels (MODE 1: per channel register r;
// MODE 2: per channel (gp, e) of its two pieces), runs TWO halving steps over its lane quad (DPP, no LDS) and adds the 4 survivors to
// rs0 / rs1; ws_sums_flush runs the other 3 steps once per kernel.
struct WsEpi {
    bf16_t* y; const bf16_t* bnb_y; const bf16_t* bnb_a;
    int y_cs, bnb_cs, bnb_acs, relu, accumulate, bnb_relu;
    bool has_affine, sums;
};

template <int NI, int MODE>
__device__ __forceinline__ void ws_epilogue_tile(const WsEpi& p, f32x16 (&acc)[2][NI], const unsigned (&pix)[2], int n0, const float* cst,
                                                 float (&rs0)[NI][4], float (&rs1)[NI][4], int khalf, int l31) {
    typedef bf16_t T;
    constexpr int MI = 2, BN = 32 * NI;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        // operand tiles of the (+)= / BatchNorm-backward epilogue: 16-byte pieces at this lane's store addresses
        u32x4 oldv[MI][2], yv[MI][2], av[MI][2];
        if (MODE != 1 && p.accumulate) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp)
                    oldv[i][gp] = *reinterpret_cast<const u32x4*>(p.y + (pix[i] * (unsigned)p.y_cs + n0 + 8 * khalf + 32 * j + 16 * gp));
        }
        if (MODE == 2 && p.sums) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    yv[i][gp] = *reinterpret_cast<const u32x4*>(p.bnb_y + (pix[i] * (unsigned)p.bnb_cs + n0 + 8 * khalf + 32 * j + 16 * gp));
                    if (p.bnb_a) av[i][gp] = *reinterpret_cast<const u32x4*>(p.bnb_a + (pix[i] * (unsigned)p.bnb_acs + n0 + 8 * khalf + 32 * j + 16 * gp));
                }
        }
        float t0[16], t1[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { t0[e] = 0.f; t1[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const unsigned yo = pix[i] * (unsigned)p.y_cs + n0 + 8 * khalf + 32 * j;     // + 16 gp: this lane's piece gp of block j
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r];
            if (MODE != 2 && p.has_affine) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {                                   // 4 consecutive channels 32 j + 8 q + 4 khalf ..
                    const int ch0 = 32 * j + 8 * q + 4 * khalf;
                    const f32x4 bi = *reinterpret_cast<const f32x4*>(cst + ch0), sc = *reinterpret_cast<const f32x4*>(cst + BN + ch0),
                                sh = *reinterpret_cast<const f32x4*>(cst + 2 * BN + ch0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = (v[4 * q + e] + bi[e]) * sc[e] + sh[e];
                        if (p.relu) t = fmaxf(t, 0.f);
                        v[4 * q + e] = t;
                    }
                }
            }
            if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { t0[r] += v[r]; t1[r] += v[r] * v[r]; }
            }
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                // channel groups 2 gp (registers 8 gp .. 8 gp + 3) and 2 gp + 1 of both half-waves -> one 16-byte piece per lane:
                // lanes 0-31 channels 32 j + 16 gp + 0..7, lanes 32-63 channels 32 j + 16 gp + 8..15 of the same pixel
                const unsigned ax = f2bf_pk(v[8 * gp + 0], v[8 * gp + 1]), ay = f2bf_pk(v[8 * gp + 2], v[8 * gp + 3]);
                const unsigned bx = f2bf_pk(v[8 * gp + 4], v[8 * gp + 5]), by = f2bf_pk(v[8 * gp + 6], v[8 * gp + 7]);
                const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                u32x4 stored = {rx[0], ry[0], rx[1], ry[1]};
                if (MODE != 1 && p.accumulate) {
                    float f8[8], o8[8];
                    unpack16<T>(stored, f8); unpack16<T>(oldv[i][gp], o8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f8[e] += o8[e];
                    stored = pack16<T>(f8);
                }
                *reinterpret_cast<u32x4*>(p.y + (yo + 16 * gp)) = stored;
                if (MODE == 2 && p.sums) {
                    const int ch0 = 32 * j + 16 * gp + 8 * khalf;
                    float mu[8], is[8], gq[8], yc[8];
                    *reinterpret_cast<f32x4*>(mu) = *reinterpret_cast<const f32x4*>(cst + ch0);
                    *reinterpret_cast<f32x4*>(mu + 4) = *reinterpret_cast<const f32x4*>(cst + ch0 + 4);
                    *reinterpret_cast<f32x4*>(is) = *reinterpret_cast<const f32x4*>(cst + BN + ch0);
                    *reinterpret_cast<f32x4*>(is + 4) = *reinterpret_cast<const f32x4*>(cst + BN + ch0 + 4);
                    unpack16<T>(stored, gq); unpack16<T>(yv[i][gp], yc);
                    if (p.bnb_a) {                                              // residual layer: the mask is the sign of the block output
                        float a8[8];
                        unpack16<T>(av[i][gp], a8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float gg = (!p.bnb_relu || a8[e] > 0.f) ? gq[e] : 0.f;
                            t0[gp * 8 + e] += gg; t1[gp * 8 + e] += gg * (yc[e] - mu[e]) * is[e];
                        }
                    } else {
                        float ks[8], sh[8];
                        *reinterpret_cast<f32x4*>(ks) = *reinterpret_cast<const f32x4*>(cst + 2 * BN + ch0);
                        *reinterpret_cast<f32x4*>(ks + 4) = *reinterpret_cast<const f32x4*>(cst + 2 * BN + ch0 + 4);
                        *reinterpret_cast<f32x4*>(sh) = *reinterpret_cast<const f32x4*>(cst + 3 * BN + ch0);
                        *reinterpret_cast<f32x4*>(sh + 4) = *reinterpret_cast<const f32x4*>(cst + 3 * BN + ch0 + 4);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float gg = (!p.bnb_relu || yc[e] * ks[e] + sh[e] > 0.f) ? gq[e] : 0.f;
                            t0[gp * 8 + e] += gg; t1[gp * 8 + e] += gg * (yc[e] - mu[e]) * is[e];
                        }
                    }
                }
            }
        }
        if (MODE != 0 && p.sums) {
            ws_halve<16, 0>(t0, l31); ws_halve<8, 1>(t0, l31);
            ws_halve<16, 0>(t1, l31); ws_halve<8, 1>(t1, l31);
#pragma unroll
            for (int e = 0; e < 4; ++e) { rs0[j][e] += t0[e]; rs1[j][e] += t1[e]; }
        }
    }
}

// The remaining butterfly steps of the running sums, the merge of the NW waves that hold sums through LDS (red: [NW][2][BN] floats;
// the caller synchronised the workgroup and every wave is done with that memory), and the fp64 shard atomics of the workgroup:
// MODE 1 -> fin_acc [8][2 C + 1] (sum, sum of squares, count), MODE 2 -> bnb_acc [8][2][C]; n0 = first channel, C = channels of the layer.
template <int NI, int MODE, int NW>
__device__ __forceinline__ void ws_sums_flush(float (&rs0)[NI][4], float (&rs1)[NI][4], bool holder, int hw, float* red, int n0, int C,
                                              double* fin_acc, double* bnb_acc, double count, int khalf, int l31) {
    constexpr int BN = 32 * NI;
    const int tid = threadIdx.x;
    // value index 8 b0 + 4 b1 + 2 b2 + b3 of block j (b = bits of l31).  MODE 1: index = channel register r; MODE 2: index = 8 gp + e
    const int idx = 8 * (l31 & 1) + 4 * ((l31 >> 1) & 1) + 2 * ((l31 >> 2) & 1) + ((l31 >> 3) & 1);
    if (holder) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            ws_halve<4, 2>(rs0[j], l31); ws_halve<2, 3>(rs0[j], l31);
            ws_halve<4, 2>(rs1[j], l31); ws_halve<2, 3>(rs1[j], l31);
            const float s0 = rs0[j][0] + __shfl_xor(rs0[j][0], 16), s1 = rs1[j][0] + __shfl_xor(rs1[j][0], 16);
            const int ch = MODE == 1 ? 32 * j + (idx & 3) + 8 * (idx >> 2) + 4 * khalf : 32 * j + 16 * (idx >> 3) + 8 * khalf + (idx & 7);
            red[(hw * 2 + 0) * BN + ch] = s0;                               // (lanes l and l ^ 16 write the same value)
            red[(hw * 2 + 1) * BN + ch] = s1;
        }
    }
    __syncthreads();
    if (tid < 2 * BN) {
        const int st = tid / BN, n = tid - st * BN;
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += (double)red[(w * 2 + st) * BN + n];
        if (MODE == 1) {
            double* a = fin_acc + (blockIdx.x & 7) * (2 * C + 1);
            fin_add(a + st * C + n0 + n, t);
            if (tid == 0 && count > 0.0) fin_add(a + 2 * C, count);
        } else {
            fin_add(bnb_acc + ((blockIdx.x & 7) * 2 + st) * C + n0 + n, t);
        }
    }
}

// MODE: see ws_epilogue_tile
template <int NCH, int NI, int MODE>
__global__ __launch_bounds__(512) void conv_ws_kernel(WsKP p) {
    typedef bf16_t T;
    constexpr int BN = 32 * NI, NT = 9, MI = 2;
    constexpr int HPC = 21;                            // halo DMA pieces (16 rows x 64 B) per chunk: 18 x 18 = 324 rows, padded to 336
    constexpr int HP = NCH * HPC;                      // halo pieces per tile
    constexpr int WP = NCH * NT * BN / 16;             // weight pieces
    constexpr int NSW = (WP + 7) / 8;                  // weight DMA instructions per wave (8 waves)
    constexpr int NSH = (HP + 3) / 4;                  // halo DMA instructions per wave and tile (4 waves of a group)
    constexpr int W_BYTES = WP * 1024, H_BYTES = HP * 1024, HC_BYTES = HPC * 1024;
    constexpr int OFF_H = W_BYTES, OFF_DUMMY = OFF_H + 2 * H_BYTES;
    constexpr int OFF_CONST = OFF_DUMMY + 1024;        // [4][BN] floats: MODE 0 / 1 bias, scale, shift; MODE 2 mean, invstd, gamma invstd, beta - mean gamma invstd
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wm = wave & 3;
    const int khalf = lane >> 5, l31 = lane & 31;

    // ---- tiles of this workgroup: XCD x (= block id % 8, where consecutive block ids go) owns a contiguous range of tiles, so that
    // neighbouring tiles share their halo rows through that XCD's L2
    const int xcd = blockIdx.x & 7, jwg = blockIdx.x >> 3;
    const int t_lo = xcd * p.per_xcd;
    const int t_hi = min(t_lo + p.per_xcd, p.ntiles);
    const int n_my = (t_lo + jwg < t_hi) ? (t_hi - t_lo - jwg + p.wg_per_xcd - 1) / p.wg_per_xcd : 0;
    if (n_my <= 0) return;
    auto tile_of = [&](int k) { return t_lo + jwg + k * p.wg_per_xcd; };
    struct TC { int b, oy0, ox0; };
    auto coords = [&](int t) {
        TC c; const int tx = t % p.tiles_x; const int r = t / p.tiles_x;
        c.ox0 = tx << 4; c.oy0 = (r % p.tiles_y) << 4; c.b = r / p.tiles_y; return c;
    };
#if SALT_WS_CLK
    unsigned long long clk[24]; int nclk = 0;
    auto stamp = [&]() { if (nclk < 24) clk[nclk++] = __builtin_readcyclecounter(); };
#else
    auto stamp = [&]() {};
#endif
    stamp();

    // ---- LDS-DMA issue
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_ws_zero);
    auto dma = [&](const void* src, int dst) {
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(smem + dst), 16, 0, 0);
    };
    auto issue_weights = [&]() {
#pragma unroll
        for (int i = 0; i < NSW; ++i) {
            const int q = wave + 8 * i;                                   // piece q = packed weight rows 16 q .. 16 q + 15
            const bool real = q < WP;
            const int R = q * 16 + (lane >> 2);
            const int slot = (lane ^ (R >> 2)) & 3;
            const unsigned char* src = real ? reinterpret_cast<const unsigned char*>(p.w + (R * 32 + slot * 8)) : zp;
            dma(src, real ? q * 1024 : OFF_DUMMY);
        }
    };
    auto issue_halo = [&](const TC& c, int g) {
        int ln = lane;
        asm volatile("" : "+v"(ln));             // keeps the per-piece index math INSIDE the tile loop: hoisted to kernel entry it is spilled around the MFMA phases
        const T* xb = p.x + (int64_t)c.b * p.H * p.W * p.x_cs;
        const int iy0 = c.oy0 + p.min_dy, ix0 = c.ox0 + p.min_dx;
        const bool clamp = p.pad_mode != 0;
#pragma unroll
        for (int i = 0; i < NSH; ++i) {
            const int pidx = wm + 4 * i;                                       // round-robin over the group's 4 waves
            const bool real = pidx < HP;
            const int ch = (NCH > 1 && pidx >= HPC) ? 1 : 0;
            const int row = (pidx - ch * HPC) * 16 + (ln >> 2);
            const int hy = (int)__umulhi((unsigned)row, 238609295u);          // row / 18 for row < 2^16 (2^32 / 18 + 1)
            const int hx = row - hy * 18;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);       // == (iy, ix) for an inside pixel
            const bool inside = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            const bool valid = real & (row < 324) & (clamp | inside);          // branch-free: an invalid piece reads 16 zero bytes
            const int slot = (ln ^ (row >> 2)) & 3;                            // LDS slot lane & 3 of row `row` holds channel slot `slot`
            const unsigned char* src = reinterpret_cast<const unsigned char*>(xb + ((iyc * p.W + ixc) * p.x_cs + ch * 32 + slot * 8));
            dma(valid ? src : zp, real ? OFF_H + g * H_BYTES + pidx * 1024 : OFF_DUMMY);
        }
    };

    // ---- prologue: the weights (all waves), then each group's first halo tile; per-channel epilogue constants into LDS
    issue_weights();
    if (grp < n_my) issue_halo(coords(tile_of(grp)), grp);
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
    }                                                                      // (read after the first phase barrier at the earliest)

    // ---- fragment addressing: the lane's halo pixel per M sub-tile is tile invariant
    int pbase[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm * 64 + i * 32 + ws_perm(l31);
        pbase[i] = (m >> 4) * 18 + (m & 15);
    }
    // acc[i][j]: D = W X^T of pixel sub-tile i, channel block j: lane = pixel ws_perm(l31) of the sub-tile, register r = channel
    // 32 j + (r & 3) + 8 (r >> 2) + 4 khalf
    f32x16 acc[MI][NI];
    struct Frag { u32x4 a[MI], b[NI]; };
    auto mma_tile = [&](int g) {
        const unsigned char* hb = smem + OFF_H + g * H_BYTES;
        int a_addr[NT][MI], b_addr[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            int pb = pbase[i];
            asm volatile("" : "+v"(pb));                                     // recomputed per tile: kept live across the epilogue these 36 addresses were spilled
#pragma unroll
            for (int t = 0; t < NT; ++t) a_addr[t][i] = ws_swz(pb + p.tap_off[t], khalf);
        }
        {
            int lb = l31;
            asm volatile("" : "+v"(lb));
#pragma unroll
            for (int j = 0; j < NI; ++j) b_addr[j] = ws_swz(j * 32 + lb, khalf);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        auto load_frag = [&](int s, Frag& f) {                              // s = (chunk, tap, k-step), a constant after unrolling
            const int c = s / (2 * NT), t = (s % (2 * NT)) >> 1, hx = (s & 1) << 5;
#pragma unroll
            for (int i = 0; i < MI; ++i) f.a[i] = *reinterpret_cast<const u32x4*>(hb + c * HC_BYTES + (a_addr[t][i] ^ hx));
#pragma unroll
            for (int j = 0; j < NI; ++j) f.b[j] = *reinterpret_cast<const u32x4*>(smem + (c * NT + t) * (BN * 64) + (b_addr[j] ^ hx));
        };
        auto mma_frag = [&](const Frag& f) {                                 // operands swapped: rows of D = output channels
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.b[j]), __builtin_bit_cast(bf16x8, f.a[i]), acc[i][j], 0, 0, 0);
        };
        // fragment ring of 3 stages: the reads of stage s + 2 are issued before the MFMAs of stage s.  One wave per SIMD feeds the matrix
        // pipe here (its partner on the SIMD is in the epilogue role), so nobody else covers its LDS latency: with a 2-stage ring the
        // phase ran at 64 % of the MFMA rate alone on the CU (in-kernel clocks, profiles/r03_ws_clocks_v1.txt)
        constexpr int NST = NCH * NT * 2;
        Frag f[3];
        load_frag(0, f[0]);
        load_frag(1, f[1]);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MI + NI), 0);
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            if (s + 2 < NST) load_frag(s + 2, f[(s + 2) % 3]);
            mma_frag(f[s % 3]);
            if (s + 2 < NST) __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
        }
    };

    // ---- running per-channel sums over all tiles of the workgroup (ws_epilogue_tile)
    float rs0[NI][4], rs1[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { rs0[j][e] = 0.f; rs1[j][e] = 0.f; }
    const bool sums = (MODE == 1 && p.fin_acc) || (MODE == 2 && p.bnb_acc);
    const WsEpi ep = {p.y, p.bnb_y, p.bnb_a, p.y_cs, p.bnb_cs, p.bnb_acs, p.relu, p.accumulate, p.bnb_relu,
                      p.bias || p.scale || p.shift || p.relu, sums};

    // epilogue of the tile in `acc` (computed by this wave one phase ago); issues the halo DMA of `next` (if any) into buffer g
    auto epilogue = [&](const TC& c, int g, bool has_next, const TC& next) {
        unsigned pix[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = wm * 64 + i * 32 + ws_perm(l31);
            pix[i] = (unsigned)((c.b * p.OH + c.oy0 + (m >> 4)) * p.OW + c.ox0 + (m & 15));
        }
        ws_epilogue_tile<NI, MODE>(ep, acc, pix, 0, reinterpret_cast<const float*>(smem + OFF_CONST), rs0, rs1, khalf, l31);
        stamp();
        if (has_next) {
            // the halo of this group's next tile goes into the buffer the group released at the phase barrier - LAST, so that the
            // plain vmcnt(0) in front of the next phase barrier covers it whatever else shares the counter
            __builtin_amdgcn_sched_barrier(0);
            issue_halo(next, g);
        }
    };

    // ---- phases.  Phase k: group k & 1 computes tile k, the other group finishes tile k - 1.  Before every phase barrier a wave
    // waits for the DMA pieces IT issued for the data the next phase reads.
    for (int k = 0; k <= n_my; ++k) {
        const int g_mma = k & 1;
        if (k == 0) { if (grp == 0 || n_my < 2) ws_wait_vm<0>(); else ws_wait_vm<NSH>(); }   // weights (+ tile 0) landed; group 1's own tile may fly on
        else if (grp == g_mma && k < n_my) ws_wait_vm<0>();                  // this group's halo tile k (prologue / its previous epilogue) landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // (k == 0: the epilogue constants written above)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp();
        if (grp == g_mma) {
            if (k < n_my) { __builtin_amdgcn_s_setprio(1); mma_tile(g_mma); __builtin_amdgcn_s_setprio(0); }
        } else if (k >= 1) {
            const bool has_next = k + 1 < n_my;
            const TC cur = coords(tile_of(k - 1));
            const TC nxt = has_next ? coords(tile_of(k + 1)) : cur;
            epilogue(cur, grp, has_next, nxt);
        }
        stamp();
    }

    // ---- per-workgroup sums -> fp64 shard atomics (the weights region is free: every MFMA phase ended before the last barrier)
    if (MODE != 0 && sums) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ws_sums_flush<NI, MODE, 8>(rs0, rs1, true, wave, reinterpret_cast<float*>(smem), 0, BN, p.fin_acc, p.bnb_acc, (double)n_my * 256.0, khalf, l31);
    }
#if SALT_WS_CLK
    stamp();
    if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 256) {
        unsigned long long* o = g_ws_clk + (blockIdx.x * 2 + grp) * 24;
        for (int i = 0; i < 24; ++i) o[i] = i < nclk ? clk[i] : 0ull;
    }
#endif
}


int ws_cus() {
    static int cus = 0;
    if (!cus) {
        hipDeviceProp_t pr; int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) cus = pr.multiProcessorCount;
        if (cus < 8) cus = 256;
    }
    return cus;
}

template <int NCH, int NI, int MODE>
int ws_launch_mode(const WsKP& k, hipStream_t st) {
    constexpr int BN = 32 * NI, HP = NCH * 21, WP = NCH * 9 * BN / 16;
    constexpr int LDS = WP * 1024 + 2 * HP * 1024 + 1024 + 4 * BN * 4;
    static_assert(LDS <= 160 * 1024 && 8 * 2 * BN * 4 <= WP * 1024, "LDS budget");
    auto kern = conv_ws_kernel<NCH, NI, MODE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(k.wg_per_xcd * 8)), dim3(512), LDS, st, k);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

template <int NCH, int NI>
int ws_launch(const WsKP& k, hipStream_t st) {
    if (k.fin_acc) return ws_launch_mode<NCH, NI, 1>(k, st);
    if (k.bnb_acc) return ws_launch_mode<NCH, NI, 2>(k, st);
    return ws_launch_mode<NCH, NI, 0>(k, st);
}


// ------------------------------------------------------------------------------------------ conv_ls_kernel
// Loader-specialised streaming kernel for the 3x3 layers whose weights do NOT fit in LDS (K = 9 Cin >= 1152: ResNet34 layer2 / layer3
// BasicBlocks via architectures/encoders.py:6-45, the deeper DecoderBlocks of architectures/base.py:7-37, and their data gradients).
// What conv_ws_kernel's clocks showed: an LDS-DMA piece costs its ISSUING wave ~250 cycles (index math + M0 + the instruction), so a
// wave that feeds the matrix pipe must not issue DMA - conv_glds_kernel interleaves both in every wave and its DMA time ADDS to its MFMA
// time.  Here the 8 waves of the one workgroup per CU are 4 LOADER waves and 4 MFMA waves (one of each per SIMD):
//   * loaders stream 32-channel chunks (18 x 18 halo rows of a 16 x 16 pixel tile + the chunk's weights of all 9 taps for the item's
//     32 NI output channels) through a ring of D chunk buffers that owns the whole LDS (D = 4 at NI = 1), with precomputed per-lane
//     source offsets (a piece is two 64-bit adds + the DMA), counted vmcnt and ONE raw s_barrier per chunk;
//   * MFMA waves compute 64 pixels x 32 NI channels each (2 x NI register blocking, 3-stage fragment ring), then run
//     ws_epilogue_tile while the loaders are already D - 1 chunks into the next item;
//   * a workgroup walks items (pixel tile, channel block) of ONE channel block, so the per-channel sums stay in registers, and a pixel
//     tile's channel blocks sit on one XCD (shared halo rows in that L2).  256 pixels x 32 channels per item is the traffic-minimal
//     shape for 256 CUs on the 9.66-GFLOP ResNet layers: 313 KB per CU against 956 KB with conv_mfma_kernel's 128 x 32 tiles at 3 per CU.
struct LsKP {
    const bf16_t* x; const bf16_t* w; bf16_t* y;
    const float* bias; const float* scale; const float* shift;
    int B, H, W, x_cs, y_cs, OH, OW, Cout;
    int tiles_x, tiles_y, ntiles, per_xcd, n_tiles, slots, nchunk;
    int min_dy, min_dx, pad_mode;
    int tap_off[9];
    int relu, accumulate;
    const bf16_t* bnb_y; const bf16_t* bnb_a; int bnb_cs, bnb_acs, bnb_relu;
    const float* bnb_mean; const float* bnb_invstd; const float* bnb_gamma; const float* bnb_beta;
    double* fin_acc; double* bnb_acc;
};

template <int NI, int MODE>
__global__ __launch_bounds__(512) void conv_ls_kernel(LsKP p) {
    typedef bf16_t T;
    constexpr int BN = 32 * NI, NT = 9, MI = 2;
    constexpr int HPC = 21, WPC = NT * BN / 16, PC = HPC + WPC;           // DMA pieces (1 KB) of one chunk: halo rows, then weights
    constexpr int NS = (PC + 3) / 4;                                    // DMA instructions per loader wave and chunk
    constexpr int D = NI == 1 ? 4 : 2;                                  // ring depth
    constexpr int H_BYTES = HPC * 1024, CH_BYTES = PC * 1024;
    constexpr int OFF_DUMMY = D * CH_BYTES, OFF_CONST = OFF_DUMMY + 1024;
    static_assert(OFF_CONST + 4 * BN * 4 <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int wm = wave & 3;
    const int khalf = lane >> 5, l31 = lane & 31;

    // ---- items of this workgroup: XCD x owns a contiguous range of pixel tiles; its workgroup j works on channel block j % n_tiles and
    // walks every `slots`-th tile of the range
    const int xcd = blockIdx.x & 7, jwg = blockIdx.x >> 3;
    if (jwg >= p.slots * p.n_tiles) return;
    const int nt = jwg % p.n_tiles, slot = jwg / p.n_tiles, n0 = nt * BN;
    const int t_lo = xcd * p.per_xcd;
    const int t_hi = min(t_lo + p.per_xcd, p.ntiles);
    const int n_items = (t_lo + slot < t_hi) ? (t_hi - t_lo - slot + p.slots - 1) / p.slots : 0;
    if (n_items <= 0) return;
    const int G = n_items * p.nchunk;                                    // chunks of this workgroup, in stream order
    struct TC { int b, oy0, ox0; };
    auto coords = [&](int k) {
        const int t = t_lo + slot + k * p.slots;
        TC c; const int tx = t % p.tiles_x; const int r = t / p.tiles_x;
        c.ox0 = tx << 4; c.oy0 = (r % p.tiles_y) << 4; c.b = r / p.tiles_y; return c;
    };
    if (tid < BN) {
        float* sc = reinterpret_cast<float*>(smem + OFF_CONST);
        if (MODE == 2) {
            if (p.bnb_acc) {
                const float mu = p.bnb_mean[n0 + tid], is = p.bnb_invstd[n0 + tid], k = p.bnb_gamma[n0 + tid] * is;
                sc[tid] = mu; sc[BN + tid] = is; sc[2 * BN + tid] = k; sc[3 * BN + tid] = p.bnb_beta[n0 + tid] - mu * k;
            }
        } else {
            sc[tid] = p.bias ? p.bias[n0 + tid] : 0.f; sc[BN + tid] = p.scale ? p.scale[n0 + tid] : 1.f; sc[2 * BN + tid] = p.shift ? p.shift[n0 + tid] : 0.f;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    float rs0[NI][4], rs1[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { rs0[j][e] = 0.f; rs1[j][e] = 0.f; }
    const bool sums = (MODE == 1 && p.fin_acc) || (MODE == 2 && p.bnb_acc);

    if (loader) {
        // ================================================================== loader waves
        const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_ws_zero);
        auto dma = [&](const void* src, int dst) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(smem + dst), 16, 0, 0);
        };
        // tile- and chunk-invariant part of this lane's source offset per slot (elements): weights only; the halo offsets depend on
        // the tile (image border) and are refreshed once per item
        int w_rel[NS], h_off[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int pi = wm + 4 * i;
            w_rel[i] = 0; h_off[i] = -1;
            if (pi >= HPC && pi < PC) {
                const int R = (pi - HPC) * 16 + (lane >> 2);               // row t * BN + n of the chunk's weight block
                const int t = R / BN, n = R - t * BN;
                w_rel[i] = (t * p.Cout + n0 + n) * 32 + ((lane ^ (R >> 2)) & 3) * 8;
            }
        }
        const T* xb = p.x;
        auto item_offsets = [&](int k) {                                  // halo source offsets of item k relative to its image
            const TC c = coords(k);
            xb = p.x + (int64_t)c.b * p.H * p.W * p.x_cs;
            const int iy0 = c.oy0 + p.min_dy, ix0 = c.ox0 + p.min_dx;
            const bool clamp = p.pad_mode != 0;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int pi = wm + 4 * i;
                if (pi < HPC) {
                    const int row = pi * 16 + (lane >> 2);
                    const int hy = (int)__umulhi((unsigned)row, 238609295u);      // row / 18
                    const int hx = row - hy * 18;
                    const int iy = iy0 + hy, ix = ix0 + hx;
                    const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
                    const bool inside = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                    const bool valid = (row < 324) & (clamp | inside);
                    h_off[i] = valid ? (iyc * p.W + ixc) * p.x_cs + ((lane ^ (row >> 2)) & 3) * 8 : -1;
                }
            }
        };
        int ik = 0, ic = 0, ig = 0;                                       // next chunk to issue: item, chunk, stream index
        auto issue_next = [&]() {
            const bool live = ig < G;
            if (live && ic == 0) item_offsets(ik);
            const int buf = (ig % D) * CH_BYTES;
            const T* wc = p.w + (int64_t)ic * NT * p.Cout * 32;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int pi = wm + 4 * i;
                const void* src = zp;
                int dst = OFF_DUMMY;
                if (live && pi < HPC) { if (h_off[i] >= 0) src = xb + (h_off[i] + ic * 32); dst = buf + pi * 1024; }
                else if (live && pi < PC) { src = wc + w_rel[i]; dst = buf + pi * 1024; }
                dma(src, dst);
            }
            if (live) { ++ig; if (++ic == p.nchunk) { ic = 0; ++ik; } }
        };
#pragma unroll 1
        for (int d = 0; d < D - 1; ++d) issue_next();
#pragma unroll 1
        for (int g = 0; g < G; ++g) {
            ws_wait_vm<(D - 2) * NS>();                                   // this wave's pieces of chunk g have landed
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                                  // ... everybody's; and the MFMA waves are done with chunk g - 1
            asm volatile("" ::: "memory");
            issue_next();                                                  // chunk g + D - 1 into the buffer chunk g - 1 released
        }
        ws_wait_vm<0>();                                                   // the trailing (dummy) pieces: no DMA may outlive the workgroup's LDS
    } else {
        // ================================================================== MFMA waves
        int pbase[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = wm * 64 + i * 32 + ws_perm(l31);
            pbase[i] = (m >> 4) * 18 + (m & 15);
        }
        const WsEpi ep = {p.y, p.bnb_y, p.bnb_a, p.y_cs, p.bnb_cs, p.bnb_acs, p.relu, p.accumulate, p.bnb_relu,
                          p.bias || p.scale || p.shift || p.relu, sums};
        struct Frag { u32x4 a[MI], b[NI]; };
        int g = 0;
#pragma unroll 1
        for (int k = 0; k < n_items; ++k) {
            f32x16 acc[MI][NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
            int a_addr[NT][MI], b_addr[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                int pb = pbase[i];
                asm volatile("" : "+v"(pb));                               // per item: not kept live across the epilogue
#pragma unroll
                for (int t = 0; t < NT; ++t) a_addr[t][i] = ws_swz(pb + p.tap_off[t], khalf);
            }
            {
                int lb = l31;
                asm volatile("" : "+v"(lb));
#pragma unroll
                for (int j = 0; j < NI; ++j) b_addr[j] = H_BYTES + ws_swz(j * 32 + lb, khalf);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
            for (int c = 0; c < p.nchunk; ++c, ++g) {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();                              // chunk g landed
                asm volatile("" ::: "memory");
                const unsigned char* hb = smem + (g % D) * CH_BYTES;
                auto load_frag = [&](int s, Frag& f) {                      // s = (tap, k-step), a constant after unrolling
                    const int t = s >> 1, hx = (s & 1) << 5;
#pragma unroll
                    for (int i = 0; i < MI; ++i) f.a[i] = *reinterpret_cast<const u32x4*>(hb + (a_addr[t][i] ^ hx));
#pragma unroll
                    for (int j = 0; j < NI; ++j) f.b[j] = *reinterpret_cast<const u32x4*>(hb + t * (BN * 64) + (b_addr[j] ^ hx));
                };
                auto mma_frag = [&](const Frag& f) {                         // operands swapped: rows of D = output channels
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.b[j]), __builtin_bit_cast(bf16x8, f.a[i]), acc[i][j], 0, 0, 0);
                };
                constexpr int NST = NT * 2;
                Frag f[3];
                load_frag(0, f[0]);
                load_frag(1, f[1]);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * (MI + NI), 0);
#pragma unroll
                for (int s2 = 0; s2 < NST; ++s2) {
                    if (s2 + 2 < NST) load_frag(s2 + 2, f[(s2 + 2) % 3]);
                    mma_frag(f[s2 % 3]);
                    if (s2 + 2 < NST) __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            const TC cc = coords(k);
            unsigned pix[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = wm * 64 + i * 32 + ws_perm(l31);
                pix[i] = (unsigned)((cc.b * p.OH + cc.oy0 + (m >> 4)) * p.OW + cc.ox0 + (m & 15));
            }
            ws_epilogue_tile<NI, MODE>(ep, acc, pix, n0, reinterpret_cast<const float*>(smem + OFF_CONST), rs0, rs1, khalf, l31);
        }
    }
    if (MODE != 0 && sums) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ws_sums_flush<NI, MODE, 4>(rs0, rs1, !loader, wm, reinterpret_cast<float*>(smem), n0, p.Cout, p.fin_acc, p.bnb_acc,
                                   nt == 0 ? (double)n_items * 256.0 : 0.0, khalf, l31);
    }
}

template <int NI, int MODE>
int ls_launch_mode(const LsKP& k, int wgs, hipStream_t st) {
    constexpr int BN = 32 * NI, PC = 21 + 9 * BN / 16, D = NI == 1 ? 4 : 2;
    constexpr int LDS = D * PC * 1024 + 1024 + 4 * BN * 4;
    auto kern = conv_ls_kernel<NI, MODE>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(512), LDS, st, k);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

template <int NI>
int ls_launch(const LsKP& k, int wgs, hipStream_t st) {
    if (k.fin_acc) return ls_launch_mode<NI, 1>(k, wgs, st);
    if (k.bnb_acc) return ls_launch_mode<NI, 2>(k, wgs, st);
    return ls_launch_mode<NI, 0>(k, wgs, st);
}

}  // namespace

// ---- host interface (conv_mfma.hip: salt_conv / salt_conv_stats_parts try this first)
// SALT_CONV_WS = 0: off unless asked for per launch; 1 (default): on for launches with at least half a tile per CU.
// cfg & 0xff == 9 asks for this kernel wherever it applies, whatever the tile count (tests); cfg >> 8 (if non-zero) caps the
// workgroups per XCD, so that small test tensors exercise the multi-tile pipeline.  A launch it does not apply to falls back to
// conv_mfma_kernel's own heuristic; salt_conv_kernel_id tells which kernel a launch gets.
bool conv_ws_eligible(const salt_conv_args* a) {
    static const int env = getenv("SALT_CONV_WS") ? atoi(getenv("SALT_CONV_WS")) : 1;
    if (!a || a->dtype != SALT_BF16) return false;
    const bool asked = (a->cfg & 0xff) == 9;
    if (a->cfg != 0 && !asked) return false;
    if (!asked && !env) return false;
    if (a->ntaps != 9 || a->in_step != 1 || a->out_step != 1 || a->out_oy || a->out_ox || a->nphase > 1) return false;
    if (a->strip || a->fold_top || a->fold_bottom || a->fold_left || a->fold_right) return false;
    if (a->stats || a->fin_ticket || a->bnb_partials || a->bnb_ticket) return false;
    const int Cin = a->x.C, Cout = a->y.C;
    if (!((Cin == 64 && (Cout == 64 || Cout == 32)) || (Cin == 32 && Cout == 64))) return false;
    if (a->x.cs % 8 || a->y.cs % 8 || ((reinterpret_cast<uintptr_t>(a->x.p) | reinterpret_cast<uintptr_t>(a->y.p) | reinterpret_cast<uintptr_t>(a->w)) & 15)) return false;
    if (a->OH != a->y.H || a->OW != a->y.W || a->OH % 16 || a->OW % 16 || a->x.B != a->y.B) return false;
    int min_dy = 1 << 30, max_dy = -(1 << 30), min_dx = 1 << 30, max_dx = -(1 << 30);
    for (int t = 0; t < 9; ++t) {
        min_dy = a->tap_dy[t] < min_dy ? a->tap_dy[t] : min_dy; max_dy = a->tap_dy[t] > max_dy ? a->tap_dy[t] : max_dy;
        min_dx = a->tap_dx[t] < min_dx ? a->tap_dx[t] : min_dx; max_dx = a->tap_dx[t] > max_dx ? a->tap_dx[t] : max_dx;
    }
    if (max_dy - min_dy != 2 || max_dx - min_dx != 2) return false;
    auto small = [](const salt_view& v) { return !v.p || (int64_t)v.B * v.H * v.W * v.cs < (int64_t)1 << 31; };
    if (!small(a->x) || !small(a->y) || !small(a->bnb_y) || !small(a->bnb_a)) return false;
    if (a->bnb_acc) {
        if (!view_ok(a->bnb_y) || a->bnb_y.B != a->y.B || a->bnb_y.H != a->y.H || a->bnb_y.W != a->y.W || a->bnb_y.C != Cout || a->bnb_y.cs % 8 ||
            (reinterpret_cast<uintptr_t>(a->bnb_y.p) & 15) || !a->bnb_mean || !a->bnb_invstd || !a->bnb_gamma || !a->bnb_beta) return false;
        if (a->bnb_a.p && (a->bnb_a.B != a->y.B || a->bnb_a.H != a->y.H || a->bnb_a.W != a->y.W || a->bnb_a.C != Cout || a->bnb_a.cs % 8 ||
                           (reinterpret_cast<uintptr_t>(a->bnb_a.p) & 15))) return false;
        if (a->fin_acc) return false;
    }
    const int64_t ntiles = (int64_t)a->x.B * (a->OH / 16) * (a->OW / 16);
    // a launch must give most CUs at least one tile; below that the per-CU weight load has nothing to amortise over
    if (!asked && ntiles < ws_cus() / 2) return false;
    return true;
}

int conv_ws_tiles(const salt_conv_args* a) { return a->x.B * (a->OH / 16) * (a->OW / 16); }

int conv_ws_launch(const salt_conv_args* a, hipStream_t st) {
    WsKP k;
    k.x = reinterpret_cast<const bf16_t*>(a->x.p); k.w = reinterpret_cast<const bf16_t*>(a->w); k.y = reinterpret_cast<bf16_t*>(a->y.p);
    k.bias = a->bias; k.scale = a->scale; k.shift = a->shift;
    k.B = a->x.B; k.H = a->x.H; k.W = a->x.W; k.x_cs = a->x.cs; k.y_cs = a->y.cs; k.OH = a->OH; k.OW = a->OW;
    k.tiles_x = a->OW / 16; k.tiles_y = a->OH / 16; k.ntiles = k.B * k.tiles_x * k.tiles_y;
    int min_dy = 1 << 30, min_dx = 1 << 30;
    for (int t = 0; t < 9; ++t) { min_dy = a->tap_dy[t] < min_dy ? a->tap_dy[t] : min_dy; min_dx = a->tap_dx[t] < min_dx ? a->tap_dx[t] : min_dx; }
    k.min_dy = min_dy; k.min_dx = min_dx; k.pad_mode = a->pad_mode;
    for (int t = 0; t < 9; ++t) k.tap_off[t] = (a->tap_dy[t] - min_dy) * 18 + (a->tap_dx[t] - min_dx);
    k.relu = a->relu; k.accumulate = a->accumulate;
    k.bnb_y = reinterpret_cast<const bf16_t*>(a->bnb_y.p); k.bnb_a = reinterpret_cast<const bf16_t*>(a->bnb_a.p);
    k.bnb_cs = a->bnb_y.cs; k.bnb_acs = a->bnb_a.cs; k.bnb_relu = a->bnb_relu;
    k.bnb_mean = a->bnb_mean; k.bnb_invstd = a->bnb_invstd; k.bnb_gamma = a->bnb_gamma; k.bnb_beta = a->bnb_beta;
    k.fin_acc = a->fin_acc; k.bnb_acc = a->bnb_acc;
    if (!a->bnb_acc) { k.bnb_y = nullptr; k.bnb_a = nullptr; }
    const int cus = ws_cus();
    k.per_xcd = cdiv(k.ntiles, 8);
    int wpx = cus / 8;                                   // workgroups per XCD: one per CU, fewer when the launch has fewer tiles
    if (wpx > k.per_xcd) wpx = k.per_xcd;
    const int cap = (a->cfg >> 8) & 0xff;
    if ((a->cfg & 0xff) == 9 && cap && wpx > cap) wpx = cap;
    k.wg_per_xcd = wpx;
    const int Cin = a->x.C, Cout = a->y.C;
    if (Cin == 64 && Cout == 64) return ws_launch<2, 2>(k, st);
    if (Cin == 64 && Cout == 32) return ws_launch<2, 1>(k, st);
    if (Cin == 32 && Cout == 64) return ws_launch<1, 2>(k, st);
    SALT_FAIL(SALT_E_BADARG, "conv_ws: channels %d -> %d", Cin, Cout);
}

// ---- conv_ls_kernel host side.  SALT_CONV_LS = 0: off unless asked for per launch (cfg & 0xff == 10); 1 (default): on.
// Returns NI (1 | 2) when the launch runs on conv_ls_kernel, else 0.
static int ls_common_ok(const salt_conv_args* a) {
    if (!a || a->dtype != SALT_BF16) return 0;
    if (a->ntaps != 9 || a->in_step != 1 || a->out_step != 1 || a->out_oy || a->out_ox || a->nphase > 1) return 0;
    if (a->strip || a->fold_top || a->fold_bottom || a->fold_left || a->fold_right) return 0;
    if (a->stats || a->fin_ticket || a->bnb_partials || a->bnb_ticket) return 0;
    const int Cin = a->x.C, Cout = a->y.C;
    if (Cin % 32 || Cin < 64 || Cout % 32) return 0;
    if (a->x.cs % 8 || a->y.cs % 8 || ((reinterpret_cast<uintptr_t>(a->x.p) | reinterpret_cast<uintptr_t>(a->y.p) | reinterpret_cast<uintptr_t>(a->w)) & 15)) return 0;
    if (a->OH != a->y.H || a->OW != a->y.W || a->OH % 16 || a->OW % 16 || a->x.B != a->y.B) return 0;
    int min_dy = 1 << 30, max_dy = -(1 << 30), min_dx = 1 << 30, max_dx = -(1 << 30);
    for (int t = 0; t < 9; ++t) {
        min_dy = a->tap_dy[t] < min_dy ? a->tap_dy[t] : min_dy; max_dy = a->tap_dy[t] > max_dy ? a->tap_dy[t] : max_dy;
        min_dx = a->tap_dx[t] < min_dx ? a->tap_dx[t] : min_dx; max_dx = a->tap_dx[t] > max_dx ? a->tap_dx[t] : max_dx;
    }
    if (max_dy - min_dy != 2 || max_dx - min_dx != 2) return 0;
    auto small = [](const salt_view& v) { return !v.p || (int64_t)v.B * v.H * v.W * v.cs < (int64_t)1 << 31; };
    if (!small(a->x) || !small(a->y) || !small(a->bnb_y) || !small(a->bnb_a)) return 0;
    if (a->bnb_acc) {
        if (!view_ok(a->bnb_y) || a->bnb_y.B != a->y.B || a->bnb_y.H != a->y.H || a->bnb_y.W != a->y.W || a->bnb_y.C != Cout || a->bnb_y.cs % 8 ||
            (reinterpret_cast<uintptr_t>(a->bnb_y.p) & 15) || !a->bnb_mean || !a->bnb_invstd || !a->bnb_gamma || !a->bnb_beta) return 0;
        if (a->bnb_a.p && (a->bnb_a.B != a->y.B || a->bnb_a.H != a->y.H || a->bnb_a.W != a->y.W || a->bnb_a.C != Cout || a->bnb_a.cs % 8 ||
                           (reinterpret_cast<uintptr_t>(a->bnb_a.p) & 15))) return 0;
        if (a->fin_acc) return 0;
    }
    return 1;
}

int conv_ls_variant(const salt_conv_args* a) {
    static const int env = getenv("SALT_CONV_LS") ? atoi(getenv("SALT_CONV_LS")) : 1;
    if (!ls_common_ok(a)) return 0;
    const bool asked = (a->cfg & 0xff) == 10;
    if (a->cfg != 0 && !asked) return 0;
    if (!asked && !env) return 0;
    const int Cout = a->y.C;
    int wpx = ws_cus() / 8;
    const int64_t ntiles = (int64_t)a->x.B * (a->OH / 16) * (a->OW / 16);
    const int force_ni = asked ? (a->cfg >> 16) & 3 : 0;                  // tests: cfg = 10 | cap << 8 | NI << 16
    // 64 output channels per item where that still gives (nearly) every CU an item, else 32
    int ni = (Cout % 64 == 0 && ntiles * (Cout / 64) * 10 >= (int64_t)ws_cus() * 9) ? 2 : 1;
    if (force_ni == 1 || force_ni == 2) ni = force_ni;
    if (ni == 2 && Cout % 64) ni = 1;
    if (Cout / (32 * ni) > wpx) return 0;                                 // more channel blocks than workgroups per XCD
    if (!asked && ntiles * (Cout / (32 * ni)) < ws_cus() / 2) return 0;   // too few items to fill the chip
    return ni;
}

int conv_ls_launch(const salt_conv_args* a, hipStream_t st) {
    const int ni = conv_ls_variant(a);
    if (!ni) SALT_FAIL(SALT_E_UNSUPPORTED, "conv_ls: not applicable");
    LsKP k;
    k.x = reinterpret_cast<const bf16_t*>(a->x.p); k.w = reinterpret_cast<const bf16_t*>(a->w); k.y = reinterpret_cast<bf16_t*>(a->y.p);
    k.bias = a->bias; k.scale = a->scale; k.shift = a->shift;
    k.B = a->x.B; k.H = a->x.H; k.W = a->x.W; k.x_cs = a->x.cs; k.y_cs = a->y.cs; k.OH = a->OH; k.OW = a->OW; k.Cout = a->y.C;
    k.tiles_x = a->OW / 16; k.tiles_y = a->OH / 16; k.ntiles = k.B * k.tiles_x * k.tiles_y;
    k.nchunk = a->x.C / 32;
    int min_dy = 1 << 30, min_dx = 1 << 30;
    for (int t = 0; t < 9; ++t) { min_dy = a->tap_dy[t] < min_dy ? a->tap_dy[t] : min_dy; min_dx = a->tap_dx[t] < min_dx ? a->tap_dx[t] : min_dx; }
    k.min_dy = min_dy; k.min_dx = min_dx; k.pad_mode = a->pad_mode;
    for (int t = 0; t < 9; ++t) k.tap_off[t] = (a->tap_dy[t] - min_dy) * 18 + (a->tap_dx[t] - min_dx);
    k.relu = a->relu; k.accumulate = a->accumulate;
    k.bnb_y = reinterpret_cast<const bf16_t*>(a->bnb_y.p); k.bnb_a = reinterpret_cast<const bf16_t*>(a->bnb_a.p);
    k.bnb_cs = a->bnb_y.cs; k.bnb_acs = a->bnb_a.cs; k.bnb_relu = a->bnb_relu;
    k.bnb_mean = a->bnb_mean; k.bnb_invstd = a->bnb_invstd; k.bnb_gamma = a->bnb_gamma; k.bnb_beta = a->bnb_beta;
    k.fin_acc = a->fin_acc; k.bnb_acc = a->bnb_acc;
    if (!a->bnb_acc) { k.bnb_y = nullptr; k.bnb_a = nullptr; }
    int wpx = ws_cus() / 8;
    const int cap = (a->cfg >> 8) & 0xff;
    k.n_tiles = k.Cout / (32 * ni);
    if ((a->cfg & 0xff) == 10 && cap && wpx > cap) wpx = cap > k.n_tiles ? cap : k.n_tiles;      // (never fewer than one workgroup per channel block)
    if (k.n_tiles > wpx) SALT_FAIL(SALT_E_UNSUPPORTED, "conv_ls: %d channel blocks for %d workgroups per XCD", k.n_tiles, wpx);
    k.per_xcd = cdiv(k.ntiles, 8);
    k.slots = wpx / k.n_tiles;
    if (k.slots > k.per_xcd) k.slots = k.per_xcd;
    const int wgs = k.slots * k.n_tiles * 8;
    return ni == 2 ? ls_launch<2>(k, wgs, st) : ls_launch<1>(k, wgs, st);
}

extern "C" int salt_debug_ws_clk(unsigned long long* host_out, int n) {
#if SALT_WS_CLK
    if (n > 256 * 48) n = 256 * 48;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ws_clk), (size_t)n * sizeof(unsigned long long));
#else
    (void)host_out; (void)n;
    return SALT_E_UNSUPPORTED;
#endif
}
