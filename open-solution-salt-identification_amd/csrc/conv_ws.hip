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
//   * First version of this kernel (kept in git history, profiles/r03_ws_clocks.txt): lane = channel, transposition of the tile
//     through wave-private LDS slices.  In-kernel clocks: epilogue 16.7 k cycles per 256 x 64 tile against 7.2 k for its MFMAs
//     (register spills around 64 ds_write_b16 + the store loop) - 25 us per 64 -> 64 @64x64 launch against 17.7 for conv_mfma_kernel.
//
// Scope (host: conv_ws_eligible): bf16, 9 taps inside a 3 x 3 window, unit steps, Cin in {32, 64}, Cout in {32, 64}, output grid a
// multiple of 16 x 16 and equal to y, zero or replicate (clamp) padding, bias / folded BN / ReLU / accumulate / statistics shards /
// BatchNorm-backward shards.  Everything else stays on conv_mfma_kernel.
#include <cstdlib>
#include <type_traits>
#include "common.h"

#ifndef SALT_LS_D1
#define SALT_LS_D1 4             // chunk-ring depth of conv_ls_kernel<NI = 1> (3 leaves 43 KB of LDS to a neighbour: DESIGN 10)
#endif
#ifndef SALT_LS_NLW
#define SALT_LS_NLW 4            // loader waves of conv_ls_kernel (8 measured)
#endif
#ifndef SALT_LS_PREFETCH
#define SALT_LS_PREFETCH 1       // conv_ls_kernel MODE 2: the (+)= / BatchNorm-backward operand tiles of an item are requested BEFORE its chunk loop (0: in the epilogue)
#endif
#ifndef SALT_LS_ABLATE
#define SALT_LS_ABLATE 0         // conv_ls_kernel timing ablations (SRC=conv_ws tools/build_variant.sh <name> -DSALT_LS_ABLATE=n; results are wrong): 1 no fragment reads / MFMAs, 2 no DMA, 4 no epilogue
#endif
#ifndef SALT_WS_CLK
#define SALT_WS_CLK 0            // 1: per-workgroup s_memtime stamps into g_ws_clk (tools/ws_clocks.py; timing build only)
#endif

namespace {

struct WsKP {
    const bf16_t* x; const bf16_t* w; bf16_t* y;
    const float* bias; const float* scale; const float* shift;
    int B, H, W, x_cs, y_cs, OH, OW;     // x is [B,H,W,*]; OH x OW: the launch's output grid (the extended grid of a fused fold)
    int yH, yW, fold_top, ox_shift;      // y is [B,yH,yW,*]: grid row oy -> y row oy - fold_top; tile columns start at -ox_shift
    int Cout, n_tiles, slots;            // channel blocks of 32 NI; workgroup j of an XCD: block j % n_tiles, every `slots`-th tile
    int tiles_x, tiles_y, ntiles, per_xcd;
    int min_dy, min_dx, pad_mode;
    int tap_off[9];
    int relu, accumulate;
    const bf16_t* bnb_y; const bf16_t* bnb_a; int bnb_cs, bnb_acs, bnb_relu;
    const float* bnb_mean; const float* bnb_invstd; const float* bnb_gamma; const float* bnb_beta;
    double* fin_acc;             // [8][2 Cout + 1] forward statistics shards (nullptr: off)
    double* bnb_acc;             // [8][2][Cout] BatchNorm-backward shards (nullptr: off)
    const bf16_t* res; int res_cs;   // residual epilogue (salt_conv_args.res), MODE 0 only
    long long y_plane;               // salt_conv_args.y_plane: channel block b of y is the dense plane y + b * y_plane (0: channel-interleaved rows)
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

__device__ __forceinline__ int ws_perm_inv(int q) { return (int)((0x74216530u >> ((q >> 2) * 4)) & 0xfu) * 4 + (q & 3); }      // ws_perm(ws_perm_inv(q)) == q

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
    const bf16_t* res; int res_cs;        // MODE 0 residual epilogue (salt_conv_args.res): y = relu?(bf16(affine) + res)
};
// The operand tiles of the (+)= / BatchNorm-backward epilogue (MODE 2) of a whole wave tile: 16-byte pieces at the lane's store addresses.
// conv_ls_kernel requests them before the item's chunk loop (its MFMA waves run loop and epilogue back to back: requested in the
// epilogue, the matrix pipe waits for 3 x 32 KB of HBM reads per item; the whole-CU workgroup leaves each wave 256 VGPRs to hold them).
template <int NI> struct WsOps { u32x4 oldv[2][NI][2], yv[2][NI][2], av[2][NI][2]; };
template <int NI, bool OLD>
__device__ __forceinline__ void ws_prefetch_operands(const WsEpi& p, const unsigned (&pix)[2], int n0, int khalf, WsOps<NI>& o) {
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const unsigned c = n0 + 8 * khalf + 32 * j + 16 * gp;
                if (OLD && p.accumulate) o.oldv[i][j][gp] = *reinterpret_cast<const u32x4*>(p.y + (pix[i] * (unsigned)p.y_cs + c));
                if (OLD && p.res) o.oldv[i][j][gp] = *reinterpret_cast<const u32x4*>(p.res + (pix[i] * (unsigned)p.res_cs + c));      // (host: never with accumulate)
                if (p.sums) {
                    o.yv[i][j][gp] = *reinterpret_cast<const u32x4*>(p.bnb_y + (pix[i] * (unsigned)p.bnb_cs + c));
                    if (p.bnb_a) o.av[i][j][gp] = *reinterpret_cast<const u32x4*>(p.bnb_a + (pix[i] * (unsigned)p.bnb_acs + c));
                }
            }
}

// Per-lane geometry of the wave tile: which of the lane's two pixels are stored (ragged grids, pad-ring pixels of a fused fold), and
// - FOLD, the data gradient of a replicate-padded convolution on the extended grid (salt_conv_args.fold_top / fold_right = 2, tile
// columns right-aligned) - where the pad-ring values it must add come from.  The ring pixels a pixel folds share its WAVE: the two
// ring columns are tile columns 14 / 15 of the edge pixel's own row, the two ring rows are rows 0 / 1 of the wave that holds image
// row 0 in row 2.  So the fold is a lane permutation (ds_bpermute) of fp32 accumulators: first columns 14, 15 onto column 13 in every
// row (ring rows included), then rows 0, 1 (sub-tile 0) onto row 2 (sub-tile 1) - which also carries the corner.
struct WsLaneGeo {
    bool valid[2];
    bool do_right, do_top;        // wave-uniform: the tile is in the last tile column / this wave holds image row 0 and the pad rows
    bool fr_on, ft_on;            // this lane's pixel takes the right / (sub-tile 1 only) the top fold
    int fr_src0, fr_src1;         // lanes of tile columns 14 / 15 of this lane's row (same sub-tile)
    int ft_src0, ft_src1;         // lanes of rows 0 / 1 of this lane's column in sub-tile 0
};

template <int NI, int MODE, bool FOLD>
__device__ __forceinline__ void ws_epilogue_tile(const WsEpi& p, f32x16 (&acc)[2][NI], const unsigned (&pix)[2], const WsLaneGeo& geo, int n0,
                                                 const float* cst, float (&rs0)[NI][4], float (&rs1)[NI][4], int khalf, int l31,
                                                 const WsOps<NI>* pre = nullptr, bool pre_old = true) {
    typedef bf16_t T;
    constexpr int MI = 2, BN = 32 * NI;
    typedef const __attribute__((address_space(3))) float* cst_lds_t;
    cst_lds_t cst_l = (cst_lds_t)cst + 8 * khalf;                              // this lane's half of every 16-channel group (MODE 2 reads below)
    if (MODE == 2) asm volatile("" : "+v"(cst_l));
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        // operand tiles of the (+)= / BatchNorm-backward epilogue: 16-byte pieces at this lane's store addresses
        u32x4 oldv[MI][2], yv[MI][2], av[MI][2];
        auto load_operands = [&](int i) {                                       // i is a constant after unrolling
            if (!geo.valid[i]) return;
            if (pre) {                                                          // (MODE 2, conv_ls_kernel: requested before the chunk loop)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    if (pre_old) oldv[i][gp] = pre->oldv[i][j][gp];
                    else if (p.accumulate) oldv[i][gp] = *reinterpret_cast<const u32x4*>(p.y + (pix[i] * (unsigned)p.y_cs + n0 + 8 * khalf + 32 * j + 16 * gp));
                    yv[i][gp] = pre->yv[i][j][gp]; av[i][gp] = pre->av[i][j][gp];
                }
                return;
            }
            if (MODE != 1 && p.accumulate) {
#pragma unroll
                for (int gp = 0; gp < 2; ++gp)
                    oldv[i][gp] = *reinterpret_cast<const u32x4*>(p.y + (pix[i] * (unsigned)p.y_cs + n0 + 8 * khalf + 32 * j + 16 * gp));
            }
            if (MODE == 0 && p.res) {                                           // (host: never together with accumulate)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp)
                    oldv[i][gp] = *reinterpret_cast<const u32x4*>(p.res + (pix[i] * (unsigned)p.res_cs + n0 + 8 * khalf + 32 * j + 16 * gp));
            }
            if (MODE == 2 && p.sums) {
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    yv[i][gp] = *reinterpret_cast<const u32x4*>(p.bnb_y + (pix[i] * (unsigned)p.bnb_cs + n0 + 8 * khalf + 32 * j + 16 * gp));
                    if (p.bnb_a) av[i][gp] = *reinterpret_cast<const u32x4*>(p.bnb_a + (pix[i] * (unsigned)p.bnb_acs + n0 + 8 * khalf + 32 * j + 16 * gp));
                }
            }
        };
        if (!FOLD) {                                                            // (FOLD holds both sub-tiles in fp32 across the lane
#pragma unroll                                                                  //  permutation: its operands are requested per sub-tile below)
            for (int i = 0; i < MI; ++i) load_operands(i);
        }
        float v[MI][16];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[i][r] = acc[i][j][r];
            if (MODE != 2 && p.has_affine) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {                                   // 4 consecutive channels 32 j + 8 q + 4 khalf ..
                    const int ch0 = 32 * j + 8 * q + 4 * khalf;
                    const f32x4 bi = *reinterpret_cast<const f32x4*>(cst + ch0), sc = *reinterpret_cast<const f32x4*>(cst + BN + ch0),
                                sh = *reinterpret_cast<const f32x4*>(cst + 2 * BN + ch0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float t = (v[i][4 * q + e] + bi[e]) * sc[e] + sh[e];
                        if (p.relu && !(MODE == 0 && p.res)) t = fmaxf(t, 0.f);  // with a residual the ReLU follows the add
                        v[i][4 * q + e] = t;
                    }
                }
            }
        }
        if (FOLD) {
            // (a ds_bpermute costs the wave ~45 cycles here: only the tiles / waves that hold pad-ring pixels run them - in-kernel
            // clocks, profiles/r03_ws_clocks.txt: 12.4 k cycles per tile epilogue with the permutations unconditional, 3.7 k without)
            const int hb = khalf << 5;                                          // the source lane holds the same channels: same half-wave
            if (geo.do_right) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float add = __shfl(v[i][r], hb + geo.fr_src0) + __shfl(v[i][r], hb + geo.fr_src1);
                        if (geo.fr_on) v[i][r] += add;
                    }
            }
            if (geo.do_top) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float add = __shfl(v[0][r], hb + geo.ft_src0) + __shfl(v[0][r], hb + geo.ft_src1);
                    if (geo.ft_on) v[1][r] += add;
                }
            }
        }
        float t0[16], t1[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) { t0[e] = 0.f; t1[e] = 0.f; }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const unsigned yo = pix[i] * (unsigned)p.y_cs + n0 + 8 * khalf + 32 * j;     // + 16 gp: this lane's piece gp of block j
            if (FOLD) load_operands(i);
            if (MODE == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { t0[r] += v[i][r]; t1[r] += v[i][r] * v[i][r]; }      // (host: MODE 1 launches have whole tiles only)
            }
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                // channel groups 2 gp (registers 8 gp .. 8 gp + 3) and 2 gp + 1 of both half-waves -> one 16-byte piece per lane:
                // lanes 0-31 channels 32 j + 16 gp + 0..7, lanes 32-63 channels 32 j + 16 gp + 8..15 of the same pixel
                const unsigned ax = f2bf_pk(v[i][8 * gp + 0], v[i][8 * gp + 1]), ay = f2bf_pk(v[i][8 * gp + 2], v[i][8 * gp + 3]);
                const unsigned bx = f2bf_pk(v[i][8 * gp + 4], v[i][8 * gp + 5]), by = f2bf_pk(v[i][8 * gp + 6], v[i][8 * gp + 7]);
                const auto rx = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
                const auto ry = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
                u32x4 stored = {rx[0], ry[0], rx[1], ry[1]};
                if (!geo.valid[i]) continue;                                    // (after the swap: its partner lane may be valid)
                if (MODE != 1 && p.accumulate) {
                    float f8[8], o8[8];
                    unpack16<T>(stored, f8); unpack16<T>(oldv[i][gp], o8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f8[e] += o8[e];
                    stored = pack16<T>(f8);
                }
                if (MODE == 0 && p.res) {
                    float f8[8], o8[8];
                    unpack16<T>(stored, f8); unpack16<T>(oldv[i][gp], o8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { f8[e] += o8[e]; if (p.relu) f8[e] = fmaxf(f8[e], 0.f); }
                    stored = pack16<T>(f8);
                }
                *reinterpret_cast<u32x4*>(p.y + (yo + 16 * gp)) = stored;
                if (MODE == 2 && p.sums) {
                    // the constants are read through ONE LDS base register + immediates: as generic-pointer arithmetic the compiler kept a
                    // register per (array, j, gp) address - the table sits above the 64 KB an immediate reaches - spilled them at 256 VGPRs and
                    // reloaded each from scratch behind `s_waitcnt vmcnt(0)`, i.e. behind the tile's own stores, eight times per tile
                    const int ch0 = 32 * j + 16 * gp;
                    const cst_lds_t cl = cst_l;
                    float mu[8], is[8], gq[8], yc[8];
                    *reinterpret_cast<f32x4*>(mu) = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(cl + ch0);
                    *reinterpret_cast<f32x4*>(mu + 4) = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(cl + ch0 + 4);
                    *reinterpret_cast<f32x4*>(is) = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(cl + BN + ch0);
                    *reinterpret_cast<f32x4*>(is + 4) = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(cl + BN + ch0 + 4);
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
                        *reinterpret_cast<f32x4*>(ks) = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(cl + 2 * BN + ch0);
                        *reinterpret_cast<f32x4*>(ks + 4) = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(cl + 2 * BN + ch0 + 4);
                        *reinterpret_cast<f32x4*>(sh) = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(cl + 3 * BN + ch0);
                        *reinterpret_cast<f32x4*>(sh + 4) = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(cl + 3 * BN + ch0 + 4);
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
template <int NCH, int NI, int MODE, bool FOLD>
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

    // ---- tiles of this workgroup: XCD x (= block id % 8, where consecutive block ids go) owns a contiguous range of pixel tiles, so
    // that neighbouring tiles share their halo rows through that XCD's L2; its workgroup j keeps the weights of channel block
    // j % n_tiles resident and walks every `slots`-th tile of the range
    const int xcd = blockIdx.x & 7, jwg = blockIdx.x >> 3;
    const int nt = jwg % p.n_tiles, slot = jwg / p.n_tiles, n0 = nt * BN;
    const int t_lo = xcd * p.per_xcd;
    const int t_hi = min(t_lo + p.per_xcd, p.ntiles);
    const int n_my = (t_lo + slot < t_hi) ? (t_hi - t_lo - slot + p.slots - 1) / p.slots : 0;
    if (n_my <= 0) return;
    auto tile_of = [&](int k) { return t_lo + slot + k * p.slots; };
    struct TC { int b, oy0, ox0; };
    auto coords = [&](int t) {
        TC c; const int tx = t % p.tiles_x; const int r = t / p.tiles_x;
        c.ox0 = (tx << 4) - p.ox_shift; c.oy0 = (r % p.tiles_y) << 4; c.b = r / p.tiles_y; return c;
    };
#if SALT_WS_CLK
    unsigned long long clk[24]; int nclk = 0;
    auto stamp = [&]() { if (nclk < 22) clk[nclk++] = __builtin_readcyclecounter(); };
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
            const int q = wave + 8 * i;                                   // piece q = rows 16 q .. 16 q + 15 of this block's [chunk][tap][BN] rows
            const bool real = q < WP;
            const int R = q * 16 + (lane >> 2);
            const int ct = R / BN, n = R - ct * BN;                       // (chunk, tap) and channel of the row inside the packed [chunk][tap][Cout][32]
            const int cslot = (lane ^ (R >> 2)) & 3;
            const unsigned char* src = real ? reinterpret_cast<const unsigned char*>(p.w + ((ct * p.Cout + n0 + n) * 32 + cslot * 8)) : zp;
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
                const float mu = p.bnb_mean[n0 + tid], is = p.bnb_invstd[n0 + tid], k = p.bnb_gamma[n0 + tid] * is;
                sc[tid] = mu; sc[BN + tid] = is; sc[2 * BN + tid] = k; sc[3 * BN + tid] = p.bnb_beta[n0 + tid] - mu * k;
            }
        } else {
            sc[tid] = p.bias ? p.bias[n0 + tid] : 0.f; sc[BN + tid] = p.scale ? p.scale[n0 + tid] : 1.f; sc[2 * BN + tid] = p.shift ? p.shift[n0 + tid] : 0.f;
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
        // phase ran at 64 % of the MFMA rate alone on the CU (in-kernel clocks, profiles/r03_ws_clocks.txt)
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
    // planar y (y_plane != 0): the epilogue addresses y + pixel * y_cs + n0 + ...; with planes of BN channels that is plane nt at offset 0
    const WsEpi ep = {p.y_plane ? p.y + ((long long)nt * p.y_plane - n0) : p.y, p.bnb_y, p.bnb_a, p.y_cs, p.bnb_cs, p.bnb_acs, p.relu, p.accumulate, p.bnb_relu,
                      p.bias || p.scale || p.shift || p.relu, sums, p.res, p.res_cs};

    // epilogue of the tile in `acc` (computed by this wave one phase ago); issues the halo DMA of `next` (if any) into buffer g
    auto epilogue = [&](const TC& c, int g, bool has_next, const TC& next) {
        unsigned pix[MI];
        WsLaneGeo geo = {{true, true}, false, false, false, false, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = wm * 64 + i * 32 + ws_perm(l31);
            const int oy = c.oy0 + (m >> 4), ox = c.ox0 + (m & 15);
            const int iy = oy - p.fold_top;                                   // y row (pad-ring rows of a fused fold: < 0, never stored)
            geo.valid[i] = ((unsigned)iy < (unsigned)p.yH) & ((unsigned)ox < (unsigned)p.yW);
            pix[i] = (unsigned)((c.b * p.yH + iy) * p.yW + ox);
        }
        if (FOLD) {
            const int q = ws_perm(l31), rs = q >> 4, col = q & 15;             // the lane's row (0 | 1) inside a sub-tile and its tile column
            geo.fr_src0 = ws_perm_inv(rs * 16 + 14); geo.fr_src1 = ws_perm_inv(rs * 16 + 15);
            geo.ft_src0 = ws_perm_inv(col); geo.ft_src1 = ws_perm_inv(16 + col);
            geo.do_right = c.ox0 + 16 == p.OW;                                 // the last tile column holds the pad columns (tile columns 14, 15)
            geo.do_top = wm == 0 && c.oy0 == 0;                                // wave 0 of the first tile row holds the pad rows (rows 0, 1)
            geo.fr_on = col == 13 && geo.do_right;                             // image column yW - 1
            geo.ft_on = rs == 0 && geo.do_top;                                 // (sub-tile 1) image row 0 = grid row 2
        }
        ws_epilogue_tile<NI, MODE, FOLD>(ep, acc, pix, geo, n0, reinterpret_cast<const float*>(smem + OFF_CONST), rs0, rs1, khalf, l31);
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
        ws_sums_flush<NI, MODE, 8>(rs0, rs1, true, wave, reinterpret_cast<float*>(smem), n0, p.Cout, p.fin_acc, p.bnb_acc,
                                   nt == 0 ? (double)n_my * 256.0 : 0.0, khalf, l31);
    }
#if SALT_WS_CLK
    if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 256) {
        unsigned long long* o = g_ws_clk + (blockIdx.x * 2 + grp) * 24;
        for (int i = 0; i < 22; ++i) o[i] = i < nclk ? clk[i] : 0ull;
        o[22] = (unsigned long long)n_my;                        // tiles of this workgroup
        o[23] = __builtin_readcyclecounter();                    // kernel end
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

template <int NCH, int NI, int MODE, bool FOLD>
int ws_launch_mode(const WsKP& k, hipStream_t st) {
    constexpr int BN = 32 * NI, HP = NCH * 21, WP = NCH * 9 * BN / 16;
    constexpr int LDS = WP * 1024 + 2 * HP * 1024 + 1024 + 4 * BN * 4;
    static_assert(LDS <= 160 * 1024 && 8 * 2 * BN * 4 <= WP * 1024, "LDS budget");
    auto kern = conv_ws_kernel<NCH, NI, MODE, FOLD>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(k.slots * k.n_tiles * 8)), dim3(512), LDS, st, k);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

template <int NCH, int NI>
int ws_launch(const WsKP& k, hipStream_t st) {
    const bool fold = k.fold_top != 0;
    if (k.fin_acc) return ws_launch_mode<NCH, NI, 1, false>(k, st);
    if (k.bnb_acc) return fold ? ws_launch_mode<NCH, NI, 2, true>(k, st) : ws_launch_mode<NCH, NI, 2, false>(k, st);
    return fold ? ws_launch_mode<NCH, NI, 0, true>(k, st) : ws_launch_mode<NCH, NI, 0, false>(k, st);
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
    const bf16_t* res; int res_cs;
    long long x_plane;               // salt_conv_args.x_plane: chunk c is the half (c & 1) of the dense 64-channel plane x + (c >> 1) * x_plane
};

// NLW loader waves (waves 4 .. 4 + NLW - 1) + 4 MFMA waves (waves 0 .. 3).
// MT = 2 (round 5, MODE 0, NI = 2: the eval-mode layers with hundreds of input channels): an item is TWO pixel tiles of the workgroup's
// walk under ONE stream of weight chunks - a 64-channel chunk of weights is 36 of the 57 KB a chunk moves, the loader waves' issue rate
// (and a ring of two 57 KB chunks against the DMA latency) is what held these layers at 46 - 50 % of the MFMA peak; per (tap, k-step) stage
// 6 fragment reads feed 8 MFMAs instead of 4 : 4, and a chunk is 144 MFMAs per wave between two barriers instead of 72.
template <int NI, int MODE, int NLW = SALT_LS_NLW, int MT = 1>
__global__ __launch_bounds__(256 + 64 * NLW) void conv_ls_kernel(LsKP p) {
    typedef bf16_t T;
    constexpr int BN = 32 * NI, NT = 9, MI = 2;
    constexpr int HPC = 21, WPC = NT * BN / 16, PC = MT * HPC + WPC;      // DMA pieces (1 KB) of one chunk: halo rows (of MT tiles), then weights
    constexpr int NS = (PC + NLW - 1) / NLW;                            // DMA instructions per loader wave and chunk
    constexpr int D = NI == 1 ? SALT_LS_D1 : 2;                                  // ring depth
    constexpr int HT_BYTES = HPC * 1024, H_BYTES = MT * HT_BYTES, CH_BYTES = PC * 1024;
    static_assert(MT == 1 || (MODE == 0 && NI == 2), "two-tile items: plain epilogue, 64-channel blocks");
    constexpr int OFF_DUMMY = D * CH_BYTES, OFF_CONST = OFF_DUMMY + 1024;
    static_assert(OFF_CONST + 4 * BN * 4 <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int wm = wave & 3;                                            // MFMA waves: pixel quarter of the tile
    const int lw = wave - 4;                                            // loader waves: piece lane of the chunk
    const int khalf = lane >> 5, l31 = lane & 31;

    // ---- items of this workgroup: XCD x owns a contiguous range of pixel tiles; its workgroup j works on channel block j % n_tiles and
    // walks every `slots`-th tile of the range
    const int xcd = blockIdx.x & 7, jwg = blockIdx.x >> 3;
    if (jwg >= p.slots * p.n_tiles) return;
    const int nt = jwg % p.n_tiles, slot = jwg / p.n_tiles, n0 = nt * BN;
    const int t_lo = xcd * p.per_xcd;
    const int t_hi = min(t_lo + p.per_xcd, p.ntiles);
    const int n_my = (t_lo + slot < t_hi) ? (t_hi - t_lo - slot + p.slots - 1) / p.slots : 0;      // pixel tiles of this workgroup
    const int n_items = (n_my + MT - 1) / MT;
    if (n_items <= 0) return;
    const int G = n_items * p.nchunk;                                    // chunks of this workgroup, in stream order
    struct TC { int b, oy0, ox0; };
    auto coords = [&](int k) {                                           // k-th pixel tile of the walk (item k / MT, tile k % MT of it)
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
            const int pi = lw + NLW * i;
            w_rel[i] = 0; h_off[i] = -1;
            if (pi >= MT * HPC && pi < PC) {
                const int R = (pi - MT * HPC) * 16 + (lane >> 2);          // row t * BN + n of the chunk's weight block
                const int t = R / BN, n = R - t * BN;
                w_rel[i] = (t * p.Cout + n0 + n) * 32 + ((lane ^ (R >> 2)) & 3) * 8;
            }
        }
        const T* xb = p.x;
        const T* xb1 = p.x;                                               // MT = 2: the image of the item's second tile
        auto item_offsets = [&](int k) {                                  // halo source offsets of item k relative to its image(s)
            const bool clamp = p.pad_mode != 0;
            TC cm[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) cm[m] = coords(MT == 1 ? k : min(MT * k + m, n_my - 1));
            xb = p.x + (int64_t)cm[0].b * p.H * p.W * p.x_cs;
            if (MT == 2) xb1 = p.x + (int64_t)cm[MT - 1].b * p.H * p.W * p.x_cs;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int pi = lw + NLW * i;
                if (pi < MT * HPC) {
                    const int m = MT == 1 ? 0 : (pi >= HPC ? 1 : 0), hp = pi - m * HPC;
                    const TC c = cm[m];
                    const int iy0 = c.oy0 + p.min_dy, ix0 = c.ox0 + p.min_dx;
                    const int row = hp * 16 + (lane >> 2);
                    const int hy = (int)__umulhi((unsigned)row, 238609295u);      // row / 18
                    const int hx = row - hy * 18;
                    const int iy = iy0 + hy, ix = ix0 + hx;
                    const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
                    const bool inside = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                    const bool valid = (row < 324) & (clamp | inside) & (MT == 1 || MT * k + m < n_my);
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
            const long long coff = p.x_plane ? (long long)(ic >> 1) * p.x_plane + (ic & 1) * 32 : (long long)ic * 32;
            const T* xc = xb + coff;                                          // wave-uniform
            const T* xc1 = xb1 + coff;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int pi = lw + NLW * i;
                const void* src = zp;
                int dst = OFF_DUMMY;
                if (live && pi < MT * HPC) { if (h_off[i] >= 0) src = ((MT == 2 && pi >= HPC) ? xc1 : xc) + h_off[i]; dst = buf + pi * 1024; }
                else if (live && pi < PC) { src = wc + w_rel[i]; dst = buf + pi * 1024; }
                if (!(SALT_LS_ABLATE & 2)) dma(src, dst);
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
                          p.bias || p.scale || p.shift || p.relu, sums, p.res, p.res_cs};
        struct Frag { u32x4 a[MT][MI], b[NI]; };
        int g = 0;
#pragma unroll 1
        for (int k = 0; k < n_items; ++k) {
            f32x16 acc[MT][MI][NI];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[m][i][j][r] = 0.f;
            int a_addr[MT == 1 ? NT : 1][MI], b_addr[NI];
            if constexpr (MT == 1) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    int pb = pbase[i];
                    asm volatile("" : "+v"(pb));                           // per item: not kept live across the epilogue
#pragma unroll
                    for (int t = 0; t < NT; ++t) a_addr[t][i] = ws_swz(pb + p.tap_off[t], khalf);
                }
            }
            {
                int lb = l31;
                asm volatile("" : "+v"(lb));
#pragma unroll
                for (int j = 0; j < NI; ++j) b_addr[j] = H_BYTES + ws_swz(j * 32 + lb, khalf);
            }
            unsigned pix[MT][MI];
            auto set_pix = [&](int lx) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const TC cc = coords(MT == 1 ? k : min(MT * k + mt, n_my - 1));
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int m = wm * 64 + i * 32 + ws_perm(lx);
                        pix[mt][i] = (unsigned)((cc.b * p.OH + cc.oy0 + (m >> 4)) * p.OW + cc.ox0 + (m & 15));
                    }
                }
            };
            constexpr bool PRE = MODE == 2 && SALT_LS_PREFETCH != 0;
            if (PRE || MT == 1) set_pix(l31);
            WsOps<PRE ? NI : 1> ops;
            constexpr bool PRE_OLD = NI == 1;                              // (NI = 2: 96 more registers do not fit beside the accumulators)
            if constexpr (PRE) ws_prefetch_operands<NI, PRE_OLD>(ep, pix[0], n0, khalf, ops);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
            for (int c = 0; c < p.nchunk; ++c, ++g) {
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();                              // chunk g landed
                asm volatile("" ::: "memory");
                if (SALT_LS_ABLATE & 1) continue;
                const unsigned char* hb = smem + (g % D) * CH_BYTES;
                int pbc[MI];
#pragma unroll
                for (int i = 0; i < MI; ++i) { pbc[i] = pbase[i]; if (MT != 1) asm volatile("" : "+v"(pbc[i])); }      // (per chunk: or the addresses below are hoisted out of the chunk loop and spilled again)
                auto load_frag = [&](int s, Frag& f) {                      // s = (tap, k-step), a constant after unrolling
                    const int t = s >> 1, hx = (s & 1) << 5;
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        // (two-tile items: the 18 tap addresses do not fit beside 128 accumulator registers - kept, they were spilled and reloaded
                        //  from scratch INSIDE this loop, 58 dependent loads per chunk - so they are formed per stage: 7 VALU per 4 MFMAs)
                        const int ad = (MT == 1 ? a_addr[t][i] : ws_swz(pbc[i] + p.tap_off[t], khalf)) ^ hx;
#pragma unroll
                        for (int m = 0; m < MT; ++m) f.a[m][i] = *reinterpret_cast<const u32x4*>(hb + m * HT_BYTES + ad);
                    }
#pragma unroll
                    for (int j = 0; j < NI; ++j) f.b[j] = *reinterpret_cast<const u32x4*>(hb + t * (BN * 64) + (b_addr[j] ^ hx));
                };
                auto mma_frag = [&](const Frag& f) {                         // operands swapped: rows of D = output channels
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < MI; ++i)
#pragma unroll
                            for (int j = 0; j < NI; ++j)
                                acc[m][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.b[j]), __builtin_bit_cast(bf16x8, f.a[m][i]), acc[m][i][j], 0, 0, 0);
                };
                constexpr int NST = NT * 2;
                constexpr int FR = MT == 1 ? 3 : 2;                         // fragment ring: two-tile items have 8 MFMAs (256 cycles) per stage to cover one stage of
                Frag f[FR];                                                // LDS latency, and no registers for a third (the 24 it costs were spilled around the loop)
                load_frag(0, f[0]);
                if (FR == 3) load_frag(1, f[1]);
                __builtin_amdgcn_sched_group_barrier(0x100, (FR - 1) * (MT * MI + NI), 0);
#pragma unroll
                for (int s2 = 0; s2 < NST; ++s2) {
                    if (s2 + FR - 1 < NST) load_frag(s2 + FR - 1, f[(s2 + FR - 1) % FR]);
                    mma_frag(f[s2 % FR]);
                    if (s2 + FR - 1 < NST) __builtin_amdgcn_sched_group_barrier(0x100, MT * MI + NI, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, MT * MI * NI, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (!(PRE || MT == 1)) {                                       // two-tile items: the store addresses are derived AFTER the chunk loop (hoisted above it they
                int lx = l31;                                              // were spilled across it and reloaded one by one in the epilogue: 96 scratch loads per item)
                asm volatile("" : "+v"(lx));
                set_pix(lx);
            }
            const WsLaneGeo geo = {{true, true}, false, false, false, false, 0, 0, 0, 0};
            if (SALT_LS_ABLATE & 4) {                                      // keep the accumulators alive without the epilogue
                float tsum = 0.f;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) tsum += acc[m][i][j][r];
                if (tsum == 123.456f) p.y[0] = f2bf(tsum);
                continue;
            }
            if constexpr (PRE) ws_epilogue_tile<NI, MODE, false>(ep, acc[0], pix[0], geo, n0, reinterpret_cast<const float*>(smem + OFF_CONST), rs0, rs1, khalf, l31, &ops, PRE_OLD);
            else {
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    if (MT == 1 || MT * k + m < n_my)                       // (wave-uniform: an odd walk ends on half an item)
                        ws_epilogue_tile<NI, MODE, false>(ep, acc[m], pix[m], geo, n0, reinterpret_cast<const float*>(smem + OFF_CONST), rs0, rs1, khalf, l31);
            }
        }
    }
    if (MODE != 0 && sums) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ws_sums_flush<NI, MODE, 4>(rs0, rs1, !loader, wm, reinterpret_cast<float*>(smem), n0, p.Cout, p.fin_acc, p.bnb_acc,
                                   nt == 0 ? (double)n_my * 256.0 : 0.0, khalf, l31);
    }
}

template <int NI, int MODE, int MT = 1>
int ls_launch_mode(const LsKP& k, int wgs, hipStream_t st) {
    constexpr int BN = 32 * NI, PC = MT * 21 + 9 * BN / 16, D = NI == 1 ? SALT_LS_D1 : 2;
    constexpr int LDS = D * PC * 1024 + 1024 + 4 * BN * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = conv_ls_kernel<NI, MODE, SALT_LS_NLW, MT>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(256 + 64 * SALT_LS_NLW), LDS, st, k);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

template <int NI>
int ls_launch(const LsKP& k, int wgs, hipStream_t st, int mt = 1) {
    if (k.fin_acc) return ls_launch_mode<NI, 1>(k, wgs, st);
    if (k.bnb_acc) return ls_launch_mode<NI, 2>(k, wgs, st);
    if constexpr (NI == 2) { if (mt == 2) return ls_launch_mode<2, 0, 2>(k, wgs, st); }
    return ls_launch_mode<NI, 0>(k, wgs, st);
}

// ------------------------------------------------------------------------------------------ conv1x1_ls_kernel (bf16, 1x1, unit steps)
// The Bottleneck 1x1 convolutions of ResNet101 / 152 (architectures/encoders.py:13-19 through torchvision's Bottleneck) are streaming
// operators: the 64 -> 256 expansion at 128 x 128 moves 1.2 GB (input, residual, output) for 17 GFLOP per 64 images.  On
// conv_mfma_kernel (load -> MFMA -> stage through LDS -> store, one tile per workgroup) they ran at 2.3 - 2.5 TB/s (C4: 8.5 of 61 ms).
// Same structure as conv_ls_kernel with the halo gone: 4 loader waves stream CHUNKS of 64 input channels - a 256-pixel tile (2 x 16 KB
// sub-tiles of 32 channels, 64-byte rows, XOR slot swizzle on the DMA's source address) + the chunk's weights for the item's 32 NI
// output channels - through a ring of D buffers, one raw s_barrier per chunk; 4 MFMA waves (one per SIMD) compute 64 pixels x 32 NI
// channels each and run ws_epilogue_tile (swapped operands: whole 16-byte NHWC pieces straight to HBM, the residual read as 16-byte
// pieces at the store addresses) while the loaders are already D - 1 chunks into the next items.  Pixels are taken 256 at a time from
// the flattened [B H W] index (a 1x1 convolution has no geometry): the host requires B H W % 256 == 0, so no lane is ever masked.
struct L1KP {
    const bf16_t* x; const bf16_t* w; bf16_t* y;
    const float* bias; const float* scale; const float* shift;
    int x_cs, y_cs, Cout, nsc;                   // nsc: chunks of 64 input channels
    int ntiles, per_xcd, n_tiles, slots;
    int relu, accumulate;
    const bf16_t* res; int res_cs;
    int step, IH, IW, OH, OW, tiles_x, tiles_y;   // step 2 (projection shortcuts): 16 x 16 OUTPUT-pixel tiles, input pixel = 2 x output pixel
    double* fin_acc;                              // MODE 1: train-mode BatchNorm statistics of the result (fp64 shards, salt_conv_args.fin_acc)
    int item_major;                               // more channel blocks than workgroups per XCD (the hypercolumn's tap GEMMs, C -> 9 C): `slots` workgroups per XCD
                                                  // walk the XCD's (pixel tile, channel block) items tile-major with stride `slots` (MODE 0, no per-channel constants)
};

// MODE 0: eval / plain epilogue; 1: + train-mode BatchNorm statistics (the projection shortcuts of a ResNet in training)
template <int NI, int MODE>
__global__ __launch_bounds__(512) void conv1x1_ls_kernel(L1KP p) {
    typedef bf16_t T;
    constexpr int BN = 32 * NI, V = 2, MI = 2, NLW = 4;
    constexpr int XPC = 16 * V, WPC = V * BN / 16, PC = XPC + WPC;        // DMA pieces (1 KB) of one chunk: pixel rows, then weights
    constexpr int NS = (PC + NLW - 1) / NLW;
    constexpr int D = NI == 1 ? 4 : 3;
    constexpr int X_BYTES = XPC * 1024, CH_BYTES = PC * 1024;
    constexpr int OFF_DUMMY = D * CH_BYTES, OFF_CONST = OFF_DUMMY + 1024;
    static_assert(OFF_CONST + 4 * BN * 4 <= 160 * 1024 && (D - 2) * NS <= 63, "LDS / vmcnt budget");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int wm = wave & 3, lw = wave - 4;
    const int khalf = lane >> 5, l31 = lane & 31;

    const int xcd = blockIdx.x & 7, jwg = blockIdx.x >> 3;
    const bool im = p.item_major != 0;
    if (jwg >= (im ? p.slots : p.slots * p.n_tiles)) return;
    const int nt = im ? 0 : jwg % p.n_tiles, slot = im ? 0 : jwg / p.n_tiles;
    int n0 = nt * BN;                             // (item-major: set per item)
    const int t_lo = xcd * p.per_xcd;
    const int t_hi = min(t_lo + p.per_xcd, p.ntiles);
    const int n_items = im ? (jwg < (t_hi - t_lo) * p.n_tiles ? ((t_hi - t_lo) * p.n_tiles - jwg + p.slots - 1) / p.slots : 0)
                           : ((t_lo + slot < t_hi) ? (t_hi - t_lo - slot + p.slots - 1) / p.slots : 0);
    if (n_items <= 0) return;
    const int G = n_items * p.nsc;
    if (tid < BN) {
        float* sc = reinterpret_cast<float*>(smem + OFF_CONST);
        sc[tid] = p.bias ? p.bias[n0 + tid] : 0.f; sc[BN + tid] = p.scale ? p.scale[n0 + tid] : 1.f; sc[2 * BN + tid] = p.shift ? p.shift[n0 + tid] : 0.f;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float rs0[NI][4], rs1[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { rs0[j][e] = 0.f; rs1[j][e] = 0.f; }

    if (loader) {
        const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_ws_zero);
        auto dma = [&](const void* src, int dst) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(smem + dst), 16, 0, 0);
        };
        // item- and chunk-invariant lane offsets (elements) of this wave's pieces
        int off[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int pi = lw + NLW * i;
            off[i] = 0;
            if (pi < XPC) {
                const int sub = pi >> 4, row = (pi & 15) * 16 + (lane >> 2);
                // step 1: row = flattened pixel of the tile; step 2: row = (ty, tx) of a 16 x 16 output tile -> input pixel (2 ty, 2 tx)
                const int prow = p.step == 2 ? ((row >> 4) * 2 * p.IW + (row & 15) * 2) : row;
                off[i] = prow * p.x_cs + sub * 32 + ((lane ^ (row >> 2)) & 3) * 8;
            } else if (pi < PC) {
                const int R = (pi - XPC) * 16 + (lane >> 2);
                const int sub = R / BN, n = R - sub * BN;
                off[i] = (sub * p.Cout + n) * 32 + ((lane ^ (R >> 2)) & 3) * 8;
            }
        }
        int ik = 0, ic = 0, ig = 0;
        auto issue_next = [&]() {
            const bool live = ig < G;
            const int buf = (ig % D) * CH_BYTES;
            long long tile = t_lo + slot + (long long)ik * p.slots;
            int n0i = n0;
            if (im) { const int item = jwg + ik * p.slots, q = item / p.n_tiles; tile = t_lo + q; n0i = (item - q * p.n_tiles) * BN; }
            long long pix0 = tile * 256;                                              // first input pixel of the tile (wave-uniform)
            if (p.step == 2) {
                const int tx = (int)(tile % p.tiles_x); const long long r = tile / p.tiles_x;
                const int ty = (int)(r % p.tiles_y); const long long b = r / p.tiles_y;
                pix0 = (b * p.IH + ty * 32) * p.IW + tx * 32;
            }
            const T* xc = p.x + pix0 * p.x_cs + ic * (32 * V);
            const T* wc = p.w + ((long long)ic * V * p.Cout + n0i) * 32;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int pi = lw + NLW * i;
                const void* src = zp;
                int dst = OFF_DUMMY;
                if (live && pi < XPC) { src = xc + off[i]; dst = buf + pi * 1024; }
                else if (live && pi < PC) { src = wc + off[i]; dst = buf + pi * 1024; }
                dma(src, dst);
            }
            if (live) { ++ig; if (++ic == p.nsc) { ic = 0; ++ik; } }
        };
#pragma unroll 1
        for (int d = 0; d < D - 1; ++d) issue_next();
#pragma unroll 1
        for (int g = 0; g < G; ++g) {
            ws_wait_vm<(D - 2) * NS>();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_next();
        }
        ws_wait_vm<0>();
    } else {
    // ================================================================== MFMA waves
    const WsEpi ep = {p.y, nullptr, nullptr, p.y_cs, 0, 0, p.relu, p.accumulate, 0, p.bias || p.scale || p.shift || p.relu, MODE == 1, p.res, p.res_cs};
    int a_addr[MI], b_addr[NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) a_addr[i] = ws_swz(wm * 64 + i * 32 + l31, khalf);
#pragma unroll
    for (int j = 0; j < NI; ++j) b_addr[j] = X_BYTES + ws_swz(j * 32 + l31, khalf);
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
        unsigned tile = (unsigned)(t_lo + slot + k * p.slots);
        if (im) { const int item = jwg + k * p.slots, q = item / p.n_tiles; tile = (unsigned)(t_lo + q); n0 = (item - q * p.n_tiles) * BN; }
        unsigned pix[MI];
        if (p.step == 2) {
            const unsigned tx = tile % (unsigned)p.tiles_x, r = tile / (unsigned)p.tiles_x, ty = r % (unsigned)p.tiles_y, b = r / (unsigned)p.tiles_y;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const unsigned m = (unsigned)(wm * 64 + i * 32 + l31);
                pix[i] = (b * (unsigned)p.OH + ty * 16u + (m >> 4)) * (unsigned)p.OW + tx * 16u + (m & 15u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i) pix[i] = tile * 256u + (unsigned)(wm * 64 + i * 32 + l31);
        }
        // the residual / (+)= operand tile of the item is requested before its chunks are waited for: a 64 -> 256 expansion is ONE chunk,
        // so an item was DMA wait -> 8 MFMAs -> residual loads -> wait -> stores, with the HBM latency of the residual exposed per item
        constexpr bool PRE = MODE == 0 && SALT_LS_PREFETCH != 0;
        WsOps<PRE ? NI : 1> ops;
        const bool pre_on = PRE && (p.res != nullptr || p.accumulate);
        if constexpr (PRE) { if (pre_on) ws_prefetch_operands<NI, true>(ep, pix, n0, khalf, ops); }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
        for (int c = 0; c < p.nsc; ++c, ++g) {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const unsigned char* hb = smem + (g % D) * CH_BYTES;
            auto load_frag = [&](int s, Frag& f) {                          // s = (sub-chunk, k-step), a constant after unrolling
                const int t = s >> 1, hx = (s & 1) << 5;
#pragma unroll
                for (int i = 0; i < MI; ++i) f.a[i] = *reinterpret_cast<const u32x4*>(hb + t * 16384 + (a_addr[i] ^ hx));
#pragma unroll
                for (int j = 0; j < NI; ++j) f.b[j] = *reinterpret_cast<const u32x4*>(hb + t * (BN * 64) + (b_addr[j] ^ hx));
            };
            constexpr int NST = V * 2;
            Frag f[NST];
#pragma unroll
            for (int s2 = 0; s2 < NST; ++s2) load_frag(s2, f[s2]);
#pragma unroll
            for (int s2 = 0; s2 < NST; ++s2)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[s2].b[j]), __builtin_bit_cast(bf16x8, f[s2].a[i]), acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        const WsLaneGeo geo = {{true, true}, false, false, false, false, 0, 0, 0, 0};
        if constexpr (PRE) {
            if (pre_on) ws_epilogue_tile<NI, MODE, false>(ep, acc, pix, geo, n0, reinterpret_cast<const float*>(smem + OFF_CONST), rs0, rs1, khalf, l31, &ops, true);
            else ws_epilogue_tile<NI, MODE, false>(ep, acc, pix, geo, n0, reinterpret_cast<const float*>(smem + OFF_CONST), rs0, rs1, khalf, l31);
        } else ws_epilogue_tile<NI, MODE, false>(ep, acc, pix, geo, n0, reinterpret_cast<const float*>(smem + OFF_CONST), rs0, rs1, khalf, l31);
    }
    }
    if (MODE == 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ws_sums_flush<NI, 1, 4>(rs0, rs1, !loader, wm, reinterpret_cast<float*>(smem), n0, p.Cout, p.fin_acc, nullptr,
                                nt == 0 ? (double)n_items * 256.0 : 0.0, khalf, l31);
    }
}

// ------------------------------------------------------------------------------------------ conv1x1_xs_kernel (bf16, 1x1, x-stationary)
// The hypercolumn's tap GEMMs at the inference shapes (engine.hyper_level: C -> 9 C, e.g. 256 -> 2304 over 64 x 64 x 64 pixels) are
// write-bound - 1.2 GB out for 0.13 GB in - but conv1x1_ls_kernel streams the 128 KB input tile of EVERY (pixel tile, 64-channel block)
// item through its loader waves again: 36 times per tile, ~250 issue cycles per 1 KB piece, 5.8 us per item (0.83 ms; the stores alone
// would take 0.3).  Here a workgroup keeps the input tile (256 pixels x <= 256 channels = <= 128 KB) resident in LDS and walks ALL
// channel blocks of it: the loader waves only stream the 8 KB weight chunks (ring of 3), the MFMA waves run 16 MFMAs per chunk and
// conv1x1_ls_kernel's epilogue.  Two extra barriers per tile fence the tile buffer (all reads of the old tile done -> DMA -> landed).
struct XsKP {
    const bf16_t* x; const bf16_t* w; bf16_t* y;
    int x_cs, y_cs, Cout, nsc;                   // nsc: chunks of 64 input channels (<= 4)
    int ntiles, n_blocks;                        // 256-pixel tiles, 64-channel blocks
    const float* bias; const float* scale; const float* shift;      // eval epilogue (the Bottleneck expansions 64 -> 256 .. 256 -> 1024 of ResNet101 / 152:
    int relu, accumulate;                        //  the same input tile under 4 channel blocks - 195 -> 137 us plain at 128 x 128 x 64 images)
    const bf16_t* res; int res_cs;
};

__global__ __launch_bounds__(512) void conv1x1_xs_kernel(XsKP p) {
    typedef bf16_t T;
    constexpr int NI = 2, BN = 32 * NI, V = 2, MI = 2, NLW = 4, D = 3;
    constexpr int WPC = V * BN / 16, NSW = WPC / NLW;                 // 1 KB weight pieces per chunk (8), per loader wave (2)
    constexpr int XCH = 32 * 1024, W_BYTES = WPC * 1024;              // one input chunk (2 sub-tiles of 256 x 64 B), one weight chunk
    static_assert(WPC % NLW == 0 && (D - 2) * NSW <= 63, "pieces / vmcnt");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int XR = p.nsc * XCH;                                       // weight ring behind the tile
    const int OFF_DUMMY = XR + D * W_BYTES, OFF_CONST = OFF_DUMMY + 1024;     // constants: [n_blocks][4][BN] floats (bias, scale, shift, -) when there is an affine epilogue, else one block's defaults

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;
    const int wm = wave & 3, lw = wave - 4;
    const int khalf = lane >> 5, l31 = lane & 31;
    const int n_my = ((int)blockIdx.x < p.ntiles) ? (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (n_my <= 0) return;
    const int cpt = p.n_blocks * p.nsc;                               // chunks per tile
    const int G = n_my * cpt;
    const bool affine = p.bias || p.scale || p.shift || p.relu;
    {
        float* sc = reinterpret_cast<float*>(smem + OFF_CONST);
        const int nc = affine ? p.n_blocks * 4 * BN : 4 * BN;
        for (int e = tid; e < nc; e += 512) {
            const int blk = e / (4 * BN), r = e - blk * 4 * BN, which = r / BN, n = blk * BN + (r - which * BN);
            sc[e] = which == 0 ? (p.bias ? p.bias[n] : 0.f) : which == 1 ? (p.scale ? p.scale[n] : 1.f) : which == 2 ? (p.shift ? p.shift[n] : 0.f) : 0.f;
        }
    }

    if (loader) {
        const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_ws_zero);
        auto dma = [&](const void* src, int dst) {
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(smem + dst), 16, 0, 0);
        };
        int woff[NSW];                                                // lane offsets (elements) of this wave's weight pieces inside a chunk's block
#pragma unroll
        for (int i = 0; i < NSW; ++i) {
            const int R = (lw + NLW * i) * 16 + (lane >> 2);
            const int sub = R / BN, n = R - sub * BN;
            woff[i] = (sub * p.Cout + n) * 32 + ((lane ^ (R >> 2)) & 3) * 8;
        }
        int ig = 0, inb = 0, ic = 0;                                   // next weight chunk to request: global index, channel block, chunk (tile-independent: every tile uses all blocks)
        auto issue_w = [&]() {
            const bool live = ig < G;
            const int buf = XR + (ig % D) * W_BYTES;
            const T* wc = p.w + ((long long)ic * V * p.Cout + inb * BN) * 32;
#pragma unroll
            for (int i = 0; i < NSW; ++i) {
                const void* src = zp; int dst = OFF_DUMMY;
                if (live) { src = wc + woff[i]; dst = buf + (lw + NLW * i) * 1024; }
                dma(src, dst);
            }
            if (live) { ++ig; if (++ic == p.nsc) { ic = 0; if (++inb == p.n_blocks) inb = 0; } }
        };
#pragma unroll 1
        for (int d = 0; d < D - 1; ++d) issue_w();
#pragma unroll 1
        for (int k = 0; k < n_my; ++k) {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                              // T1: every read of the previous tile is done
            asm volatile("" ::: "memory");
            const long long pix0 = ((long long)blockIdx.x + (long long)k * gridDim.x) * 256;
#pragma unroll 1
            for (int c = 0; c < p.nsc; ++c) {
                const T* xc = p.x + pix0 * p.x_cs + c * (32 * V);
#pragma unroll
                for (int i = 0; i < 32 / NLW; ++i) {                   // 32 pieces per chunk: sub-tile (pi >> 4), 16 pixel rows x 4 slots per piece
                    const int pi = lw + NLW * i;
                    const int sub = pi >> 4, row = (pi & 15) * 16 + (lane >> 2);
                    dma(xc + row * p.x_cs + sub * 32 + ((lane ^ (row >> 2)) & 3) * 8, c * XCH + pi * 1024);
                }
            }
            ws_wait_vm<0>();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                              // T2: the tile (and the ring's first chunks) landed
            asm volatile("" ::: "memory");
#pragma unroll 1
            for (int g = 0; g < cpt; ++g) {
                ws_wait_vm<(D - 2) * NSW>();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issue_w();
            }
        }
        ws_wait_vm<0>();
    } else {
        const WsEpi ep = {p.y, nullptr, nullptr, p.y_cs, 0, 0, p.relu, p.accumulate, 0, affine, false, p.res, p.res_cs};
        int a_addr[MI], b_addr[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) a_addr[i] = ws_swz(wm * 64 + i * 32 + l31, khalf);
#pragma unroll
        for (int j = 0; j < NI; ++j) b_addr[j] = ws_swz(j * 32 + l31, khalf);
        float rs0[NI][4], rs1[NI][4];
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { rs0[j][e] = 0.f; rs1[j][e] = 0.f; }
        struct Frag { u32x4 a[MI], b[NI]; };
        int g = 0;
#pragma unroll 1
        for (int k = 0; k < n_my; ++k) {
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                              // T1
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                              // T2
            asm volatile("" ::: "memory");
            const unsigned tile = blockIdx.x + (unsigned)k * gridDim.x;
            unsigned pix[MI];
#pragma unroll
            for (int i = 0; i < MI; ++i) pix[i] = tile * 256u + (unsigned)(wm * 64 + i * 32 + l31);
#pragma unroll 1
            for (int nb = 0; nb < p.n_blocks; ++nb) {
                f32x16 acc[MI][NI];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
                // the residual / (+)= operand tile of the block is requested before its chunks (conv1x1_ls_kernel: a 64 -> 256 expansion is ONE chunk)
                WsOps<NI> ops;
                const bool pre_on = p.res != nullptr || p.accumulate;
                if (pre_on) ws_prefetch_operands<NI, true>(ep, pix, nb * BN, khalf, ops);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
                for (int c = 0; c < p.nsc; ++c, ++g) {
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    const unsigned char* xb = smem + c * XCH;
                    const unsigned char* wb = smem + XR + (g % D) * W_BYTES;
                    constexpr int NST = V * 2;
                    Frag f[NST];
#pragma unroll
                    for (int s2 = 0; s2 < NST; ++s2) {
                        const int t = s2 >> 1, hx = (s2 & 1) << 5;
#pragma unroll
                        for (int i = 0; i < MI; ++i) f[s2].a[i] = *reinterpret_cast<const u32x4*>(xb + t * 16384 + (a_addr[i] ^ hx));
#pragma unroll
                        for (int j = 0; j < NI; ++j) f[s2].b[j] = *reinterpret_cast<const u32x4*>(wb + t * (BN * 64) + (b_addr[j] ^ hx));
                    }
#pragma unroll
                    for (int s2 = 0; s2 < NST; ++s2)
#pragma unroll
                        for (int i = 0; i < MI; ++i)
#pragma unroll
                            for (int j = 0; j < NI; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[s2].b[j]), __builtin_bit_cast(bf16x8, f[s2].a[i]), acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
                const WsLaneGeo geo = {{true, true}, false, false, false, false, 0, 0, 0, 0};
                const float* cst = reinterpret_cast<const float*>(smem + OFF_CONST) + (affine ? nb * 4 * BN : 0);
                if (pre_on) ws_epilogue_tile<NI, 0, false>(ep, acc, pix, geo, nb * BN, cst, rs0, rs1, khalf, l31, &ops, true);
                else ws_epilogue_tile<NI, 0, false>(ep, acc, pix, geo, nb * BN, cst, rs0, rs1, khalf, l31);
            }
        }
    }
}

int xs_launch(const XsKP& k, int wgs, hipStream_t st) {
    const bool affine = k.bias || k.scale || k.shift || k.relu;
    const int lds = k.nsc * 32 * 1024 + 3 * 8 * 1024 + 1024 + (affine ? k.n_blocks : 1) * 4 * 64 * 4;
    if (lds > 160 * 1024) SALT_FAIL(SALT_E_LDS, "conv1x1_xs: %d bytes of LDS", lds);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv1x1_xs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(conv1x1_xs_kernel, dim3((unsigned)wgs), dim3(512), lds, st, k);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

template <int NI, int MODE>
int l1_launch_mode(const L1KP& k, int wgs, hipStream_t st) {
    constexpr int BN = 32 * NI, PC = 32 + 2 * BN / 16, D = NI == 1 ? 4 : 3;
    constexpr int LDS = D * PC * 1024 + 1024 + 4 * BN * 4;
    auto kern = conv1x1_ls_kernel<NI, MODE>;
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
int l1_launch(const L1KP& k, int wgs, hipStream_t st) {
    return k.fin_acc ? l1_launch_mode<NI, 1>(k, wgs, st) : l1_launch_mode<NI, 0>(k, wgs, st);
}


// ------------------------------------------------------------------------------------------ conv_stem16_kernel (bf16, 16 input channels)
// The ResNet stem after the 2 x 2 space-to-depth (engine.py _stem_s2d: 16 taps over 16 channels -> 64, torchvision's conv1 via
// architectures/encoders.py:20-27): K = 16 per tap is exactly ONE v_mfma_f32_32x32x16_bf16 step, but the general kernels stage 32-channel
// chunks (half of them zero padding) and one tile per workgroup: 45 us for 4.3 GFLOP (of which 2.1 padding) / 21 MB at B = 32.  Here a
// persistent 4-wave workgroup keeps the weights of all 16 taps in LDS as 32-byte rows (32 KB), streams 16 x 16-pixel tiles through a
// double-buffered 19 x 19 halo of 32-byte pixel rows (LDS-DMA, one tile ahead; 1-KB wave reads of 32 consecutive rows are conflict free
// without a swizzle) and runs conv_ws_kernel's swapped-operand epilogue (ws_epilogue_tile: 16-byte stores, statistics in registers).
struct S16KP {
    const bf16_t* x; const bf16_t* w; bf16_t* y;
    const float* bias; const float* scale; const float* shift;
    int B, H, W, x_cs, y_cs, OH, OW;
    int tiles_x, tiles_y, ntiles;
    int min_dy, min_dx;
    int tap_off[16];
    int relu;
    double* fin_acc;
};

template <int MODE>
__global__ __launch_bounds__(256) void conv_stem16_kernel(S16KP p) {
    constexpr int NI = 2, MI = 2, BN = 64, NT = 16;
    constexpr int WPIECES = NT * BN / 32;                // 32 pieces of 1 KB = 32 rows of 32 bytes
    constexpr int HPIECES = 12;                          // 19 x 19 = 361 rows, padded to 384
    constexpr int W_BYTES = WPIECES * 1024, H_BYTES = HPIECES * 1024;
    constexpr int OFF_H = W_BYTES, OFF_DUMMY = OFF_H + 2 * H_BYTES, OFF_CONST = OFF_DUMMY + 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
    const int n_my = ((int)blockIdx.x < p.ntiles) ? (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    if (n_my <= 0) return;
    struct TC { int b, oy0, ox0; };
    auto coords = [&](int k) {
        const int t = (int)blockIdx.x + k * (int)gridDim.x;
        TC c; const int tx = t % p.tiles_x; const int r = t / p.tiles_x;
        c.ox0 = tx << 4; c.oy0 = (r % p.tiles_y) << 4; c.b = r / p.tiles_y; return c;
    };
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_ws_zero);
    auto dma = [&](const void* src, int dst) {
        __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(smem + dst), 16, 0, 0);
    };
    // weights: LDS row (tap, n) = the first 32 bytes (16 channels) of the packed 64-byte row [tap][Cout][32]
#pragma unroll
    for (int i = 0; i < WPIECES / 4; ++i) {
        const int q = wave + 4 * i;
        const int R = q * 32 + (lane >> 1);
        dma(reinterpret_cast<const unsigned char*>(p.w + (R * 32 + (lane & 1) * 8)), q * 1024);
    }
    auto issue_halo = [&](const TC& c, int buf) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const bf16_t* xb = p.x + (int64_t)c.b * p.H * p.W * p.x_cs;
        const int iy0 = c.oy0 + p.min_dy, ix0 = c.ox0 + p.min_dx;
#pragma unroll
        for (int i = 0; i < HPIECES / 4; ++i) {
            const int pidx = wave + 4 * i;
            const int row = pidx * 32 + (ln >> 1);
            const int hy = (int)__umulhi((unsigned)row, 226050911u);         // row / 19
            const int hx = row - hy * 19;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool valid = (row < 361) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            const unsigned char* src = reinterpret_cast<const unsigned char*>(xb + ((iy * p.W + ix) * p.x_cs + (ln & 1) * 8));
            dma(valid ? src : zp, OFF_H + buf * H_BYTES + pidx * 1024);
        }
    };
    issue_halo(coords(0), 0);
    if (tid < BN) {
        float* sc = reinterpret_cast<float*>(smem + OFF_CONST);
        sc[tid] = p.bias ? p.bias[tid] : 0.f; sc[BN + tid] = p.scale ? p.scale[tid] : 1.f; sc[2 * BN + tid] = p.shift ? p.shift[tid] : 0.f;
    }
    float rs0[NI][4], rs1[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) { rs0[j][e] = 0.f; rs1[j][e] = 0.f; }
    const bool sums = MODE == 1 && p.fin_acc;
    const WsEpi ep = {p.y, nullptr, nullptr, p.y_cs, 0, 0, p.relu, 0, 0, p.bias || p.scale || p.shift || p.relu, sums, nullptr, 0};
    int pbase[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wave * 64 + i * 32 + ws_perm(l31);
        pbase[i] = (m >> 4) * 19 + (m & 15);
    }
#pragma unroll 1
    for (int k = 0; k < n_my; ++k) {
        if (k == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const TC cur = coords(k);
        if (k + 1 < n_my) issue_halo(coords(k + 1), (k + 1) & 1);
        const unsigned char* hb = smem + OFF_H + (k & 1) * H_BYTES;
        f32x16 acc[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        struct Frag { u32x4 a[MI], b[NI]; };
        auto load_frag = [&](int t, Frag& f) {
#pragma unroll
            for (int i = 0; i < MI; ++i) f.a[i] = *reinterpret_cast<const u32x4*>(hb + (pbase[i] + p.tap_off[t]) * 32 + khalf * 16);
#pragma unroll
            for (int j = 0; j < NI; ++j) f.b[j] = *reinterpret_cast<const u32x4*>(smem + (t * BN + j * 32 + l31) * 32 + khalf * 16);
        };
        Frag f[2];
        load_frag(0, f[0]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t + 1 < NT) load_frag(t + 1, f[(t + 1) & 1]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[t & 1].b[j]), __builtin_bit_cast(bf16x8, f[t & 1].a[i]), acc[i][j], 0, 0, 0);
        }
        if (k + 1 < n_my) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile's pieces landed; this tile's stores stay out of the wait
        unsigned pix[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = wave * 64 + i * 32 + ws_perm(l31);
            pix[i] = (unsigned)((cur.b * p.OH + cur.oy0 + (m >> 4)) * p.OW + cur.ox0 + (m & 15));
        }
        const WsLaneGeo geo = {{true, true}, false, false, false, false, 0, 0, 0, 0};
        ws_epilogue_tile<NI, MODE, false>(ep, acc, pix, geo, 0, reinterpret_cast<const float*>(smem + OFF_CONST), rs0, rs1, khalf, l31);
    }
    if (MODE == 1 && sums) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ws_sums_flush<NI, 1, 4>(rs0, rs1, true, wave, reinterpret_cast<float*>(smem), 0, BN, p.fin_acc, nullptr, (double)n_my * 256.0, khalf, l31);
    }
}

template <int MODE>
int stem16_launch_mode(const S16KP& k, hipStream_t st) {
    constexpr int LDS = 32 * 1024 + 2 * 12 * 1024 + 1024 + 4 * 64 * 4;
    int wgs = ws_cus() * 2;
    if (wgs > k.ntiles) wgs = k.ntiles;
    hipLaunchKernelGGL(conv_stem16_kernel<MODE>, dim3((unsigned)wgs), dim3(256), LDS, st, k);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

}  // namespace

// ---- host interface (conv_mfma.hip: salt_conv / salt_conv_stats_parts try this first)
// SALT_CONV_WS = 0: off unless asked for per launch; 1 (default): on for launches with at least half a tile per CU.
// cfg & 0xff == 9 asks for this kernel wherever it applies, whatever the tile count (tests); cfg >> 8 (if non-zero) caps the
// workgroups per XCD, so that small test tensors exercise the multi-tile pipeline.  A launch it does not apply to falls back to
// conv_mfma_kernel's own heuristic; salt_conv_kernel_id tells which kernel a launch gets.
int conv_ws_tiles(const salt_conv_args* a);

bool conv_ws_eligible(const salt_conv_args* a) {
    static const int env = getenv("SALT_CONV_WS") ? atoi(getenv("SALT_CONV_WS")) : 1;
    if (!a || a->dtype != SALT_BF16) return false;
    const bool asked = (a->cfg & 0xff) == 9;
    if ((a->cfg & 0xff) != 0 && !asked) return false;
    if (!asked && !env) return false;
    if (a->ntaps != 9 || a->in_step != 1 || a->out_step != 1 || a->out_oy || a->out_ox || a->nphase > 1) return false;
    // plain launches, or the FUSED fold of a 3x3 replicate-padded convolution's data gradient (two pad rows on top, two columns right)
    const bool fold = !a->strip && (a->fold_top || a->fold_right);
    if (a->strip || a->fold_bottom || a->fold_left) return false;
    if (fold && (a->fold_top != 2 || a->fold_right != 2 || a->fin_acc || a->OH != a->y.H + 2 || a->OW != a->y.W + 2)) return false;
    // the extended grid is tiled 16 x 16 from its top-right corner: on a 34 x 34 grid that is 9 tiles for 4 tiles of image - measured
    // slower than conv_mfma_kernel's fused fold below 64 x 64 (profiles/r03_ws_clocks.txt)
    if (fold && !asked && (a->y.H < 64 || a->y.W < 64)) return false;
    if (!fold && (a->OH != a->y.H || a->OW != a->y.W)) return false;
    if (a->stats || a->fin_ticket || a->bnb_partials || a->bnb_ticket) return false;
    // the kernels' MODE dispatch: MODE 1 (forward statistics) has no accumulate, MODE 2 (BatchNorm-backward sums) no bias / scale /
    // shift / ReLU epilogue - such launches go to conv_mfma_kernel, which honours every combination
    if (a->fin_acc && a->accumulate) return false;
    if (a->bnb_acc && (a->bias || a->scale || a->shift || a->relu)) return false;
    const int Cin = a->x.C, Cout = a->y.C;
    if (a->x_plane) return false;
    // planar y: planes of exactly one channel block (64 channels)
    if (a->y_plane && (a->y.cs != 64 || Cout % 64 || a->y_plane < (int64_t)a->y.B * a->y.H * a->y.W * 64 || a->y_plane % 8 || a->bnb_acc || a->res.p)) return false;
    // all taps x all input channels of ONE block of 32 | 64 output channels stay in LDS: Cin <= 64
    if (!((Cin == 64 && (Cout % 64 == 0 || Cout == 32)) || (Cin == 32 && Cout % 64 == 0))) return false;
    if (a->x.cs % 8 || a->y.cs % 8 || ((reinterpret_cast<uintptr_t>(a->x.p) | reinterpret_cast<uintptr_t>(a->y.p) | reinterpret_cast<uintptr_t>(a->w)) & 15)) return false;
    if (a->x.B != a->y.B) return false;
    if (a->fin_acc && (a->OH % 16 || a->OW % 16)) return false;           // forward statistics: whole tiles only (no per-pixel mask there)
    int min_dy = 1 << 30, max_dy = -(1 << 30), min_dx = 1 << 30, max_dx = -(1 << 30);
    for (int t = 0; t < 9; ++t) {
        min_dy = a->tap_dy[t] < min_dy ? a->tap_dy[t] : min_dy; max_dy = a->tap_dy[t] > max_dy ? a->tap_dy[t] : max_dy;
        min_dx = a->tap_dx[t] < min_dx ? a->tap_dx[t] : min_dx; max_dx = a->tap_dx[t] > max_dx ? a->tap_dx[t] : max_dx;
    }
    if (max_dy - min_dy != 2 || max_dx - min_dx != 2) return false;
    auto small = [](const salt_view& v) { return !v.p || (int64_t)v.B * v.H * v.W * v.cs < (int64_t)1 << 31; };
    if (!small(a->x) || !small(a->y) || !small(a->bnb_y) || !small(a->bnb_a) || !small(a->res)) return false;
    if (a->res.p && (a->res.cs % 8 || (reinterpret_cast<uintptr_t>(a->res.p) & 15) || a->accumulate || a->fin_acc || a->bnb_acc || fold)) return false;
    if (a->bnb_acc) {
        if (!view_ok(a->bnb_y) || a->bnb_y.B != a->y.B || a->bnb_y.H != a->y.H || a->bnb_y.W != a->y.W || a->bnb_y.C != Cout || a->bnb_y.cs % 8 ||
            (reinterpret_cast<uintptr_t>(a->bnb_y.p) & 15) || !a->bnb_mean || !a->bnb_invstd || !a->bnb_gamma || !a->bnb_beta) return false;
        if (a->bnb_a.p && (a->bnb_a.B != a->y.B || a->bnb_a.H != a->y.H || a->bnb_a.W != a->y.W || a->bnb_a.C != Cout || a->bnb_a.cs % 8 ||
                           (reinterpret_cast<uintptr_t>(a->bnb_a.p) & 15))) return false;
        if (a->fin_acc) return false;
    }
    const int bn = Cout % 64 == 0 ? 64 : 32;
    if (Cout / bn > ws_cus() / 8) return false;
    const int64_t nitems = (int64_t)conv_ws_tiles(a) * (Cout / bn);
    // a launch must give most CUs at least one item; below that the per-CU weight load has nothing to amortise over
    if (!asked && nitems < ws_cus() / 2) return false;
    return true;
}

int conv_ws_tiles(const salt_conv_args* a) { return a->x.B * cdiv(a->OH, 16) * cdiv(a->OW, 16); }

int conv_ws_launch(const salt_conv_args* a, hipStream_t st) {
    WsKP k;
    k.x = reinterpret_cast<const bf16_t*>(a->x.p); k.w = reinterpret_cast<const bf16_t*>(a->w); k.y = reinterpret_cast<bf16_t*>(a->y.p);
    k.bias = a->bias; k.scale = a->scale; k.shift = a->shift;
    k.B = a->x.B; k.H = a->x.H; k.W = a->x.W; k.x_cs = a->x.cs; k.y_cs = a->y.cs; k.OH = a->OH; k.OW = a->OW;
    k.yH = a->y.H; k.yW = a->y.W; k.fold_top = a->fold_top;
    k.tiles_x = cdiv(a->OW, 16); k.tiles_y = cdiv(a->OH, 16); k.ntiles = k.B * k.tiles_x * k.tiles_y;
    // fused fold: tile columns are RIGHT-aligned with the extended grid, so that the two pad columns share a tile (and a wave) with the
    // last image column; rows start at 0: the two pad rows share the first tile row's first wave with image row 0
    k.ox_shift = a->fold_right ? k.tiles_x * 16 - a->OW : 0;
    int min_dy = 1 << 30, min_dx = 1 << 30;
    for (int t = 0; t < 9; ++t) { min_dy = a->tap_dy[t] < min_dy ? a->tap_dy[t] : min_dy; min_dx = a->tap_dx[t] < min_dx ? a->tap_dx[t] : min_dx; }
    k.min_dy = min_dy; k.min_dx = min_dx; k.pad_mode = a->pad_mode;
    for (int t = 0; t < 9; ++t) k.tap_off[t] = (a->tap_dy[t] - min_dy) * 18 + (a->tap_dx[t] - min_dx);
    k.relu = a->relu; k.accumulate = a->accumulate;
    k.bnb_y = reinterpret_cast<const bf16_t*>(a->bnb_y.p); k.bnb_a = reinterpret_cast<const bf16_t*>(a->bnb_a.p);
    k.bnb_cs = a->bnb_y.cs; k.bnb_acs = a->bnb_a.cs; k.bnb_relu = a->bnb_relu;
    k.bnb_mean = a->bnb_mean; k.bnb_invstd = a->bnb_invstd; k.bnb_gamma = a->bnb_gamma; k.bnb_beta = a->bnb_beta;
    k.fin_acc = a->fin_acc; k.bnb_acc = a->bnb_acc;
    k.res = reinterpret_cast<const bf16_t*>(a->res.p); k.res_cs = a->res.cs;
    k.y_plane = a->y_plane;
    if (!a->bnb_acc) { k.bnb_y = nullptr; k.bnb_a = nullptr; }
    const int Cin = a->x.C, Cout = a->y.C, bn = Cout % 64 == 0 ? 64 : 32;
    k.Cout = Cout; k.n_tiles = Cout / bn;
    int wpx = ws_cus() / 8;                              // workgroups per XCD: one per CU, fewer when the launch has fewer items
    const int cap = (a->cfg >> 8) & 0xff;
    if (cap && wpx > cap) wpx = cap > k.n_tiles ? cap : k.n_tiles;
    k.per_xcd = cdiv(k.ntiles, 8);
    k.slots = wpx / k.n_tiles;
    if (k.slots > k.per_xcd) k.slots = k.per_xcd;
    if (k.slots < 1) SALT_FAIL(SALT_E_UNSUPPORTED, "conv_ws: %d channel blocks for %d workgroups per XCD", k.n_tiles, wpx);
    if (Cin == 64 && bn == 64) return ws_launch<2, 2>(k, st);
    if (Cin == 64 && bn == 32) return ws_launch<2, 1>(k, st);
    if (Cin == 32 && bn == 64) return ws_launch<1, 2>(k, st);
    SALT_FAIL(SALT_E_BADARG, "conv_ws: channels %d -> %d", Cin, Cout);
}

// ---- conv_ls_kernel host side.  SALT_CONV_LS = 0: off unless asked for per launch (cfg & 0xff == 10); 1 (default): on.
// Returns NI (1 | 2) when the launch runs on conv_ls_kernel, else 0.
static int ls_common_ok(const salt_conv_args* a) {
    if (!a || a->dtype != SALT_BF16) return 0;
    if ((a->fin_acc && a->accumulate) || (a->bnb_acc && (a->bias || a->scale || a->shift || a->relu))) return 0;      // see conv_ws_eligible: MODE dispatch
    if (a->ntaps != 9 || a->in_step != 1 || a->out_step != 1 || a->out_oy || a->out_ox || a->nphase > 1) return 0;
    if (a->strip || a->fold_top || a->fold_bottom || a->fold_left || a->fold_right) return 0;
    if (a->stats || a->fin_ticket || a->bnb_partials || a->bnb_ticket) return 0;
    const int Cin = a->x.C, Cout = a->y.C;
    if (Cin % 32 || Cin < 64 || Cout % 32) return 0;
    if (a->y_plane) return 0;
    if (a->x_plane && (a->x.cs != 64 || Cin % 64 || a->x_plane % 8 || a->x_plane < (int64_t)a->x.B * a->x.H * a->x.W * 64)) return 0;
    if (a->x.cs % 8 || a->y.cs % 8 || ((reinterpret_cast<uintptr_t>(a->x.p) | reinterpret_cast<uintptr_t>(a->y.p) | reinterpret_cast<uintptr_t>(a->w)) & 15)) return 0;
    if (a->OH != a->y.H || a->OW != a->y.W || a->OH % 16 || a->OW % 16 || a->x.B != a->y.B) return 0;
    int min_dy = 1 << 30, max_dy = -(1 << 30), min_dx = 1 << 30, max_dx = -(1 << 30);
    for (int t = 0; t < 9; ++t) {
        min_dy = a->tap_dy[t] < min_dy ? a->tap_dy[t] : min_dy; max_dy = a->tap_dy[t] > max_dy ? a->tap_dy[t] : max_dy;
        min_dx = a->tap_dx[t] < min_dx ? a->tap_dx[t] : min_dx; max_dx = a->tap_dx[t] > max_dx ? a->tap_dx[t] : max_dx;
    }
    if (max_dy - min_dy != 2 || max_dx - min_dx != 2) return 0;
    auto small = [](const salt_view& v) { return !v.p || (int64_t)v.B * v.H * v.W * v.cs < (int64_t)1 << 31; };
    // x: the loaders address an IMAGE with 32-bit lane offsets behind a 64-bit base (round 5: the 512 -> 256 convolution over the 64-image
    // 256 x 256 hypercolumn is 2^31 elements); everything the epilogue touches is addressed from the tensor's base with 32 bits
    if ((int64_t)a->x.H * a->x.W * a->x.cs >= (int64_t)1 << 31 || (int64_t)a->x.B * a->x.H * a->x.W * a->x.cs >= (int64_t)1 << 40) return 0;
    if (!small(a->y) || !small(a->bnb_y) || !small(a->bnb_a) || !small(a->res)) return 0;
    if (a->res.p && (a->res.cs % 8 || (reinterpret_cast<uintptr_t>(a->res.p) & 15) || a->accumulate || a->fin_acc || a->bnb_acc)) return 0;
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
    if ((a->cfg & 0xff) != 0 && !asked) return 0;
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
    k.res = reinterpret_cast<const bf16_t*>(a->res.p); k.res_cs = a->res.cs;
    k.x_plane = a->x_plane;
    if (!a->bnb_acc) { k.bnb_y = nullptr; k.bnb_a = nullptr; }
    int wpx = ws_cus() / 8;
    const int cap = (a->cfg >> 8) & 0xff;
    k.n_tiles = k.Cout / (32 * ni);
    if (cap && wpx > cap) wpx = cap > k.n_tiles ? cap : k.n_tiles;      // (never fewer than one workgroup per channel block)
    if (k.n_tiles > wpx) SALT_FAIL(SALT_E_UNSUPPORTED, "conv_ls: %d channel blocks for %d workgroups per XCD", k.n_tiles, wpx);
    k.per_xcd = cdiv(k.ntiles, 8);
    k.slots = wpx / k.n_tiles;
    if (k.slots > k.per_xcd) k.slots = k.per_xcd;
    const int wgs = k.slots * k.n_tiles * 8;
    // two-tile items (MT = 2): plain epilogue, 64-channel blocks, >= 4 chunks of input channels and >= 16 tiles per workgroup (>= 4 from 24 chunks).  Same-box
    // per-layer A/B on the ResNet152 pass: every layer of the class 2 - 12 % faster, the class 17.7 -> 16.0 ms.  (The first
    // version kept the 18 tap addresses in registers beside 128 accumulators: they were spilled and reloaded INSIDE the chunk loop and
    // the layers with few chunks ran 12 - 20 % slower.)  SALT_CONV_LS_MT=1: off; SALT_CONV_LS_MT_MINCHUNK; cfg bit 20 asks, bit 21 forbids
    static const int mt_env = getenv("SALT_CONV_LS_MT") ? atoi(getenv("SALT_CONV_LS_MT")) : 2;
    static const int mt_minchunk = getenv("SALT_CONV_LS_MT_MINCHUNK") ? atoi(getenv("SALT_CONV_LS_MT_MINCHUNK")) : 4;
    // (a bias / folded-BatchNorm / ReLU / residual epilogue marks the forward layers of an eval-mode network: the plain data gradients of a B = 32 training step have 4 - 8 tiles per
    //  workgroup and came out 0.25 % slower per step with two-tile items; SALT_CONV_LS_MT_TRAIN=1 lifts the restriction)
    static const bool mt_train = getenv("SALT_CONV_LS_MT_TRAIN") != nullptr;
    static const int mt_mintiles = getenv("SALT_CONV_LS_MT_MINTILES") ? atoi(getenv("SALT_CONV_LS_MT_MINTILES")) : 16;
    const bool asked = (a->cfg & 0xff) == 10;
    int mt = 1;
    if (ni == 2 && !k.fin_acc && !k.bnb_acc) {
        if (asked && ((a->cfg >> 20) & 1)) mt = 2;
        else if (!(asked && ((a->cfg >> 21) & 1)) && mt_env == 2 && k.nchunk >= mt_minchunk && (a->scale || a->bias || a->relu || a->res.p || mt_train) &&
                 (k.per_xcd >= mt_mintiles * k.slots || (k.per_xcd >= 4 * k.slots && k.nchunk >= 24))) mt = 2;       // (8 tiles per workgroup x 4 chunks - the training step's 128 -> 64 over the two full-resolution hypercolumn planes - came out 50 % slower)
    }
    return ni == 2 ? ls_launch<2>(k, wgs, st, mt) : ls_launch<1>(k, wgs, st);
}


// ---- conv1x1_ls_kernel host side.  SALT_CONV_1X1_LS = 0: off unless asked for per launch (cfg & 0xff == 11).  Returns NI (1 | 2) or 0.
int conv1x1_ls_variant(const salt_conv_args* a) {
    static const int env = getenv("SALT_CONV_1X1_LS") ? atoi(getenv("SALT_CONV_1X1_LS")) : 1;
    if (!a || a->dtype != SALT_BF16 || a->ntaps != 1 || a->tap_dy[0] || a->tap_dx[0]) return 0;
    const bool asked = (a->cfg & 0xff) == 11;
    if ((a->cfg & 0xff) != 0 && !asked) return 0;
    if (!asked && !env) return 0;
    if ((a->in_step != 1 && a->in_step != 2) || a->out_step != 1 || a->out_oy || a->out_ox || a->nphase > 1) return 0;
    if (a->strip || a->fold_top || a->fold_bottom || a->fold_left || a->fold_right) return 0;
    // eval / plain epilogues, or the train-mode statistics through the fp64 shards finalized by the consumer (as conv_ls_kernel's MODE 1)
    if (a->stats || a->fin_ticket || a->bnb_acc || a->bnb_partials || a->bnb_ticket || a->in_scale || a->in_fin_acc) return 0;
    if (a->fin_acc && (a->accumulate || a->res.p || a->bias || a->scale || a->shift || a->relu)) return 0;
    if (a->x_plane || a->y_plane) return 0;
    const int Cin = a->x.C, Cout = a->y.C;
    if (Cin % 64 || Cout % 32) return 0;
    if (a->x.B != a->y.B || a->OH != a->y.H || a->OW != a->y.W) return 0;
    if (a->in_step == 1 && (a->x.H != a->y.H || a->x.W != a->y.W)) return 0;
    // stride 2 (ResNet projection shortcuts): 16 x 16 output tiles, every tap inside the input
    if (a->in_step == 2 && (a->y.H % 16 || a->y.W % 16 || 2 * a->y.H > a->x.H + 1 || 2 * a->y.W > a->x.W + 1)) return 0;
    const int64_t npix = (int64_t)a->y.B * a->y.H * a->y.W;
    if (npix % 256) return 0;
    if (a->x.cs % 8 || a->y.cs % 8 || ((reinterpret_cast<uintptr_t>(a->x.p) | reinterpret_cast<uintptr_t>(a->y.p) | reinterpret_cast<uintptr_t>(a->w)) & 15)) return 0;
    if (a->res.p && (a->res.cs % 8 || (reinterpret_cast<uintptr_t>(a->res.p) & 15) || a->accumulate)) return 0;
    auto small = [&](const salt_view& v) { return !v.p || (int64_t)v.B * v.H * v.W * v.cs < (int64_t)1 << 31; };
    if (!small(a->x) || !small(a->y) || !small(a->res)) return 0;
    const int wpx = ws_cus() / 8;
    int ni = Cout % 64 == 0 ? 2 : 1;
    const int force_ni = asked ? (a->cfg >> 16) & 3 : 0;
    if (force_ni == 1 || (force_ni == 2 && Cout % 64 == 0)) ni = force_ni;
    // more channel blocks than workgroups per XCD: the item-major walk (plain epilogue only - no per-channel constants, no statistics)
    if (Cout / (32 * ni) > wpx && (a->fin_acc || a->bias || a->scale || a->shift)) return 0;
    if (!asked && (npix / 256) * (Cout / (32 * ni)) < ws_cus() / 2) return 0;        // too few items to fill the chip
    return ni;
}

// conv1x1_xs_kernel instead: plain epilogue, unit step, <= 256 input channels, many 64-channel blocks per pixel tile and enough tiles for
// the chip.  SALT_CONV_1X1_XS=0: off; cfg bit 19 (with cfg & 0xff == 11): asked for (tests walk it on small tensors).
static bool conv1x1_xs_ok(const salt_conv_args* a) {
    static const int env = getenv("SALT_CONV_1X1_XS") ? atoi(getenv("SALT_CONV_1X1_XS")) : 1;
    const bool asked = (a->cfg & 0xff) == 11 && ((a->cfg >> 19) & 1);
    if (!env && !asked) return false;
    if (a->in_step != 1 || a->fin_acc) return false;
    const int Cin = a->x.C, Cout = a->y.C;
    if (Cin % 64 || Cin > 256 || Cout % 64) return false;
    const bool affine = a->bias || a->scale || a->shift || a->relu;
    if (Cin / 64 * 32 * 1024 + 25 * 1024 + (affine ? Cout / 64 : 1) * 1024 > 160 * 1024) return false;
    const int64_t tiles = (int64_t)a->y.B * a->y.H * a->y.W / 256;
    if (tiles >= (1 << 22)) return false;
    // by size: the tap GEMMs (>= 8 blocks), and the Bottleneck expansions with <= 128 input channels per 4 blocks... measured (plain
    // epilogue, 64 images): 64 -> 256 @128^2 195 -> 137 us, 128 -> 512 @64^2 93 -> 92, 256 -> 1024 @32^2 55 -> 52, 256 -> 64 @128^2 128 -> 137
    return asked || (tiles >= ws_cus() && (Cout / 64 >= 8 || (Cout >= 4 * Cin && Cout / 64 >= 4)));
}

int conv1x1_ls_launch(const salt_conv_args* a, hipStream_t st) {
    const int ni = conv1x1_ls_variant(a);
    if (!ni) SALT_FAIL(SALT_E_UNSUPPORTED, "conv1x1_ls: not applicable");
    if (conv1x1_xs_ok(a)) {
        XsKP k;
        k.x = reinterpret_cast<const bf16_t*>(a->x.p); k.w = reinterpret_cast<const bf16_t*>(a->w); k.y = reinterpret_cast<bf16_t*>(a->y.p);
        k.x_cs = a->x.cs; k.y_cs = a->y.cs; k.Cout = a->y.C; k.nsc = a->x.C / 64;
        k.ntiles = (int)((int64_t)a->y.B * a->y.H * a->y.W / 256); k.n_blocks = a->y.C / 64;
        k.bias = a->bias; k.scale = a->scale; k.shift = a->shift; k.relu = a->relu; k.accumulate = a->accumulate;
        k.res = reinterpret_cast<const bf16_t*>(a->res.p); k.res_cs = a->res.cs;
        int wgs = ws_cus();
        const int cap = (a->cfg >> 8) & 0xff;
        if (cap) wgs = cap * 8;
        if (wgs > k.ntiles) wgs = k.ntiles;
        return xs_launch(k, wgs, st);
    }
    L1KP k;
    k.x = reinterpret_cast<const bf16_t*>(a->x.p); k.w = reinterpret_cast<const bf16_t*>(a->w); k.y = reinterpret_cast<bf16_t*>(a->y.p);
    k.bias = a->bias; k.scale = a->scale; k.shift = a->shift;
    k.x_cs = a->x.cs; k.y_cs = a->y.cs; k.Cout = a->y.C; k.nsc = a->x.C / 64;
    k.ntiles = (int)((int64_t)a->y.B * a->y.H * a->y.W / 256);
    k.step = a->in_step; k.IH = a->x.H; k.IW = a->x.W; k.OH = a->y.H; k.OW = a->y.W; k.tiles_x = a->y.W / 16; k.tiles_y = a->y.H / 16;
    if (k.step == 2 && (k.tiles_x < 1 || k.tiles_y < 1)) SALT_FAIL(SALT_E_UNSUPPORTED, "conv1x1_ls: stride 2 needs 16 x 16 output tiles");
    k.relu = a->relu; k.accumulate = a->accumulate;
    k.res = reinterpret_cast<const bf16_t*>(a->res.p); k.res_cs = a->res.cs;
    k.fin_acc = a->fin_acc;
    int wpx = ws_cus() / 8;
    const int cap = (a->cfg >> 8) & 0xff;
    k.n_tiles = k.Cout / (32 * ni);
    k.item_major = k.n_tiles > wpx || ((a->cfg & 0xff) == 11 && ((a->cfg >> 18) & 1) && !(a->fin_acc || a->bias || a->scale || a->shift));   // (bit 18: tests walk it on small tensors)
    if (cap && wpx > cap) wpx = (k.item_major || cap > k.n_tiles) ? cap : k.n_tiles;
    k.per_xcd = cdiv(k.ntiles, 8);
    k.slots = k.item_major ? wpx : wpx / k.n_tiles;
    if (!k.item_major && k.slots > k.per_xcd) k.slots = k.per_xcd;
    if (k.slots < 1) k.slots = 1;
    const int wgs = (k.item_major ? k.slots : k.slots * k.n_tiles) * 8;
    return ni == 2 ? l1_launch<2>(k, wgs, st) : l1_launch<1>(k, wgs, st);
}


// ---- conv_stem16_kernel host side.  SALT_CONV_STEM16 = 0: off unless asked for per launch (cfg & 0xff == 13).
int conv_stem16_variant(const salt_conv_args* a) {
    static const int env = getenv("SALT_CONV_STEM16") ? atoi(getenv("SALT_CONV_STEM16")) : 1;
    if (!a || a->dtype != SALT_BF16 || a->ntaps != 16) return 0;
    const bool asked = (a->cfg & 0xff) == 13;
    if ((a->cfg & 0xff) != 0 && !asked) return 0;
    if (!asked && !env) return 0;
    if (a->in_step != 1 || a->out_step != 1 || a->out_oy || a->out_ox || a->nphase > 1 || a->pad_mode != 0) return 0;
    if (a->strip || a->fold_top || a->fold_bottom || a->fold_left || a->fold_right || a->x_plane || a->y_plane || a->res.p || a->accumulate) return 0;
    if (a->stats || a->fin_ticket || a->bnb_acc || a->bnb_partials || a->bnb_ticket || a->in_scale || a->in_fin_acc) return 0;
    if (a->fin_acc && (a->bias || a->scale || a->shift || a->relu)) { /* MODE 1 applies the affine part before the sums, as conv_ws_kernel */ }
    if (a->x.C != 16 || a->y.C != 64 || a->x.B != a->y.B || a->OH != a->y.H || a->OW != a->y.W || a->OH % 16 || a->OW % 16) return 0;
    int min_dy = 1 << 30, max_dy = -(1 << 30), min_dx = 1 << 30, max_dx = -(1 << 30);
    for (int t = 0; t < 16; ++t) {
        min_dy = a->tap_dy[t] < min_dy ? a->tap_dy[t] : min_dy; max_dy = a->tap_dy[t] > max_dy ? a->tap_dy[t] : max_dy;
        min_dx = a->tap_dx[t] < min_dx ? a->tap_dx[t] : min_dx; max_dx = a->tap_dx[t] > max_dx ? a->tap_dx[t] : max_dx;
    }
    if (max_dy - min_dy > 3 || max_dx - min_dx > 3) return 0;
    if (a->x.cs % 8 || a->y.cs % 8 || ((reinterpret_cast<uintptr_t>(a->x.p) | reinterpret_cast<uintptr_t>(a->y.p) | reinterpret_cast<uintptr_t>(a->w)) & 15)) return 0;
    if ((int64_t)a->x.B * a->x.H * a->x.W * a->x.cs >= (int64_t)1 << 31 || (int64_t)a->y.B * a->y.H * a->y.W * a->y.cs >= (int64_t)1 << 31) return 0;
    if (!asked && (int64_t)a->y.B * (a->OH / 16) * (a->OW / 16) < ws_cus() / 2) return 0;
    return 1;
}

int conv_stem16_launch(const salt_conv_args* a, hipStream_t st) {
    if (!conv_stem16_variant(a)) SALT_FAIL(SALT_E_UNSUPPORTED, "conv_stem16: not applicable");
    S16KP k;
    k.x = reinterpret_cast<const bf16_t*>(a->x.p); k.w = reinterpret_cast<const bf16_t*>(a->w); k.y = reinterpret_cast<bf16_t*>(a->y.p);
    k.bias = a->bias; k.scale = a->scale; k.shift = a->shift;
    k.B = a->x.B; k.H = a->x.H; k.W = a->x.W; k.x_cs = a->x.cs; k.y_cs = a->y.cs; k.OH = a->OH; k.OW = a->OW;
    k.tiles_x = a->OW / 16; k.tiles_y = a->OH / 16; k.ntiles = a->y.B * k.tiles_x * k.tiles_y;
    int min_dy = 1 << 30, min_dx = 1 << 30;
    for (int t = 0; t < 16; ++t) { min_dy = a->tap_dy[t] < min_dy ? a->tap_dy[t] : min_dy; min_dx = a->tap_dx[t] < min_dx ? a->tap_dx[t] : min_dx; }
    k.min_dy = min_dy; k.min_dx = min_dx;
    for (int t = 0; t < 16; ++t) k.tap_off[t] = (a->tap_dy[t] - min_dy) * 19 + (a->tap_dx[t] - min_dx);
    k.relu = a->relu; k.fin_acc = a->fin_acc;
    return a->fin_acc ? stem16_launch_mode<1>(k, st) : stem16_launch_mode<0>(k, st);
}

#if SALT_WS_CLK      // clock-instrumented variant builds only (tools/build_variant.sh -DSALT_WS_CLK=1): not part of the C-ABI of the shipped library
extern "C" int salt_debug_ws_clk(unsigned long long* host_out, int n) {
    if (n > 256 * 48) n = 256 * 48;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ws_clk), (size_t)n * sizeof(unsigned long long));
}
#endif
