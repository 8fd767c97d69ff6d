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
//       EPI(tile k): the same waves, one phase later, with the tile still in their accumulators: BatchNorm statistics from the
//                    registers, bf16 rounding, transposition of the wave's own 64 x Cout block through a WAVE-PRIVATE LDS region
//                    (no cross-wave dependency, hence no barrier), whole 128-byte NHWC pixel rows to HBM, (+)= / BatchNorm-backward
//                    sums for data gradients, and the LDS-DMA of the halo of tile k+2 into the buffer tile k just released.
//     So the epilogue + next-tile load of one group hide under the MFMAs of the other; each SIMD hosts one wave of either group.
//     ONE raw s_barrier per phase (counted vmcnt, never __syncthreads: DMA stays in flight across it).
//   * 64 -> 64 leaves no LDS for a separate transposition buffer: a wave transposes inside the slice of ITS group's halo buffer
//     that its OWN next-tile DMA pieces will overwrite (contiguous piece ownership), and issues those pieces only after its reads.
//   * Per-workgroup sums (BatchNorm forward statistics, BatchNorm-backward sums) are carried in registers across all tiles of the
//     workgroup and leave as ONE set of fp64 shard atomics per workgroup (salt_conv_args.fin_acc / bnb_acc without ticket).
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
__device__ unsigned long long g_ws_clk[256 * 32];
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

template <int NCH, int NI>
__global__ __launch_bounds__(512) void conv_ws_kernel(WsKP p) {
    typedef bf16_t T;
    constexpr int BN = 32 * NI, NT = 9, MI = 2;
    constexpr int HPC = 21;                            // halo DMA pieces (16 rows x 64 B) per chunk: 18 x 18 = 324 rows, padded to 336
    constexpr int HP = NCH * HPC;                      // halo pieces per tile
    constexpr int WP = NCH * NT * BN / 16;             // weight pieces
    constexpr int NSW = (WP + 7) / 8;                  // weight DMA instructions per wave (8 waves)
    constexpr int NSH = (HP + 3) / 4;                  // halo DMA instructions per wave and tile (4 waves of a group)
    constexpr bool ALIAS = (NCH == 2 && NI == 2);      // no room for a separate transposition buffer
    constexpr int W_BYTES = WP * 1024, H_BYTES = HP * 1024, HC_BYTES = HPC * 1024;
    constexpr int PITCHB = BN * 2 + 16;                // staged pixel row: BN bf16 + 16 bytes (bank spread)
    constexpr int STG_WAVE = 64 * PITCHB;
    constexpr int OFF_H = W_BYTES, OFF_STG = OFF_H + 2 * H_BYTES;
    constexpr int OFF_DUMMY = ALIAS ? OFF_STG : OFF_STG + 4 * STG_WAVE;
    constexpr int OFF_BNB = OFF_DUMMY + 1024;          // [4][BN] floats: mean, invstd, gamma * invstd, beta - mean * gamma * invstd
    constexpr int PPO = BN / 8;                        // 16-byte pieces per output pixel row
    constexpr int NPC = 64 * PPO / 64;                 // pieces per lane of a wave's 64 x BN block (= PPO)
    static_assert(!ALIAS || (HP == 42 && STG_WAVE <= 10 * 1024), "aliased transposition slices");
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
    unsigned long long clk[16]; int nclk = 0;
    auto stamp = [&]() { if (nclk < 16) clk[nclk++] = __builtin_readcyclecounter(); };
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
    // halo piece owned by slot i of wave wm: ALIAS - a contiguous range (11, 11, 10, 10 pieces), so that the wave's transposition
    // slice is exactly what its own DMA overwrites; otherwise round-robin
    const int a_start = wm * 10 + (wm < 2 ? wm : 2), a_cnt = wm < 2 ? 11 : 10;
    auto issue_halo = [&](const TC& c, int g) {
        int ln = lane;
        asm volatile("" : "+v"(ln));             // keeps the per-piece index math INSIDE the tile loop: hoisted to kernel entry it is spilled around the MFMA phases
        const T* xb = p.x + (int64_t)c.b * p.H * p.W * p.x_cs;
        const int iy0 = c.oy0 + p.min_dy, ix0 = c.ox0 + p.min_dx;
        const bool clamp = p.pad_mode != 0;
#pragma unroll
        for (int i = 0; i < NSH; ++i) {
            const int pidx = ALIAS ? a_start + i : wm + 4 * i;
            const bool real = ALIAS ? i < a_cnt : pidx < HP;
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

    // ---- prologue: the weights (all waves), then each group's first halo tile
    issue_weights();
    if (grp < n_my) issue_halo(coords(tile_of(grp)), grp);

    // ---- fragment addressing: the lane's halo pixel per M sub-tile is tile invariant; the 18 + 18 A addresses derived from it are
    // recomputed at the head of every MFMA phase (kept live across the epilogue they were spilled to scratch)
    int pbase[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm * 64 + i * 32 + ws_perm(l31);
        pbase[i] = (m >> 4) * 18 + (m & 15);
    }
    f32x16 acc[MI][NI];
    struct Frag { u32x4 a[MI], b[NI]; };
    auto mma_tile = [&](int g) {
        const unsigned char* hb = smem + OFF_H + g * H_BYTES;
        int a_addr[NT][MI], b_addr[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            int pb = pbase[i];
            asm volatile("" : "+v"(pb));                                     // not hoistable: see above
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
        auto mma_frag = [&](const Frag& f) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[i]), __builtin_bit_cast(bf16x8, f.b[j]), acc[i][j], 0, 0, 0);
        };
        constexpr int NST = NCH * NT * 2;
        Frag f0, f1;
        load_frag(0, f0);
        __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
#pragma unroll
        for (int s = 0; s < NST; s += 2) {
            load_frag(s + 1, f1);
            mma_frag(f0);
            __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);         // the reads of stage s+1 ...
            __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);         // ... then the MFMAs of stage s
            if (s + 2 < NST) load_frag(s + 2, f0);
            mma_frag(f1);
            if (s + 2 < NST) __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
        }
    };

    // ---- per-lane epilogue constants and per-workgroup running sums
    const bool want_stats = p.fin_acc != nullptr;
    const bool bnb = p.bnb_acc != nullptr;
    const bool has_affine = p.bias || p.scale || p.shift || p.relu;
    float ep_bias[NI], ep_sc[NI], ep_sh[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = j * 32 + l31;
        ep_bias[j] = p.bias ? p.bias[n] : 0.f; ep_sc[j] = p.scale ? p.scale[n] : 1.f; ep_sh[j] = p.shift ? p.shift[n] : 0.f;
    }
    double st_s[NI], st_q[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) { st_s[j] = 0.0; st_q[j] = 0.0; }
    const int pc = lane % PPO, prow = lane / PPO;                          // this lane's channel piece / first pixel row of its pieces
    float b1[8], b2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { b1[e] = 0.f; b2[e] = 0.f; }
    if (bnb && tid < BN) {                                                 // BatchNorm constants of the layer whose backward sums ride along
        float* sb = reinterpret_cast<float*>(smem + OFF_BNB);
        const float mu = p.bnb_mean[tid], is = p.bnb_invstd[tid], sc = p.bnb_gamma[tid] * is;
        sb[tid] = mu; sb[BN + tid] = is; sb[2 * BN + tid] = sc; sb[3 * BN + tid] = p.bnb_beta[tid] - mu * sc;
    }                                                                      // (read after the first phase barrier at the earliest)

    // epilogue of the tile in `acc` (computed by this wave one phase ago); issues the halo DMA of `next` (if any) into buffer g
    auto epilogue = [&](const TC& c, int g, bool has_next, const TC& next) {
        unsigned char* stg = smem + (ALIAS ? OFF_H + g * H_BYTES + a_start * 1024 : OFF_STG + wm * STG_WAVE);
        // global element offsets of this lane's pieces: piece it = pixel rows it * (64 / PPO) + prow of the wave's 64 pixels
        unsigned goff[NPC];
#pragma unroll
        for (int it = 0; it < NPC; ++it) {
            const int m = wm * 64 + it * (64 / PPO) + prow;
            goff[it] = (unsigned)((c.b * p.OH + c.oy0 + (m >> 4)) * p.OW + c.ox0 + (m & 15));
        }
        // affine / ReLU (eval, or the convolution bias), statistics of the fp32 values, bf16 transposition through the wave's slice
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            float ssum = 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r];
                    if (has_affine) {
                        v = (v + ep_bias[j]) * ep_sc[j] + ep_sh[j];
                        if (p.relu) v = fmaxf(v, 0.f);
                        acc[i][j][r] = v;
                    }
                    ssum += v;
                    const int ml = i * 32 + ws_perm((r & 3) + 8 * (r >> 2) + 4 * khalf);
                    *reinterpret_cast<T*>(stg + ml * PITCHB + (j * 32 + l31) * 2) = f2bf(v);
                }
            if (want_stats) {
                const float s = ssum + __shfl_xor(ssum, 32);
                const float mean = s * (1.f / 64.f);
                float m2 = 0.f;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const float d = acc[i][j][r] - mean; m2 += d * d; }
                m2 += __shfl_xor(m2, 32);
                st_s[j] += (double)s;
                st_q[j] += (double)m2 + (double)s * (double)s * (1.0 / 64.0);
            }
        }
        // the accumulators are dead from here.  Two batches of NPC / 2 pieces (bounds the live registers): operand loads of the (+)= /
        // BatchNorm-backward epilogue, staged pieces, sums, stores
        float bmu[8], bis[8], bsc[8], bsh[8];
        if (bnb) {
            const float* sb = reinterpret_cast<const float*>(smem + OFF_BNB) + pc * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) { bmu[e] = sb[e]; bis[e] = sb[BN + e]; bsc[e] = sb[2 * BN + e]; bsh[e] = sb[3 * BN + e]; }
        }
        constexpr int HB = NPC / 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 oldv[HB], yv[HB], av[HB], sv[HB];
            if (p.accumulate) {
#pragma unroll
                for (int u = 0; u < HB; ++u) oldv[u] = *reinterpret_cast<const u32x4*>(p.y + (goff[h * HB + u] * (unsigned)p.y_cs + pc * 8));
            }
            if (bnb) {
#pragma unroll
                for (int u = 0; u < HB; ++u) yv[u] = *reinterpret_cast<const u32x4*>(p.bnb_y + (goff[h * HB + u] * (unsigned)p.bnb_cs + pc * 8));
                if (p.bnb_a) {
#pragma unroll
                    for (int u = 0; u < HB; ++u) av[u] = *reinterpret_cast<const u32x4*>(p.bnb_a + (goff[h * HB + u] * (unsigned)p.bnb_acs + pc * 8));
                }
            }
#pragma unroll
            for (int u = 0; u < HB; ++u) sv[u] = *reinterpret_cast<const u32x4*>(stg + ((h * HB + u) * (64 / PPO) + prow) * PITCHB + pc * 16);
#pragma unroll
            for (int u = 0; u < HB; ++u) {
                u32x4 stored = sv[u];
                if (p.accumulate) {
                    float f[8], o[8];
                    unpack16<T>(stored, f); unpack16<T>(oldv[u], o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += o[e];
                    stored = pack16<T>(f);
                }
                *reinterpret_cast<u32x4*>(p.y + (goff[h * HB + u] * (unsigned)p.y_cs + pc * 8)) = stored;
                if (bnb) {
                    float gq[8], yc[8];
                    unpack16<T>(stored, gq); unpack16<T>(yv[u], yc);
                    if (p.bnb_a) {
                        float a8[8];
                        unpack16<T>(av[u], a8);
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float gg = (!p.bnb_relu || a8[e] > 0.f) ? gq[e] : 0.f;
                            b1[e] += gg; b2[e] += gg * (yc[e] - bmu[e]) * bis[e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float gg = (!p.bnb_relu || yc[e] * bsc[e] + bsh[e] > 0.f) ? gq[e] : 0.f;
                            b1[e] += gg; b2[e] += gg * (yc[e] - bmu[e]) * bis[e];
                        }
                    }
                }
            }
        }
        if (has_next) {
            // the halo of this group's next tile goes into the buffer the group released at the phase barrier - LAST, so that the wave's
            // reads of its transposition slice (ALIAS: a slice of that very buffer) have returned and the plain vmcnt(0) in front of the
            // next phase barrier covers it whatever else (stores, compiler scratch traffic) shares the counter
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
        asm volatile("" ::: "memory");
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
    if (want_stats || bnb) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int C = BN;
        if (want_stats) {
            double* red = reinterpret_cast<double*>(smem);                  // [8 waves][BN][2]
            if (khalf == 0) {
#pragma unroll
                for (int j = 0; j < NI; ++j) { red[(wave * BN + j * 32 + l31) * 2] = st_s[j]; red[(wave * BN + j * 32 + l31) * 2 + 1] = st_q[j]; }
            }
            __syncthreads();
            if (tid < 2 * BN) {
                const int st = tid / BN, n = tid - st * BN;
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < 8; ++w) t += red[(w * BN + n) * 2 + st];
                double* a = p.fin_acc + (blockIdx.x & 7) * (2 * C + 1);
                fin_add(a + st * C + n, t);
                if (tid == 0) fin_add(a + 2 * C, (double)n_my * 256.0);
            }
            __syncthreads();
        }
        if (bnb) {
            float* red = reinterpret_cast<float*>(smem);                    // [8 waves][64 / PPO rows][BN][2]
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[((wave * (64 / PPO) + prow) * BN + pc * 8 + e) * 2] = b1[e];
                red[((wave * (64 / PPO) + prow) * BN + pc * 8 + e) * 2 + 1] = b2[e];
            }
            __syncthreads();
            if (tid < 2 * BN) {
                const int st = tid / BN, n = tid - st * BN;
                float t = 0.f;
                for (int r = 0; r < 8 * (64 / PPO); ++r) t += red[(r * BN + n) * 2 + st];
                fin_add(p.bnb_acc + ((blockIdx.x & 7) * 2 + st) * C + n, (double)t);
            }
        }
    }
#if SALT_WS_CLK
    stamp();
    if (lane == 0 && (wave == 0 || wave == 4) && blockIdx.x < 256) {
        unsigned long long* o = g_ws_clk + (blockIdx.x * 2 + grp) * 16;
        for (int i = 0; i < 16; ++i) o[i] = i < nclk ? clk[i] : 0ull;
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

template <int NCH, int NI>
int ws_launch(const WsKP& k, hipStream_t st) {
    constexpr int BN = 32 * NI, HP = NCH * 21, WP = NCH * 9 * BN / 16;
    constexpr bool ALIAS = (NCH == 2 && NI == 2);
    constexpr int LDS = WP * 1024 + 2 * HP * 1024 + (ALIAS ? 0 : 4 * 64 * (BN * 2 + 16)) + 1024 + 4 * BN * 4;
    static_assert(LDS <= 160 * 1024, "LDS budget");
    auto kern = conv_ws_kernel<NCH, NI>;
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

extern "C" int salt_debug_ws_clk(unsigned long long* host_out, int n) {
#if SALT_WS_CLK
    if (n > 256 * 32) n = 256 * 32;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ws_clk), (size_t)n * sizeof(unsigned long long));
#else
    (void)host_out; (void)n;
    return SALT_E_UNSUPPORTED;
#endif
}
