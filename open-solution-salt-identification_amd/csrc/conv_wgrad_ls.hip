// conv_wgrad_ls.hip — loader-specialised, row-streaming weight gradient of the 3x3 unit-step convolutions (bf16, gfx950).
//
//   dW[t][a][b] = sum_p P[p, a] * Q[pad(p + tap_t), b]      (P = dL/dy, Q = x; loss.backward() of common_blocks/models.py:133 over
//                                                            architectures/base.py:7-37, unet_models.py:21-30, torchvision BasicBlock)
//
// Why a third weight-gradient structure.  conv_wgrad_fast8_kernel (conv_mfma.hip) stages every 128-pixel tile global -> registers ->
// LDS inside the waves that feed the matrix pipe: its k-loop takes 15.5 us with and 9.3 us without the global loads - the
// ~115 issue cycles of every global_load_dwordx4 ADD to the MFMA time, exactly what conv_ls_kernel's clocks showed for the forward
// convolutions.  Same cure here:
//   * 8 waves = 4 LOADER waves + 4 MFMA waves (one of each per SIMD).  Loaders move global -> LDS by LDS-DMA (global_load_lds_dwordx4,
//     1 KB per wave instruction, no staging registers, no ds_write pass) from precomputed per-lane offsets; MFMA waves issue only
//     ds_read_b64_tr_b16 + v_alignbit + v_mfma_f32_32x32x16_bf16 (64 x 64 channel block x 9 taps, 144 accumulator registers each).
//   * ROW STREAMING instead of pixel tiles: a workgroup walks 16-pixel-wide column strips of an image top to bottom.  A unit is KU
//     pixel rows (KU k-steps of 16 pixels): KU x 2 KB of P and KU NEW halo rows of Q (KU x 18 pixels) - the two halo rows a unit shares
//     with its predecessor stay in the predecessor's ring slot, so Q is read 18/16 times instead of 180/128 times per pixel and a ring
//     slot is 17 KB (KU = 4) instead of a 39 KB tile.  A strip (or a split that starts mid-strip) opens with a "pre" entry that loads
//     only the two halo rows above its first row.  Maps of width <= 8 (the 512-channel levels) put TWO images side by side in a
//     k-step (NB = 2: 2 x (8 + 2) halo pixels per row).
//   * Ring of NS slots, ONE raw s_barrier per entry, counted vmcnt (never 0 inside the loop): the loaders run NS - 2 entries ahead.
//   * LDS rows are 128 B (64 channels) with the 16-byte slots XOR-swizzled by bit 1 of the row index (applied to the per-lane SOURCE
//     address of the DMA): the transposed reads of 4 consecutive pixel rows x 64 B hit 64 distinct banks (tools/wgrad_ls_model.py
//     checks the loader map, every fragment address and the bank sets on the CPU) - no 192-byte row padding, which LDS-DMA's
//     lane-linear destination could not produce.
//   * Workgroup -> XCD: the (split, a-block, b-block) triples are laid out so that the triples of one XCD share a split (and an
//     a-block): P / Q of that split are fetched into that XCD's L2 once.
// Partial-slab format, operand swap (16-byte slab stores) and salt_wgrad_reduce are those of conv_wgrad_fast_kernel.
#include <cstdlib>
#include <type_traits>
#include "common.h"

#ifndef SALT_WL_NS4
#define SALT_WL_NS4 6            // ring slots, KU = 4 (17 / 18 KB each)
#endif
#ifndef SALT_WL_NS8
#define SALT_WL_NS8 4            // ring slots, KU = 8 (34 / 36 KB each)
#endif
#ifndef SALT_WL_NS2S
#define SALT_WL_NS2S 6           // ring slots of the stride-2 variant (KU = 2: 22 KB each)
#endif
#ifndef SALT_WL_ABLATE
#define SALT_WL_ABLATE 0         // timing ablations (results are wrong): 1 no fragment reads / MFMAs, 2 no DMA, 4 no slab stores
#endif

#ifndef SALT_WL_CLK
#define SALT_WL_CLK 0            // 1: per-workgroup s_memtime stamps into g_wl_clk (tools/wl_clocks.py; timing build only)
#endif

namespace {

#if SALT_WL_CLK
__device__ unsigned long long g_wl_clk[1024 * 16];
#define WL_T() __builtin_amdgcn_s_memtime()
#endif

struct WlKP {
    const bf16_t* P; const bf16_t* Q; float* partials;
    int B, PH, PW, Ca, p_cs;
    int QH, QW, Cb, q_cs;
    int min_dy, min_dx;
    int ncol, U, total, upw, nsplit;     // strips = image groups x ncol column blocks, U units each; units per workgroup
    int a_blocks, b_blocks, V, per_xcd;  // V = nsplit * a_blocks * b_blocks workgroup triples, per_xcd of them on every XCD
    long long q_plane;
};

__device__ __attribute__((aligned(16))) unsigned int g_wl_zero[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

template <int N> __device__ __forceinline__ void wl_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}

// position in the entry stream of a workgroup: unit u (global index), its row block k inside strip (img, cx); `pre`: the two-halo-row
// entry in front of unit u is still to come
struct WlCur { int u, k, cx, img; bool pre; };

// ST = 2: the stride-2 3x3 convolutions (ResNet layer2-4 conv1; Q = x at twice P's resolution, zero pad 1).  A Q image row is stored as
// TWO 18-pixel halo rows - its even-column plane E (hx = 2 k) and its odd one O (hx = 2 k + 1), de-interleaved by the loader's source
// addresses - so the fragment reads stay unit-stride and conflict-free: tap dx = 0 -> E[k], dx = 1 -> O[k], dx = 2 -> E[k + 1].  A
// k-step (P row y) needs image rows 2 y - 1, 2 y, 2 y + 1: two new ones per k-step, the third is the previous k-step's last.  A unit
// of KU k-steps brings 2 KU image rows = 4 KU halo rows; a pre entry the image row above the first k-step.
template <bool PAD, int NB, int KU, int ST = 1>
__global__ __launch_bounds__(512) void conv_wgrad_ls_kernel(WlKP p) {
    constexpr int TW = 16 / NB, HP = ST == 2 ? 18 : NB * (TW + 2), KHS = NB == 1 ? 8 : (ST == 2 ? TW + 1 : TW + 2);
    constexpr int QIR = ST == 2 ? 2 * KU : KU;                          // Q image rows a unit brings
    constexpr int P_BYTES = KU * 16 * 128, QROWS = (ST == 2 ? 2 : 1) * QIR * HP, Q_BYTES = QROWS * 128, SLOT = P_BYTES + Q_BYTES;
    constexpr int NPP = P_BYTES / 1024, NQP = QROWS / 8, PCS = NPP + NQP;
    constexpr int PRE_FIRST = ((ST == 2 ? 2 * (QIR - 1) : KU - 2) * HP) / 8;  // first Q piece a pre entry loads
    constexpr int NS = ST == 2 ? SALT_WL_NS2S : (KU == 4 ? SALT_WL_NS4 : SALT_WL_NS8);
    static_assert(ST == 1 || !PAD, "stride 2: zero padding only");
    constexpr int OFF_DUMMY = NS * SLOT;
    constexpr int INVALID = (int)0x80000000;                             // voffset >= num_records: the lane reads zeros
    constexpr int NPS_ = (NPP + 3) / 4, NQS_ = (NQP + 3) / 4, PPW = NPS_ + NQS_;
    static_assert(QROWS % 8 == 0 && NS >= 3 && (NS - 3) * PPW <= 63, "ring geometry");
    static_assert(OFF_DUMMY + 1024 <= 160 * 1024, "LDS budget");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;

    // ---- which (split, a-block, b-block) triple: XCD x owns the triples [x per_xcd, (x + 1) per_xcd)
    const int v = (int)(blockIdx.x & 7) * p.per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= p.per_xcd || v >= p.V) return;
    const int bb = v % p.b_blocks;
    const int ab = (v / p.b_blocks) % p.a_blocks;
    const int split = v / (p.b_blocks * p.a_blocks);
    const int a0 = ab * 64, c0 = bb * 64;
    const int u0 = min(split * p.upw, p.total), u1 = min(u0 + p.upw, p.total);
    const int G = u1 > u0 ? (u1 - u0) + (u1 - 1) / p.U - u0 / p.U + 1 : 0;       // units + one pre entry per strip touched

    auto cur_init = [&]() {
        WlCur c; c.u = u0; const int strip = u0 / p.U; c.k = u0 - strip * p.U; c.img = strip / p.ncol; c.cx = strip - c.img * p.ncol; c.pre = true;
        return c;
    };
    auto cur_next = [&](WlCur& c) {
        if (c.pre) { c.pre = false; return; }
        ++c.u;
        if (++c.k == p.U) { c.k = 0; c.pre = true; if (++c.cx == p.ncol) { c.cx = 0; ++c.img; } }
    };

    if (loader) {
        // ================================================================== loader waves
        // Buffer-addressed LDS-DMA (buffer_load_dwordx4 ... lds): ONE descriptor per tensor built from kernel arguments (provably
        // uniform: no waterfall loop), a 32-bit per-lane byte offset, and out-of-range lanes (offset >= num_records) write ZEROS to
        // LDS (tools/probes/oob_lds_probe.hip) - zero padding, ragged edges and channel tails cost one v_cndmask instead of a
        // 64-bit pointer select.  A piece is ~8 instructions (version 1, flat addresses + full index math per piece: ~280 issue
        // cycles per piece, the loaders were the critical path at 11 B/clk/CU).
        const int lw = wave - 4;
        const bf16_t* Qg = p.Q + (p.q_plane ? (long long)bb * p.q_plane - c0 : 0);     // planar Q: channel c0 + j of block bb is element j of plane bb
        const auto rsP = __builtin_amdgcn_make_buffer_rsrc((void*)p.P, 0, INVALID, 0x00020000);
        const auto rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qg, 0, INVALID, 0x00020000);
        const int rowP2 = p.PW * p.p_cs * 2, rowQ2 = p.QW * p.q_cs * 2;               // row pitches in bytes
        // Piece slots of a wave: slots 0 .. NPS - 1 are P pieces lw + 4 i, slots NPS .. PPW - 1 are Q pieces lw + 4 (i - NPS) - the
        // tensor (hence the descriptor) of a slot is known at compile time: a runtime choice between the two descriptors made the
        // compiler select them lane-wise and wrap every DMA in a waterfall loop.
        // kernel-invariant lane constants: row (P: k-step, Q: halo row), x | image << 8, channel element offset (-1: beyond the
        // tensor's channels)
        constexpr int NPS = NPS_, NQS = NQS_;
        static_assert(NPS + NQS == PPW, "piece slots");
        int krow[PPW], kx[PPW], kch[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            krow[i] = 0; kx[i] = 0; kch[i] = -1;
            if (i < NPS) {
                if (lw + 4 * i < NPP) {
                    const int r = (lw + 4 * i) * 8 + (lane >> 3);
                    const int cs = (lane & 7) ^ (((r >> 1) & 1) << 2);
                    const int px = r & 15;
                    krow[i] = r >> 4;
                    kx[i] = NB == 1 ? px : ((px & 7) | ((px >> 3) << 8));
                    if (a0 + cs * 8 < p.Ca) kch[i] = a0 + cs * 8;
                }
            } else if (lw + 4 * (i - NPS) < NQP) {
                const int r = (lw + 4 * (i - NPS)) * 8 + (lane >> 3);
                const int cs = (lane & 7) ^ (((r >> 1) & 1) << 2);
                const int hrl = r / HP, hx = r - hrl * HP;
                if (ST == 1) {
                    const int im = NB == 1 ? 0 : hx / (TW + 2);
                    krow[i] = hrl;
                    kx[i] = (hx - im * (TW + 2)) | (im << 8);
                    if (c0 + cs * 8 < p.Cb) kch[i] = c0 + cs * 8;
                } else {
                    // halo row hrl = plane (hrl & 1) of image row hrl >> 1; plane pixel k of image im -> image column 2 (x0 + k) + plane + min_dx
                    const int im = NB == 1 ? 0 : hx / (TW + 1), k = hx - im * (TW + 1);
                    krow[i] = hrl >> 1;
                    kx[i] = k | (im << 8) | ((hrl & 1) << 16);
                    if (c0 + cs * 8 < p.Cb && k <= TW) kch[i] = c0 + cs * 8;
                }
            }
        }
        // per strip (rebuilt at its pre entry): byte offset of the lane's piece from the tensor base without the entry's row term,
        // and the lane's row number, 255 where the lane is masked for the whole strip (channel tail, column / image beyond the tensor)
        int voff[PPW], vrow[PPW];
        auto strip_tables = [&](const WlCur& c) {
            const int b0 = c.img * NB, x0 = c.cx * TW;
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int x = kx[i] & 255, im = (kx[i] >> 8) & 255;
                if (i < NPS) {
                    const bool ok = kch[i] >= 0 && x0 + x < p.PW && b0 + im < p.B;
                    voff[i] = (b0 + im) * p.PH * rowP2 + ((x0 + x) * p.p_cs + kch[i]) * 2 + krow[i] * rowP2;
                    vrow[i] = ok ? krow[i] : 255;
                } else {
                    int ix = ST == 2 ? 2 * (x0 + x) + (kx[i] >> 16) + p.min_dx : x0 + p.min_dx + x;
                    bool ok = kch[i] >= 0 && b0 + im < p.B;
                    if (PAD) ix = min(max(ix, 0), p.QW - 1);
                    else ok = ok && (unsigned)ix < (unsigned)p.QW;
                    voff[i] = (b0 + im) * p.QH * rowQ2 + (ix * p.q_cs + kch[i]) * 2 + (PAD ? 0 : krow[i] * rowQ2);
                    vrow[i] = ok ? krow[i] : 255;
                }
            }
        };
        auto issue = [&](bool live, const WlCur& c, int slot) {
            if (live && c.pre) strip_tables(c);
            const int y0 = c.k * KU;
            // image row of the slot's Q row 0 (a pre entry is the tail of the unit that would precede the segment)
            const int yq = ST == 2 ? 2 * (c.pre ? y0 - KU : y0) + 1 + p.min_dy : (c.pre ? y0 - KU : y0) + 2 + p.min_dy;
            const int base = slot * SLOT;
            const bool p_on = live && !c.pre;
            const int entP = y0 * rowP2, hiP = p_on ? min(KU, p.PH - y0) : 0;
            const int entQ = yq * rowQ2;
            const int loQ = max(max(0, -yq), c.pre ? (ST == 2 ? QIR - 1 : KU - 2) : 0), hiQ = max(loQ, min(QIR, p.QH - yq));
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                if (i < NPS) {
                    const int dst = p_on && lw + 4 * i < NPP ? base + (lw + 4 * i) * 1024 : OFF_DUMMY;
                    const int vo = (unsigned)vrow[i] < (unsigned)hiP ? voff[i] + entP : INVALID;
                    if (!(SALT_WL_ABLATE & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (lds_ptr_t)(smem + dst), 16, vo, 0, 0, 0);
                } else {
                    const int qi = lw + 4 * (i - NPS);
                    const bool q_on = live && qi < NQP && (!c.pre || qi >= PRE_FIRST);
                    const int dst = q_on ? base + P_BYTES + qi * 1024 : OFF_DUMMY;
                    int vo;
                    if (PAD) {
                        const int iyc = min(max(yq + vrow[i], 0), p.QH - 1);          // rows a pre entry does not need are loaded anyway (clamped: valid)
                        vo = (unsigned)vrow[i] < (q_on ? 255u : 0u) ? voff[i] + iyc * rowQ2 : INVALID;
                    } else {
                        vo = (unsigned)(vrow[i] - loQ) < (unsigned)(q_on ? hiQ - loQ : 0) ? voff[i] + entQ : INVALID;
                    }
                    if (!(SALT_WL_ABLATE & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (lds_ptr_t)(smem + dst), 16, vo, 0, 0, 0);
                }
            }
        };
        WlCur lc = cur_init();
        int ig = 0;
#if SALT_WL_CLK
        const unsigned long long lt0 = WL_T(); unsigned long long lt_vm = 0, lt_bar = 0, lt_iss = 0;
#endif
#pragma unroll 1
        for (int d = 0; d < NS - 2; ++d) { issue(ig < G, lc, ig % NS); if (ig < G) { cur_next(lc); ++ig; } }
#if SALT_WL_CLK
        const unsigned long long lt1 = WL_T();
#endif
#pragma unroll 1
        for (int g = 0; g < G; ++g) {
#if SALT_WL_CLK
            const unsigned long long ta = WL_T();
#endif
            wl_wait_vm<(NS - 3) * PPW>();                                  // this wave's pieces of entry g have landed
#if SALT_WL_CLK
            const unsigned long long tb = WL_T();
#endif
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();                                  // ... everybody's; and the MFMA waves are done with entry g - 2
            asm volatile("" ::: "memory");
#if SALT_WL_CLK
            const unsigned long long tc = WL_T();
#endif
            issue(ig < G, lc, ig % NS);                                    // entry g + NS - 2 into the slot entry g - 2 released
            if (ig < G) { cur_next(lc); ++ig; }
#if SALT_WL_CLK
            const unsigned long long td = WL_T();
            lt_vm += tb - ta; lt_bar += tc - tb; lt_iss += td - tc;
#endif
        }
        wl_wait_vm<0>();                                                   // trailing dummy pieces: no DMA may outlive the workgroup's LDS
#if SALT_WL_CLK
        if (wave == 4 && lane == 0 && blockIdx.x < 1024) {
            unsigned long long* o = g_wl_clk + blockIdx.x * 16 + 8;
            o[0] = lt0; o[1] = lt1; o[2] = lt_vm; o[3] = lt_bar; o[4] = lt_iss; o[5] = WL_T();
        }
#endif
        return;
    }

    // ================================================================== MFMA waves
    // A k-step (16 pixels of one P row) needs halo rows j, j + 1, j + 2 of Q; consecutive k-steps share two of them, so the wave
    // keeps FOUR row buffers in registers (three in use, one being filled) and reads every halo row from LDS once: per k-step 2
    // transposed reads of P + 4 of Q (pixels 0-3, 4-7 of the 12-pixel run for tap dx = 0, pixels 2-5, 6-9 for dx = 2 - aligned
    // register quads, no v_mov - and dx = 1 by four v_alignbit), all issued ONE K-STEP AHEAD of their MFMAs.  The prefetch crosses
    // entry boundaries: the barrier of entry g + 1 is taken at the head of the LAST k-step of entry g, whose MFMAs then cover
    // the first reads of entry g + 1 (4 buffers x KU = 4 / 8 k-steps: the buffer rotation is static per unit).
    const int wa = wave >> 1, wb = wave & 1;
    const int l31 = lane & 31, khalf = lane >> 5, g16 = (lane >> 4) & 1, i16 = lane & 15, prow = i16 >> 2, pcol = (i16 & 3) * 4;
    const int bytec = (pcol & 7) * 2;
    const int csA = wa * 4 + g16 * 2 + (pcol >> 3), csB = wb * 4 + g16 * 2 + (pcol >> 3);
    const int sA = (prow >> 1) & 1;
    const int paL = (khalf * 8 + prow) * 128 + ((csA ^ (sA << 2)) << 4) + bytec;
    int qL0[2], qL2[2];                                                  // runs from pixel 0 / pixel 2, by the parity of the halo row inside its slot
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        // bit 1 of the LDS row index of this lane's pixel: halo row hrl starts at row hrl HP (HP / 2 odd: its parity enters)
        const int s = (((khalf * KHS + prow) >> 1) & 1) ^ (((HP / 2) & 1) ? par : 0);
        qL0[par] = P_BYTES + (khalf * KHS + prow) * 128 + ((csB ^ (s << 2)) << 4) + bytec;
        qL2[par] = P_BYTES + (khalf * KHS + prow + 2) * 128 + ((csB ^ ((s ^ 1) << 2)) << 4) + bytec;
    }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    struct Row { u32x2 a0, a1, c0, c1; u32x4 sh; };                       // pixels 0-3, 4-7, 2-5, 6-9 of the wave's run (x 2 k-halves); sh: the dx = 1 operand
    struct AFr { s16x4 lo, hi; };
    auto tr = [&](int addr) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + addr)); };
    auto read_row = [&](int sb, int hrl, Row& r) {                        // hrl constant after unrolling
        const int par = hrl & 1;
        r.a0 = __builtin_bit_cast(u32x2, tr(sb + qL0[par] + (hrl * HP) * 128));
        r.a1 = __builtin_bit_cast(u32x2, tr(sb + qL0[par] + (hrl * HP + 4) * 128));
        r.c0 = __builtin_bit_cast(u32x2, tr(sb + qL2[par] + (hrl * HP) * 128));
        r.c1 = __builtin_bit_cast(u32x2, tr(sb + qL2[par] + (hrl * HP + 4) * 128));
    };
    auto read_a = [&](int sb, int j, AFr& a) {
        a.lo = tr(sb + paL + (j * 16) * 128);
        a.hi = tr(sb + paL + (j * 16 + 4) * 128);
    };
    // the dx = 1 operand (pixels 1 .. 8) of a row: four v_alignbit, made ONCE per row at the head of the k-step that uses the row first -
    // one k-step after its reads were issued (left to itself the compiler makes them right behind the reads and waits for those)
    auto make_sh = [&](Row& r) {
        r.sh = u32x4{__builtin_amdgcn_alignbit(r.a0.y, r.a0.x, 16), __builtin_amdgcn_alignbit(r.a1.x, r.a0.y, 16),
                     __builtin_amdgcn_alignbit(r.a1.y, r.a1.x, 16), __builtin_amdgcn_alignbit(r.c1.y, r.a1.y, 16)};
    };
    auto b_operand = [&](const Row& r, int dx) -> bf16x8 {                // dx constant after unrolling
        u32x4 vv;
        if (dx == 0) vv = u32x4{r.a0.x, r.a0.y, r.a1.x, r.a1.y};
        else if (dx == 2) vv = u32x4{r.c0.x, r.c0.y, r.c1.x, r.c1.y};
        else vv = r.sh;
        return __builtin_bit_cast(bf16x8, vv);
    };
    typedef short s16x8 __attribute__((ext_vector_type(8)));

    AFr ab2[2];
#if SALT_WL_CLK
    const unsigned long long mt0 = WL_T(); unsigned long long mt_first = 0, mt_bar = 0;
#define WL_BAR_T0() const unsigned long long tb0_ = WL_T()
#define WL_BAR_T1() do { const unsigned long long tb1_ = WL_T(); mt_bar += tb1_ - tb0_; if (!mt_first) mt_first = tb1_; } while (0)
#else
#define WL_BAR_T0()
#define WL_BAR_T1()
#endif
    auto barrier = [&]() {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    if constexpr (ST == 1) {
    Row rb[4];
    int g = 0, u = u0;
#pragma unroll 1
    while (g < G) {
        // ---- pre entry of this segment: halo rows -2, -1 (slot rows KU - 2, KU - 1) into row buffers 0, 1
        const int k0 = u % p.U;
        const int n = min(p.U - k0, u1 - u);                              // units of this segment (>= 1)
        { WL_BAR_T0(); barrier(); WL_BAR_T1(); }
        int sb = (g % NS) * SLOT;
        if (!(SALT_WL_ABLATE & 1)) { read_row(sb, KU - 2, rb[0]); read_row(sb, KU - 1, rb[1]); make_sh(rb[0]); make_sh(rb[1]); }
        ++g;
        { WL_BAR_T0(); barrier(); WL_BAR_T1(); }                          // its first unit
        sb = (g % NS) * SLOT;
        if (!(SALT_WL_ABLATE & 1)) { read_row(sb, 0, rb[2]); read_a(sb, 0, ab2[0]); }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
        for (int i = 0; i < n; ++i) {
            const bool has_next = i + 1 < n;
            int sbn = sb;
#pragma unroll
            for (int j = 0; j < KU; ++j) {
                // prefetch for k-step j + 1: P rows of k-step j + 1, halo row (j + 1) of this slot = row 0 of the next one after the last k-step
                if (j == KU - 1) {
                    if (has_next) {
                        { WL_BAR_T0(); barrier(); WL_BAR_T1(); }          // entry g + 1 landed
                        sbn = ((g + 1) % NS) * SLOT;
                    }
                    // (always issued - after the last unit of a segment from this slot again, unused - so that both paths carry the
                    //  same number of LDS reads: the compiler's s_waitcnt insertion is conservative at the join otherwise)
                    if (!(SALT_WL_ABLATE & 1)) { read_a(sbn, 0, ab2[(j + 1) & 1]); read_row(sbn, 0, rb[(j + 3) & 3]); }
                } else if (!(SALT_WL_ABLATE & 1)) {
                    read_a(sb, j + 1, ab2[(j + 1) & 1]);
                    read_row(sb, j + 1, rb[(j + 3) & 3]);
                }
                if (!(SALT_WL_ABLATE & 1)) {
                    make_sh(rb[(j + 2) & 3]);                             // the row whose reads were issued one k-step ago
                    const AFr& af = ab2[j & 1];
                    const s16x8 av = {af.lo[0], af.lo[1], af.lo[2], af.lo[3], af.hi[0], af.hi[1], af.hi[2], af.hi[3]};
                    // the six taps whose operands are register quads as they were read first, the three dx = 1 taps (v_alignbit) last
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        const int dy = q < 6 ? q >> 1 : q - 6, dx = q < 6 ? (q & 1) * 2 : 1, t = dy * 3 + dx;
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_operand(rb[(j + dy) & 3], dx), __builtin_bit_cast(bf16x8, av), acc[t], 0, 0, 0);
                    }
                    // pin: the six LDS reads of the NEXT k-step behind the first three MFMAs, this k-step's four v_alignbit behind the
                    // next two; nothing moves across the k-step boundary (the scheduler otherwise pulls the next row's v_alignbit up to
                    // its reads and waits for them: 1 510 instead of 1 152 cycles per 36-MFMA unit)
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (q < 3) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        if (q >= 3 && q < 5) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            sb = sbn; ++g;
        }
        __builtin_amdgcn_s_setprio(0);
        u += n;
    }
    } else {
    int g = 0, u = u0;
    // ---- stride 2: image-row buffers {E: pixels 0-3, 4-7, (6-)8-9; O: pixels 0-3, 4-7}; A = the row above the k-step (the previous
    // k-step's last row, copied), S[j & 1] = the k-step's two new rows, S[(j + 1) & 1] being filled for the next one
    struct Row2 { u32x2 ea0, ea1, ec1, oa0, oa1; };
    auto read_row2 = [&](int sb, int ir, Row2& r) {                      // ir (image row of the unit) constant after unrolling
        const int he = 2 * ir, ho = 2 * ir + 1;                           // halo rows of its planes: even parity / odd parity
        r.ea0 = __builtin_bit_cast(u32x2, tr(sb + qL0[0] + (he * HP) * 128));
        r.ea1 = __builtin_bit_cast(u32x2, tr(sb + qL0[0] + (he * HP + 4) * 128));
        r.ec1 = __builtin_bit_cast(u32x2, tr(sb + qL2[0] + (he * HP + 4) * 128));
        r.oa0 = __builtin_bit_cast(u32x2, tr(sb + qL0[1] + (ho * HP) * 128));
        r.oa1 = __builtin_bit_cast(u32x2, tr(sb + qL0[1] + (ho * HP + 4) * 128));
    };
    auto b_operand2 = [&](const Row2& r, int dx) -> bf16x8 {
        u32x4 vv;
        if (dx == 0) vv = u32x4{r.ea0.x, r.ea0.y, r.ea1.x, r.ea1.y};
        else if (dx == 1) vv = u32x4{r.oa0.x, r.oa0.y, r.oa1.x, r.oa1.y};
        else vv = u32x4{__builtin_amdgcn_alignbit(r.ea0.y, r.ea0.x, 16), __builtin_amdgcn_alignbit(r.ea1.x, r.ea0.y, 16),
                        __builtin_amdgcn_alignbit(r.ea1.y, r.ea1.x, 16), __builtin_amdgcn_alignbit(r.ec1.y, r.ea1.y, 16)};
        return __builtin_bit_cast(bf16x8, vv);
    };
    Row2 ra, rs2[2][2];                                                   // rs2[set][0 = even image row 2 y, 1 = odd image row 2 y + 1]
#pragma unroll 1
    while (g < G) {
        const int k0 = u % p.U;
        const int n = min(p.U - k0, u1 - u);
        { WL_BAR_T0(); barrier(); WL_BAR_T1(); }                          // pre entry: image row 2 y0 - 1 = the slot's last image row
        int sb = (g % NS) * SLOT;
        if (!(SALT_WL_ABLATE & 1)) read_row2(sb, QIR - 1, ra);
        ++g;
        { WL_BAR_T0(); barrier(); WL_BAR_T1(); }
        sb = (g % NS) * SLOT;
        if (!(SALT_WL_ABLATE & 1)) { read_row2(sb, 0, rs2[0][0]); read_row2(sb, 1, rs2[0][1]); read_a(sb, 0, ab2[0]); }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll 1
        for (int i = 0; i < n; ++i) {
            const bool has_next = i + 1 < n;
            int sbn = sb;
#pragma unroll
            for (int j = 0; j < KU; ++j) {
                if (j == KU - 1) {
                    if (has_next) {
                        { WL_BAR_T0(); barrier(); WL_BAR_T1(); }
                        sbn = ((g + 1) % NS) * SLOT;
                        if (!(SALT_WL_ABLATE & 1)) {
                            read_a(sbn, 0, ab2[(j + 1) & 1]);
                            read_row2(sbn, 0, rs2[(j + 1) & 1][0]); read_row2(sbn, 1, rs2[(j + 1) & 1][1]);
                        }
                    }
                } else if (!(SALT_WL_ABLATE & 1)) {
                    read_a(sb, j + 1, ab2[(j + 1) & 1]);
                    read_row2(sb, 2 * (j + 1), rs2[(j + 1) & 1][0]); read_row2(sb, 2 * (j + 1) + 1, rs2[(j + 1) & 1][1]);
                }
                if (!(SALT_WL_ABLATE & 1)) {
                    const AFr& af = ab2[j & 1];
                    const s16x8 av = {af.lo[0], af.lo[1], af.lo[2], af.lo[3], af.hi[0], af.hi[1], af.hi[2], af.hi[3]};
#pragma unroll
                    for (int q = 0; q < 9; ++q) {                         // dx = 0, 1 (register quads as read) first, dx = 2 (v_alignbit) last
                        const int dy = q < 6 ? q >> 1 : q - 6, dx = q < 6 ? (q & 1) : 2, t = dy * 3 + dx;
                        const Row2& rr = dy == 0 ? ra : rs2[j & 1][dy - 1];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_operand2(rr, dx), __builtin_bit_cast(bf16x8, av), acc[t], 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 9; ++q) {                         // pin: the 12 LDS reads of the next k-step behind the first six MFMAs
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (q < 6 && j != KU - 1) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        if (q < 6) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    }
                    ra = rs2[j & 1][1];                                   // the next k-step's row above
                }
            }
            sb = sbn; ++g;
        }
        __builtin_amdgcn_s_setprio(0);
        u += n;
    }
    }
#undef WL_BAR_T0
#undef WL_BAR_T1

#if SALT_WL_CLK
    const unsigned long long mt_loop = WL_T();
#endif
    // ---- partial slab partials[split][t][a][b]: lane = a-row, 4 consecutive registers = 4 consecutive b (operands swapped)
    const int a = a0 + wa * 32 + l31;
    if (a < p.Ca && !(SALT_WL_ABLATE & 4)) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float* row = p.partials + (((long long)split * 9 + t) * p.Ca + a) * p.Cb;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int b = c0 + wb * 32 + 8 * gq + 4 * khalf;
                if (b < p.Cb) *reinterpret_cast<f32x4*>(row + b) = f32x4{acc[t][4 * gq], acc[t][4 * gq + 1], acc[t][4 * gq + 2], acc[t][4 * gq + 3]};
            }
        }
    }
    if (SALT_WL_ABLATE & 4) {                                             // keep every accumulator alive without the stores
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[t][r];
        if (s == 123.456f) p.partials[0] = s;
    }
#if SALT_WL_CLK
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wave == 0 && lane == 0 && blockIdx.x < 1024) {
        unsigned long long* o = g_wl_clk + blockIdx.x * 16;
        o[0] = mt0; o[1] = mt_first; o[2] = mt_bar; o[3] = mt_loop; o[4] = WL_T(); o[5] = (unsigned long long)G;
    }
#endif
}

template <bool PAD, int NB, int KU, int ST = 1>
int wl_launch_inst(const WlKP& k, hipStream_t st) {
    constexpr int TW = 16 / NB, HP = ST == 2 ? 18 : NB * (TW + 2);
    constexpr int SLOT = KU * 16 * 128 + (ST == 2 ? 4 : 1) * KU * HP * 128, NS = ST == 2 ? SALT_WL_NS2S : (KU == 4 ? SALT_WL_NS4 : SALT_WL_NS8);
    constexpr int LDS = NS * SLOT + 1024;
    auto kern = conv_wgrad_ls_kernel<PAD, NB, KU, ST>;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(8 * k.per_xcd)), dim3(512), LDS, st, k);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

}  // namespace

#if SALT_WL_CLK      // clock-instrumented variant builds only (tools/build_variant.sh -DSALT_WL_CLK=1): not part of the C-ABI of the shipped library
extern "C" int salt_debug_wl_clk(unsigned long long* host_out, int n) {
    if (n > 1024 * 16) n = 1024 * 16;
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wl_clk), (size_t)n * sizeof(unsigned long long));
}
#endif

// salt_conv_wgrad / salt_conv_wgrad_nsplit try this first (conv_mfma.hip).  Returns 0 when the launch is not one of this kernel's
// shapes, else nsplit (>= 1); with `launch` it also enqueues the kernel (a->partials, a->nsplit as returned here) and puts the
// launch status into *rc.
int conv_wgrad_ls(const salt_conv_wgrad_args* a, bool launch, hipStream_t st, int* rc) {
    static const bool off = getenv("SALT_WGRAD_LS") && atoi(getenv("SALT_WGRAD_LS")) == 0;
    if (off || a->dtype != SALT_BF16 || a->ntaps != 9 || (a->q_step != 1 && a->q_step != 2)) return 0;
    static const bool no_s2 = getenv("SALT_WGRAD_LS_S2") && atoi(getenv("SALT_WGRAD_LS_S2")) == 0;       // A/B: stride-2 layers on the previous kernels
    if (a->q_step == 2 && (no_s2 || a->pad_mode != 0)) return 0;
    for (int t = 0; t < 9; ++t)
        if (a->tap_dy[t] != a->tap_dy[0] + t / 3 || a->tap_dx[t] != a->tap_dx[0] + t % 3) return 0;     // raster 3 x 3 window
    if (a->p.cs % 8 || a->q.cs % 8 || a->p.C % 8 || a->q.C % 8 || a->p.B != a->q.B) return 0;
    if ((reinterpret_cast<uintptr_t>(a->p.p) | reinterpret_cast<uintptr_t>(a->q.p)) & 15) return 0;
    if ((long long)a->q.B * a->q.H * a->q.W * (a->q_plane ? a->q.C : a->q.cs) * 2 >= (1ll << 31) || (long long)a->p.B * a->p.H * a->p.W * a->p.cs * 2 >= (1ll << 31)) return 0;
    // (salt_conv_wgrad_nsplit reported THIS kernel's split count for the shape: a misaligned slab pointer must not silently fall through
    //  to the generic kernel, whose split count differs - ADVICE r4)
    if (launch && (reinterpret_cast<uintptr_t>(a->partials) & 15)) {
        salt_set_error("conv_wgrad: the partials workspace must be 16-byte aligned for this shape (conv_wgrad_ls_kernel)");
        if (rc) *rc = SALT_E_BADARG;
        return 1;
    }
    const char* ku_env = getenv("SALT_WL_KU");                            // read per call: the tests switch it inside one process
    const int KU = a->q_step == 2 ? 2 : ((ku_env && atoi(ku_env) == 8) ? 8 : 4);
    const int NB = a->p.W <= 8 ? 2 : 1, TW = 16 / NB;
    WlKP k;
    k.P = reinterpret_cast<const bf16_t*>(a->p.p); k.Q = reinterpret_cast<const bf16_t*>(a->q.p); k.partials = a->partials;
    k.B = a->p.B; k.PH = a->p.H; k.PW = a->p.W; k.Ca = a->p.C; k.p_cs = a->p.cs;
    k.QH = a->q.H; k.QW = a->q.W; k.Cb = a->q.C; k.q_cs = a->q.cs;
    k.min_dy = a->tap_dy[0]; k.min_dx = a->tap_dx[0];
    k.q_plane = a->q_plane;
    k.ncol = cdiv(k.PW, TW); k.U = cdiv(k.PH, KU);
    k.total = cdiv(k.B, NB) * k.ncol * k.U;
    k.a_blocks = cdiv(k.Ca, 64); k.b_blocks = cdiv(k.Cb, 64);
    // split rule of wgrad_plan: a workgroup budget per launch and a minimum of pixels per split (every split costs a slab round trip)
    // round 6: 128 workgroups per launch instead of 512 (wgrad_plan in conv_mfma.hip has the measurements: backward is bound by the
    // kernel time of both queues together, a weight-gradient launch on every CU slows the data-gradient chain beside it)
    static const int target_wgs = getenv("SALT_WGRAD_WGS") ? atoi(getenv("SALT_WGRAD_WGS")) : 128;
    static const int wgs_big = getenv("SALT_WGRAD_WGS_BIG") ? atoi(getenv("SALT_WGRAD_WGS_BIG")) : 0;      // A/B: the 128 x 128 maps
    static const int tpw = getenv("SALT_WGRAD_TPW") ? atoi(getenv("SALT_WGRAD_TPW")) : 8;          // in 128-pixel tiles, as before
    const int blocks = k.a_blocks * k.b_blocks;
    int upw_min = tpw * 8 / KU; if (upw_min < 1) upw_min = 1;
    int ns = ((wgs_big > 0 && (long long)k.total * KU >= 16384) ? wgs_big : target_wgs) / blocks;
    if (ns > k.total / upw_min) ns = k.total / upw_min;
    if (ns < 1) ns = 1;
    k.upw = cdiv(k.total, ns);
    ns = cdiv(k.total, k.upw);
    if (!launch) return ns;
    // any split count the caller sized `partials` for works (a split beyond the last unit writes a zero slab): the engine passes the
    // planned one, the tests walk mid-strip split starts with others
    if (a->nsplit < 1) { salt_set_error("wgrad: nsplit %d", a->nsplit); *rc = SALT_E_BADARG; return ns; }
    ns = a->nsplit;
    k.upw = cdiv(k.total, ns);
    k.nsplit = ns;
    k.V = ns * blocks; k.per_xcd = cdiv(k.V, 8);
    const bool pad = a->pad_mode != 0;
#define SALT_WL(NB_, KU_) (pad ? wl_launch_inst<true, NB_, KU_>(k, st) : wl_launch_inst<false, NB_, KU_>(k, st))
    if (a->q_step == 2) *rc = NB == 1 ? wl_launch_inst<false, 1, 2, 2>(k, st) : wl_launch_inst<false, 2, 2, 2>(k, st);
    else if (NB == 1) *rc = KU == 4 ? SALT_WL(1, 4) : SALT_WL(1, 8);
    else *rc = KU == 4 ? SALT_WL(2, 4) : SALT_WL(2, 8);
#undef SALT_WL
    return ns;
}
