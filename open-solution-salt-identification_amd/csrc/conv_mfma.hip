// conv_mfma.hip — im2col-free direct convolution on the CDNA4 matrix cores (gfx950).
//
// One kernel family, parameterised by a tap list, serves every dense convolution of the path:
//   nn.Conv2d 3x3 / 1x1, stride 1|2, zero pad           (unet_models.py:24, torchvision ResNet blocks)
//   replicate-pad top/right 3x3 (and (k,1)/(1,k))       (architectures/base.py:21-27)
//   one output-parity phase of ConvTranspose2d k3/k4 s2 (unet_models.py:44,60; base.py:48-49)
//   the data-gradients of all of the above (transposed packed weights, mirrored taps)
//   the weight-gradients (conv_wgrad_kernel below).
//
// Data movement (forward/dgrad): per workgroup one output tile of BM pixels x BN channels.  Per 64-byte
// channel chunk the NHWC input HALO tile (all pixels any tap of the tile touches) and the chunk's
// weights for all taps are staged ONCE in LDS as 64-byte rows (XOR-swizzled 16-byte slots, conflict-free
// for ds_read_b128 A/B fragment reads); the 9 taps then re-read the same LDS pixels at shifted
// addresses, so HBM/L2 sees each input element once per tile instead of 9 times and no im2col
// matrix ever exists.  The inner contraction over (tap, channel) runs on MFMA:
//   bf16: v_mfma_f32_32x32x16_bf16 (one per 16-byte k-step),  f32: 4 x v_mfma_f32_32x32x2_f32
//   (exact f32, bitwise an fmaf chain) with the k-slots permuted so both operands are 16-byte LDS reads.
// Epilogue: bias, optional folded-BN affine + ReLU (eval), optional accumulate, and deterministic
// per-wave BatchNorm partial statistics (sum, M2) for train mode.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#ifndef SALT_K1_DBG
#define SALT_K1_DBG 0
#endif
#ifndef SALT_K1_FAST_STORE
#define SALT_K1_FAST_STORE 1
#endif

namespace {

struct ConvKP {
    const void* x; const void* w; void* y;
    const float* bias; const float* scale; const float* shift;
    float* stats; float* stats_cnt;
    int B, H, W, Cin, x_cs;
    int Cout, y_cs, OHf, OWf, OH, OW, out_step, out_oy, out_ox;
    int ntaps; int tap_off[SALT_MAX_TAPS];
    int in_step, pad_mode, min_dy, min_dx;
    int th_log2, tw_log2, nb;
    int gen, tw, th, twth;           // tile = nb images x th rows x tw columns; gen: not powers of two (nb * th * tw <= BM, see tile_pix)
    unsigned tw_magic, twth_magic;
    int hh, hw;
    int tiles_y, tiles_x;
    int nchunk, a_bytes;
    int relu, accumulate, stats_part0;
    void* strip; int strip_cs, fold_top, fold_bottom, fold_left, fold_right;     // fold mode (strip != nullptr)
    int nphase, m_tiles_ph; int64_t w_phase_elems;   // phase-fused stride-2 launch: 4 output-parity phases, one packed weight block each
    int fold_fused, ox_shift;        // fused fold (strip == nullptr, fold_top / fold_right > 0): tile columns start at -ox_shift
    unsigned hhw_magic, hw_magic;                                                // x / d == umulhi(x, 2^32 / d + 1) for x * d < 2^32
    int m_tiles, n_tiles;
    int vt;                          // virtual taps of a 1x1 convolution: vt channel chunks staged per barrier round (1 = off)
    int nbuf;                        // conv_glds_kernel: depth of the LDS chunk ring (2 | 3)
    int no_perm8;                    // conv_glds_kernel: A/B - natural lane -> pixel order on the 8 x 8 tiles (SALT_GLDS_NO_PERM8)
    int y_small;                     // y / bnb_y / bnb_a hold < 2^31 elements each: the epilogue may use 32-bit offsets
    // BatchNorm-backward sums of the stored tile (saltnet.h, salt_conv_args.bnb_*); bnb_partials == nullptr: off
    const void* bnb_y; const void* bnb_a; int bnb_cs, bnb_acs, bnb_relu;
    const float* bnb_mean; const float* bnb_invstd; const float* bnb_gamma; const float* bnb_beta; float* bnb_partials;
    const void* res; int res_cs;     // residual epilogue (salt_conv_args.res): y = relu?(stored + res); nullptr: off
    BnFin fin; BnbFin bnbf;          // in-launch BatchNorm finalize (saltnet.h, salt_conv_args.fin / bnb_fin; common.h); acc == nullptr: off
    // input transform (salt_conv_args.in_*): x' = relu?(x * in_scale[c] + in_shift[c]) applied by the loader between the global load and
    // the LDS store; in_fin.acc != nullptr: the coefficients are finalized from the producer's statistics shards in the prologue
    const float* in_scale; const float* in_shift; int in_relu, in_tab_off; BnFin in_fin;
};

template <typename T> struct Mma;
template <> struct Mma<float> {
    static __device__ __forceinline__ void step(const u32x4& a, const u32x4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
};
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void step(const u32x4& a, const u32x4& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};

// 16-byte piece load from global with channel-tail / alignment handling.
template <typename T>
__device__ __forceinline__ u32x4 load_piece(const T* base, int64_t off, int ch0, int C, bool vec_ok) {
    constexpr int VE = Elem<T>::VE;
    if (vec_ok && ch0 + VE <= C) return *reinterpret_cast<const u32x4*>(base + off);
    float f[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) f[j] = (ch0 + j < C) ? Elem<T>::ld(base + off + j) : 0.f;
    if (sizeof(T) == 4) return pack16<T>(f);
    // bf16: repack exactly (values came from bf16, conversion is lossless)
    return pack16<T>(f);
}

__device__ __forceinline__ int swz_addr(int row, int slot) {        // 64-byte rows, 4 x 16-byte slots
    return row * 64 + (((slot ^ (row >> 2)) & 3) << 4);
}

// Row m of a pixel tile -> (column, row, image) inside the tile.  Power-of-two tiles: shifts.  General tiles (gen; the fused-fold data
// gradients of the small decoder maps: full-width strips of the 10 / 18 / 34-wide extended grid instead of 16-wide tiles that are
// 40 - 60 % empty): two multiply-high divisions; rows past nb * th * tw belong to no pixel (bl = 2^20: fails every b0 + bl < B test).
struct TilePix { int tx, ty, bl; };
__device__ __forceinline__ TilePix tile_pix(const ConvKP& p, int m) {
    TilePix t;
    if (p.gen) {
        const int bl = (int)__umulhi((unsigned)m, p.twth_magic);
        const int r = m - bl * p.twth;
        t.ty = (int)__umulhi((unsigned)r, p.tw_magic);
        t.tx = r - t.ty * p.tw;
        t.bl = bl < p.nb ? bl : (1 << 20);
    } else {
        t.tx = m & ((1 << p.tw_log2) - 1);
        t.ty = (m >> p.tw_log2) & ((1 << p.th_log2) - 1);
        t.bl = m >> (p.tw_log2 + p.th_log2);
    }
    return t;
}

// 16 zero bytes in device memory: a masked-out piece of a branch-free loader reads them instead of zero-initialising its
// destination registers under a branch (see conv_wgrad_fast_kernel)
__device__ __attribute__((aligned(16))) unsigned int g_zero_piece[4] = {0u, 0u, 0u, 0u};

constexpr int MAXA = 9;     // halo pieces per thread: P_halo*4 <= 9*256 (6*256 for the 256-pixel tile: keeps its prefetch registers in budget)
constexpr int maxa_for(int bm) { return bm >= 256 ? 6 : MAXA; }

// Epilogue of one output tile of conv_mfma_kernel.  C layout (32x32): col n = lane&31,
// row m = (r&3) + 8*(r>>2) + 4*(lane>>5).  The accumulator tile goes through LDS (free after the main loop) so that the global stores
// are whole 16-byte pieces of NHWC pixel rows (a wave writes full 128-B lines) instead of 2/4-byte scattered stores.
struct TileCoord { int m_tile, n_tile, oy0, ox0, b0, n0, out_oy, out_ox; };

template <typename T, int MI, int NI, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const ConvKP& p, const TileCoord& tc, f32x16 (&acc)[MI][NI], unsigned char* smem) {
    constexpr int BN = 32 * NI * WN;
    constexpr int VE = Elem<T>::VE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int khalf = lane >> 5, l31 = lane & 31;
    const int m_tile = tc.m_tile, n_tile = tc.n_tile, oy0 = tc.oy0, ox0 = tc.ox0, b0 = tc.b0, n0 = tc.n0, out_oy = tc.out_oy, out_ox = tc.out_ox;
    int nrow[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) nrow[j] = (wn * NI + j) * 32 + l31;
    constexpr int BM = 32 * MI * WM;
    constexpr int PITCH = BN + 16 / (int)sizeof(T);               // elements per LDS row (16-byte pad: no bank aliasing of rows)
    constexpr int PPO = BN / VE;                                   // 16-byte pieces per output pixel row
    T* sO = reinterpret_cast<T*>(smem);
    float ssum[NI], cntf = 0.f;
#pragma unroll
    for (int j = 0; j < NI; ++j) ssum[j] = 0.f;
    unsigned vmask[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) vmask[i] = 0xffffu;
    // everything below is specialised on three workgroup-uniform facts so that the common launches do not pay for the rare ones:
    // statistics wanted (train-mode forward only), an affine / ReLU epilogue present (eval only), tile fully inside the output grid
    const bool want_stats = p.stats != nullptr || p.fin.acc != nullptr;
    const bool has_affine = p.bias || p.scale || p.shift || p.relu;
    const bool full_tile = !p.gen && (b0 + p.nb <= p.B) && (oy0 + p.th <= p.OH) && ox0 >= 0 && (ox0 + p.tw <= p.OW);
    if (want_stats) {
        if (full_tile) {
            cntf = 16.f * MI;
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                vmask[i] = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    const TilePix tp = tile_pix(p, m);
                    const int tx = tp.tx, ty = tp.ty, bl = tp.bl;
                    const bool valid = (b0 + bl < p.B) && (oy0 + ty < p.OH) && (ox0 + tx < p.OW) && (ox0 + tx >= 0);
                    if (valid) { vmask[i] |= 1u << r; cntf += 1.f; }
                }
            }
        }
    }
    __syncthreads();                                               // every wave is done reading sA / sB
    auto stage = [&](auto affine_c, auto stats_c) {
        constexpr bool AFF = decltype(affine_c)::value, ST = decltype(stats_c)::value;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            float bias = 0.f, sc = 1.f, sh = 0.f;
            if constexpr (AFF) {
                const int n = n0 + nrow[j];
                const bool nok = n < p.Cout;
                bias = (nok && p.bias) ? p.bias[n] : 0.f;
                sc = (nok && p.scale) ? p.scale[n] : 1.f;
                sh = (nok && p.shift) ? p.shift[n] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r];
                    if constexpr (AFF) {
                        v = (v + bias) * sc + sh;
                        if (p.relu && !p.res) v = fmaxf(v, 0.f);           // with a residual the ReLU follows the add (store loop)
                        acc[i][j][r] = v;
                    }
                    if constexpr (ST) { if ((vmask[i] >> r) & 1u) ssum[j] += v; }
                    const int ml = (wm * MI + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    Elem<T>::st(sO + ml * PITCH + nrow[j], v);
                }
            }
        }
    };
    if (has_affine) { if (want_stats) stage(std::true_type{}, std::true_type{}); else stage(std::true_type{}, std::false_type{}); }
    else { if (want_stats) stage(std::false_type{}, std::true_type{}); else stage(std::false_type{}, std::false_type{}); }
    // BatchNorm statistics of this workgroup: per-wave (sum, M2 about the wave's mean, count) from the registers, then the WM wave
    // rows are merged in fixed order (Chan) through LDS - in a region BEHIND the staged output tile, so that the merge shares the
    // barrier of the tile and (in-launch finalize) the atomics are in flight while the tile is stored.
    constexpr int STATS_OFF = (BM * PITCH * (int)sizeof(T) + 15) & ~15;
    float* sS = reinterpret_cast<float*>(smem + STATS_OFF);            // [WM][BN][2] then [WM] counts
    float* sC = sS + WM * BN * 2;
    if (want_stats) {
        cntf += __shfl_xor(cntf, 32);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            float s = ssum[j] + __shfl_xor(ssum[j], 32);
            const float mean = cntf > 0.f ? s / cntf : 0.f;
            float m2 = 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((vmask[i] >> r) & 1u) { const float d = acc[i][j][r] - mean; m2 += d * d; }
            m2 += __shfl_xor(m2, 32);
            if (khalf == 0) { sS[(wm * BN + nrow[j]) * 2 + 0] = s; sS[(wm * BN + nrow[j]) * 2 + 1] = m2; }
        }
        if (lane == 0 && wn == 0) sC[wm] = cntf;
    }
    __syncthreads();
    if (want_stats && tid < BN) {
        const int part = p.stats_part0 + m_tile;
        float N = 0.f, S = 0.f, M2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) {
            const float nk = sC[w];
            if (nk > 0.f) {
                const float sk = sS[(w * BN + tid) * 2 + 0], mk = sS[(w * BN + tid) * 2 + 1];
                if (N == 0.f) { N = nk; S = sk; M2 = mk; }
                else {
                    const float d = sk / nk - S / N;
                    M2 += mk + d * d * (N * nk / (N + nk));
                    S += sk; N += nk;
                }
            }
        }
        const int n = n0 + tid;
        if (p.fin.acc) {
            double* a = p.fin.acc + (blockIdx.x & 7) * (2 * p.Cout + 1);
            if (n < p.Cout && N > 0.f) {
                fin_add(a + n, (double)S);
                fin_add(a + p.Cout + n, (double)M2 + (double)S * (double)S / (double)N);
            }
            if (tid == 0 && n_tile == 0) fin_add(a + 2 * p.Cout, (double)N);
        } else {
            if (n < p.Cout) {
                p.stats[((int64_t)part * 2 + 0) * p.Cout + n] = S;
                p.stats[((int64_t)part * 2 + 1) * p.Cout + n] = M2;
            }
            if (tid == 0 && n_tile == 0) p.stats_cnt[part] = N;
        }
    }
    {
        T* yg = reinterpret_cast<T*>(p.y);
        const bool y_vec = ((p.y_cs % VE) == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0);
        // BatchNorm-backward sums of the stored values: a thread's channel piece (tid % PPO) is the same for all of its pixels
        const bool bnb = p.bnb_partials != nullptr || p.bnbf.acc != nullptr;
        float b1[VE], b2[VE], bmu[VE], bis[VE], bsc[VE], bsh[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) { b1[e] = 0.f; b2[e] = 0.f; bmu[e] = 0.f; bis[e] = 0.f; bsc[e] = 0.f; bsh[e] = 0.f; }
        if (bnb) {
            const int nb0 = n0 + (tid % PPO) * VE;
#pragma unroll
            for (int e = 0; e < VE; ++e)
                if (nb0 + e < p.Cout) {
                    bmu[e] = p.bnb_mean[nb0 + e]; bis[e] = p.bnb_invstd[nb0 + e];
                    bsc[e] = p.bnb_gamma[nb0 + e] * bis[e]; bsh[e] = p.bnb_beta[nb0 + e] - bmu[e] * bsc[e];
                }
        }
        // Fast path (whole tile inside the grid, whole aligned channel pieces, no fold): 32-bit offsets, no per-piece bounds / mode
        // tests, fully unrolled so that the LDS reads and (accumulate / BatchNorm-backward) loads of all pieces are in flight together.
        // The general loop below spent ~70 instructions per 16-byte piece: the epilogue of a 9.66-GFLOP layer took 4.5 us of 19 us
        // WITHOUT its stores (round-2 variant builds: tools/build_variant.sh -DSALT_K1_DBG=n).
        const bool fast_store = SALT_K1_FAST_STORE && full_tile && !p.fold_fused && !p.strip && y_vec && (n0 + BN <= p.Cout) && p.y_small;
        if (fast_store) {
            const int pc = tid % PPO, n = n0 + pc * VE;
            const bool accum = p.accumulate != 0;
            const T* yb = reinterpret_cast<const T*>(p.bnb_y);
            const T* ab = reinterpret_cast<const T*>(p.bnb_a);
#pragma unroll
            for (int it = 0; it < BM * PPO / 256; ++it) {
                const int m = it * (256 / PPO) + tid / PPO;
                const TilePix tp = tile_pix(p, m);
                const int tx = tp.tx, ty = tp.ty, bl = tp.bl;
                const unsigned pix = ((unsigned)(b0 + bl) * p.OHf + (oy0 + ty) * p.out_step + out_oy) * p.OWf + (ox0 + tx) * p.out_step + out_ox;
                T* dst = yg + (pix * (unsigned)p.y_cs + n);
                u32x4 stored = *reinterpret_cast<const u32x4*>(sO + m * PITCH + pc * VE);
                if (accum) {
                    float f[VE], o[VE];
                    unpack16<T>(stored, f);
                    unpack16<T>(*reinterpret_cast<const u32x4*>(dst), o);
#pragma unroll
                    for (int e = 0; e < VE; ++e) f[e] += o[e];
                    stored = pack16<T>(f);
                }
                if (p.res) {                                       // host: out_step 1, full grid - pix is the residual's pixel index too
                    float f[VE], o[VE];
                    unpack16<T>(stored, f);
                    unpack16<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.res) + (pix * (unsigned)p.res_cs + n)), o);
#pragma unroll
                    for (int e = 0; e < VE; ++e) { f[e] += o[e]; if (p.relu) f[e] = fmaxf(f[e], 0.f); }
                    stored = pack16<T>(f);
                }
                *reinterpret_cast<u32x4*>(dst) = stored;
                if (bnb) {                                         // host: out_step 1 (pix is the pixel index of bnb_y / bnb_a too)
                    float g[VE], yc[VE];
                    unpack16<T>(stored, g);
                    unpack16<T>(*reinterpret_cast<const u32x4*>(yb + (pix * (unsigned)p.bnb_cs + n)), yc);
                    if (p.bnb_a) {
                        float av[VE];
                        unpack16<T>(*reinterpret_cast<const u32x4*>(ab + (pix * (unsigned)p.bnb_acs + n)), av);
#pragma unroll
                        for (int e = 0; e < VE; ++e) {
                            const float gg = (!p.bnb_relu || av[e] > 0.f) ? g[e] : 0.f;
                            b1[e] += gg; b2[e] += gg * (yc[e] - bmu[e]) * bis[e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < VE; ++e) {
                            const float gg = (!p.bnb_relu || yc[e] * bsc[e] + bsh[e] > 0.f) ? g[e] : 0.f;
                            b1[e] += gg; b2[e] += gg * (yc[e] - bmu[e]) * bis[e];
                        }
                    }
                }
            }
        } else
        for (int q = tid; q < BM * PPO; q += 256) {
            const int m = q / PPO, pc = q - m * PPO;
            const TilePix tp = tile_pix(p, m);
            const int tx = tp.tx, ty = tp.ty, bl = tp.bl;
            const int oy = oy0 + ty, ox = ox0 + tx, b = b0 + bl;
            const int n = n0 + pc * VE;
            if (b >= p.B || oy >= p.OH || ox >= p.OW || ox < 0 || n >= p.Cout) continue;
            T* dst = yg + (((int64_t)b * p.OHf + oy * p.out_step + out_oy) * p.OWf + ox * p.out_step + out_ox) * p.y_cs + n;
            bool accum = p.accumulate != 0;
            bool dvec = y_vec;
            int fold_rows = 0, fold_cols = 0;                      // fused fold: ring pixels above / right of this edge pixel
            if (p.fold_fused) {
                // replicate-pad adjoint inside the tile: the pad ring (top rows, right columns of the extended grid) is never stored;
                // the edge pixel it folds onto sums its ring pixels from the staged tile (the tile grid is laid out so that they
                // share a tile: rows start at 0 with th > fold_top, columns at -ox_shift)
                const int iy = oy - p.fold_top;
                if (iy < 0 || ox >= p.OWf) continue;
                dst = yg + (((int64_t)b * p.OHf + iy) * p.OWf + ox) * p.y_cs + n;
                fold_rows = iy == 0 ? p.fold_top : 0;
                fold_cols = ox == p.OWf - 1 ? p.fold_right : 0;
            }
            if (p.strip) {                                         // fold mode: interior -> y (unpadded), pad ring -> strip
                const int iy = oy - p.fold_top, ix = ox - p.fold_left;
                if (iy >= 0 && iy < p.OHf && ix >= 0 && ix < p.OWf) {
                    dst = yg + (((int64_t)b * p.OHf + iy) * p.OWf + ix) * p.y_cs + n;
                } else {
                    const int64_t ring = (int64_t)(p.fold_top + p.fold_bottom) * p.OW + (int64_t)p.OHf * (p.fold_left + p.fold_right);
                    dst = reinterpret_cast<T*>(p.strip) + ((int64_t)b * ring + fold_ring_index(oy, ox, p.OHf, p.OWf, p.fold_top, p.fold_bottom, p.fold_left, p.fold_right)) * p.strip_cs + n;
                    accum = false;
                    dvec = (p.strip_cs % VE) == 0;
                }
            }
            u32x4 v = *reinterpret_cast<const u32x4*>(sO + m * PITCH + pc * VE);
            if (fold_rows | fold_cols) {                           // host: fused fold implies whole aligned channel pieces
                float f[VE], o[VE];
                unpack16<T>(v, f);
                for (int ky = 0; ky <= fold_rows; ++ky)
                    for (int kx = 0; kx <= fold_cols; ++kx) {
                        if ((ky | kx) == 0) continue;
                        unpack16<T>(*reinterpret_cast<const u32x4*>(sO + (m - ky * p.tw + kx) * PITCH + pc * VE), o);
#pragma unroll
                        for (int e = 0; e < VE; ++e) f[e] += o[e];
                    }
                v = pack16<T>(f);
            }
            if (dvec && n + VE <= p.Cout) {
                u32x4 stored = v;
                if (accum) {
                    float f[VE], o[VE];
                    unpack16<T>(v, f);
                    unpack16<T>(*reinterpret_cast<const u32x4*>(dst), o);
#pragma unroll
                    for (int e = 0; e < VE; ++e) f[e] += o[e];
                    stored = pack16<T>(f);
                }
                if (p.res) {                                       // host: res implies whole aligned pieces, out_step 1, full grid, no fold
                    float f[VE], o[VE];
                    unpack16<T>(stored, f);
                    unpack16<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.res) + (((int64_t)b * p.OHf + oy) * p.OWf + ox) * p.res_cs + n), o);
#pragma unroll
                    for (int e = 0; e < VE; ++e) { f[e] += o[e]; if (p.relu) f[e] = fmaxf(f[e], 0.f); }
                    stored = pack16<T>(f);
                }
#if defined(SALT_K1_DBG) && (SALT_K1_DBG & 8)
                if (stored.x == 0x12345678u && stored.y == 0x9abcdef0u) *reinterpret_cast<u32x4*>(dst) = stored;   // ablation: staging without the stores
#else
                *reinterpret_cast<u32x4*>(dst) = stored;
#endif
                if (bnb) {                                         // host: bnb implies whole aligned pieces, out_step 1, no strip
                    float g[VE], yc[VE];
                    unpack16<T>(stored, g);
                    unpack16<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.bnb_y) + (((int64_t)b * p.OHf + oy * p.out_step + out_oy - (p.fold_fused ? p.fold_top : 0)) * p.OWf + ox * p.out_step + out_ox) * p.bnb_cs + n), yc);
                    if (p.bnb_a) {                                 // residual layer: the mask is the sign of the forward output
                        float av[VE];
                        unpack16<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.bnb_a) + (((int64_t)b * p.OHf + oy * p.out_step + out_oy - (p.fold_fused ? p.fold_top : 0)) * p.OWf + ox * p.out_step + out_ox) * p.bnb_acs + n), av);
#pragma unroll
                        for (int e = 0; e < VE; ++e) {
                            const float gg = (!p.bnb_relu || av[e] > 0.f) ? g[e] : 0.f;
                            b1[e] += gg; b2[e] += gg * (yc[e] - bmu[e]) * bis[e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < VE; ++e) {
                            const float gg = (!p.bnb_relu || yc[e] * bsc[e] + bsh[e] > 0.f) ? g[e] : 0.f;
                            b1[e] += gg; b2[e] += gg * (yc[e] - bmu[e]) * bis[e];
                        }
                    }
                }
            } else {
                float f[VE];
                unpack16<T>(v, f);
#pragma unroll
                for (int e = 0; e < VE; ++e)
                    if (n + e < p.Cout) Elem<T>::st(dst + e, accum ? f[e] + Elem<T>::ld(dst + e) : f[e]);
            }
        }
        if (bnb) {
            // cross-row sums: 256/PPO rows of BN channels x 2 statistics through LDS, rows added in ascending order
            float* sR = reinterpret_cast<float*>(smem);
            const int row = tid / PPO, cl0 = (tid % PPO) * VE;
            __syncthreads();                                       // the output tile in LDS has been stored
#pragma unroll
            for (int e = 0; e < VE; ++e) { sR[(row * BN + cl0 + e) * 2] = b1[e]; sR[(row * BN + cl0 + e) * 2 + 1] = b2[e]; }
            __syncthreads();
            for (int e = tid; e < 2 * BN; e += 256) {
                const int st = e >= BN ? 1 : 0, cl = e - st * BN;
                float t = 0.f;
                for (int r = 0; r < 256 / PPO; ++r) t += sR[(r * BN + cl) * 2 + st];
                if (n0 + cl < p.Cout) {
                    if (p.bnbf.acc) fin_add(p.bnbf.acc + ((blockIdx.x & 7) * 2 + st) * p.Cout + n0 + cl, (double)t);
                    else p.bnb_partials[((int64_t)m_tile * 2 + st) * p.Cout + n0 + cl] = t;
                }
            }
            if (p.bnbf.acc && p.bnbf.ticket) {
                unsigned* flag = reinterpret_cast<unsigned*>(sR + (256 / PPO) * BN * 2);
                if (fin_arrive(p.bnbf.ticket, (unsigned)(p.m_tiles * p.n_tiles), flag)) fin_backward(p.bnbf, p.bnb_gamma, p.bnb_invstd, p.Cout);
            }
        }
    }
    if (p.fin.acc && p.fin.ticket) {                               // in-launch finalize: the atomics were issued before the tile stores
        unsigned* flag = reinterpret_cast<unsigned*>(smem + STATS_OFF) + WM * BN * 2 + WM + 2;
        if (fin_arrive(p.fin.ticket, (unsigned)(p.m_tiles * p.n_tiles), flag)) fin_forward(p.fin, p.Cout, nullptr);
    }
}

// NT > 0: tap count known at compile time (9 for every 3x3): the tap loop is fully unrolled so the compiler keeps the tap
// offsets in SGPRs, folds the weight-row offsets into ds_read immediates and hoists the next taps' fragment reads above the
// current MFMAs (the runtime-loop form serialised s_load -> address VALU -> ds_read -> MFMA per tap).
// MAXA_T > 0 selects the INTERLEAVED loader (host: aligned input view, Cin a multiple of the 16-byte piece, halo <= MAXA_T pieces
// per thread): the global loads of chunk c+1 are not issued in one burst after the barrier - where the 8-12 waves of a CU queue up
// behind its 64 B/clk address path and nobody feeds the matrix cores (in-kernel clocks on 256->256 @32x32: 0.98 us of load issue
// per 0.94 us of MFMAs) - but one piece per (tap, k-step) stage between the MFMAs, branch-free (masked pieces read g_zero_piece).
template <typename T, int MI, int NI, int WM, int WN, int NT, int MAXA_T = 0, bool INA = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvKP p) {
    constexpr int BN = 32 * NI * WN;
    constexpr int KCE = 64 / (int)sizeof(T);
    constexpr int VE = Elem<T>::VE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;
    unsigned char* sB = smem + p.a_bytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int khalf = lane >> 5, l31 = lane & 31;

    // 1-D grid, XCD-aware order: consecutive block ids go round-robin to the 8 XCDs, so within one XCD (id % 8) the channel tiles of
    // ONE pixel tile are adjacent in dispatch order and share its halo rows through that XCD's L2 (re-read n_tiles times otherwise).
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int n_tile = local % p.n_tiles;
    const int m_tile = (local / p.n_tiles) * 8 + xcd;
    if (m_tile >= p.m_tiles) return;
    int tile = m_tile, out_oy = p.out_oy, out_ox = p.out_ox;
    int64_t w_off = 0;
    if (p.nphase > 1) {                       // phase-fused stride-2 launch: the tile's output parity selects offset and weight block
        const int ph = m_tile / p.m_tiles_ph;
        tile = m_tile - ph * p.m_tiles_ph;
        out_oy = ph >> 1; out_ox = ph & 1;
        w_off = ph * p.w_phase_elems;
    }
    const int txi = tile % p.tiles_x; tile /= p.tiles_x;
    const int tyi = tile % p.tiles_y;
    const int tbi = tile / p.tiles_y;
    const int oy0 = tyi * p.th, ox0 = txi * p.tw - p.ox_shift, b0 = tbi * p.nb;
    const int n0 = n_tile * BN;
    const int hhw = p.hh * p.hw;
    const int phalo = p.nb * hhw;
    const int iy0 = oy0 * p.in_step + p.min_dy, ix0 = ox0 * p.in_step + p.min_dx;
    const T* xg = reinterpret_cast<const T*>(p.x);
    const T* wg = reinterpret_cast<const T*>(p.w) + w_off;
    const bool x_vec = ((p.x_cs % VE) == 0) && ((reinterpret_cast<uintptr_t>(p.x) & 15) == 0);
    // ---- input transform table [2][Cin] behind the staging buffers (INA: bf16, whole 16-byte pieces - checked by the host)
    float* in_tab = reinterpret_cast<float*>(smem + p.in_tab_off);
    if constexpr (INA) {
        if (p.in_fin.acc) fin_forward_consumer(p.in_fin, p.Cin, in_tab, in_tab + p.Cin, blockIdx.x == 0);
        else for (int c = tid; c < p.Cin; c += 256) { in_tab[c] = p.in_scale[c]; in_tab[p.Cin + c] = p.in_shift[c]; }
    }

    // ---- per-thread halo staging pieces (fixed for the whole chunk loop); offsets are relative to image b0 (fit 32 bits)
    constexpr bool ILV = MAXA_T > 0;
    static_assert(!ILV || NT > 0, "the interleaved loader rides on the unrolled tap pipeline");
    constexpr int MAXA = ILV ? MAXA_T : maxa_for(32 * MI * WM);
    const T* xg0 = xg + (int64_t)b0 * p.H * p.W * p.x_cs;
    int a_goff[MAXA];
    // A 1x1 convolution stages vt consecutive channel chunks per round as "virtual taps": LDS row = vtap * phalo + pixel, the packed
    // weights [chunk][1][Cout][KC] are the same bytes as [chunk / vt][vt][Cout][KC], and the pipelined NT = vt tap loop applies.
    const int npa = (phalo * p.vt * 4 + 255) >> 8;
#pragma unroll
    for (int k = 0; k < MAXA; ++k) {
        a_goff[k] = -2;
        if (k < npa) {
            const int q = tid + (k << 8);
            const int prow = q >> 2;
            if (prow < phalo * p.vt) {
                int vtap = 0, pix = prow;
                if (p.vt > 1) { vtap = prow / phalo; pix = prow - vtap * phalo; }
                const int bl = (int)__umulhi((unsigned)pix, p.hhw_magic);
                const int r = pix - bl * hhw;
                const int hy = (int)__umulhi((unsigned)r, p.hw_magic);
                const int hx = r - hy * p.hw;
                int iy = iy0 + hy, ix = ix0 + hx;
                bool valid = b0 + bl < p.B;
                if (p.pad_mode) {
                    iy = min(max(iy, 0), p.H - 1);
                    ix = min(max(ix, 0), p.W - 1);
                } else {
                    valid = valid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                }
                a_goff[k] = valid ? ((bl * p.H + iy) * p.W + ix) * p.x_cs + vtap * KCE + (q & 3) * VE : -1;
            }
        }
    }
    // ---- per-lane A fragment rows (halo pixel index of the lane's output pixel, per M sub-tile)
    int pbase[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = (wm * MI + i) * 32 + l31;
        const TilePix tp = tile_pix(p, m);
        const int tx = tp.tx, ty = tp.ty, bl = tp.bl;
        pbase[i] = bl < p.nb ? bl * hhw + ty * p.in_step * p.hw + tx * p.in_step : 0;      // (general tiles: a row past the last pixel reads halo row 0)
    }
    int nrow[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) nrow[j] = (wn * NI + j) * 32 + l31;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // compile-time ablations (round-2 variant builds: tools/build_variant.sh -DSALT_K1_DBG=n): 1 return after the prologue, 2 no chunk loop, 4 no epilogue
    if (SALT_K1_DBG & 1) {
        int t = 0;
#pragma unroll
        for (int k = 0; k < MAXA; ++k) t += a_goff[k];
#pragma unroll
        for (int i = 0; i < MI; ++i) t += pbase[i];
        if (t == 0x7fffffff) reinterpret_cast<int*>(p.y)[0] = t;
        return;
    }
    const int nbp = p.ntaps * BN * 4;          // weight pieces per chunk
    constexpr int MAXBP = (9 * BN * 4 + 255) / 256;          // weight pieces per thread that fit the register prefetch (<= 9 taps)

    // per-lane fragment base addresses.  A: row r = pbase + tap offset, byte = r*64 + (((slot ^ (r>>2)) & 3) << 4); the
    // second k-step (slot | 2) is the same address XOR 32.  B: rows t*BN + n with BN % 16 == 0, so the swizzle term does
    // not depend on the tap: byte = t*BN*64 + const.
    int b_addr[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) b_addr[j] = swz_addr(nrow[j], khalf);
    auto tap_mma = [&](int t, int toff) {
        u32x4 a0[MI], a1[MI], b0[NI], b1[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ad = swz_addr(pbase[i] + toff, khalf);
            a0[i] = *reinterpret_cast<const u32x4*>(sA + ad);
            a1[i] = *reinterpret_cast<const u32x4*>(sA + (ad ^ 32));
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            b0[j] = *reinterpret_cast<const u32x4*>(sB + t * (BN * 64) + b_addr[j]);
            b1[j] = *reinterpret_cast<const u32x4*>(sB + t * (BN * 64) + (b_addr[j] ^ 32));
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) Mma<T>::step(a0[i], b0[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) Mma<T>::step(a1[i], b1[j], acc[i][j]);
    };
    // NT > 0: fragment software pipeline over the 2 NT (tap, k-step) stages - the LDS reads of stage s+1 are issued before the
    // MFMAs of stage s, and sched_group_barrier pins that interleave (the default schedule sinks every read to its first use,
    // which exposes the full LDS latency to a wave that has only one partner on its SIMD).
    struct Frag { u32x4 a[MI], b[NI]; };
    constexpr int NTA = NT > 0 ? NT : 1;
    int a_addr[NTA][MI];                                          // LDS address of the lane's A fragment per (tap, M sub-tile): chunk invariant
    if constexpr (NT > 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < MI; ++i) a_addr[t][i] = swz_addr(pbase[i] + p.tap_off[t], khalf);
    }
    auto load_frag = [&](int s, Frag& f) {
        const int t = s >> 1, hx = (s & 1) << 5;
#pragma unroll
        for (int i = 0; i < MI; ++i) f.a[i] = *reinterpret_cast<const u32x4*>(sA + (a_addr[NT > 0 ? t : 0][i] ^ hx));
#pragma unroll
        for (int j = 0; j < NI; ++j) f.b[j] = *reinterpret_cast<const u32x4*>(sB + t * (BN * 64) + (b_addr[j] ^ hx));
    };
    auto mma_frag = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) Mma<T>::step(f.a[i], f.b[j], acc[i][j]);
    };
    // hook(s): extra work of stage s (the interleaved loader's global loads), nvm = how many VMEM instructions that is
    auto compute_chunk = [&](auto&& hook, auto nvm_c) {
        constexpr int NVM = decltype(nvm_c)::value;
        if constexpr (NT > 0) {
            Frag f0, f1;
            load_frag(0, f0);
            __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);         // stage 0 reads (prologue)
#pragma unroll
            for (int s = 0; s < 2 * NT; s += 2) {
                load_frag(s + 1, f1);
                hook(s);
                mma_frag(f0);
                __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);     // DS reads of stage s+1 ...
                if (NVM > 0) __builtin_amdgcn_sched_group_barrier(0x020, NVM, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);     // ... then the MFMAs of stage s
                if (s + 2 < 2 * NT) load_frag(s + 2, f0);
                hook(s + 1);
                mma_frag(f1);
                if (s + 2 < 2 * NT) __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
                if (NVM > 0) __builtin_amdgcn_sched_group_barrier(0x020, NVM, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
            }
        } else {
            for (int t = 0; t < p.ntaps; ++t) tap_mma(t, p.tap_off[t]);
        }
    };
    auto no_hook = [](int) {};
    const std::integral_constant<int, 0> no_vm{};

    if (NT > 0 || p.ntaps <= 9) {
        // Software pipeline: the global loads of chunk c+1 are issued right after the barrier that releases the MFMA
        // phase of chunk c and land in registers while the matrix cores work; they are written to LDS only after the
        // next barrier.  HBM/L2 latency is hidden behind compute instead of being paid once per chunk.
        u32x4 ra[MAXA], rb[MAXBP];
        // Piece q = tid + 256 k sits in LDS row (tid >> 2) + 64 k, slot tid & 3.  64 % 4 == 0, so the swizzle term is the same for every
        // k: LDS store address = lane constant + 4096 k.  The weight row of piece k is t * BN + n with (64 k + (tid >> 2)) = t * BN + n:
        // its global offset splits into a lane constant and a part that only depends on k (uniform) - no per-piece index math per chunk.
        const int lds_lane = swz_addr(tid >> 2, tid & 3);
        constexpr int RPP = 64;                                     // LDS rows per 256-thread pass
        const int lrow = tid >> 2;
        int b_lane_off, b_lane_n;                                   // element offset / output channel of the lane inside a pass
        if constexpr (BN >= RPP) { b_lane_n = lrow; b_lane_off = 0; }
        else { b_lane_n = lrow % BN; b_lane_off = (lrow / BN) * p.Cout * KCE; }
        b_lane_off += (n0 + b_lane_n) * KCE + (tid & 3) * VE;
        auto b_uniform = [&](int k) -> int {                        // element offset of pass k relative to the chunk's weights
            if constexpr (BN >= RPP) return ((k * RPP) / BN) * p.Cout * KCE + ((k * RPP) % BN) * KCE;
            else return (k * (RPP / BN)) * p.Cout * KCE;
        };
        auto b_ok = [&](int k) -> bool {
            if constexpr (BN > RPP) return n0 + b_lane_n + (k * RPP) % BN < p.Cout;
            else return n0 + b_lane_n < p.Cout;
        };
        auto load_chunk = [&](int c) {
            const int ch_base = c * KCE * p.vt;
#pragma unroll
            for (int k = 0; k < MAXA; ++k) {
                ra[k] = u32x4{0u, 0u, 0u, 0u};
                if (k < npa && a_goff[k] >= 0) {
                    const int ch0 = ch_base + (tid & 3) * VE;
                    if (ch0 < p.Cin) ra[k] = load_piece<T>(xg0, a_goff[k] + ch_base, ch0, p.Cin, x_vec);
                }
            }
            const T* wc = wg + (int64_t)c * p.ntaps * p.Cout * KCE;            // uniform
#pragma unroll
            for (int k = 0; k < MAXBP; ++k) {
                rb[k] = u32x4{0u, 0u, 0u, 0u};
                if (tid + (k << 8) < nbp && b_ok(k)) rb[k] = *reinterpret_cast<const u32x4*>(wc + b_uniform(k) + b_lane_off);
            }
        };
        // interleaved loader: piece i < MAXA is halo piece i, the rest are weight pieces; branch-free (masked: the zero piece)
        constexpr int NPC = MAXA + MAXBP, PPS = NT > 0 ? (NPC + 2 * NT - 1) / (2 * NT) : 1;
        const T* zp = reinterpret_cast<const T*>(g_zero_piece);
        auto issue_piece = [&](int i, int c, bool live) {          // i is a constant after unrolling
            if (i < MAXA) {
                const int k = i;
                const int ch_base = c * KCE * p.vt;
                const bool ok = live && k < npa && a_goff[k] >= 0 && ch_base + (tid & 3) * VE < p.Cin;
                ra[k] = *reinterpret_cast<const u32x4*>(ok ? xg0 + (a_goff[k] + ch_base) : zp);
            } else {
                const int k = i - MAXA;
                const T* wc = wg + (int64_t)c * p.ntaps * p.Cout * KCE;
                const bool ok = live && tid + (k << 8) < nbp && b_ok(k);
                rb[k] = *reinterpret_cast<const u32x4*>(ok ? wc + b_uniform(k) + b_lane_off : zp);
            }
        };
        if constexpr (ILV) {
#pragma unroll
            for (int i = 0; i < NPC; ++i) issue_piece(i, 0, true);
        } else {
            load_chunk(0);
        }
        for (int c = 0; c < ((SALT_K1_DBG & 2) ? 0 : p.nchunk); ++c) {
            __syncthreads();                    // fragment reads of chunk c-1 are done (c == 0: the input transform table is complete)
            if constexpr (INA) {
                const int ch0 = c * KCE + (tid & 3) * VE;
                if (ch0 < p.Cin) {
                    float sc[8], sh[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { sc[j] = in_tab[ch0 + j]; sh[j] = in_tab[p.Cin + ch0 + j]; }
#pragma unroll
                    for (int k = 0; k < MAXA; ++k) {
                        if (k < npa && a_goff[k] >= 0) {
                            unsigned o[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const unsigned wv = ra[k][j];
                                float lo = __uint_as_float(wv << 16) * sc[2 * j] + sh[2 * j];
                                float hi = __uint_as_float(wv & 0xffff0000u) * sc[2 * j + 1] + sh[2 * j + 1];
                                if (p.in_relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
                                o[j] = f2bf_pk(lo, hi);
                            }
                            ra[k] = u32x4{o[0], o[1], o[2], o[3]};
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < MAXA; ++k)
                if (k < npa && a_goff[k] != -2) *reinterpret_cast<u32x4*>(sA + lds_lane + k * (RPP * 64)) = ra[k];
#pragma unroll
            for (int k = 0; k < MAXBP; ++k)
                if (tid + (k << 8) < nbp) *reinterpret_cast<u32x4*>(sB + lds_lane + k * (RPP * 64)) = rb[k];
            __syncthreads();
            if constexpr (ILV) {
                const bool live = c + 1 < p.nchunk;
                compute_chunk([&](int st) {
#pragma unroll
                    for (int u = 0; u < PPS; ++u)
                        if (st * PPS + u < NPC) issue_piece(st * PPS + u, c + 1, live);
                }, std::integral_constant<int, PPS>{});
            } else {
                if (c + 1 < p.nchunk) load_chunk(c + 1);
                compute_chunk(no_hook, no_vm);
            }
        }
    } else {
        for (int c = 0; c < p.nchunk; ++c) {
            const int ch_base = c * KCE * p.vt;
            __syncthreads();                        // previous chunk's fragment reads are done
#pragma unroll
            for (int k = 0; k < MAXA; ++k) {
                if (k < npa && a_goff[k] != -2) {
                    const int q = tid + (k << 8);
                    u32x4 v = {0u, 0u, 0u, 0u};
                    const int ch0 = ch_base + (q & 3) * VE;
                    if (a_goff[k] >= 0 && ch0 < p.Cin) v = load_piece<T>(xg0, a_goff[k] + ch_base, ch0, p.Cin, x_vec);
                    *reinterpret_cast<u32x4*>(sA + swz_addr(q >> 2, q & 3)) = v;
                }
            }
            for (int q = tid; q < nbp; q += 256) {
                const int row = q >> 2;                       // t*BN + n
                const int t = row / BN, n = row - t * BN;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (n0 + n < p.Cout)
                    v = *reinterpret_cast<const u32x4*>(wg + (((int64_t)c * p.ntaps + t) * p.Cout + n0 + n) * KCE + (q & 3) * VE);
                *reinterpret_cast<u32x4*>(sB + swz_addr(row, q & 3)) = v;
            }
            __syncthreads();
            compute_chunk(no_hook, no_vm);
        }
    }

    if (SALT_K1_DBG & 4) {
        float tsum = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) tsum += acc[i][j][r];
        if (tsum == 123.456f) reinterpret_cast<float*>(p.y)[0] = tsum;
        return;
    }
    conv_epilogue<T, MI, NI, WM, WN>(p, TileCoord{m_tile, n_tile, oy0, ox0, b0, n0, out_oy, out_ox}, acc, smem);
}

// ------------------------------------------------------------------------------------------ conv_glds_kernel (bf16, stride 1)
// Second kernel of the family, for the layers whose cost is the staging, not the contraction (every 9.66-GFLOP ResNet layer,
// the decoder layers): ONE 512-thread workgroup per CU that owns the whole LDS.
//   * Staging is LDS-DMA (global_load_lds_dwordx4): no staging registers, no ds_write pass.  A ring of NBUF chunk buffers
//     (halo rows + the chunk's weights of all taps) is filled NBUF-1 chunks ahead; one raw s_barrier per chunk, counted vmcnt
//     (never 0 inside the loop), so HBM/L2 latency is covered by the chunks in flight instead of by co-resident workgroups.
//     The DMA destination is lane-linear (M0 + lane*16), so the XOR slot swizzle of the 64-byte rows is applied to the per-lane
//     SOURCE address (slot s of row r holds channel slot s ^ (r>>2)); the fragment reads use the same involution.
//   * The 8 waves are MB pixel blocks (32*MI pixels each) x KS K-slices: every wave accumulates a (32*MI) x BN block over ITS
//     share of the (tap, k-step) stages of each chunk (register blocking MI x NI = 2 x 2: one ds_read_b128 per MFMA instead of
//     two on the 128x32 tile), and the two waves of a SIMD (w, w+4) carry the two tap halves (5 + 4 taps), so every matrix
//     pipe sees the same 9 taps.  After the loop the K-slices are reduced through the (then free) ring in fixed slice order.
//   * Epilogue as in conv_mfma_kernel (bias / folded BN / ReLU / accumulate / BN partials / fold mode / BN-backward sums).
// MFMA row -> pixel of a 32-pixel sub-tile.  ds_read_b128 is serviced in 16-lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31};
// with the natural order (lane = 2 image rows x 16 pixels) a group reads LDS rows a..a+3, a+12..a+15 and hw+a+4..hw+a+11, and
// because the halo pitch hw = 18 is not a multiple of 16 two of them share a bank (rows congruent mod 16): +50 % LDS cycles
// (SQ_LDS_BANK_CONFLICT).  Permuting the blocks of 4 rows so that each lane group covers 16 CONSECUTIVE pixels of one image row
// makes every fragment read conflict free for any tap shift.  Pure relabelling: the epilogue stages row m at pixel perm(m).
__device__ __forceinline__ int lane_pixel_perm(int m) {                  // m in [0, 32)
    return (int)((0x73261540u >> ((m >> 2) * 4)) & 0xfu) * 4 + (m & 3);
}

// The same for 8 x 8 tiles of a 3 x 3 layer (halo pitch 10: the 8 x 8 maps, a whole image per 64 pixels; round 6 - the SQ counters of
// the step showed SQ_LDS_BANK_CONFLICT at 55 % of this kernel's LDS cycles, profiles/r06_pmc_sq.json): image rows y and y + 4 are
// 40 = 8 (mod 16) LDS rows apart, so a 16-lane group that covers 8 pixels of row y and 8 of row y + 4 touches 16 distinct rows mod 16
// for any tap shift.  The two 32-pixel sub-tiles of an image take rows {0, 1, 4, 5} and {2, 3, 6, 7}; s = sub-tile parity.
__device__ __forceinline__ int lane_pixel_perm8(int s, int m) {           // m in [0, 32) -> pixel of the 8 x 8 image
    const int g = m >> 2;
    const int row = 2 * s + (int)((0x50054114u >> (g * 4)) & 0xfu);
    return row * 8 + ((g >> 1) & 1) * 4 + (m & 3);
}
// tile pixel of MFMA row `mr` of 32-pixel sub-tile `sub`: mode 0 natural, 1 lane_pixel_perm, 2 lane_pixel_perm8
__device__ __forceinline__ int frag_pixel(int mode, int sub, int mr) {
    return mode == 2 ? (sub >> 1) * 64 + lane_pixel_perm8(sub & 1, mr) : sub * 32 + (mode == 1 ? lane_pixel_perm(mr) : mr);
}

__device__ __forceinline__ void wait_vmcnt_upto(int n) {               // n is wave-uniform
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
        case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

// halo rows (padded to 16) a tile may have: 128-pixel tiles <= 208 (8x16 pixels + 3x3 halo = 180, two 8x8 images = 200),
// 256-pixel tiles <= 400 (16x16 + halo = 324, four 8x8 images = 400)
constexpr int v2_namax(int bm) { return bm <= 128 ? 13 : 25; }
// compile-time ablation switches of conv_glds_kernel (tools/build_variant.sh -DSALT_V2_DBG=N; 0 in the shipped library):
// 1 no DMA, 2 no MFMA, 4 no fragment reads, 8 return after the main loop, 16 return after the prologue
#ifndef SALT_V2_DBG
#define SALT_V2_DBG 0
#endif
constexpr int DBG = SALT_V2_DBG;

template <int MI, int NI, int MB, int KS, int NT>
__global__ __launch_bounds__(512, 2) void conv_glds_kernel(ConvKP p) {
    constexpr int NAMAX = v2_namax(32 * MI * MB);
    typedef bf16_t T;
    static_assert(MB * KS == 8, "8 waves");
    static_assert(KS == 2 || KS == 4, "K slices");
    constexpr int NTHR = 512, VE = 8, KCE = 32;
    constexpr int BM = 32 * MI * MB, BN = 32 * NI;
    constexpr int NSUB = MI * NI;                                        // 32x32 sub-blocks of a wave's block
    constexpr int OWN = NSUB >= KS ? NSUB / KS : 1;                      // sub-blocks a wave finalises after the slice reduction
    static_assert(NSUB >= KS ? (NSUB % KS == 0) : (KS % NSUB == 0), "slice reduction");
    constexpr int T0N = (NT + 1) / 2;                                    // taps of the first tap half
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int khalf = lane >> 5, l31 = lane & 31;
    const int mb = wave % MB, sl = wave / MB;
    const int taphalf = sl / (KS / 2), hsel = sl % (KS / 2);

    // ---- tile coordinates (same XCD-aware order as conv_mfma_kernel)
    const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3;
    const int n_tile = local % p.n_tiles;
    const int m_tile = (local / p.n_tiles) * 8 + xcd;
    if (m_tile >= p.m_tiles) return;
    int tile = m_tile, out_oy = p.out_oy, out_ox = p.out_ox;
    int64_t w_off = 0;
    if (p.nphase > 1) {                       // phase-fused stride-2 launch: the tile's output parity selects offset and weight block
        const int ph = m_tile / p.m_tiles_ph;
        tile = m_tile - ph * p.m_tiles_ph;
        out_oy = ph >> 1; out_ox = ph & 1;
        w_off = ph * p.w_phase_elems;
    }
    const int txi = tile % p.tiles_x; tile /= p.tiles_x;
    const int tyi = tile % p.tiles_y;
    const int tbi = tile / p.tiles_y;
    const int oy0 = tyi * p.th, ox0 = txi * p.tw - p.ox_shift, b0 = tbi * p.nb;
    const int n0 = n_tile * BN;
    const int hhw = p.hh * p.hw;
    const int phalo = p.nb * hhw;
    const int iy0 = oy0 + p.min_dy, ix0 = ox0 + p.min_dx;
    const T* xg0 = reinterpret_cast<const T*>(p.x) + (int64_t)b0 * p.H * p.W * p.x_cs;
    const T* wg = reinterpret_cast<const T*>(p.w) + w_off;
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_zero_piece);

    // ---- DMA slots.  A chunk is NA + NB wave-instructions of 1 KB (16 LDS rows each): NA halo pieces (rows padded to a multiple
    // of 16; pad rows and out-of-image pixels read the zero piece) and NB weight pieces.  Wave w owns the wave-instructions
    // w, w + 8, ...; EVERY wave issues exactly NS instructions per chunk - a surplus slot copies the zero piece into a per-wave
    // scratch KB behind the ring - so the vmcnt bookkeeping is a compile-time constant and the issue is branch free: the slots ride
    // between the MFMA stages (one or two per stage) instead of in a burst after the barrier, where all eight waves queued behind
    // the CU's address path and nobody fed the matrix cores (measured: DMA time + MFMA time, no overlap at all).
    constexpr int NB_I = NT * BN / 16;                                   // weight wave-instructions per chunk
    constexpr int NS = (NAMAX + NB_I + 7) / 8;                           // DMA instructions per wave and chunk
    const int na = p.a_bytes >> 10;                                      // halo wave-instructions (host: a_bytes = 1 KB * ceil(phalo / 16))
    const int buf_bytes = p.a_bytes + NB_I * 1024;
    const int nbuf = p.nbuf;                                             // ring depth (host: 2 or 3)
    const int dummy_off = nbuf * buf_bytes + wave * 1024;
    const int w_chunk_bytes = NT * p.Cout * KCE * (int)sizeof(T);
    const unsigned char* d_src[NS];
    int d_inc[NS], d_off[NS];                                            // d_off < 0: surplus slot (scratch destination)
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int j = wave + 8 * i;
        d_src[i] = zp; d_inc[i] = 0; d_off[i] = -1;
        if (j < na) {
            d_off[i] = j * 1024;
            const int row = j * 16 + (lane >> 2);
            if (row < phalo) {
                const int bl = (int)__umulhi((unsigned)row, p.hhw_magic);
                const int r = row - bl * hhw;
                const int hy = (int)__umulhi((unsigned)r, p.hw_magic);
                const int hx = r - hy * p.hw;
                int iy = iy0 + hy, ix = ix0 + hx;
                bool valid = b0 + bl < p.B;
                if (p.pad_mode) { iy = min(max(iy, 0), p.H - 1); ix = min(max(ix, 0), p.W - 1); }
                else valid = valid && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                if (valid) {
                    const int slot = (lane ^ (row >> 2)) & 3;            // LDS slot lane&3 of row `row` holds channel slot `slot`
                    d_src[i] = reinterpret_cast<const unsigned char*>(xg0 + ((bl * p.H + iy) * p.W + ix) * p.x_cs + slot * VE);
                    d_inc[i] = KCE * (int)sizeof(T);
                }
            }
        } else if (j < na + NB_I) {
            d_off[i] = p.a_bytes + (j - na) * 1024;
            const int row = (j - na) * 16 + (lane >> 2);                 // t * BN + n
            const int t = row / BN, n = row - t * BN;
            if (n0 + n < p.Cout) {
                const int slot = (lane ^ (n >> 2)) & 3;                  // BN % 16 == 0: (row >> 2) & 3 == (n >> 2) & 3
                d_src[i] = reinterpret_cast<const unsigned char*>(wg + ((int64_t)t * p.Cout + n0 + n) * KCE + slot * VE);
                d_inc[i] = w_chunk_bytes;
            }
        }
    }
    // issue slot i of the next chunk into ring slot `buf` (`live` false past the last chunk: zero piece -> scratch, same count)
    auto issue_slot = [&](int i, int buf, bool live) {                   // i is a constant after unrolling
        if (DBG & 1) return;
        const bool real = live && d_off[i] >= 0;
        const int dst = real ? buf * buf_bytes + d_off[i] : dummy_off;
        const unsigned char* src = live ? d_src[i] : zp;
        __builtin_amdgcn_global_load_lds((glb_void_ptr)src, (lds_void_ptr)(smem + dst), 16, 0, 0);
        d_src[i] += d_inc[i];
    };

    // ---- fragment addressing
    // conflict-free lane -> pixel order: 16-pixel tile rows (1), 8 x 8 tiles at halo pitch 10 (2)
    const int perm = p.gen ? 0 : p.tw_log2 == 4 ? 1 : (MI % 2 == 0 && p.tw_log2 == 3 && p.th_log2 == 3 && p.hw == 10 && !p.no_perm8) ? 2 : 0;
    int pbase[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = frag_pixel(perm, mb * MI + i, l31);
        const TilePix tp = tile_pix(p, m);
        const int tx = tp.tx, ty = tp.ty, bl = tp.bl;
        pbase[i] = bl < p.nb ? bl * hhw + ty * p.hw + tx : 0;                         // (general tiles: a row past the last pixel reads halo row 0)
    }
    int b_addr[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) b_addr[j] = p.a_bytes + swz_addr(j * 32 + l31, khalf);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    struct Frag { u32x4 a[MI], b[NI]; };
    auto mma_frag = [&](const Frag& f) {
        if (DBG & 2) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j][0] += __uint_as_float(f.a[i].x ^ f.b[j].y);
            return;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[i]), __builtin_bit_cast(bf16x8, f.b[j]), acc[i][j], 0, 0, 0);
    };

    // main loop body for one tap half (TA .. TB-1 compile-time); KS == 2: both k-steps, KS == 4: the k-step hsel
    auto run = [&](auto ta_c, auto tb_c) {
        constexpr int TA = decltype(ta_c)::value, TB = decltype(tb_c)::value, NTL = TB - TA;
        constexpr int NST = NTL * (KS == 2 ? 2 : 1);                     // stages per chunk of this wave
        int a_addr[NTL][MI];
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int i = 0; i < MI; ++i) a_addr[t][i] = swz_addr(pbase[i] + p.tap_off[TA + t], khalf);
        const int hx4 = (KS == 4) ? (hsel << 5) : 0;
        auto load_frag = [&](int s, const unsigned char* base, Frag& f) {  // s is a constant after unrolling
            const int tl = (KS == 2) ? (s >> 1) : s;
            const int hx = (KS == 2) ? ((s & 1) << 5) : hx4;
            if (DBG & 4) {
#pragma unroll
                for (int i = 0; i < MI; ++i) f.a[i] = u32x4{(unsigned)tl, 1u, 2u, 3u};
#pragma unroll
                for (int j = 0; j < NI; ++j) f.b[j] = u32x4{(unsigned)hx, 5u, 6u, 7u};
                return;
            }
#pragma unroll
            for (int i = 0; i < MI; ++i) f.a[i] = *reinterpret_cast<const u32x4*>(base + (a_addr[tl][i] ^ hx));
#pragma unroll
            for (int j = 0; j < NI; ++j) f.b[j] = *reinterpret_cast<const u32x4*>(base + (TA + tl) * (BN * 64) + (b_addr[j] ^ hx));
        };
        // prologue: ring depth - 1 chunks in flight
#pragma unroll 1
        for (int c0 = 0; c0 < nbuf - 1; ++c0) {
            const bool live = c0 < p.nchunk;
#pragma unroll
            for (int i = 0; i < NS; ++i) issue_slot(i, c0, live);
        }
        int slot_c = 0, slot_i = nbuf - 1;                                // ring slot being computed / filled
        for (int c = 0; c < p.nchunk; ++c) {
            // chunk c landed: this wave's share by the counted vmcnt (the nbuf - 2 younger chunks may stay in flight), everybody's
            // by the barrier; the barrier also says every wave is done reading the slot the DMA below overwrites (chunk c-1's)
            if (!(DBG & 1)) {
                if (nbuf == 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NS) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const bool live = c + nbuf - 1 < p.nchunk;
            const int fill = slot_i;
            slot_i = (slot_i + 1 == nbuf) ? 0 : slot_i + 1;
            const unsigned char* base = smem + slot_c * buf_bytes;
            slot_c = (slot_c + 1 == nbuf) ? 0 : slot_c + 1;
            Frag f0, f1;
            load_frag(0, base, f0);
            __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
            constexpr int SPS = (NS + NST - 1) / NST;                    // DMA slots per stage
#pragma unroll
            for (int s = 0; s < NST; s += 2) {
                if (s + 1 < NST) load_frag(s + 1, base, f1);
#pragma unroll
                for (int u = 0; u < SPS; ++u) if (s * SPS + u < NS) issue_slot(s * SPS + u, fill, live);
                mma_frag(f0);
                if (s + 1 < NST) __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
                if (!(DBG & 1)) __builtin_amdgcn_sched_group_barrier(0x010, SPS, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
                if (s + 1 < NST) {
                    if (s + 2 < NST) load_frag(s + 2, base, f0);
#pragma unroll
                    for (int u = 0; u < SPS; ++u) if ((s + 1) * SPS + u < NS) issue_slot((s + 1) * SPS + u, fill, live);
                    mma_frag(f1);
                    if (s + 2 < NST) __builtin_amdgcn_sched_group_barrier(0x100, MI + NI, 0);
                    if (!(DBG & 1)) __builtin_amdgcn_sched_group_barrier(0x010, SPS, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, MI * NI, 0);
                }
            }
        }
    };
    if (DBG & 16) { if (tid == 0 && blockIdx.x == 0x7fffffff) reinterpret_cast<int*>(p.y)[0] = pbase[0] + (int)(size_t)d_src[0] + d_off[1] + b_addr[0]; return; }
    if (taphalf == 0) run(std::integral_constant<int, 0>{}, std::integral_constant<int, T0N>{});
    else run(std::integral_constant<int, T0N>{}, std::integral_constant<int, NT>{});
    if (DBG & 8) {
        float tsum = 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) tsum += acc[i][j][r];
        if (tsum == 123.456f) reinterpret_cast<float*>(p.y)[0] = tsum;
        return;
    }

    // ---- K-slice reduction through LDS (the ring is free): sub-block u of a pixel block is finalised by slice u % KS (NSUB >= KS)
    // or by slice u (NSUB < KS); partial sums are added in ascending slice order whatever the owner is (deterministic)
    f32x16 fin[OWN];
    int rowoff[OWN], coloff[OWN];
    bool owner;
    {
        float* red = reinterpret_cast<float*>(smem);                     // [mb][u][slice][16][64] floats
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                 // every wave is done with the ring
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int u = i * NI + j;
                const int own_sl = NSUB >= KS ? (u % KS) : u;
                if (sl != own_sl) {
                    float* dst = red + ((((mb * NSUB + u) * KS + sl) * 16) * 64) + lane * 4;
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 v = {acc[i][j][r4 * 4], acc[i][j][r4 * 4 + 1], acc[i][j][r4 * 4 + 2], acc[i][j][r4 * 4 + 3]};
                        *reinterpret_cast<f32x4*>(dst + r4 * 256) = v;
                    }
                }
            }
        __syncthreads();
        owner = NSUB >= KS ? true : (sl < NSUB);
#pragma unroll
        for (int o = 0; o < OWN; ++o) {
            // NSUB >= KS: owned sub-blocks u = sl + o * KS;  NSUB < KS: u = sl (owners only)
            const int u = NSUB >= KS ? (sl + o * KS) : (sl < NSUB ? sl : 0);
            const int i = u / NI, j = u % NI;
            rowoff[o] = (mb * MI + i) * 32; coloff[o] = j * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) fin[o][r] = 0.f;
            if (owner) {
#pragma unroll
                for (int s2 = 0; s2 < KS; ++s2) {
                    if (s2 == sl) {
                        // own partial: select acc[i][j] with compile-time indices
#pragma unroll
                        for (int ii = 0; ii < MI; ++ii)
#pragma unroll
                            for (int jj = 0; jj < NI; ++jj)
                                if (ii * NI + jj == u) {
#pragma unroll
                                    for (int r = 0; r < 16; ++r) fin[o][r] += acc[ii][jj][r];
                                }
                    } else {
                        const float* src = red + ((((mb * NSUB + u) * KS + s2) * 16) * 64) + lane * 4;
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(src + r4 * 256);
                            fin[o][r4 * 4] += v.x; fin[o][r4 * 4 + 1] += v.y; fin[o][r4 * 4 + 2] += v.z; fin[o][r4 * 4 + 3] += v.w;
                        }
                    }
                }
            }
        }
    }

    if (DBG & 32) {
        float tsum = 0.f;
#pragma unroll
        for (int o = 0; o < OWN; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) tsum += fin[o][r];
        if (tsum == 123.456f) reinterpret_cast<float*>(p.y)[0] = tsum;
        return;
    }
    // ---- epilogue (conv_mfma_kernel's, generalised to "a wave owns OWN 32x32 sub-blocks at (rowoff, coloff)" and 512 threads)
    constexpr int PITCH = BN + 16 / (int)sizeof(T);
    constexpr int PPO = BN / VE;
    constexpr int RB = BM / 32;                                          // 32-row blocks of the tile
    T* sO = reinterpret_cast<T*>(smem);
    float ssum[OWN], cntf[OWN];
    unsigned vmask[OWN];
#pragma unroll
    for (int o = 0; o < OWN; ++o) { ssum[o] = 0.f; cntf[o] = 0.f; vmask[o] = 0xffffu; }
    const bool want_stats = p.stats != nullptr || p.fin.acc != nullptr;
    const bool has_affine = p.bias || p.scale || p.shift || p.relu;
    const bool full_tile = !p.gen && (b0 + p.nb <= p.B) && (oy0 + p.th <= p.OH) && ox0 >= 0 && (ox0 + p.tw <= p.OW);
    if (want_stats) {
#pragma unroll
        for (int o = 0; o < OWN; ++o) {
            if (full_tile) { cntf[o] = 16.f; }
            else {
                vmask[o] = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mr = (r & 3) + 8 * (r >> 2) + 4 * khalf;
                    const int m = frag_pixel(perm, rowoff[o] >> 5, mr);
                    const TilePix tp = tile_pix(p, m);
                    const int tx = tp.tx, ty = tp.ty, bl = tp.bl;
                    const bool valid = (b0 + bl < p.B) && (oy0 + ty < p.OH) && (ox0 + tx < p.OW) && (ox0 + tx >= 0);
                    if (valid) { vmask[o] |= 1u << r; cntf[o] += 1.f; }
                }
            }
        }
    }
    __syncthreads();                                                     // the reduction scratch has been read
    if (owner) {
#pragma unroll
        for (int o = 0; o < OWN; ++o) {
            float bias = 0.f, sc = 1.f, sh = 0.f;
            const int n = n0 + coloff[o] + l31;
            if (has_affine) {
                const bool nok = n < p.Cout;
                bias = (nok && p.bias) ? p.bias[n] : 0.f;
                sc = (nok && p.scale) ? p.scale[n] : 1.f;
                sh = (nok && p.shift) ? p.shift[n] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = fin[o][r];
                if (has_affine) {
                    v = (v + bias) * sc + sh;
                    if (p.relu && !p.res) v = fmaxf(v, 0.f);
                    fin[o][r] = v;
                }
                if (want_stats && ((vmask[o] >> r) & 1u)) ssum[o] += v;
                const int mr = (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const int ml = frag_pixel(perm, rowoff[o] >> 5, mr);
                Elem<T>::st(sO + ml * PITCH + coloff[o] + l31, v);
            }
        }
    }
    __syncthreads();
    {
        T* yg = reinterpret_cast<T*>(p.y);
        const bool y_vec = ((p.y_cs % VE) == 0) && ((reinterpret_cast<uintptr_t>(p.y) & 15) == 0);
        const bool bnb = p.bnb_partials != nullptr || p.bnbf.acc != nullptr;
        float b1[VE], b2[VE], bmu[VE], bis[VE], bsc[VE], bsh[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) { b1[e] = 0.f; b2[e] = 0.f; bmu[e] = 0.f; bis[e] = 0.f; bsc[e] = 0.f; bsh[e] = 0.f; }
        if (bnb) {
            const int nb0 = n0 + (tid % PPO) * VE;
#pragma unroll
            for (int e = 0; e < VE; ++e)
                if (nb0 + e < p.Cout) {
                    bmu[e] = p.bnb_mean[nb0 + e]; bis[e] = p.bnb_invstd[nb0 + e];
                    bsc[e] = p.bnb_gamma[nb0 + e] * bis[e]; bsh[e] = p.bnb_beta[nb0 + e] - bmu[e] * bsc[e];
                }
        }
        if (DBG & 64) { if (sO[tid] == 12345) reinterpret_cast<T*>(p.y)[0] = sO[tid + 1]; return; }
        for (int q = tid; q < BM * PPO; q += NTHR) {
            const int m = q / PPO, pc = q - m * PPO;
            const TilePix tp = tile_pix(p, m);
            const int tx = tp.tx, ty = tp.ty, bl = tp.bl;
            const int oy = oy0 + ty, ox = ox0 + tx, b = b0 + bl;
            const int n = n0 + pc * VE;
            if (b >= p.B || oy >= p.OH || ox >= p.OW || ox < 0 || n >= p.Cout) continue;
            T* dst = yg + (((int64_t)b * p.OHf + oy * p.out_step + out_oy) * p.OWf + ox * p.out_step + out_ox) * p.y_cs + n;
            bool accum = p.accumulate != 0;
            bool dvec = y_vec;
            int fold_rows = 0, fold_cols = 0;                      // fused fold: ring pixels above / right of this edge pixel
            if (p.fold_fused) {
                // replicate-pad adjoint inside the tile: the pad ring (top rows, right columns of the extended grid) is never stored;
                // the edge pixel it folds onto sums its ring pixels from the staged tile (the tile grid is laid out so that they
                // share a tile: rows start at 0 with th > fold_top, columns at -ox_shift)
                const int iy = oy - p.fold_top;
                if (iy < 0 || ox >= p.OWf) continue;
                dst = yg + (((int64_t)b * p.OHf + iy) * p.OWf + ox) * p.y_cs + n;
                fold_rows = iy == 0 ? p.fold_top : 0;
                fold_cols = ox == p.OWf - 1 ? p.fold_right : 0;
            }
            if (p.strip) {
                const int iy = oy - p.fold_top, ix = ox - p.fold_left;
                if (iy >= 0 && iy < p.OHf && ix >= 0 && ix < p.OWf) {
                    dst = yg + (((int64_t)b * p.OHf + iy) * p.OWf + ix) * p.y_cs + n;
                } else {
                    const int64_t ring = (int64_t)(p.fold_top + p.fold_bottom) * p.OW + (int64_t)p.OHf * (p.fold_left + p.fold_right);
                    dst = reinterpret_cast<T*>(p.strip) + ((int64_t)b * ring + fold_ring_index(oy, ox, p.OHf, p.OWf, p.fold_top, p.fold_bottom, p.fold_left, p.fold_right)) * p.strip_cs + n;
                    accum = false;
                    dvec = (p.strip_cs % VE) == 0;
                }
            }
            u32x4 v = *reinterpret_cast<const u32x4*>(sO + m * PITCH + pc * VE);
            if (fold_rows | fold_cols) {                           // host: fused fold implies whole aligned channel pieces
                float f[VE], o[VE];
                unpack16<T>(v, f);
                for (int ky = 0; ky <= fold_rows; ++ky)
                    for (int kx = 0; kx <= fold_cols; ++kx) {
                        if ((ky | kx) == 0) continue;
                        unpack16<T>(*reinterpret_cast<const u32x4*>(sO + (m - ky * p.tw + kx) * PITCH + pc * VE), o);
#pragma unroll
                        for (int e = 0; e < VE; ++e) f[e] += o[e];
                    }
                v = pack16<T>(f);
            }
            if (dvec && n + VE <= p.Cout) {
                u32x4 stored = v;
                if (accum) {
                    float f[VE], o[VE];
                    unpack16<T>(v, f);
                    unpack16<T>(*reinterpret_cast<const u32x4*>(dst), o);
#pragma unroll
                    for (int e = 0; e < VE; ++e) f[e] += o[e];
                    stored = pack16<T>(f);
                }
                if (p.res) {
                    float f[VE], o[VE];
                    unpack16<T>(stored, f);
                    unpack16<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.res) + (((int64_t)b * p.OHf + oy) * p.OWf + ox) * p.res_cs + n), o);
#pragma unroll
                    for (int e = 0; e < VE; ++e) { f[e] += o[e]; if (p.relu) f[e] = fmaxf(f[e], 0.f); }
                    stored = pack16<T>(f);
                }
                *reinterpret_cast<u32x4*>(dst) = stored;
                if (bnb) {
                    float g[VE], yc[VE];
                    unpack16<T>(stored, g);
                    unpack16<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.bnb_y) + (((int64_t)b * p.OHf + oy * p.out_step + out_oy - (p.fold_fused ? p.fold_top : 0)) * p.OWf + ox * p.out_step + out_ox) * p.bnb_cs + n), yc);
                    if (p.bnb_a) {
                        float av[VE];
                        unpack16<T>(*reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(p.bnb_a) + (((int64_t)b * p.OHf + oy * p.out_step + out_oy - (p.fold_fused ? p.fold_top : 0)) * p.OWf + ox * p.out_step + out_ox) * p.bnb_acs + n), av);
#pragma unroll
                        for (int e = 0; e < VE; ++e) {
                            const float gg = (!p.bnb_relu || av[e] > 0.f) ? g[e] : 0.f;
                            b1[e] += gg; b2[e] += gg * (yc[e] - bmu[e]) * bis[e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < VE; ++e) {
                            const float gg = (!p.bnb_relu || yc[e] * bsc[e] + bsh[e] > 0.f) ? g[e] : 0.f;
                            b1[e] += gg; b2[e] += gg * (yc[e] - bmu[e]) * bis[e];
                        }
                    }
                }
            } else {
                float f[VE];
                unpack16<T>(v, f);
#pragma unroll
                for (int e = 0; e < VE; ++e)
                    if (n + e < p.Cout) Elem<T>::st(dst + e, accum ? f[e] + Elem<T>::ld(dst + e) : f[e]);
            }
        }
        if (bnb) {
            float* sR = reinterpret_cast<float*>(smem);
            const int row = tid / PPO, cl0 = (tid % PPO) * VE;
            __syncthreads();
#pragma unroll
            for (int e = 0; e < VE; ++e) { sR[(row * BN + cl0 + e) * 2] = b1[e]; sR[(row * BN + cl0 + e) * 2 + 1] = b2[e]; }
            __syncthreads();
            for (int e = tid; e < 2 * BN; e += NTHR) {
                const int st = e >= BN ? 1 : 0, cl = e - st * BN;
                float t = 0.f;
                for (int r = 0; r < NTHR / PPO; ++r) t += sR[(r * BN + cl) * 2 + st];
                if (n0 + cl < p.Cout) {
                    if (p.bnbf.acc) fin_add(p.bnbf.acc + ((blockIdx.x & 7) * 2 + st) * p.Cout + n0 + cl, (double)t);
                    else p.bnb_partials[((int64_t)m_tile * 2 + st) * p.Cout + n0 + cl] = t;
                }
            }
            if (p.bnbf.acc && p.bnbf.ticket) {
                unsigned* flag = reinterpret_cast<unsigned*>(sR + (NTHR / PPO) * BN * 2);
                if (fin_arrive(p.bnbf.ticket, (unsigned)(p.m_tiles * p.n_tiles), flag)) fin_backward(p.bnbf, p.bnb_gamma, p.bnb_invstd, p.Cout);
            }
        }
    }
    if (want_stats) {
        // per 32-row block and channel: (sum, M2 about the block's own mean, count), merged over the tile's RB blocks in fixed order
        float* sS = reinterpret_cast<float*>(smem);                      // [RB][BN][2] then [RB] counts
        float* sC = sS + RB * BN * 2;
        __syncthreads();
        if (owner) {
#pragma unroll
            for (int o = 0; o < OWN; ++o) {
                const float cnt = cntf[o] + __shfl_xor(cntf[o], 32);
                const float s = ssum[o] + __shfl_xor(ssum[o], 32);
                const float mean = cnt > 0.f ? s / cnt : 0.f;
                float m2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((vmask[o] >> r) & 1u) { const float d = fin[o][r] - mean; m2 += d * d; }
                m2 += __shfl_xor(m2, 32);
                const int rb = rowoff[o] >> 5;
                if (khalf == 0) { sS[(rb * BN + coloff[o] + l31) * 2 + 0] = s; sS[(rb * BN + coloff[o] + l31) * 2 + 1] = m2; }
                if (lane == 0 && coloff[o] == 0) sC[rb] = cnt;
            }
        }
        __syncthreads();
        const int part = p.stats_part0 + m_tile;
        if (tid < BN) {
            float N = 0.f, S = 0.f, M2 = 0.f;
#pragma unroll
            for (int w = 0; w < RB; ++w) {
                const float nk = sC[w];
                if (nk > 0.f) {
                    const float sk = sS[(w * BN + tid) * 2 + 0], mk = sS[(w * BN + tid) * 2 + 1];
                    if (N == 0.f) { N = nk; S = sk; M2 = mk; }
                    else {
                        const float d = sk / nk - S / N;
                        M2 += mk + d * d * (N * nk / (N + nk));
                        S += sk; N += nk;
                    }
                }
            }
            const int n = n0 + tid;
            if (p.fin.acc) {
                double* a = p.fin.acc + (blockIdx.x & 7) * (2 * p.Cout + 1);
                if (n < p.Cout && N > 0.f) {
                    fin_add(a + n, (double)S);
                    fin_add(a + p.Cout + n, (double)M2 + (double)S * (double)S / (double)N);
                }
                if (tid == 0 && n_tile == 0) fin_add(a + 2 * p.Cout, (double)N);
            } else {
                if (n < p.Cout) {
                    p.stats[((int64_t)part * 2 + 0) * p.Cout + n] = S;
                    p.stats[((int64_t)part * 2 + 1) * p.Cout + n] = M2;
                }
                if (tid == 0 && n_tile == 0) p.stats_cnt[part] = N;
            }
        }
        if (p.fin.acc && p.fin.ticket) {
            unsigned* flag = reinterpret_cast<unsigned*>(sC + RB + 2);
            if (fin_arrive(p.fin.ticket, (unsigned)(p.m_tiles * p.n_tiles), flag)) fin_forward(p.fin, p.Cout, reinterpret_cast<double*>(smem));
        }
    }
}

struct TileCfg { int id, MI, NI, WM, WN, KS; };   // KS > 0: conv_glds_kernel (WM = pixel blocks MB, WN = 1)
// id -> (BM, BN): 1: 128x64, 2: 256x64, 3: 128x128, 4: 128x32, 5: 64x64
// conv_glds_kernel: 6: 256x64 (4 blocks x 2 slices), 7: 128x64 (2 x 4), 8: 128x32 (2 x 4)
const TileCfg kCfgs[] = {{1, 2, 1, 2, 2, 0}, {2, 2, 2, 4, 1, 0}, {3, 2, 2, 2, 2, 0}, {4, 1, 1, 4, 1, 0}, {5, 1, 1, 2, 2, 0},
                         {6, 2, 2, 4, 1, 2}, {7, 2, 2, 2, 1, 4}, {8, 2, 1, 2, 1, 4}};

struct Plan {
    TileCfg cfg; ConvKP kp; dim3 grid; size_t lds; int parts;
};

int make_plan(const salt_conv_args* a, Plan* pl) {
    if (!a) SALT_FAIL(SALT_E_BADARG, "null args");
    {
        salt_view yv = a->y, xv = a->x;
        if (a->y_plane) yv.cs = yv.C;            // planar y (conv_ws_kernel only): y.cs is the plane's channel count, checked there
        if (a->x_plane) xv.cs = xv.C;            // planar x (conv_ls_kernel only)
        if (!view_ok(xv) || !view_ok(yv) || !a->w) SALT_FAIL(SALT_E_BADARG, "conv: bad view");
        if (a->y_plane < 0 || a->x_plane < 0) SALT_FAIL(SALT_E_BADARG, "conv: x_plane / y_plane");
    }
    if (a->ntaps < 1 || a->ntaps > SALT_MAX_TAPS) SALT_FAIL(SALT_E_BADARG, "conv: ntaps %d", a->ntaps);
    if (a->in_step < 1 || a->in_step > 2 || a->out_step < 1 || a->out_step > 2) SALT_FAIL(SALT_E_BADARG, "conv: steps");
    if (a->OH < 1 || a->OW < 1) SALT_FAIL(SALT_E_BADARG, "conv: empty output grid");
    const bool fold_fused = !a->strip && (a->fold_top > 0 || a->fold_right > 0);
    if (fold_fused) {
        const int ve = a->dtype == SALT_F32 ? 4 : 8;
        if (a->fold_bottom || a->fold_left || a->fold_top < 0 || a->fold_right < 0 || a->out_step != 1 || a->out_oy || a->out_ox || a->stats || a->fin_acc ||
            a->OH != a->y.H + a->fold_top || a->OW != a->y.W + a->fold_right)
            SALT_FAIL(SALT_E_BADARG, "conv: fused fold needs OH/OW = y.H + top / y.W + right (top / right pads only), out_step 1, no stats");
        if (a->y.C % ve || a->y.cs % ve || (reinterpret_cast<uintptr_t>(a->y.p) & 15))
            SALT_FAIL(SALT_E_BADARG, "conv: fused fold needs 16-byte aligned whole channel pieces");
    }
    if (!a->strip && !fold_fused && ((a->OH - 1) * a->out_step + a->out_oy >= a->y.H || (a->OW - 1) * a->out_step + a->out_ox >= a->y.W))
        SALT_FAIL(SALT_E_BADARG, "conv: output grid exceeds buffer");
    if (a->x.B != a->y.B) SALT_FAIL(SALT_E_BADARG, "conv: batch mismatch");
    const int Cout = a->y.C;
    int min_dy = 1 << 30, max_dy = -(1 << 30), min_dx = 1 << 30, max_dx = -(1 << 30);
    for (int t = 0; t < a->ntaps; ++t) {
        min_dy = a->tap_dy[t] < min_dy ? a->tap_dy[t] : min_dy; max_dy = a->tap_dy[t] > max_dy ? a->tap_dy[t] : max_dy;
        min_dx = a->tap_dx[t] < min_dx ? a->tap_dx[t] : min_dx; max_dx = a->tap_dx[t] > max_dx ? a->tap_dx[t] : max_dx;
    }
    const int64_t pixels = (int64_t)a->x.B * a->OH * a->OW * (a->nphase > 1 ? a->nphase : 1);
    // 1x1 stride-1 convolution with Cin a multiple of 4 chunks: 4 channel chunks per barrier round (virtual taps, see the kernel)
    const int KCE_ = a->dtype == SALT_F32 ? 16 : 32;
    static const bool vt_off = getenv("SALT_CONV_NO_VT") != nullptr;
    const int vt = (!vt_off && a->ntaps == 1 && a->in_step == 1 && a->x.C % (4 * KCE_) == 0) ? 4 : 1;
    // ---- conv_glds_kernel (one 512-thread workgroup per CU, LDS-DMA ring): bf16, unit input step, 9 or 4 real taps, whole aligned
    // 64-byte channel chunks.  Decided from the geometry only (salt_conv_stats_parts plans with the same rule before the epilogue
    // fields are known).
    static const int v2_env = getenv("SALT_CONV_V2") ? atoi(getenv("SALT_CONV_V2")) : 1;     // 0: off, N >= 6: force config N
    const bool in_tf = a->in_scale || a->in_fin_acc;            // the input transform lives in conv_mfma_kernel's register-staged loader
    const bool v2_ok = a->dtype == SALT_BF16 && a->in_step == 1 && (a->ntaps == 9 || a->ntaps == 4) && a->x.C % 32 == 0 &&
                       a->x.cs % 8 == 0 && (reinterpret_cast<uintptr_t>(a->x.p) & 15) == 0 && !in_tf;
    // ---- tile config heuristic (overridable for tests/tuning)
    int id = ((a->cfg & 0xff) >= 9 && (a->cfg & 0xff) <= 13) ? 0 : (a->cfg & 0xff);   // 9 = "conv_ws_kernel where it applies" (conv_ws.hip): the heuristic decides for the rest
    if (id >= 6 && !v2_ok) id = 0;            // a forced conv_glds config applies where the kernel does (tests force one config per graph)
    if (id == 2 && vt > 1) id = 1;            // 256-pixel tiles x 4 virtual taps exceed the halo-piece budget
    if (id == 0 && v2_ok && v2_env && ((a->cfg & 0xff) == 0 || ((a->cfg & 0xff) >= 9 && (a->cfg & 0xff) <= 13))) {
        if (v2_env >= 6) id = v2_env;
        else {
            // measured (round-2 variant builds, DESIGN_history.md): the LDS-DMA kernel wins where conv_mfma_kernel's 128x32 tiles cannot
            // fill the chip with 64-channel tiles - the 8x8 maps (2048-3200 pixels, 512-768 channels: 24.0 -> 18.1 us) - ties on the
            // 16x16 / 32x32 maps and loses on the large maps, whose few channel chunks leave nothing to pipeline.  1 (default): the
            // few-pixel layers only; 2: + the 16x16 maps; 3: every eligible layer.
            const int64_t t256 = ((pixels + 255) / 256) * cdiv(Cout, 64), t128 = ((pixels + 127) / 128) * cdiv(Cout, 64);
            if (t128 < 192 && Cout > 32) id = 8;
            else if (v2_env >= 2 && t256 < 256 && Cout > 32) id = 7;
            else if (v2_env >= 3) id = Cout <= 32 ? 8 : (t256 >= 256 ? 6 : 7);
        }
    }
    if (id == 0) {
        const bool big = a->in_step == 1 && a->OH >= 16 && a->OW >= 16 && pixels >= 256 * 256 * 2;
        if (Cout <= 32) id = 4;
        else if (big && vt == 1) id = 2;      // 256-pixel tiles: the chunk's weights are staged once per 256 pixels
        else id = 1;
        // few pixels, many channels (ResNet stages 3-4, the 8x8 / 16x16 decoder levels): 128x32 tiles double the workgroup count and
        // run at 3 workgroups per CU - measured 8-15 % faster than 128x64 / 64x64 there (tools/cfg_sweep.sh)
        const int64_t wgs = ((pixels + 127) / 128) * cdiv(Cout, 64);
        static const int small_cfg = getenv("SALT_CONV_SMALL_CFG") ? atoi(getenv("SALT_CONV_SMALL_CFG")) : 4;
        if (id == 1 && wgs < 512) id = (small_cfg == 5 && wgs >= 256) ? 1 : small_cfg;
    }
    const TileCfg* cfg = nullptr;
    for (const auto& c : kCfgs) if (c.id == id) cfg = &c;
    if (!cfg) SALT_FAIL(SALT_E_BADARG, "conv: unknown cfg %d", id);
    for (int attempt = 0; attempt < 4; ++attempt) {
        const int BM = 32 * cfg->MI * cfg->WM;
        ConvKP& k = pl->kp;
        k.tw_log2 = ilog2_ceil(a->OW) < 4 ? ilog2_ceil(a->OW) : 4;
        int rem = ilog2_ceil(BM) - k.tw_log2;
        k.th_log2 = ilog2_ceil(a->OH) < rem ? ilog2_ceil(a->OH) : rem;
        k.nb = BM >> (k.tw_log2 + k.th_log2);
        int th = 1 << k.th_log2, tw = 1 << k.tw_log2;
        k.gen = 0;
        // General tiles for the fused-fold data gradients (the extended grids are 2^n + 2 wide: 16-wide tiles of the 10 / 18 / 34-wide
        // grids are 40 - 60 % empty): full-width strips of th rows, or whole images when one fits.  Taken when they save >= 15 % of the
        // tile slots and their halo fits the loader.
        static const bool gen_off = getenv("SALT_CONV_GEN_TILES") && atoi(getenv("SALT_CONV_GEN_TILES")) == 0;
        if (fold_fused && !gen_off && a->in_step == 1 && vt == 1 && a->OW * (a->fold_top + 1) <= BM) {
            const int gtw = a->OW;
            int gth = BM / gtw, gnb = 1;
            if (gth >= a->OH) { gth = a->OH; gnb = BM / (a->OH * a->OW); }
            const int64_t slots_p2 = (int64_t)cdiv(a->x.B, k.nb) * cdiv(a->OH, th) * cdiv(a->OW, tw) * BM;
            const int64_t slots_g = (int64_t)cdiv(a->x.B, gnb) * cdiv(a->OH, gth) * BM;
            const int gphalo = gnb * (gth + max_dy - min_dy) * (gtw + max_dx - min_dx);
            const bool fits = cfg->KS > 0 ? cdiv(gphalo, 16) <= v2_namax(BM) : gphalo * 4 <= maxa_for(BM) * 256;
            if (fits && slots_g * 115 <= slots_p2 * 100) {
                k.gen = 1; th = gth; tw = gtw; k.nb = gnb; k.tw_log2 = 0; k.th_log2 = 0;
            }
        }
        k.tw = tw; k.th = th; k.twth = tw * th;
        k.tw_magic = (unsigned)((1ull << 32) / (unsigned)tw) + 1u; k.twth_magic = (unsigned)((1ull << 32) / (unsigned)(tw * th)) + 1u;
        k.hh = (th - 1) * a->in_step + (max_dy - min_dy) + 1;
        k.hw = (tw - 1) * a->in_step + (max_dx - min_dx) + 1;
        const int phalo = k.nb * k.hh * k.hw * vt;
        if (cfg->KS > 0) {
            // LDS-DMA ring: halo rows + the chunk's weights of all taps per slot, 3 slots when they fit (2 chunks in flight), else 2
            // (halo rows padded to whole 1-KB DMA instructions; 8 KB behind the ring take the surplus slots' copies)
            const int64_t buf = (int64_t)cdiv(phalo, 16) * 1024 + (int64_t)a->ntaps * (32 * cfg->NI) * 64;
            k.nbuf = 3 * buf + 8192 <= 160 * 1024 ? 3 : 2;
            static const bool no_perm8 = getenv("SALT_GLDS_NO_PERM8") != nullptr;
            k.no_perm8 = no_perm8 ? 1 : 0;
            if (cdiv(phalo, 16) > v2_namax(BM) || 2 * buf + 8192 > 160 * 1024) {
                if (attempt == 0 && cfg->id != 8) { for (const auto& c : kCfgs) if (c.id == 8) cfg = &c; continue; }
                // tiny maps (4 x 4 and below: a 128-pixel tile is 8+ images, each with its own halo ring): conv_mfma_kernel's tiles
                // (round 3: a ResNet50 at 64 x 64 in bf16 failed here instead of falling back)
                if ((a->cfg & 0xff) < 6) { for (const auto& c : kCfgs) if (c.id == 4) cfg = &c; continue; }
                SALT_FAIL(SALT_E_UNSUPPORTED, "conv: halo tile of %d pixels too large for the LDS ring", phalo);
            }
            break;
        }
        if (phalo * 4 > maxa_for(BM) * 256) {
            if (cfg->id != 5) { for (const auto& c : kCfgs) if (c.id == 5) cfg = &c; continue; }
            SALT_FAIL(SALT_E_UNSUPPORTED, "conv: halo tile of %d pixels too large", phalo);
        }
        break;
    }
    pl->cfg = *cfg;
    ConvKP& k = pl->kp;
    const int BN = 32 * cfg->NI * cfg->WN;
    const int KCE = a->dtype == SALT_F32 ? 16 : 32;
    k.x = a->x.p; k.w = a->w; k.y = a->y.p;
    k.bias = a->bias; k.scale = a->scale; k.shift = a->shift; k.stats = a->stats; k.stats_cnt = a->stats_cnt;
    k.B = a->x.B; k.H = a->x.H; k.W = a->x.W; k.Cin = a->x.C; k.x_cs = a->x.cs;
    k.Cout = Cout; k.y_cs = a->y.cs; k.OHf = a->y.H; k.OWf = a->y.W; k.OH = a->OH; k.OW = a->OW;
    k.out_step = a->out_step; k.out_oy = a->out_oy; k.out_ox = a->out_ox;
    k.ntaps = a->ntaps; k.in_step = a->in_step; k.pad_mode = a->pad_mode; k.min_dy = min_dy; k.min_dx = min_dx;
    for (int t = 0; t < a->ntaps; ++t) k.tap_off[t] = (a->tap_dy[t] - min_dy) * k.hw + (a->tap_dx[t] - min_dx);
    k.vt = vt;
    if (vt > 1) { k.ntaps = vt; for (int t = 0; t < vt; ++t) k.tap_off[t] = t * k.nb * k.hh * k.hw; }
    k.fold_fused = fold_fused ? 1 : 0;
    // fused fold: the tile columns are RIGHT-aligned with the extended grid (the last tile ends at OW), so the right pad columns
    // share a tile with the last image column whatever W is; rows start at 0, the top pad rows share the first tile with row 0
    k.tiles_y = cdiv(a->OH, k.th); k.tiles_x = cdiv(a->OW, k.tw);
    k.ox_shift = (fold_fused && a->fold_right > 0) ? k.tiles_x * k.tw - a->OW : 0;
    if (fold_fused && (k.th <= a->fold_top || k.tw <= a->fold_right))
        SALT_FAIL(SALT_E_UNSUPPORTED, "conv: fused fold needs tiles larger than the pad (%d x %d)", k.th, k.tw);
    const int tiles_b = cdiv(a->x.B, k.nb);
    k.nchunk = cdiv(a->x.C, KCE * vt);
    k.a_bytes = k.nb * k.hh * k.hw * 64 * vt;
    if (cfg->KS > 0) k.a_bytes = cdiv(k.nb * k.hh * k.hw, 16) * 1024;
    k.hhw_magic = (unsigned)((1ull << 32) / (unsigned)(k.hh * k.hw)) + 1u;
    k.hw_magic = (unsigned)((1ull << 32) / (unsigned)k.hw) + 1u;
    k.relu = a->relu; k.accumulate = a->accumulate; k.stats_part0 = a->stats_part0;
    k.strip = a->strip; k.strip_cs = a->strip_cs; k.fold_top = a->fold_top; k.fold_bottom = a->fold_bottom; k.fold_left = a->fold_left; k.fold_right = a->fold_right;
    k.m_tiles = tiles_b * k.tiles_y * k.tiles_x; k.n_tiles = cdiv(Cout, BN);
    k.nphase = a->nphase > 1 ? a->nphase : 1; k.m_tiles_ph = k.m_tiles; k.w_phase_elems = a->w_phase_elems;
    if (k.nphase > 1) {
        if (k.nphase != 4 || a->out_step != 2 || a->strip || fold_fused || a->w_phase_elems <= 0 ||
            (a->OH - 1) * 2 + 1 >= a->y.H || (a->OW - 1) * 2 + 1 >= a->y.W)
            SALT_FAIL(SALT_E_BADARG, "conv: a phase-fused launch is 4 output-parity phases of an out_step 2 grid that fits y for every parity");
        k.m_tiles *= 4;
    }
    pl->grid = dim3((unsigned)(cdiv(k.m_tiles, 8) * 8 * k.n_tiles), 1, 1);
    pl->lds = (size_t)k.a_bytes + (size_t)k.ntaps * BN * 64;
    if (cfg->KS > 0) {
        pl->lds = pl->lds * (size_t)k.nbuf + 8192;
        const size_t red = (size_t)cfg->WM * (cfg->MI * cfg->NI) * cfg->KS * 4096;       // K-slice reduction scratch
        if (red > pl->lds) pl->lds = red;
    }
    {   // the epilogue stages the BM x BN output tile through the same LDS
        const size_t es = a->dtype == SALT_F32 ? 4 : 2;
        size_t out_bytes = (size_t)(32 * cfg->MI * cfg->WM) * (BN * es + 16);
        // conv_mfma_kernel merges the BatchNorm statistics of its waves in a region behind the staged tile ([WM][BN][2] + [WM] floats + flag)
        if (cfg->KS == 0 && (a->stats || a->fin_acc)) out_bytes += 16 + (size_t)cfg->WM * BN * 8 + (size_t)cfg->WM * 4 + 32;
        if (out_bytes > pl->lds) pl->lds = out_bytes;
    }
    pl->parts = tiles_b * k.tiles_y * k.tiles_x * k.nphase;
    k.bnb_partials = a->bnb_partials; k.bnb_y = a->bnb_y.p; k.bnb_cs = a->bnb_y.cs; k.bnb_relu = a->bnb_relu;
    k.bnb_a = a->bnb_a.p; k.bnb_acs = a->bnb_a.cs;
    k.bnb_mean = a->bnb_mean; k.bnb_invstd = a->bnb_invstd; k.bnb_gamma = a->bnb_gamma; k.bnb_beta = a->bnb_beta;
    if (a->bnb_partials || a->bnb_acc) {
        const int ve = a->dtype == SALT_F32 ? 4 : 8;
        // round 6: a phase-fused stride-2 launch (nphase 4: every output parity of an even grid, i.e. every pixel of y exactly once) also
        // completes the gradient it writes - the data gradient of layerN.0.conv1 (architectures/encoders.py:38-45) carries the sums too
        const bool all_phases = k.nphase == 4 && a->out_step == 2 && !a->out_oy && !a->out_ox && a->OH * 2 == a->y.H && a->OW * 2 == a->y.W;
        if (a->strip || a->stats || a->fin_acc || ((a->out_step != 1 || a->out_oy || a->out_ox) && !all_phases) ||
            (!fold_fused && !all_phases && (a->OH != a->y.H || a->OW != a->y.W)))
            SALT_FAIL(SALT_E_BADARG, "conv: BatchNorm-backward sums need a plain (or fused-fold, or all-phases stride-2) full-grid launch");
        if (!view_ok(a->bnb_y) || a->bnb_y.B != a->y.B || a->bnb_y.H != a->y.H || a->bnb_y.W != a->y.W || a->bnb_y.C != a->y.C)
            SALT_FAIL(SALT_E_BADARG, "conv: bnb_y shape");
        if (!a->bnb_mean || !a->bnb_invstd || !a->bnb_gamma || !a->bnb_beta) SALT_FAIL(SALT_E_BADARG, "conv: bnb parameters missing");
        if (a->bnb_a.p && (!view_ok(a->bnb_a) || a->bnb_a.B != a->y.B || a->bnb_a.H != a->y.H || a->bnb_a.W != a->y.W || a->bnb_a.C != a->y.C ||
                           a->bnb_a.cs % ve || (reinterpret_cast<uintptr_t>(a->bnb_a.p) & 15)))
            SALT_FAIL(SALT_E_BADARG, "conv: bnb_a shape / alignment");
        if (Cout % ve || a->y.cs % ve || a->bnb_y.cs % ve || ((reinterpret_cast<uintptr_t>(a->y.p) | reinterpret_cast<uintptr_t>(a->bnb_y.p)) & 15))
            SALT_FAIL(SALT_E_BADARG, "conv: BatchNorm-backward sums need 16-byte aligned whole channel pieces");
        const size_t red_bytes = (size_t)((cfg->KS > 0 ? 512 : 256) / (BN / ve)) * BN * 2 * sizeof(float) + 16;
        if (red_bytes > pl->lds) pl->lds = red_bytes;
    }
    k.res = a->res.p; k.res_cs = a->res.cs;
    if (a->res.p) {
        const int ve = a->dtype == SALT_F32 ? 4 : 8;
        if (a->strip || fold_fused || a->stats || a->fin_acc || a->bnb_partials || a->bnb_acc || a->accumulate || a->out_step != 1 || a->out_oy || a->out_ox ||
            k.nphase > 1 || a->OH != a->y.H || a->OW != a->y.W)
            SALT_FAIL(SALT_E_BADARG, "conv: the residual epilogue needs a plain full-grid launch");
        if (!view_ok(a->res) || a->res.B != a->y.B || a->res.H != a->y.H || a->res.W != a->y.W || a->res.C != a->y.C)
            SALT_FAIL(SALT_E_BADARG, "conv: res shape");
        if (Cout % ve || a->y.cs % ve || a->res.cs % ve || ((reinterpret_cast<uintptr_t>(a->y.p) | reinterpret_cast<uintptr_t>(a->res.p)) & 15))
            SALT_FAIL(SALT_E_BADARG, "conv: the residual epilogue needs 16-byte aligned whole channel pieces");
    }
    k.fin.acc = nullptr; k.bnbf.acc = nullptr;
    k.in_scale = a->in_scale; k.in_shift = a->in_shift; k.in_relu = a->in_relu; k.in_tab_off = 0; k.in_fin = BnFin{};
    if (in_tf) {
        if (a->dtype != SALT_BF16 || a->ntaps != 9 || vt != 1 || cfg->KS > 0 || cfg->MI * cfg->WM == 2)
            SALT_FAIL(SALT_E_UNSUPPORTED, "conv: the input transform needs a 9-tap bf16 launch with 128- or 256-pixel tiles");
        if (a->in_fin_acc) {
            const salt_bn_finalize_args* f = static_cast<const salt_bn_finalize_args*>(a->in_fin);
            if (!f || f->C != a->x.C || !f->gamma || !f->beta || !f->mean || !f->invstd || !f->scale || !f->shift)
                SALT_FAIL(SALT_E_BADARG, "conv: in_fin_acc needs the complete salt_bn_finalize arguments of the producer");
            k.in_fin = BnFin{const_cast<double*>(a->in_fin_acc), nullptr, f->gamma, f->beta, f->running_mean, f->running_var, f->num_batches_tracked,
                             f->momentum, f->eps, f->mean, f->invstd, f->scale, f->shift};
        } else if (!a->in_shift) SALT_FAIL(SALT_E_BADARG, "conv: in_scale without in_shift");
        k.in_tab_off = (int)((pl->lds + 15) & ~(size_t)15);
        pl->lds = (size_t)k.in_tab_off + (size_t)2 * a->x.C * sizeof(float);
    }
    {
        auto small = [](const salt_view& v) { return !v.p || (int64_t)v.B * v.H * v.W * v.cs < (int64_t)1 << 31; };
        k.y_small = small(a->y) && small(a->bnb_y) && small(a->bnb_a) && small(a->res);
    }
    if (a->fin_acc) {
        if (a->stats || a->stats_part0 || a->strip) SALT_FAIL(SALT_E_BADARG, "conv: fin_acc excludes stats partials and fold mode");
        k.fin = BnFin{};
        k.fin.acc = a->fin_acc;
        if (a->fin_ticket) {                                    // in-launch finalize by the last arriver
            const salt_bn_finalize_args* f = static_cast<const salt_bn_finalize_args*>(a->fin);
            if (!f || f->C != Cout || !f->gamma || !f->beta || !f->mean || !f->invstd || !f->scale || !f->shift)
                SALT_FAIL(SALT_E_BADARG, "conv: in-launch BatchNorm finalize needs complete finalize arguments");
            k.fin = BnFin{a->fin_acc, a->fin_ticket, f->gamma, f->beta, f->running_mean, f->running_var, f->num_batches_tracked,
                          f->momentum, f->eps, f->mean, f->invstd, f->scale, f->shift};
        }
    }
    if (a->bnb_acc) {
        k.bnbf = BnbFin{};
        k.bnbf.acc = a->bnb_acc;
        if (a->bnb_ticket) {
            const salt_bn_bwd_args* f = static_cast<const salt_bn_bwd_args*>(a->bnb_fin);
            if (!f || !f->coef || f->y.C != Cout || (f->dgamma && !f->dbeta))
                SALT_FAIL(SALT_E_BADARG, "conv: in-launch BatchNorm-backward finalize needs the salt_bn_bwd arguments");
            k.bnbf = BnbFin{a->bnb_acc, a->bnb_ticket, f->dgamma, f->dbeta, f->coef, f->accumulate_param_grads,
                            (double)f->y.B * f->y.H * f->y.W};
        }
    }
    if (pl->lds > 160 * 1024) SALT_FAIL(SALT_E_LDS, "conv: needs %zu bytes of LDS", pl->lds);
    return SALT_OK;
}

template <typename T, int MI, int NI, int WM, int WN, int NT, int MAXA_T = 0, bool INA = false>
int launch_cfg_nt(const Plan& pl, hipStream_t st) {
    auto kern = conv_mfma_kernel<T, MI, NI, WM, WN, NT, MAXA_T, INA>;
    static bool attr_set = false;
    if (pl.lds > 64 * 1024 && !attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, pl.grid, dim3(256), pl.lds, st, pl.kp);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

template <typename T, int MI, int NI, int WM, int WN>
int launch_cfg(const Plan& pl, hipStream_t st) {
    if constexpr (sizeof(T) == 2 && MI * WM != 2) {                       // bf16, 128- and 256-pixel tiles: interleaved loader
        static const bool ilv_off = getenv("SALT_CONV_NO_ILV") != nullptr;
        const ConvKP& k = pl.kp;
        const int npa = (k.nb * k.hh * k.hw * k.vt * 4 + 255) >> 8;
        const bool ok = !ilv_off && k.x_cs % 8 == 0 && k.Cin % 8 == 0 && (reinterpret_cast<uintptr_t>(k.x) & 15) == 0;
        if (k.in_scale || k.in_fin.acc) {                                  // input transform: the interleaved 9-tap loader only
            if (!ok || k.ntaps != 9 || k.vt != 1) SALT_FAIL(SALT_E_UNSUPPORTED, "conv: the input transform needs a 9-tap bf16 launch over aligned 16-byte pieces");
            if constexpr (MI * WM == 8) { if (npa <= 6) return launch_cfg_nt<T, MI, NI, WM, WN, 9, 6, true>(pl, st); }
            else {
                if (npa <= 3) return launch_cfg_nt<T, MI, NI, WM, WN, 9, 3, true>(pl, st);
                return launch_cfg_nt<T, MI, NI, WM, WN, 9, 9, true>(pl, st);
            }
            SALT_FAIL(SALT_E_UNSUPPORTED, "conv: the input transform does not fit this tile's loader");
        }
        if (ok && k.ntaps == 9) {
            if constexpr (MI * WM == 8) { if (npa <= 6) return launch_cfg_nt<T, MI, NI, WM, WN, 9, 6>(pl, st); }
            else {
                if (npa <= 3) return launch_cfg_nt<T, MI, NI, WM, WN, 9, 3>(pl, st);
                return launch_cfg_nt<T, MI, NI, WM, WN, 9, 9>(pl, st);
            }
        }
        if constexpr (MI * WM != 8) { if (ok && k.ntaps == 4) return launch_cfg_nt<T, MI, NI, WM, WN, 4, 9>(pl, st); }
    }
    if (pl.kp.in_scale || pl.kp.in_fin.acc) SALT_FAIL(SALT_E_UNSUPPORTED, "conv: the input transform needs bf16 tiles of 128 or 256 pixels");
    if (pl.kp.ntaps == 9) return launch_cfg_nt<T, MI, NI, WM, WN, 9>(pl, st);
    if (pl.kp.ntaps == 4) return launch_cfg_nt<T, MI, NI, WM, WN, 4>(pl, st);      // 1x1 with 4 virtual taps, ConvT k4 output phases
    return launch_cfg_nt<T, MI, NI, WM, WN, 0>(pl, st);
}

template <int MI, int NI, int MB, int KS>
int launch_glds(const Plan& pl, hipStream_t st) {
    auto go = [&](auto kern, bool& attr_set) -> int {
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, pl.grid, dim3(512), pl.lds, st, pl.kp);
        SALT_CHECK_LAUNCH();
        return SALT_OK;
    };
    static bool set9 = false, set4 = false;                  // per template instantiation
    if (pl.kp.ntaps == 9) return go(conv_glds_kernel<MI, NI, MB, KS, 9>, set9);
    return go(conv_glds_kernel<MI, NI, MB, KS, 4>, set4);
}

template <typename T>
int launch_T(const Plan& pl, hipStream_t st) {
    if constexpr (sizeof(T) == 2) {
        switch (pl.cfg.id) {
            case 6: return launch_glds<2, 2, 4, 2>(pl, st);
            case 7: return launch_glds<2, 2, 2, 4>(pl, st);
            case 8: return launch_glds<2, 1, 2, 4>(pl, st);
        }
    }
    switch (pl.cfg.id) {
        case 1: return launch_cfg<T, 2, 1, 2, 2>(pl, st);
        case 2: return launch_cfg<T, 2, 2, 4, 1>(pl, st);
        case 3: return launch_cfg<T, 2, 2, 2, 2>(pl, st);
        case 4: return launch_cfg<T, 1, 1, 4, 1>(pl, st);
        case 5: return launch_cfg<T, 1, 1, 2, 2>(pl, st);
    }
    SALT_FAIL(SALT_E_BADARG, "conv: cfg");
}

// ------------------------------------------------------------------------------------------ weight packing
struct PackKP {
    const float* w; void* wp; int D0, D1, KH, KW, ntaps, transpose, N, C, nchunk;
    int n_off, n_total, chunk_off;                 // placement inside a wider packed tensor (salt_pack_conv_weight_args)
    int tap_kh[SALT_MAX_TAPS], tap_kw[SALT_MAX_TAPS];
};
template <typename T>
__global__ void pack_weight_kernel(PackKP p) {
    constexpr int KCE = 64 / (int)sizeof(T);
    const int64_t total = (int64_t)p.nchunk * p.ntaps * p.N * KCE;
    T* out = reinterpret_cast<T*>(p.wp);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int kc = (int)(i % KCE);
        int64_t r = i / KCE;
        const int n = (int)(r % p.N); r /= p.N;
        const int t = (int)(r % p.ntaps);
        const int ch = (int)(r / p.ntaps) * KCE + kc;
        float v = 0.f;
        if (ch < p.C && p.tap_kh[t] >= 0) {
            const int d0 = p.transpose ? ch : n, d1 = p.transpose ? n : ch;
            v = p.w[(((int64_t)d0 * p.D1 + d1) * p.KH + p.tap_kh[t]) * p.KW + p.tap_kw[t]];
        }
        Elem<T>::st(out + ((((int64_t)(r / p.ntaps) + p.chunk_off) * p.ntaps + t) * p.n_total + n + p.n_off) * KCE + kc, v);
    }
}

constexpr int PACK_EPB = 2048;          // packed elements per block in the batched pack kernel

// Vector path of the forward packs (bf16, KH * KW = 9 | 1 taps in raster order, D1 a multiple of 32, 16-byte aligned pointers - the 3x3
// and 1x1 layers of a ResNet, 97 % of its parameters): a thread owns 8 input channels of one output channel for ALL taps - 2 KK float4
// loads of 32 KK contiguous bytes, KK 16-byte stores; consecutive threads are consecutive (chunk, n, channel group) in the packed order,
// so a wave's stores of one tap are 1 KB contiguous.  The forward pack opens every training step on the main stream (the scalar path:
// 94 us for ResNet34's 21 M parameters, 2-byte stores).  Same values: the same fp32 -> bf16 conversion of the same elements.
// The transposed (data-gradient) packs take the same thread mapping with gathered scalar loads (8 rows of KK contiguous floats, lanes
// along the packed order): the scalar path's launch (16 062 workgroups, 2-byte stores, 141 us on the side stream) kept the stem
// convolution of the forward pass waiting for wave slots - 164 us in the step against 45 us alone.
__host__ __device__ inline bool pack_vec_ok(const salt_pack_conv_weight_args& a, int dtype) {
    const int kk = a.KH * a.KW;
    if (a.d1_cnt || a.n_off || a.n_total || a.chunk_off) return false;          // sub-block jobs: scalar paths
    if (dtype != SALT_BF16 || (kk != 9 && kk != 1) || a.ntaps != kk || (a.transpose ? a.D0 : a.D1) % 32) return false;
    if ((reinterpret_cast<uintptr_t>(a.wp) & 15) || (!a.transpose && (reinterpret_cast<uintptr_t>(a.w) & 15))) return false;
    for (int t = 0; t < kk; ++t) if (a.tap_kh[t] * a.KW + a.tap_kw[t] != t) return false;
    return true;
}
template <int KK>
__device__ __forceinline__ void pack_vec(const salt_pack_conv_weight_args& a, int64_t gi) {
    const int N = a.D0, C = a.D1;
    const int q = (int)(gi & 3);
    const int64_t r = gi >> 2;
    const int chunk = (int)(r / N), n = (int)(r - (int64_t)chunk * N);
    if (chunk >= C / 32) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(a.w + ((int64_t)n * C + chunk * 32 + q * 8) * KK);
    float v[8 * KK];
#pragma unroll
    for (int i = 0; i < 2 * KK; ++i) { const f32x4 x = src[i]; v[4 * i] = x.x; v[4 * i + 1] = x.y; v[4 * i + 2] = x.z; v[4 * i + 3] = x.w; }
    bf16_t* out = reinterpret_cast<bf16_t*>(a.wp);
#pragma unroll
    for (int t = 0; t < KK; ++t) {
        u32x4 o;
        o.x = f2bf_pk(v[0 * KK + t], v[1 * KK + t]); o.y = f2bf_pk(v[2 * KK + t], v[3 * KK + t]);
        o.z = f2bf_pk(v[4 * KK + t], v[5 * KK + t]); o.w = f2bf_pk(v[6 * KK + t], v[7 * KK + t]);
        *reinterpret_cast<u32x4*>(out + (((int64_t)chunk * KK + t) * N + n) * 32 + q * 8) = o;
    }
}

template <int KK>
__device__ __forceinline__ void pack_vec_t(const salt_pack_conv_weight_args& a, int64_t gi) {
    const int N = a.D1, C = a.D0;                                       // transposed: packed rows = input channels, chunks over output channels
    const int q = (int)(gi & 3);
    const int64_t r = gi >> 2;
    const int chunk = (int)(r / N), n = (int)(r - (int64_t)chunk * N);
    if (chunk >= C / 32) return;
    const float* src = a.w + ((int64_t)(chunk * 32 + q * 8) * N + n) * KK;
    float v[8 * KK];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int t = 0; t < KK; ++t) v[j * KK + t] = src[(int64_t)j * N * KK + t];
    bf16_t* out = reinterpret_cast<bf16_t*>(a.wp);
#pragma unroll
    for (int t = 0; t < KK; ++t) {
        u32x4 o;
        o.x = f2bf_pk(v[0 * KK + t], v[1 * KK + t]); o.y = f2bf_pk(v[2 * KK + t], v[3 * KK + t]);
        o.z = f2bf_pk(v[4 * KK + t], v[5 * KK + t]); o.w = f2bf_pk(v[6 * KK + t], v[7 * KK + t]);
        *reinterpret_cast<u32x4*>(out + (((int64_t)chunk * KK + t) * N + n) * 32 + q * 8) = o;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_batched_kernel(const salt_pack_conv_weight_args* jobs, const int* job_block0, int njobs) {
    constexpr int KCE = 64 / (int)sizeof(T);
    // binary search: job j with job_block0[j] <= blockIdx.x < job_block0[j+1]
    int lo = 0, hi = njobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (job_block0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const salt_pack_conv_weight_args& a = jobs[lo];
    const int S1 = a.d1_cnt > 0 ? a.d1_cnt : a.D1;                      // channels of the D1 axis this job packs (D1 stays the row stride)
    const int N = a.transpose ? S1 : a.D0, C = a.transpose ? a.D0 : S1;
    const int NT_ = a.n_total > 0 ? a.n_total : N;                      // rows of the packed tensor the job writes into
    const int nchunk = (C + KCE - 1) / KCE;
    const int64_t total = (int64_t)nchunk * a.ntaps * N * KCE;
    const int64_t base = (int64_t)(blockIdx.x - job_block0[lo]) * PACK_EPB;
    T* out = reinterpret_cast<T*>(a.wp);
    if constexpr (sizeof(T) == 2) {
        if (pack_vec_ok(a, a.dtype)) {
            const int64_t gi = (int64_t)(blockIdx.x - job_block0[lo]) * 256 + threadIdx.x;
            if (a.transpose) { if (a.KH * a.KW == 9) pack_vec_t<9>(a, gi); else pack_vec_t<1>(a, gi); }
            else if (a.KH * a.KW == 9) pack_vec<9>(a, gi); else pack_vec<1>(a, gi);
            return;
        }
    }
    if (!a.transpose) {
        // forward packs: thread = one (output channel n, input channel ch) pair, lanes along ch.  The KH*KW weights of a pair are
        // contiguous (neighbouring lanes read neighbouring 36-byte blocks: coalesced), and for every tap the lanes of a chunk write
        // one contiguous 64-byte packed row.
        const int Cp = nchunk * KCE;
        const int64_t pi = (int64_t)(blockIdx.x - job_block0[lo]) * 256 + threadIdx.x;
        if (pi >= (int64_t)N * Cp) return;
        const int n = (int)(pi / Cp), ch = (int)(pi - (int64_t)n * Cp);
        const int chunk = ch / KCE, kc = ch - chunk * KCE;
        const float* src = a.w + ((int64_t)n * a.D1 + ch) * a.KH * a.KW;
        for (int t = 0; t < a.ntaps; ++t) {
            const float v = (ch < C && a.tap_kh[t] >= 0) ? src[a.tap_kh[t] * a.KW + a.tap_kw[t]] : 0.f;      // tap_kh < 0: a zero tap
            Elem<T>::st(out + (((int64_t)(chunk + a.chunk_off) * a.ntaps + t) * NT_ + n + a.n_off) * KCE + kc, v);
        }
        return;
    }
    for (int e = threadIdx.x; e < PACK_EPB; e += 256) {
        const int64_t i = base + e;
        if (i >= total) break;
        const int kc = (int)(i % KCE);
        int64_t r = i / KCE;
        const int n = (int)(r % N); r /= N;
        const int t = (int)(r % a.ntaps);
        const int ch = (int)(r / a.ntaps) * KCE + kc;
        float v = 0.f;
        if (ch < C && a.tap_kh[t] >= 0) {
            const int d0 = a.transpose ? ch : n, d1 = a.transpose ? n : ch;
            v = a.w[(((int64_t)d0 * a.D1 + d1) * a.KH + a.tap_kh[t]) * a.KW + a.tap_kw[t]];
        }
        Elem<T>::st(out + ((((int64_t)(r / a.ntaps) + a.chunk_off) * a.ntaps + t) * NT_ + n + a.n_off) * KCE + kc, v);
    }
}

// ------------------------------------------------------------------------------------------ Adam + forward packs in one pass
// (saltnet.h: salt_adam_pack.)  Job blocks: pack_vec's thread mapping - a thread owns 8 input channels x KK taps of one output channel,
// 8 KK contiguous fp32 of the master in the flat buffer: it runs the Adam update over them as 2 KK float4 groups (the same arithmetic, in
// the same order, as adam_kernel in loss.hip: bit-identical parameters and moments) and stores the KK packed pieces from the updated
// values.  Rest blocks: the plain update over the ranges no job covers.
struct AdamPackKP {
    float* p; const float* g; float* m; float* v; const float* hyper;
    const salt_pack_conv_weight_args* jobs; const int* job_block0; int njobs, pack_blocks;
    const int64_t* rest; const int* rest_block0; int nrest;
};
// A job block = 64 segments; a segment = the 32 input channels x 9 taps of one output channel = 288 CONTIGUOUS floats of the master.
// Phase 1: the block's 4 608 float4 groups are updated in memory order (coalesced 1 152-byte runs - a thread per (output channel, 8 input
// channels) walking its own 288 bytes left every load of the four arrays at a 288-byte lane stride: 0.76 ms for the network instead of
// 0.13) and the new parameters are parked in LDS; phase 2: pack_vec's thread mapping reads them back and stores the packed pieces.
constexpr int AP_SEG = 64, AP_SEGF = 288;
__global__ __launch_bounds__(256) void adam_pack_kernel(AdamPackKP k) {
    extern __shared__ __attribute__((aligned(16))) float ap_sm[];              // [AP_SEG][AP_SEGF]
    const float lr = k.hyper[0], b1 = k.hyper[1], b2 = k.hyper[2], eps = k.hyper[3], wd = k.hyper[4], bc1 = k.hyper[5], bc2 = k.hyper[6], gs = k.hyper[7];
    const float step_size = lr / bc1, rs = 1.f / sqrtf(bc2);
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < k.pack_blocks) {
        int lo = 0, hi = k.njobs - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (k.job_block0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
        const salt_pack_conv_weight_args& a = k.jobs[lo];
        const int N = a.D0, C = a.D1, nseg = N * (C / 32);
        const int seg0 = ((int)blockIdx.x - k.job_block0[lo]) * AP_SEG;
        const int64_t wbase = a.w - k.p;
#pragma unroll 3
        for (int it = 0; it < AP_SEG * AP_SEGF / 4 / 256; ++it) {
            const int f = tid + 256 * it, sl = f / (AP_SEGF / 4), fi = f - sl * (AP_SEGF / 4), seg = seg0 + sl;
            if (seg < nseg) {
                const int chunk = seg / N, n = seg - chunk * N;
                const int64_t off = wbase + ((int64_t)n * C + chunk * 32) * 9 + 4 * fi;
                f32x4 pp = *reinterpret_cast<const f32x4*>(k.p + off);
                const f32x4 gg = *reinterpret_cast<const f32x4*>(k.g + off);
                f32x4 mm = *reinterpret_cast<const f32x4*>(k.m + off), vv = *reinterpret_cast<const f32x4*>(k.v + off);
                adam4(pp, gg, mm, vv, b1, b2, eps, wd, gs, step_size, rs);
                *reinterpret_cast<f32x4*>(k.p + off) = pp; *reinterpret_cast<f32x4*>(k.m + off) = mm; *reinterpret_cast<f32x4*>(k.v + off) = vv;
                *reinterpret_cast<f32x4*>(ap_sm + sl * AP_SEGF + 4 * fi) = pp;
            }
        }
        __syncthreads();
        const int sl = tid >> 2, q = tid & 3, seg = seg0 + sl;
        if (seg >= nseg) return;
        const int chunk = seg / N, n = seg - chunk * N;
        const float* v = ap_sm + sl * AP_SEGF + q * 72;                          // 8 channels x 9 taps of this thread
        bf16_t* out = reinterpret_cast<bf16_t*>(a.wp);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            u32x4 o;
            o.x = f2bf_pk(v[0 * 9 + t], v[1 * 9 + t]); o.y = f2bf_pk(v[2 * 9 + t], v[3 * 9 + t]);
            o.z = f2bf_pk(v[4 * 9 + t], v[5 * 9 + t]); o.w = f2bf_pk(v[6 * 9 + t], v[7 * 9 + t]);
            *reinterpret_cast<u32x4*>(out + (((int64_t)chunk * 9 + t) * N + n) * 32 + q * 8) = o;
        }
        return;
    }
    const int rb = (int)blockIdx.x - k.pack_blocks;
    int lo = 0, hi = k.nrest - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (k.rest_block0[mid] <= rb) lo = mid; else hi = mid - 1; }
    const int64_t first = k.rest[2 * lo], count = k.rest[2 * lo + 1];
    const int64_t e = (int64_t)(rb - k.rest_block0[lo]) * 1024 + tid * 4;
    if (e >= count) return;
    const int64_t off = first + e;
    f32x4 pp = *reinterpret_cast<const f32x4*>(k.p + off);
    const f32x4 gg = *reinterpret_cast<const f32x4*>(k.g + off);
    f32x4 mm = *reinterpret_cast<const f32x4*>(k.m + off), vv = *reinterpret_cast<const f32x4*>(k.v + off);
    adam4(pp, gg, mm, vv, b1, b2, eps, wd, gs, step_size, rs);
    *reinterpret_cast<f32x4*>(k.p + off) = pp; *reinterpret_cast<f32x4*>(k.m + off) = mm; *reinterpret_cast<f32x4*>(k.v + off) = vv;
}

// ------------------------------------------------------------------------------------------ weight gradient
// dW[t][a][b] = sum_p P[p,a] * Q[pad(p*q_step + tap_t), b].  Workgroup = 64(a) x 64(b) block for all taps of
// the launch over a slice of the pixel tiles (split-K); 4 waves as 2(a) x 2(b), each 32x32 per tap.
struct WgradKP {
    const void* P; const void* Q; float* partials;
    int B, PH, PW, Ca, p_cs;        // P view (output-grid pixels)
    int QH, QW, Cb, q_cs;
    int ntaps; int tap_off[SALT_MAX_TAPS];
    int q_step, pad_mode, min_dy, min_dx;
    int th_log2, tw_log2, nb, hh, hw;
    int tiles_y, tiles_x, ntiles, nsplit;
    int a_blocks, b_blocks;
    int bmp;                        // pixels per K tile (64 | 128)
    long long q_plane;              // salt_conv_wgrad_args.q_plane: b-block bb reads the dense plane Q + bb * q_plane (0: channel-interleaved rows)
    int atomic;                     // SALT_WGRAD_ATOMIC=1 (A/B): every split ADDS into slab 0 with global_atomic_add_f32 instead of writing its own slab
    int ksplit;                     // conv_wgrad_kernel<float>: 0 off; 4: Ca, Cb <= 32 - the four waves share block (0, 0) and a quarter of the k-steps each;
                                    // 2: Ca <= 32 - waves (wa, wb) work on block (0, wb), k-step pairs of parity wa; 3: Cb <= 32 - block (wa, 0), parity wb
};

template <typename T> struct WRow { static constexpr int BYTES = 64 * (int)sizeof(T); };
__device__ __forceinline__ void slab_put4(float* dst, const f32x4& v, bool atomic) {
    if (atomic) { unsafeAtomicAdd(dst, v.x); unsafeAtomicAdd(dst + 1, v.y); unsafeAtomicAdd(dst + 2, v.z); unsafeAtomicAdd(dst + 3, v.w); }
    else *reinterpret_cast<f32x4*>(dst) = v;
}   // 64 channels per pixel row

template <typename T, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradKP p) {
    constexpr int VE = Elem<T>::VE;
    constexpr int PPR = 64 / VE;                                // 16-byte pieces per pixel row
    constexpr int ROWB = (sizeof(T) == 2) ? 192 : 256;           // bf16 rows padded to 192 B: conflict-free tr reads
    constexpr bool PIPE = sizeof(T) == 2;                        // register-prefetch pipeline (fits the VGPR budget for bf16)
    constexpr int MAXP = 128 * PPR / 256, MAXQ = 10;
    const int BMP = p.bmp;                                       // pixels per K tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sP = smem;
    unsigned char* sQ = smem + BMP * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int blk = blockIdx.x;
    const int ab = blk % p.a_blocks;
    const int bb = (blk / p.a_blocks) % p.b_blocks;
    const int split = blk / (p.a_blocks * p.b_blocks);
    const int a0 = ab * 64, c0 = bb * 64;
    const int hhw = p.hh * p.hw, phalo = p.nb * hhw;
    const T* Pg = reinterpret_cast<const T*>(p.P);
    const T* Qg = reinterpret_cast<const T*>(p.Q) + (p.q_plane ? (long long)bb * p.q_plane - c0 : 0);   // planar Q: channel c0 + j of block bb is element j of plane bb
    const bool p_vec = ((p.p_cs % VE) == 0) && ((reinterpret_cast<uintptr_t>(p.P) & 15) == 0);
    const bool q_vec = ((p.q_cs % VE) == 0) && ((reinterpret_cast<uintptr_t>(p.Q) & 15) == 0);

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- tile walk without divisions: tile = (tbi, tyi, txi), advanced by nsplit with carries
    struct TileC { int tbi, tyi, txi; };
    auto decode = [&](int tile) { TileC c; c.txi = tile % p.tiles_x; const int tt = tile / p.tiles_x; c.tyi = tt % p.tiles_y; c.tbi = tt / p.tiles_y; return c; };
    const TileC tstep = decode(p.nsplit);
    auto advance = [&](TileC& c) {
        c.txi += tstep.txi; if (c.txi >= p.tiles_x) { c.txi -= p.tiles_x; ++c.tyi; }
        c.tyi += tstep.tyi; if (c.tyi >= p.tiles_y) { c.tyi -= p.tiles_y; ++c.tbi; }
        c.tbi += tstep.tbi;
    };
    // ---- tile-invariant piece descriptors.  Piece q = tid + 256 k covers 16 bytes of pixel row q / PPR; 256 % PPR == 0, so the
    // channel piece (tid % PPR) is the same for every k and only the pixel coordinates (packed bl:ty:tx / bl:hy:hx) differ.
    const int pcx = tid % PPR;
    const int chP = a0 + pcx * VE, chQ = c0 + pcx * VE;
    const int np = BMP * PPR, nq = phalo * PPR;
    auto descP = [&](int q) -> unsigned {
        if (q >= np || chP >= p.Ca) return 0xffffffffu;
        const int m = q / PPR;
        const int tx = m & ((1 << p.tw_log2) - 1);
        const int ty = (m >> p.tw_log2) & ((1 << p.th_log2) - 1);
        const int bl = m >> (p.tw_log2 + p.th_log2);
        return (unsigned)((bl << 16) | (ty << 8) | tx);
    };
    auto descQ = [&](int q) -> unsigned {
        if (q >= nq || chQ >= p.Cb) return 0xffffffffu;
        const int pix = q / PPR;
        const int bl = pix / hhw;
        const int r = pix - bl * hhw;
        const int hy = r / p.hw, hx = r - hy * p.hw;
        return (unsigned)((bl << 16) | (hy << 8) | hx);
    };
    auto fetchP = [&](const TileC& c, unsigned d) -> u32x4 {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (d == 0xffffffffu) return v;
        const int oy = (c.tyi << p.th_log2) + (int)((d >> 8) & 255u), ox = (c.txi << p.tw_log2) + (int)(d & 255u), b = c.tbi * p.nb + (int)(d >> 16);
        if (b < p.B && oy < p.PH && ox < p.PW)
            v = load_piece<T>(Pg, (((int64_t)b * p.PH + oy) * p.PW + ox) * p.p_cs + chP, chP, p.Ca, p_vec);
        return v;
    };
    auto fetchQ = [&](const TileC& c, unsigned d) -> u32x4 {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (d == 0xffffffffu) return v;
        int iy = (c.tyi << p.th_log2) * p.q_step + p.min_dy + (int)((d >> 8) & 255u);
        int ix = (c.txi << p.tw_log2) * p.q_step + p.min_dx + (int)(d & 255u);
        const int b = c.tbi * p.nb + (int)(d >> 16);
        bool valid = b < p.B;
        if (p.pad_mode) { iy = min(max(iy, 0), p.QH - 1); ix = min(max(ix, 0), p.QW - 1); }
        else valid = valid && iy >= 0 && iy < p.QH && ix >= 0 && ix < p.QW;
        if (valid) v = load_piece<T>(Qg, (((int64_t)b * p.QH + iy) * p.QW + ix) * p.q_cs + chQ, chQ, p.Cb, q_vec);
        return v;
    };
    auto lds_p = [&](int q) -> u32x4* { return reinterpret_cast<u32x4*>(sP + (q / PPR) * ROWB + pcx * 16); };
    auto lds_q = [&](int q) -> u32x4* { return reinterpret_cast<u32x4*>(sQ + (q / PPR) * ROWB + pcx * 16); };

    int tob[NT];                                                // tap offsets in LDS bytes (uniform)
#pragma unroll
    for (int t = 0; t < NT; ++t) tob[t] = p.tap_off[t] * ROWB;  // the plan repeats tap 0 in the slots ntaps..NT-1 (never stored)

    auto compute_tile = [&]() {
        if constexpr (sizeof(T) == 4) {
            // f32: v_mfma_f32_32x32x2_f32, A[i=a][k=pixel], B[k=pixel][j=b]; one dword per lane per operand.
            const int khalf = lane >> 5, l31 = lane & 31;
            // K-split (round 4): a layer with <= 32 output or input channels fills a quarter / half of the 64 x 64 block and the exact-f32
            // MFMA takes 64 cycles per 2 pixels - the waves of the empty quadrants take a share of the k-steps of the real one instead
            const int ks = p.ksplit;
            const int wa_e = (ks == 4 || ks == 2) ? 0 : wa, wb_e = (ks == 4 || ks == 3) ? 0 : wb;
            const int kmod = ks == 4 ? 3 : (ks ? 1 : 0), kidx = ks == 4 ? wave : (ks == 2 ? wa : (ks == 3 ? wb : 0));
            const int aoff = (wa_e * 32 + l31) * 4, boff = (wb_e * 32 + l31) * 4;
            for (int k0 = 0; k0 < BMP; k0 += 2) {
                if (((k0 >> 1) & kmod) != kidx) continue;
                const int m = k0 + khalf;
                const int tx = m & ((1 << p.tw_log2) - 1);
                const int ty = (m >> p.tw_log2) & ((1 << p.th_log2) - 1);
                const int bl = m >> (p.tw_log2 + p.th_log2);
                const int qb = (bl * hhw + ty * p.q_step * p.hw + tx * p.q_step) * ROWB + boff;
                const float av = *reinterpret_cast<const float*>(sP + m * ROWB + aoff);
                float bv[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) bv[t] = *reinterpret_cast<const float*>(sQ + qb + tob[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
            }
        } else {
            // bf16: v_mfma_f32_32x32x16_bf16 needs 8 k(=pixel)-consecutive values per lane while LDS rows are
            // channel-contiguous: ds_read_b64_tr_b16 transposes a [4 pixel][16 channel] block per 16-lane group.
            // All 2 + 2 NT transposed reads of a k-step are issued before the first MFMA (no control flow in between).
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
            const int khalf = lane >> 5;
            const int g16 = (lane >> 4) & 1;                     // which 16-row half of the 32-row operand
            const int i16 = lane & 15;
            const int prow = i16 >> 2;                           // pixel within the k-quad this lane fetches
            const int pcol = (i16 & 3) * 4;                      // channel offset of its 4 contiguous elements
            const int acol = (wa * 32 + g16 * 16 + pcol) * 2;
            const int bcol = (wb * 32 + g16 * 16 + pcol) * 2;
            for (int k0 = 0; k0 < BMP; k0 += 16) {
                int pa[2], qb[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int m = k0 + khalf * 8 + h * 4 + prow;
                    const int tx = m & ((1 << p.tw_log2) - 1);
                    const int ty = (m >> p.tw_log2) & ((1 << p.th_log2) - 1);
                    const int bl = m >> (p.tw_log2 + p.th_log2);
                    pa[h] = m * ROWB + acol;
                    qb[h] = (bl * hhw + ty * p.q_step * p.hw + tx * p.q_step) * ROWB + bcol;
                }
                const s16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sP + pa[0]));
                const s16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sP + pa[1]));
                s16x4 blo[NT], bhi[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    blo[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sQ + qb[0] + tob[t]));
                    bhi[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sQ + qb[1] + tob[t]));
                }
                const s16x8 av = {alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const s16x8 bv = {blo[t][0], blo[t][1], blo[t][2], blo[t][3], bhi[t][0], bhi[t][1], bhi[t][2], bhi[t][3]};
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[t], 0, 0, 0);
                }
            }
        }
    };

    if (PIPE && nq <= MAXQ * 256) {
        // software pipeline: next tile's global loads fly while the matrix cores work on the current one.
        // Per piece only tile-invariant lane constants are kept (packed coordinates for the bounds test, a 32-bit element offset
        // relative to the tile); per tile the origin is a handful of uniform values - no per-piece 64-bit index arithmetic.
        unsigned dp[MAXP], dq[MAXQ];
        int offP[MAXP], offQ[MAXQ];
        const int rowP = p.PW * p.p_cs, rowQ = p.QW * p.q_cs;
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            dp[k] = descP(tid + (k << 8));
            const int bl = (int)(dp[k] >> 16), ty = (int)((dp[k] >> 8) & 255u), tx = (int)(dp[k] & 255u);
            offP[k] = (bl * p.PH + ty) * rowP + tx * p.p_cs + chP;
        }
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) {
            dq[k] = descQ(tid + (k << 8));
            offQ[k] = (int)(dq[k] >> 16) * p.QH * rowQ + chQ;                 // image part only: rows / columns may be clamped
        }
        u32x4 rp[MAXP], rq[MAXQ];
        auto load_tile = [&](const TileC& c) {
            const int limB = p.B - c.tbi * p.nb, limY = p.PH - (c.tyi << p.th_log2), limX = p.PW - (c.txi << p.tw_log2);
            const T* baseP = Pg + (((int64_t)c.tbi * p.nb * p.PH + (c.tyi << p.th_log2)) * p.PW + (c.txi << p.tw_log2)) * p.p_cs;
            const T* baseQ = Qg + (int64_t)c.tbi * p.nb * p.QH * rowQ;
            const int iy0 = (c.tyi << p.th_log2) * p.q_step + p.min_dy, ix0 = (c.txi << p.tw_log2) * p.q_step + p.min_dx;
#pragma unroll
            for (int k = 0; k < MAXP; ++k) {
                rp[k] = u32x4{0u, 0u, 0u, 0u};
                const unsigned d = dp[k];
                if (d != 0xffffffffu && (int)(d >> 16) < limB && (int)((d >> 8) & 255u) < limY && (int)(d & 255u) < limX)
                    rp[k] = load_piece<T>(baseP, offP[k], chP, p.Ca, p_vec);
            }
#pragma unroll
            for (int k = 0; k < MAXQ; ++k) {
                rq[k] = u32x4{0u, 0u, 0u, 0u};
                const unsigned d = dq[k];
                if (d == 0xffffffffu) continue;
                int iy = iy0 + (int)((d >> 8) & 255u), ix = ix0 + (int)(d & 255u);
                bool valid = (int)(d >> 16) < limB;
                if (p.pad_mode) { iy = min(max(iy, 0), p.QH - 1); ix = min(max(ix, 0), p.QW - 1); }
                else valid = valid && (unsigned)iy < (unsigned)p.QH && (unsigned)ix < (unsigned)p.QW;
                if (valid) rq[k] = load_piece<T>(baseQ, offQ[k] + iy * rowQ + ix * p.q_cs, chQ, p.Cb, q_vec);
            }
        };
        TileC cur = decode(split);
        if (split < p.ntiles) load_tile(cur);
        for (int tile = split; tile < p.ntiles; tile += p.nsplit) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < MAXP; ++k) { const int q = tid + (k << 8); if (q < np) *lds_p(q) = rp[k]; }
#pragma unroll
            for (int k = 0; k < MAXQ; ++k) { const int q = tid + (k << 8); if (q < nq) *lds_q(q) = rq[k]; }
            __syncthreads();
            advance(cur);
            if (tile + p.nsplit < p.ntiles) load_tile(cur);
            compute_tile();
        }
    } else {
        TileC cur = decode(split);
        for (int tile = split; tile < p.ntiles; tile += p.nsplit) {
            __syncthreads();
            for (int q = tid; q < np; q += 256) *lds_p(q) = fetchP(cur, descP(q));
            for (int q = tid; q < nq; q += 256) *lds_q(q) = fetchQ(cur, descQ(q));
            __syncthreads();
            compute_tile();
            advance(cur);
        }
    }
    // K-split (fp32): the waves that shared a channel block meet in LDS tap by tap, fixed order (leader + partner 1 [+ 2 + 3])
    int wa_o = wa, wb_o = wb;
    bool leader = true;
    if (sizeof(T) == 4 && p.ksplit) {
        const int ks = p.ksplit;
        const int step = ks == 4 ? 1 : (ks == 2 ? 2 : 1), cnt = ks == 4 ? 4 : 2;          // partners of leader w: w + step, w + 2 step, ...
        leader = ks == 4 ? wave == 0 : (ks == 2 ? wa == 0 : wb == 0);
        wa_o = (ks == 4 || ks == 2) ? 0 : wa; wb_o = (ks == 4 || ks == 3) ? 0 : wb;
        float* red = reinterpret_cast<float*>(smem);                                       // [wave][16][64] floats: 4 KB per wave
        __syncthreads();                                                                   // every wave is done with the tile buffers
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (!leader) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[t][r];
            }
            __syncthreads();
            if (leader) {
                for (int q = 1; q < cnt; ++q) {
                    const int w2 = wave + q * step;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += red[(w2 * 16 + r) * 64 + lane];
                }
            }
            __syncthreads();
        }
    }
    // write the partial slab: partials[split][t][a][b]
    const int khalf = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < p.ntaps && leader) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int a = a0 + wa_o * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
                const int b = c0 + wb_o * 32 + l31;
                if (a < p.Ca && b < p.Cb) {
                    float* d = p.partials + (((int64_t)(p.atomic ? 0 : split) * p.ntaps + t) * p.Ca + a) * p.Cb + b;
                    if (p.atomic) unsafeAtomicAdd(d, acc[t][r]); else *d = acc[t][r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ weight gradient, fast path
// Same decomposition and partial-slab format as conv_wgrad_kernel, for the shapes that dominate training: bf16, 3x3 (NT taps in
// one launch), KS*16-pixel K tiles, 16-byte aligned views whose channel counts are multiples of 8, halo of at most MAXQ*256
// pieces (the host checks all of it).  A weight-gradient workgroup is alone on its CU (144 accumulator registers per lane), so
// nothing hides latency unless the instruction stream does it itself:
//   * the k-step loop is fully unrolled and software pipelined: the transposed LDS reads of k-step j+1 and the global loads of
//     the NEXT pixel tile are issued between the MFMAs of k-step j (sched_group_barrier pins the interleave);
//   * the loader is branch free: a masked-out piece loads 16 zero bytes from g_zero_piece instead of zero-initialising its
//     destination registers (which made the compiler wait for the loads already in flight) - per piece a few VALU instructions;
//   * the MFMA operands are swapped (D^T): a lane then owns 4 consecutive b-channels of one a-row, the slab is written with
//     16-byte stores instead of 4-byte ones.

// ROW16: every k-step is one 16-pixel tile row (tw = 16, th = KS, one image per tile, 18-pixel halo rows, unit step): the halo
// address of a fragment read is a per-tap lane constant plus a compile-time multiple of the halo row - no address arithmetic.
template <int NT, int KS, bool PAD, bool ROW16>
__global__ __launch_bounds__(256) void conv_wgrad_fast_kernel(WgradKP p) {
    static_assert(!ROW16 || NT == 9, "ROW16 shares pixel runs between the three taps of a 3x3 kernel row");
    typedef bf16_t T;
    constexpr int PPR = 8, ROWB = 192, BMP = KS * 16;
    constexpr int MAXP = BMP * PPR / 256, MAXQ = 10, NPIECE = MAXP + MAXQ;
    constexpr int LSTEPS = KS / 2;                               // the next tile's loads are issued during the first k-steps
    constexpr int LPS = (NPIECE + LSTEPS - 1) / LSTEPS;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sP = smem;
    unsigned char* sQ = smem + BMP * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wa = wave >> 1, wb = wave & 1;
    const int blk = blockIdx.x;
    const int ab = blk % p.a_blocks;
    const int bb = (blk / p.a_blocks) % p.b_blocks;
    const int split = blk / (p.a_blocks * p.b_blocks);
    const int a0 = ab * 64, c0 = bb * 64;
    const int hhw = p.hh * p.hw, phalo = p.nb * hhw;
    const T* Pg = reinterpret_cast<const T*>(p.P);
    const T* Qg = reinterpret_cast<const T*>(p.Q) + (p.q_plane ? (long long)bb * p.q_plane - c0 : 0);   // planar Q: channel c0 + j of block bb is element j of plane bb
    const T* zp = reinterpret_cast<const T*>(g_zero_piece);

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- tile walk without divisions (see conv_wgrad_kernel)
    struct TileC { int tbi, tyi, txi; };
    auto decode = [&](int tile) { TileC c; c.txi = tile % p.tiles_x; const int tt = tile / p.tiles_x; c.tyi = tt % p.tiles_y; c.tbi = tt / p.tiles_y; return c; };
    const TileC tstep = decode(p.nsplit);
    auto advance = [&](TileC& c) {
        c.txi += tstep.txi; if (c.txi >= p.tiles_x) { c.txi -= p.tiles_x; ++c.tyi; }
        c.tyi += tstep.tyi; if (c.tyi >= p.tiles_y) { c.tyi -= p.tiles_y; ++c.tbi; }
        c.tbi += tstep.tbi;
    };

    // ---- tile-invariant piece constants.  Piece q = tid + 256 k is 16 bytes (8 channels) of pixel row q / 8.
    const int pcx = tid % PPR;
    const int chP = a0 + pcx * 8, chQ = c0 + pcx * 8;
    const int np = BMP * PPR, nq = phalo * PPR;
    const int rowP = p.PW * p.p_cs, rowQ = p.QW * p.q_cs;
    int offP[MAXP], pby[MAXP], pxx[MAXP];                          // element offset in the tile, (image << 8 | row), column
    int offQ[MAXQ], hyq[MAXQ], hxq[MAXQ], blq[MAXQ];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int q = tid + (k << 8), m = q / PPR;
        const int tx = m & ((1 << p.tw_log2) - 1), ty = (m >> p.tw_log2) & ((1 << p.th_log2) - 1), bl = m >> (p.tw_log2 + p.th_log2);
        const bool ok = q < np && chP < p.Ca;
        offP[k] = (bl * p.PH + ty) * rowP + tx * p.p_cs + chP;
        pby[k] = ok ? bl : (1 << 20); pxx[k] = tx | (ty << 16);
    }
#pragma unroll
    for (int k = 0; k < MAXQ; ++k) {
        const int q = tid + (k << 8), pix = q / PPR;
        const int bl = pix / hhw, r = pix - bl * hhw, hy = r / p.hw, hx = r - hy * p.hw;
        const bool ok = q < nq && chQ < p.Cb;
        blq[k] = ok ? bl : (1 << 20);                                // fails "image < limB" for every tile
        hyq[k] = hy; hxq[k] = hx;
        offQ[k] = bl * p.QH * rowQ + chQ + (PAD ? 0 : hy * rowQ + hx * p.q_cs);
    }
    u32x4 rp[MAXP], rq[MAXQ];
    struct TileCtx { const T* baseP; const T* baseQ; int iy0, ix0, limB, limY, limX; };
    auto make_ctx = [&](const TileC& c, bool live) {
        TileCtx x;
        x.limB = live ? p.B - c.tbi * p.nb : 0;                      // no next tile: every piece is masked out
        x.limY = p.PH - (c.tyi << p.th_log2); x.limX = p.PW - (c.txi << p.tw_log2);
        x.baseP = Pg + (((int64_t)c.tbi * p.nb * p.PH + (c.tyi << p.th_log2)) * p.PW + (c.txi << p.tw_log2)) * p.p_cs;
        x.iy0 = (c.tyi << p.th_log2) * p.q_step + p.min_dy; x.ix0 = (c.txi << p.tw_log2) * p.q_step + p.min_dx;
        x.baseQ = Qg + (int64_t)c.tbi * p.nb * p.QH * rowQ;
        if (!PAD) x.baseQ += (int64_t)x.iy0 * rowQ + (int64_t)x.ix0 * p.q_cs;     // may point in front of the image: in-range pieces only
        return x;
    };
    auto issue_piece = [&](int K, const TileCtx& x) {                // K is a constant after unrolling
        if (K < MAXP) {
            const int k = K;
            const bool ok = pby[k] < x.limB && (pxx[k] >> 16) < x.limY && (pxx[k] & 0xffff) < x.limX;
            rp[k] = *reinterpret_cast<const u32x4*>(ok ? x.baseP + offP[k] : zp);
        } else {
            const int k = K - MAXP;
            if (!PAD) {
                const bool ok = (unsigned)(x.iy0 + hyq[k]) < (unsigned)p.QH && (unsigned)(x.ix0 + hxq[k]) < (unsigned)p.QW && blq[k] < x.limB;
                rq[k] = *reinterpret_cast<const u32x4*>(ok ? x.baseQ + offQ[k] : zp);
            } else {
                const int iy = min(max(x.iy0 + hyq[k], 0), p.QH - 1), ix = min(max(x.ix0 + hxq[k], 0), p.QW - 1);
                rq[k] = *reinterpret_cast<const u32x4*>(blq[k] < x.limB ? x.baseQ + (offQ[k] + iy * rowQ + ix * p.q_cs) : zp);
            }
        }
    };

    // ---- fragment addressing: ds_read_b64_tr_b16 transposes a [4 pixel][16 channel] block per 16-lane group (conv_wgrad_kernel)
    const int khalf = lane >> 5, g16 = (lane >> 4) & 1, i16 = lane & 15, prow = i16 >> 2, pcol = (i16 & 3) * 4;
    const int pa_base = (khalf * 8 + prow) * ROWB + (wa * 32 + g16 * 16 + pcol) * 2;     // + (16 j + 4 h) rows: an immediate
    int qaddr[ROW16 ? 2 : KS * 2];                                    // halo-row byte address of pixel 16 j + 8 khalf + 4 h + prow
#pragma unroll
    for (int i = 0; i < (ROW16 ? 2 : KS * 2); ++i) {
        const int m = (i >> 1) * 16 + khalf * 8 + (i & 1) * 4 + prow;
        const int tx = m & ((1 << p.tw_log2) - 1), ty = (m >> p.tw_log2) & ((1 << p.th_log2) - 1), bl = m >> (p.tw_log2 + p.th_log2);
        qaddr[i] = (bl * hhw + ty * p.q_step * p.hw + tx * p.q_step) * ROWB + (wb * 32 + g16 * 16 + pcol) * 2;
    }
    int tob[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) tob[t] = p.tap_off[t] * ROWB;
    // ROW16: the three taps of a kernel row read overlapping pixel runs, so one run of 12 pixels (three transposed quads per
    // lane) serves all three: tap dx is the run shifted by dx pixels = dx 16-bit elements (dx = 1: four v_alignbit, dx = 2: a
    // register offset).  11 LDS reads per k-step instead of 20 - the LDS return path, not the MFMA, bounds this kernel.
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    struct Frag { s16x4 alo, ahi, blo[ROW16 ? 1 : NT], bhi[ROW16 ? 1 : NT]; u32x2 q[ROW16 ? 3 : 1][3]; };
    int tq[3];
    if (ROW16) {
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) tq[dy] = (khalf * 8 + prow) * ROWB + (wb * 32 + g16 * 16 + pcol) * 2 + dy * (18 * ROWB);
    }
    auto read_frags = [&](int j, Frag& fr) {
        fr.alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sP + pa_base + (j * 16) * ROWB));
        fr.ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sP + pa_base + (j * 16 + 4) * ROWB));
        if constexpr (ROW16) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    fr.q[dy][u] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sQ + tq[dy] + (j * 18 + u * 4) * ROWB)));
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                fr.blo[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sQ + qaddr[2 * j] + tob[t]));
                fr.bhi[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sQ + qaddr[2 * j + 1] + tob[t]));
            }
        }
    };
    auto b_operand = [&](const Frag& fr, int t) -> bf16x8 {
        if constexpr (ROW16) {
            const int dy = t / 3, dx = t - dy * 3;
            const unsigned w0 = fr.q[dy][0].x, w1 = fr.q[dy][0].y, w2 = fr.q[dy][1].x, w3 = fr.q[dy][1].y, w4 = fr.q[dy][2].x;
            u32x4 v;
            if (dx == 0) v = u32x4{w0, w1, w2, w3};
            else if (dx == 1) v = u32x4{__builtin_amdgcn_alignbit(w1, w0, 16), __builtin_amdgcn_alignbit(w2, w1, 16),
                                        __builtin_amdgcn_alignbit(w3, w2, 16), __builtin_amdgcn_alignbit(w4, w3, 16)};
            else v = u32x4{w1, w2, w3, w4};
            return __builtin_bit_cast(bf16x8, v);
        } else {
            const s16x8 bv = {fr.blo[t][0], fr.blo[t][1], fr.blo[t][2], fr.blo[t][3], fr.bhi[t][0], fr.bhi[t][1], fr.bhi[t][2], fr.bhi[t][3]};
            return __builtin_bit_cast(bf16x8, bv);
        }
    };

    TileC cur = decode(split);
    {
        const TileCtx x = make_ctx(cur, split < p.ntiles);
#pragma unroll
        for (int K = 0; K < NPIECE; ++K) issue_piece(K, x);
    }
    Frag f0, f1;
    for (int tile = split; tile < p.ntiles; tile += p.nsplit) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXP; ++k) { const int q = tid + (k << 8); if (q < np) *reinterpret_cast<u32x4*>(sP + (q / PPR) * ROWB + pcx * 16) = rp[k]; }
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) { const int q = tid + (k << 8); if (q < nq) *reinterpret_cast<u32x4*>(sQ + (q / PPR) * ROWB + pcx * 16) = rq[k]; }
        __syncthreads();
        advance(cur);
        const TileCtx x = make_ctx(cur, tile + p.nsplit < p.ntiles);
        read_frags(0, f0);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            Frag& fc = (j & 1) ? f1 : f0;
            Frag& fn = (j & 1) ? f0 : f1;
            if (j + 1 < KS) read_frags(j + 1, fn);
            int nld = 0;
            if (j < LSTEPS) {
#pragma unroll
                for (int u = 0; u < LPS; ++u)
                    if (j * LPS + u < NPIECE) { issue_piece(j * LPS + u, x); ++nld; }
            }
            const s16x8 av = {fc.alo[0], fc.alo[1], fc.alo[2], fc.alo[3], fc.ahi[0], fc.ahi[1], fc.ahi[2], fc.ahi[3]};
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 bv = b_operand(fc, t);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, __builtin_bit_cast(bf16x8, av), acc[t], 0, 0, 0);
            }
            if (j + 1 < KS) {                                        // pin: 2 LDS reads (+ a global load) behind every MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, ROW16 ? 1 : 2, 0);
                    if (t < nld) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
        }
    }
    // ---- partial slab partials[split][t][a][b]: lane = a-row, 4 consecutive registers = 4 consecutive b
    const int l31 = lane & 31;
    const int a = a0 + wa * 32 + l31;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < p.ntaps && a < p.Ca) {
            float* row = p.partials + (((int64_t)(p.atomic ? 0 : split) * p.ntaps + t) * p.Ca + a) * p.Cb;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int b = c0 + wb * 32 + 8 * g + 4 * khalf;
                if (b < p.Cb) slab_put4(row + b, f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]}, p.atomic != 0);
            }
        }
    }
}

// 8-wave variant of the ROW16 fast path: the same workgroup tile and LDS layout, but two waves per SIMD - waves 0-3 accumulate
// taps 0-4, waves 4-7 taps 5-8 (80 / 64 accumulator registers per lane instead of 144) - so that one wave's VMEM / LDS issue
// stalls (a global_load_dwordx4 costs ~115 cycles of issue time) overlap with the other wave's MFMAs.  Each half reads the pixel
// runs of the two kernel rows its taps touch (8 LDS reads per k-step) and moves half of the tile pieces.
template <bool PAD>
__global__ __launch_bounds__(512) void conv_wgrad_fast8_kernel(WgradKP p) {
    typedef bf16_t T;
    constexpr int KS = 8, NTH = 512, PPR = 8, ROWB = 192, BMP = KS * 16, NTA = 5;
    constexpr int MAXP = BMP * PPR / NTH, MAXQ = 5, NPIECE = MAXP + MAXQ;
    constexpr int LSTEPS = KS / 2, LPS = (NPIECE + LSTEPS - 1) / LSTEPS;
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sP = smem;
    unsigned char* sQ = smem + BMP * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wa = (wave >> 1) & 1, wb = wave & 1;
    const int blk = blockIdx.x;
    const int ab = blk % p.a_blocks;
    const int bb = (blk / p.a_blocks) % p.b_blocks;
    const int split = blk / (p.a_blocks * p.b_blocks);
    const int a0 = ab * 64, c0 = bb * 64;
    const int hhw = p.hh * p.hw, phalo = p.nb * hhw;
    const T* Pg = reinterpret_cast<const T*>(p.P);
    const T* Qg = reinterpret_cast<const T*>(p.Q) + (p.q_plane ? (long long)bb * p.q_plane - c0 : 0);   // planar Q: channel c0 + j of block bb is element j of plane bb
    const T* zp = reinterpret_cast<const T*>(g_zero_piece);

    f32x16 acc[NTA];
#pragma unroll
    for (int t = 0; t < NTA; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    struct TileC { int tbi, tyi, txi; };
    auto decode = [&](int tile) { TileC c; c.txi = tile % p.tiles_x; const int tt = tile / p.tiles_x; c.tyi = tt % p.tiles_y; c.tbi = tt / p.tiles_y; return c; };
    const TileC tstep = decode(p.nsplit);
    auto advance = [&](TileC& c) {
        c.txi += tstep.txi; if (c.txi >= p.tiles_x) { c.txi -= p.tiles_x; ++c.tyi; }
        c.tyi += tstep.tyi; if (c.tyi >= p.tiles_y) { c.tyi -= p.tiles_y; ++c.tbi; }
        c.tbi += tstep.tbi;
    };

    // ---- tile-invariant piece constants.  Piece q = tid + 512 k is 16 bytes (8 channels) of pixel row q / 8.
    const int pcx = tid % PPR;
    const int chP = a0 + pcx * 8, chQ = c0 + pcx * 8;
    const int np = BMP * PPR, nq = phalo * PPR;
    const int rowP = p.PW * p.p_cs, rowQ = p.QW * p.q_cs;
    int offP[MAXP], pby[MAXP], pxx[MAXP];
    int offQ[MAXQ], hyq[MAXQ], hxq[MAXQ], blq[MAXQ];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int q = tid + k * NTH, m = q / PPR;
        const int tx = m & ((1 << p.tw_log2) - 1), ty = (m >> p.tw_log2) & ((1 << p.th_log2) - 1), bl = m >> (p.tw_log2 + p.th_log2);
        const bool ok = q < np && chP < p.Ca;
        offP[k] = (bl * p.PH + ty) * rowP + tx * p.p_cs + chP;
        pby[k] = ok ? bl : (1 << 20); pxx[k] = tx | (ty << 16);
    }
#pragma unroll
    for (int k = 0; k < MAXQ; ++k) {
        const int q = tid + k * NTH, pix = q / PPR;
        const int bl = pix / hhw, r = pix - bl * hhw, hy = r / p.hw, hx = r - hy * p.hw;
        const bool ok = q < nq && chQ < p.Cb;
        blq[k] = ok ? bl : (1 << 20);
        hyq[k] = hy; hxq[k] = hx;
        offQ[k] = bl * p.QH * rowQ + chQ + (PAD ? 0 : hy * rowQ + hx * p.q_cs);
    }
    u32x4 rp[MAXP], rq[MAXQ];
    struct TileCtx { const T* baseP; const T* baseQ; int iy0, ix0, limB, limY, limX; };
    auto make_ctx = [&](const TileC& c, bool live) {
        TileCtx x;
        x.limB = live ? p.B - c.tbi * p.nb : 0;
        x.limY = p.PH - (c.tyi << p.th_log2); x.limX = p.PW - (c.txi << p.tw_log2);
        x.baseP = Pg + (((int64_t)c.tbi * p.nb * p.PH + (c.tyi << p.th_log2)) * p.PW + (c.txi << p.tw_log2)) * p.p_cs;
        x.iy0 = (c.tyi << p.th_log2) * p.q_step + p.min_dy; x.ix0 = (c.txi << p.tw_log2) * p.q_step + p.min_dx;
        x.baseQ = Qg + (int64_t)c.tbi * p.nb * p.QH * rowQ;
        if (!PAD) x.baseQ += (int64_t)x.iy0 * rowQ + (int64_t)x.ix0 * p.q_cs;
        return x;
    };
    auto issue_piece = [&](int K, const TileCtx& x) {
        if (K < MAXP) {
            const int k = K;
            const bool ok = pby[k] < x.limB && (pxx[k] >> 16) < x.limY && (pxx[k] & 0xffff) < x.limX;
            rp[k] = *reinterpret_cast<const u32x4*>(ok ? x.baseP + offP[k] : zp);
        } else {
            const int k = K - MAXP;
            if (!PAD) {
                const bool ok = (unsigned)(x.iy0 + hyq[k]) < (unsigned)p.QH && (unsigned)(x.ix0 + hxq[k]) < (unsigned)p.QW && blq[k] < x.limB;
                rq[k] = *reinterpret_cast<const u32x4*>(ok ? x.baseQ + offQ[k] : zp);
            } else {
                const int iy = min(max(x.iy0 + hyq[k], 0), p.QH - 1), ix = min(max(x.ix0 + hxq[k], 0), p.QW - 1);
                rq[k] = *reinterpret_cast<const u32x4*>(blq[k] < x.limB ? x.baseQ + (offQ[k] + iy * rowQ + ix * p.q_cs) : zp);
            }
        }
    };

    // ---- fragment addressing (conv_wgrad_fast_kernel, ROW16): this wave half reads kernel rows grp and grp + 1
    const int khalf = lane >> 5, g16 = (lane >> 4) & 1, i16 = lane & 15, prow = i16 >> 2, pcol = (i16 & 3) * 4;
    const int pa_base = (khalf * 8 + prow) * ROWB + (wa * 32 + g16 * 16 + pcol) * 2;
    int tq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) tq[i] = (khalf * 8 + prow) * ROWB + (wb * 32 + g16 * 16 + pcol) * 2 + (grp + i) * (18 * ROWB);
    struct Frag { s16x4 alo, ahi; u32x2 q[2][3]; };
    auto read_frags = [&](int j, Frag& fr) {
        fr.alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sP + pa_base + (j * 16) * ROWB));
        fr.ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sP + pa_base + (j * 16 + 4) * ROWB));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int u = 0; u < 3; ++u)
                fr.q[i][u] = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(sQ + tq[i] + (j * 18 + u * 4) * ROWB)));
    };
    auto b_operand = [&](const Frag& fr, int row, int dx) -> bf16x8 {       // row, dx constants after unrolling
        const unsigned w0 = fr.q[row][0].x, w1 = fr.q[row][0].y, w2 = fr.q[row][1].x, w3 = fr.q[row][1].y, w4 = fr.q[row][2].x;
        u32x4 v;
        if (dx == 0) v = u32x4{w0, w1, w2, w3};
        else if (dx == 1) v = u32x4{__builtin_amdgcn_alignbit(w1, w0, 16), __builtin_amdgcn_alignbit(w2, w1, 16),
                                    __builtin_amdgcn_alignbit(w3, w2, 16), __builtin_amdgcn_alignbit(w4, w3, 16)};
        else v = u32x4{w1, w2, w3, w4};
        return __builtin_bit_cast(bf16x8, v);
    };
    Frag f0, f1;
    // the k-steps of one tile for wave half G (taps 5 G ... 5 G + CNT - 1): static register indices need G at compile time
    auto ksteps = [&](auto gc, const TileCtx& x) {
        constexpr int G = decltype(gc)::value, CNT = G ? 4 : 5;
        read_frags(0, f0);
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            Frag& fc = (j & 1) ? f1 : f0;
            Frag& fn = (j & 1) ? f0 : f1;
            if (j + 1 < KS) read_frags(j + 1, fn);
            int nld = 0;
            if (j < LSTEPS) {
#pragma unroll
                for (int u = 0; u < LPS; ++u)
                    if (j * LPS + u < NPIECE) { issue_piece(j * LPS + u, x); ++nld; }
            }
            const s16x8 av = {fc.alo[0], fc.alo[1], fc.alo[2], fc.alo[3], fc.ahi[0], fc.ahi[1], fc.ahi[2], fc.ahi[3]};
#pragma unroll
            for (int tl = 0; tl < CNT; ++tl) {
                const int t = G * 5 + tl;
                acc[tl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b_operand(fc, t / 3 - G, t % 3), __builtin_bit_cast(bf16x8, av), acc[tl], 0, 0, 0);
            }
            if (j + 1 < KS) {                                        // pin: the 8 LDS reads (+ the global loads) between the MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int tl = 0; tl < CNT; ++tl) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (tl < 6 - CNT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (tl < nld) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
        }
    };

    TileC cur = decode(split);
    {
        const TileCtx x = make_ctx(cur, split < p.ntiles);
#pragma unroll
        for (int K = 0; K < NPIECE; ++K) issue_piece(K, x);
    }
    for (int tile = split; tile < p.ntiles; tile += p.nsplit) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXP; ++k) { const int q = tid + k * NTH; if (q < np) *reinterpret_cast<u32x4*>(sP + (q / PPR) * ROWB + pcx * 16) = rp[k]; }
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) { const int q = tid + k * NTH; if (q < nq) *reinterpret_cast<u32x4*>(sQ + (q / PPR) * ROWB + pcx * 16) = rq[k]; }
        __syncthreads();
        advance(cur);
        const TileCtx x = make_ctx(cur, tile + p.nsplit < p.ntiles);
        if (grp == 0) ksteps(std::integral_constant<int, 0>{}, x);     // wave-uniform: no barrier inside
        else ksteps(std::integral_constant<int, 1>{}, x);
    }
    // ---- partial slab partials[split][t][a][b]: lane = a-row, 4 consecutive registers = 4 consecutive b
    const int l31 = lane & 31;
    const int a = a0 + wa * 32 + l31;
    const int cnt = grp ? 4 : 5;
#pragma unroll
    for (int tl = 0; tl < NTA; ++tl) {
        if (tl < cnt && a < p.Ca) {
            float* row = p.partials + (((int64_t)(p.atomic ? 0 : split) * p.ntaps + grp * 5 + tl) * p.Ca + a) * p.Cb;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int b = c0 + wb * 32 + 8 * g + 4 * khalf;
                if (b < p.Cb) slab_put4(row + b, f32x4{acc[tl][4 * g], acc[tl][4 * g + 1], acc[tl][4 * g + 2], acc[tl][4 * g + 3]}, p.atomic != 0);
            }
        }
    }
}

// fp32 version of the ROW16 fast path (exact fp32: v_mfma_f32_32x32x2_f32, two pixels per k-step, 64 k-steps per 128-pixel tile).
// LDS rows are 64 channels x 4 B; a fragment read is one dword per lane (lanes = consecutive channels: conflict free) at a per-tap
// lane constant plus a compile-time offset of the k-step, so the unrolled loop has no address arithmetic; the branch-free loader and
// the 16-byte slab stores are those of conv_wgrad_fast_kernel.  The generic kernel spent ~34 us per tile here, most of it on one
// global load in flight at a time.
// KSPLIT / SIDE: a layer with <= 32 output or input channels fills a quarter / half of the 64 x 64 channel block, and the exact-f32
// MFMA (64 cycles for K = 2) makes the idle waves expensive (vanilla U-Net, 16-64 channels: 200 us per launch whatever the layer).
// KSPLIT = 4 (both sides <= 32): all four waves work on the SAME 32 x 32 channel block and take a quarter of the tile's 64 k-steps
// each; KSPLIT = 2 (SIDE 0: the a side <= 32, SIDE 1: the b side <= 32): the two waves that would idle split the k-steps with their
// partners.  The partial accumulators meet in LDS after the tile loop (fixed order: k-slice 0 + 1 + 2 + 3).
template <bool PAD, int KSPLIT = 1, int SIDE = 0>
__global__ __launch_bounds__(256) void conv_wgrad_fast32_kernel(WgradKP p) {
    typedef float T;
    constexpr int NT = 9, KS = 64, PPR = 16, ROWB = 256, BMP = 128;
    constexpr int JN = KS / KSPLIT;                                   // k-steps per wave and tile
    constexpr int MAXP = BMP * PPR / 256, MAXQ = 12, NPIECE = MAXP + MAXQ;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sP = smem;
    unsigned char* sQ = smem + BMP * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // wave roles: channel sub-block (wa, wb) and k-slice ks
    const int wa = KSPLIT == 4 ? 0 : (KSPLIT == 2 && SIDE == 0 ? 0 : wave >> 1);
    const int wb = KSPLIT == 4 ? 0 : (KSPLIT == 2 && SIDE == 1 ? 0 : wave & 1);
    const int ks = KSPLIT == 4 ? wave : (KSPLIT == 2 ? (SIDE == 0 ? wave >> 1 : wave & 1) : 0);
    const int blk = blockIdx.x;
    const int ab = blk % p.a_blocks;
    const int bb = (blk / p.a_blocks) % p.b_blocks;
    const int split = blk / (p.a_blocks * p.b_blocks);
    const int a0 = ab * 64, c0 = bb * 64;
    const int hhw = p.hh * p.hw, phalo = p.nb * hhw;
    const T* Pg = reinterpret_cast<const T*>(p.P);
    const T* Qg = reinterpret_cast<const T*>(p.Q) + (p.q_plane ? (long long)bb * p.q_plane - c0 : 0);   // planar Q: channel c0 + j of block bb is element j of plane bb
    const T* zp = reinterpret_cast<const T*>(g_zero_piece);

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    struct TileC { int tbi, tyi, txi; };
    auto decode = [&](int tile) { TileC c; c.txi = tile % p.tiles_x; const int tt = tile / p.tiles_x; c.tyi = tt % p.tiles_y; c.tbi = tt / p.tiles_y; return c; };
    const TileC tstep = decode(p.nsplit);
    auto advance = [&](TileC& c) {
        c.txi += tstep.txi; if (c.txi >= p.tiles_x) { c.txi -= p.tiles_x; ++c.tyi; }
        c.tyi += tstep.tyi; if (c.tyi >= p.tiles_y) { c.tyi -= p.tiles_y; ++c.tbi; }
        c.tbi += tstep.tbi;
    };

    // ---- tile-invariant piece constants.  Piece q = tid + 256 k is 16 bytes (4 channels) of pixel row q / 16.
    const int pcx = tid % PPR;
    const int chP = a0 + pcx * 4, chQ = c0 + pcx * 4;
    const int np = BMP * PPR, nq = phalo * PPR;
    const int rowP = p.PW * p.p_cs, rowQ = p.QW * p.q_cs;
    int offP[MAXP], offQ[MAXQ];
    unsigned dp[MAXP], dq[MAXQ];                                      // packed (image << 16 | row << 8 | column); 0xffffffff: no piece
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int q = tid + (k << 8), m = q / PPR;
        const int tx = m & 15, ty = (m >> 4) & 7, bl = m >> 7;
        const bool ok = q < np && chP < p.Ca;
        offP[k] = (bl * p.PH + ty) * rowP + tx * p.p_cs + chP;
        dp[k] = ok ? (unsigned)((bl << 16) | (ty << 8) | tx) : 0xffffffffu;
    }
#pragma unroll
    for (int k = 0; k < MAXQ; ++k) {
        const int q = tid + (k << 8), pix = q / PPR;
        const int bl = pix / hhw, r = pix - bl * hhw, hy = r / p.hw, hx = r - hy * p.hw;
        const bool ok = q < nq && chQ < p.Cb;
        dq[k] = ok ? (unsigned)((bl << 16) | (hy << 8) | hx) : 0xffffffffu;
        offQ[k] = bl * p.QH * rowQ + chQ + (PAD ? 0 : hy * rowQ + hx * p.q_cs);
    }
    u32x4 rp[MAXP], rq[MAXQ];
    struct TileCtx { const T* baseP; const T* baseQ; int iy0, ix0, limB, limY, limX; };
    auto make_ctx = [&](const TileC& c, bool live) {
        TileCtx x;
        x.limB = live ? p.B - c.tbi * p.nb : 0;
        x.limY = p.PH - (c.tyi << p.th_log2); x.limX = p.PW - (c.txi << p.tw_log2);
        x.baseP = Pg + (((int64_t)c.tbi * p.nb * p.PH + (c.tyi << p.th_log2)) * p.PW + (c.txi << p.tw_log2)) * p.p_cs;
        x.iy0 = (c.tyi << p.th_log2) * p.q_step + p.min_dy; x.ix0 = (c.txi << p.tw_log2) * p.q_step + p.min_dx;
        x.baseQ = Qg + (int64_t)c.tbi * p.nb * p.QH * rowQ;
        if (!PAD) x.baseQ += (int64_t)x.iy0 * rowQ + (int64_t)x.ix0 * p.q_cs;
        return x;
    };
    auto issue_piece = [&](int K, const TileCtx& x) {
        if (K < MAXP) {
            const int k = K;
            const unsigned d = dp[k];
            const bool ok = (int)(d >> 16) < x.limB && (int)((d >> 8) & 255u) < x.limY && (int)(d & 255u) < x.limX;    // no piece: image 65535
            rp[k] = *reinterpret_cast<const u32x4*>(ok ? x.baseP + offP[k] : zp);
        } else {
            const int k = K - MAXP;
            const unsigned d = dq[k];
            const int bl = (int)(d >> 16), hy = (int)((d >> 8) & 255u), hx = (int)(d & 255u);
            if (!PAD) {
                const bool ok = (unsigned)(x.iy0 + hy) < (unsigned)p.QH && (unsigned)(x.ix0 + hx) < (unsigned)p.QW && bl < x.limB;
                rq[k] = *reinterpret_cast<const u32x4*>(ok ? x.baseQ + offQ[k] : zp);
            } else {
                const int iy = min(max(x.iy0 + hy, 0), p.QH - 1), ix = min(max(x.ix0 + hx, 0), p.QW - 1);
                rq[k] = *reinterpret_cast<const u32x4*>(bl < x.limB ? x.baseQ + (offQ[k] + iy * rowQ + ix * p.q_cs) : zp);
            }
        }
    };

    // ---- fragment addressing: k-step j covers pixels 2 j + khalf = (row j >> 3, column (2 j & 15) + khalf) of the 8 x 16 tile
    const int khalf = lane >> 5, l31 = lane & 31;
    // k-slice ks starts at k-step ks * JN = pixel row 2 ks JN of the tile (JN is a multiple of 8: whole 16-pixel tile rows)
    const int pa_lane = (khalf + 2 * JN * ks) * ROWB + (wa * 32 + l31) * 4;            // + 2 j rows: an immediate
    int qt[NT];                                                        // per-tap lane constant; + ((j >> 3) * 18 + (2 j & 15)) rows: an immediate
#pragma unroll
    for (int t = 0; t < NT; ++t) qt[t] = (khalf + p.tap_off[t] + (ks * JN / 8) * 18) * ROWB + (wb * 32 + l31) * 4;
    struct Frag { float a, b[NT]; };
    auto read_frags = [&](int j, Frag& fr) {
        fr.a = *reinterpret_cast<const float*>(sP + pa_lane + (2 * j) * ROWB);
#pragma unroll
        for (int t = 0; t < NT; ++t) fr.b[t] = *reinterpret_cast<const float*>(sQ + qt[t] + ((j >> 3) * 18 + ((2 * j) & 15)) * ROWB);
    };

    TileC cur = decode(split);
    {
        const TileCtx x = make_ctx(cur, split < p.ntiles);
#pragma unroll
        for (int K = 0; K < NPIECE; ++K) issue_piece(K, x);
    }
    Frag f0, f1;
    for (int tile = split; tile < p.ntiles; tile += p.nsplit) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MAXP; ++k) { const int q = tid + (k << 8); if (q < np) *reinterpret_cast<u32x4*>(sP + (q / PPR) * ROWB + pcx * 16) = rp[k]; }
#pragma unroll
        for (int k = 0; k < MAXQ; ++k) { const int q = tid + (k << 8); if (q < nq) *reinterpret_cast<u32x4*>(sQ + (q / PPR) * ROWB + pcx * 16) = rq[k]; }
        __syncthreads();
        advance(cur);
        const TileCtx x = make_ctx(cur, tile + p.nsplit < p.ntiles);
        read_frags(0, f0);
        constexpr int LEVERY = JN >= 3 * NPIECE ? 3 : 1;              // next tile's 20 pieces: one every third k-step (every step when k is split)
        constexpr int LPS = (NPIECE * LEVERY + JN - 1) / JN;          // pieces per loading k-step
#pragma unroll
        for (int j = 0; j < JN; ++j) {
            Frag& fc = (j & 1) ? f1 : f0;
            Frag& fn = (j & 1) ? f0 : f1;
            if (j + 1 < JN) read_frags(j + 1, fn);
            const bool ld = (j % LEVERY == 0) && ((j / LEVERY) * LPS < NPIECE);
            if (ld) {
#pragma unroll
                for (int u = 0; u < LPS; ++u) if ((j / LEVERY) * LPS + u < NPIECE) issue_piece((j / LEVERY) * LPS + u, x);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fc.b[t], fc.a, acc[t], 0, 0, 0);
            if (j + 1 < JN) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (ld && t < LPS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
            }
        }
    }
    if constexpr (KSPLIT > 1) {
        // sum the k-slices of every (wa, wb) sub-block tap by tap through LDS: slice s > 0 parks its accumulator, slice 0 adds them in
        // ascending slice order.  Waves of the same sub-block: KSPLIT 4: all four; KSPLIT 2: the pair that shares (wa, wb).
        float* red = reinterpret_cast<float*>(smem);                  // [pair][slice - 1][16][64]
        const int pair = KSPLIT == 4 ? 0 : (SIDE == 0 ? (wave & 1) : (wave >> 1));
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            __syncthreads();                                           // tile loop / previous tap done with this LDS
            if (ks > 0) {
                float* dst = red + ((pair * (KSPLIT - 1) + ks - 1) * 16) * 64 + lane * 4;
#pragma unroll
                for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(dst + g * 256) = f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
            }
            __syncthreads();
            if (ks == 0) {
#pragma unroll
                for (int s2 = 1; s2 < KSPLIT; ++s2) {
                    const float* src = red + ((pair * (KSPLIT - 1) + s2 - 1) * 16) * 64 + lane * 4;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(src + g * 256);
                        acc[t][4 * g] += v.x; acc[t][4 * g + 1] += v.y; acc[t][4 * g + 2] += v.z; acc[t][4 * g + 3] += v.w;
                    }
                }
            }
        }
        if (ks > 0) return;
    }
    // ---- partial slab partials[split][t][a][b]: lane = a-row, 4 consecutive registers = 4 consecutive b
    const int a = a0 + wa * 32 + l31;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < p.ntaps && a < p.Ca) {
            float* row = p.partials + (((int64_t)(p.atomic ? 0 : split) * p.ntaps + t) * p.Ca + a) * p.Cb;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int b = c0 + wb * 32 + 8 * g + 4 * khalf;
                if (b < p.Cb) slab_put4(row + b, f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]}, p.atomic != 0);
            }
        }
    }
}

int wgrad_plan(const salt_conv_wgrad_args* a, WgradKP* k, int* nsplit_out) {
    if (!a) SALT_FAIL(SALT_E_BADARG, "wgrad: null args");
    {
        salt_view qv = a->q;
        if (a->q_plane) {                         // planar q: planes of one 64-channel b-block
            if (a->q.cs != 64 || a->q.C % 64 || a->q_plane % 8 || a->q_plane < (int64_t)a->q.B * a->q.H * a->q.W * 64)
                SALT_FAIL(SALT_E_BADARG, "wgrad: q_plane needs dense planes of 64 channels");
            qv.cs = qv.C;
        }
        if (!view_ok(a->p) || !view_ok(qv)) SALT_FAIL(SALT_E_BADARG, "wgrad: bad view");
    }
    if (a->ntaps < 1 || a->ntaps > 9) SALT_FAIL(SALT_E_BADARG, "wgrad: ntaps %d (max 9 per launch)", a->ntaps);
    if (a->p.B != a->q.B) SALT_FAIL(SALT_E_BADARG, "wgrad: batch mismatch");
    int min_dy = 1 << 30, max_dy = -(1 << 30), min_dx = 1 << 30, max_dx = -(1 << 30);
    for (int t = 0; t < a->ntaps; ++t) {
        min_dy = a->tap_dy[t] < min_dy ? a->tap_dy[t] : min_dy; max_dy = a->tap_dy[t] > max_dy ? a->tap_dy[t] : max_dy;
        min_dx = a->tap_dx[t] < min_dx ? a->tap_dx[t] : min_dx; max_dx = a->tap_dx[t] > max_dx ? a->tap_dx[t] : max_dx;
    }
    const int rowb = a->dtype == SALT_F32 ? 256 : 192;
    int th = 1, tw = 1;
    for (int bmp = 128; bmp >= 32; bmp >>= 1) {
        k->bmp = bmp;
        k->tw_log2 = ilog2_ceil(a->p.W) < 4 ? ilog2_ceil(a->p.W) : 4;
        int rem = ilog2_ceil(bmp) - k->tw_log2;
        k->th_log2 = ilog2_ceil(a->p.H) < rem ? ilog2_ceil(a->p.H) : rem;
        k->nb = bmp >> (k->tw_log2 + k->th_log2);
        th = 1 << k->th_log2; tw = 1 << k->tw_log2;
        k->hh = (th - 1) * a->q_step + (max_dy - min_dy) + 1;
        k->hw = (tw - 1) * a->q_step + (max_dx - min_dx) + 1;
        if ((size_t)(bmp + k->nb * k->hh * k->hw) * rowb <= 80 * 1024) break;     // keep >= 2 workgroups per CU
    }
    k->P = a->p.p; k->Q = a->q.p; k->partials = a->partials;
    k->q_plane = a->q_plane; k->ksplit = 0; k->atomic = 0;
    k->B = a->p.B; k->PH = a->p.H; k->PW = a->p.W; k->Ca = a->p.C; k->p_cs = a->p.cs;
    k->QH = a->q.H; k->QW = a->q.W; k->Cb = a->q.C; k->q_cs = a->q.cs;
    k->ntaps = a->ntaps; k->q_step = a->q_step; k->pad_mode = a->pad_mode; k->min_dy = min_dy; k->min_dx = min_dx;
    for (int t = 0; t < a->ntaps; ++t) k->tap_off[t] = (a->tap_dy[t] - min_dy) * k->hw + (a->tap_dx[t] - min_dx);
    for (int t = a->ntaps; t < SALT_MAX_TAPS; ++t) k->tap_off[t] = k->tap_off[0];      // padding slots of the NT-tap kernel instance
    k->tiles_y = cdiv(a->p.H, th); k->tiles_x = cdiv(a->p.W, tw);
    k->ntiles = cdiv(a->p.B, k->nb) * k->tiles_y * k->tiles_x;
    k->a_blocks = cdiv(a->p.C, 64); k->b_blocks = cdiv(a->q.C, 64);
    // every split costs a partial slab (written here, read again by the reduce): bound the workgroup count and give each
    // workgroup enough pixel tiles to amortise its slab
    static const int target_wgs = getenv("SALT_WGRAD_WGS") ? atoi(getenv("SALT_WGRAD_WGS")) : 512;
    // bf16: >= 8 pixel tiles per split (every split costs a 147 KB slab round trip).  fp32: 2 - the exact-f32 MFMA is 16x slower, so a
    // workgroup's 1024 pixels were ~200 us of matrix work on 32 workgroups (vanilla U-Net fp32: 7.47 -> 6.13 ms per step)
    static const int min_tiles_env = getenv("SALT_WGRAD_TPW") ? atoi(getenv("SALT_WGRAD_TPW")) : 0;
    const int blocks_ = k->a_blocks * k->b_blocks;
    // fp32 (round 3): at least 2 tiles per split and about 256 workgroups per launch - what SALT_WGRAD_TPW=4 gave the ResNet34 layers
    // (23.5 ms per step against 24.4 with 2) and SALT_WGRAD_TPW=2 the small levels of the vanilla net (4.17 against 4.27 ms): the 8x8 /
    // 16x16 levels run on the generic kernel, whose time is the length of a workgroup's tile chain, while more than ~256 workgroups take
    // CUs from the data-gradient chain
    const int min_tiles = min_tiles_env > 0 ? min_tiles_env : (a->dtype == SALT_F32 ? 2 : 8);
    static const bool wgs_env = getenv("SALT_WGRAD_WGS") != nullptr;
    // bf16 (round 6): 128 workgroups per launch instead of 512.  Backward is bound by the kernel time of its two queues together
    // (DESIGN 7): a weight-gradient launch that holds every CU slows the data-gradient chain beside it by more than its own shorter
    // run is worth, and every split less is a 147 KB slab less to write and reduce.  Same box, C2 step: 512 5.05 - 5.08 ms, 256 5.00,
    // 160 5.00, 128 4.97 - 4.99, 96 5.03, 64 5.13, 32 5.97 (profiles/r06_wgrad_wgs_ab.txt)
    static const int wgs_big = getenv("SALT_WGRAD_WGS_BIG") ? atoi(getenv("SALT_WGRAD_WGS_BIG")) : 0;      // A/B: the 128 x 128 maps (>= 2048 pixel tiles)
    int ns = (wgs_env ? target_wgs : (a->dtype == SALT_F32 && !min_tiles_env) ? 256 : a->dtype == SALT_F32 ? target_wgs : 128) / blocks_;
    if (wgs_big > 0 && k->ntiles >= 2048) ns = wgs_big / blocks_;
    // the stem (64 x 16 channels, launched on the MAIN stream at the very end of backward, nothing left to overlap with): finer
    // split.  NOT for the other single-block layers: their launches share the chip with the data-gradient chain, and 256 instead
    // of 128 weight-gradient workgroups cost 6.17 -> 6.24 ms per step (and halving the tiles per split wherever a launch has fewer than
    // 256 workgroups 6.17 -> 6.30)
    const int mt = (k->a_blocks * k->b_blocks == 1 && a->q.C <= 16 && min_tiles >= 2) ? min_tiles / 2 : min_tiles;
    if (ns > k->ntiles / mt) ns = k->ntiles / mt;
    if (ns < 1) ns = 1;
    if (ns > k->ntiles) ns = k->ntiles;
    *nsplit_out = ns;
    k->nsplit = ns;
    return SALT_OK;
}

struct ReduceKP {
    const float* partials; float* grad; int nsplit, ntaps, Ca, Cb, KH, KW, accumulate;
    int ldb, a_mod;                                // slice of a wider tensor / tap-GEMM slab (salt_wgrad_reduce_args)
    int tap_kh[SALT_MAX_TAPS], tap_kw[SALT_MAX_TAPS];
};
// 256 threads = 4 split-rows x 64 consecutive slab elements: coalesced 256-B reads, 4x the parallelism of one thread
// per element (the slab is small - 36 K floats for a 64->64 3x3 - while nsplit can be 512), fixed summation order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(ReduceKP p) {
    __shared__ float sm[4][64];
    const int64_t slab = (int64_t)p.ntaps * p.Ca * p.Cb;
    const int e = threadIdx.x & 63, row = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + e;
    float s0 = 0.f, s1 = 0.f;
    if (i < slab) {
        int k = row;
        for (; k + 4 < p.nsplit; k += 8) { s0 += p.partials[k * slab + i]; s1 += p.partials[(k + 4) * slab + i]; }
        for (; k < p.nsplit; k += 4) s0 += p.partials[k * slab + i];
    }
    sm[row][e] = s0 + s1;
    __syncthreads();
    if (row != 0 || i >= slab) return;
    const float s = sm[0][e] + sm[1][e] + sm[2][e] + sm[3][e];
    const int b = (int)(i % p.Cb);
    int64_t r = i / p.Cb;
    int a = (int)(r % p.Ca);
    int t = (int)(r / p.Ca);
    if (p.a_mod) { t = a / p.a_mod; a -= t * p.a_mod; }               // tap-GEMM slab: row = (tap, reference row)
    float* dst = p.grad + (((int64_t)a * p.ldb + b) * p.KH + p.tap_kh[t]) * p.KW + p.tap_kw[t];
    *dst = p.accumulate ? (*dst + s) : s;
}

// nsplit <= 8 (every 3x3 layer of the networks here): one thread per element with all split loads in flight, no LDS round trip;
// same summation order as wgrad_reduce_kernel for nsplit <= 8: (((v0 + v4) + (v1 + v5)) + (v2 + v6)) + (v3 + v7)
__global__ __launch_bounds__(256) void wgrad_reduce8_kernel(ReduceKP p) {
    const int64_t slab = (int64_t)p.ntaps * p.Ca * p.Cb;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= slab) return;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = k < p.nsplit ? p.partials[k * slab + i] : 0.f;
    const float s = (((v[0] + v[4]) + (v[1] + v[5])) + (v[2] + v[6])) + (v[3] + v[7]);
    const int b = (int)(i % p.Cb);
    int64_t r = i / p.Cb;
    int a = (int)(r % p.Ca);
    int t = (int)(r / p.Ca);
    if (p.a_mod) { t = a / p.a_mod; a -= t * p.a_mod; }               // tap-GEMM slab: row = (tap, reference row)
    float* dst = p.grad + (((int64_t)a * p.ldb + b) * p.KH + p.tap_kh[t]) * p.KW + p.tap_kw[t];
    *dst = p.accumulate ? (*dst + s) : s;
}

// (VERDICT r4 #3a, built and measured in round 5: no gain - kept opt-in behind SALT_WGRAD_REDUCE9=1.)
// nsplit <= 8 AND the nine taps of a 3x3 weight in raster order: one thread per (a, b) pair sums its nine taps (9 x nsplit coalesced
// loads in flight) and stores them as the 36 contiguous bytes they are in the reference layout - a wave writes 2304 contiguous bytes.
// wgrad_reduce8_kernel's thread-per-element form stores 4 bytes at a 36-byte stride (every 128-byte line of a 512 x 512 layer's
// 9.4 MB gradient is written nine times, a ninth each).  Same summation order per element: bit-identical results.
__global__ __launch_bounds__(256) void wgrad_reduce9_kernel(ReduceKP p) {
    const int64_t ab = (int64_t)blockIdx.x * 256 + threadIdx.x, nab = (int64_t)p.Ca * p.Cb;
    if (ab >= nab) return;
    const int a = (int)(ab / p.Cb), b = (int)(ab - (int64_t)a * p.Cb);
    const int64_t slab = 9 * nab;
    float s[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = k < p.nsplit ? p.partials[k * slab + t * nab + ab] : 0.f;
        s[t] = (((v[0] + v[4]) + (v[1] + v[5])) + (v[2] + v[6])) + (v[3] + v[7]);
    }
    float* dst = p.grad + ((int64_t)a * p.ldb + b) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) dst[t] = p.accumulate ? dst[t] + s[t] : s[t];
}

// salt_wgrad_reduce_batched: one launch over the reductions of several layers.  A block finds its job in the block-prefix table and
// runs wgrad_reduce8_kernel's (nsplit <= 8: a thread per element, 256 elements per block) or wgrad_reduce_kernel's (4 split rows x 64
// elements) body on it - the same loads, the same summation order, the same stores.
__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(const salt_wgrad_reduce_args* jobs, const int* block0, int njobs) {
    __shared__ float sm[4][64];
    int lo = 0, hi = njobs;                                     // block0[lo] <= blockIdx.x < block0[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((int)blockIdx.x >= block0[mid]) lo = mid; else hi = mid; }
    const salt_wgrad_reduce_args& a = jobs[lo];
    const int blk = (int)blockIdx.x - block0[lo];
    const int64_t slab = (int64_t)a.ntaps * a.Ca * a.Cb;
    const int ldb = a.ldb > 0 ? a.ldb : a.Cb;
    int64_t i; float s; bool writer;
    if (a.nsplit <= 8) {
        i = (int64_t)blk * 256 + threadIdx.x;
        writer = i < slab;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (writer && k < a.nsplit) ? a.partials[k * slab + i] : 0.f;
        s = (((v[0] + v[4]) + (v[1] + v[5])) + (v[2] + v[6])) + (v[3] + v[7]);
    } else {
        const int e = threadIdx.x & 63, row = threadIdx.x >> 6;
        i = (int64_t)blk * 64 + e;
        float s0 = 0.f, s1 = 0.f;
        if (i < slab) {
            int k = row;
            for (; k + 4 < a.nsplit; k += 8) { s0 += a.partials[k * slab + i]; s1 += a.partials[(k + 4) * slab + i]; }
            for (; k < a.nsplit; k += 4) s0 += a.partials[k * slab + i];
        }
        sm[row][e] = s0 + s1;
        __syncthreads();
        writer = row == 0 && i < slab;
        s = sm[0][e] + sm[1][e] + sm[2][e] + sm[3][e];
    }
    if (!writer) return;
    const int b = (int)(i % a.Cb);
    int64_t r = i / a.Cb;
    int ar = (int)(r % a.Ca);
    int t = (int)(r / a.Ca);
    if (a.a_mod) { t = ar / a.a_mod; ar -= t * a.a_mod; }
    float* dst = a.grad + (((int64_t)ar * ldb + b) * a.KH + a.tap_kh[t]) * a.KW + a.tap_kw[t];
    *dst = a.accumulate ? (*dst + s) : s;
}

}  // namespace

extern "C" int salt_wgrad_reduce_job_blocks(const salt_wgrad_reduce_args* a) {
    if (!a || a->ntaps < 1 || a->ntaps > SALT_MAX_TAPS || a->nsplit < 1 || a->Ca < 1 || a->Cb < 1 || a->KH < 1 || a->KW < 1) return -1;
    int nt_tab = a->ntaps;
    if (a->a_mod) {
        if (a->a_mod < 0 || a->ntaps != 1 || a->Ca % a->a_mod || a->Ca / a->a_mod > SALT_MAX_TAPS) return -1;
        nt_tab = a->Ca / a->a_mod;
    }
    if (a->ldb && a->ldb < a->Cb) return -1;
    for (int t = 0; t < nt_tab; ++t)
        if (a->tap_kh[t] < 0 || a->tap_kh[t] >= a->KH || a->tap_kw[t] < 0 || a->tap_kw[t] >= a->KW) return -1;
    const int64_t slab = (int64_t)a->ntaps * a->Ca * a->Cb;
    const int64_t blocks = a->nsplit <= 8 ? (slab + 255) / 256 : (slab + 63) / 64;
    return blocks > 0x3fffffff ? -1 : (int)blocks;
}

extern "C" int salt_wgrad_reduce_batched(const salt_wgrad_reduce_batched_args* a, void* stream) {
    if (!a || !a->jobs || !a->job_block0 || a->njobs < 1 || a->total_blocks < 1) SALT_FAIL(SALT_E_BADARG, "wgrad_reduce_batched: bad args");
    hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3((unsigned)a->total_blocks), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const salt_wgrad_reduce_args*>(a->jobs), a->job_block0, a->njobs);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_conv(const salt_conv_args* a, void* stream) {
    Plan pl;
    int rc = make_plan(a, &pl);
    if (rc) return rc;
    if (a->stats && !a->stats_cnt) SALT_FAIL(SALT_E_BADARG, "conv: stats without stats_cnt");
    if (a->strip) {
        if (a->stats || a->out_step != 1 || a->out_oy || a->out_ox || a->fold_top < 0 || a->fold_bottom < 0 || a->fold_left < 0 || a->fold_right < 0 ||
            a->OH != a->y.H + a->fold_top + a->fold_bottom || a->OW != a->y.W + a->fold_left + a->fold_right || a->strip_cs < a->y.C ||
            (reinterpret_cast<uintptr_t>(a->strip) & 15))
            SALT_FAIL(SALT_E_BADARG, "conv: fold mode needs OH/OW = y.H/y.W + pads, out_step 1, no stats, 16-byte aligned strip");
    }
    const bool in_tf = a->in_scale || a->in_fin_acc;                              // input transform: conv_mfma_kernel only
    if (!in_tf && conv_ws_eligible(a)) return conv_ws_launch(a, (hipStream_t)stream);      // the plan above validated the arguments
    if (a->y_plane) SALT_FAIL(SALT_E_UNSUPPORTED, "conv: planar y (y_plane) is written by conv_ws_kernel only and this launch is not eligible for it");
    if (!in_tf && conv_ls_variant(a)) return conv_ls_launch(a, (hipStream_t)stream);
    if (a->x_plane) SALT_FAIL(SALT_E_UNSUPPORTED, "conv: planar x (x_plane) is read by conv_ls_kernel only and this launch is not eligible for it");
    if (!in_tf && conv1x1_ls_variant(a)) return conv1x1_ls_launch(a, (hipStream_t)stream);
    if (!in_tf && conv_thin_variant(a)) return conv_thin_launch(a, (hipStream_t)stream);
    if (!in_tf && conv_stem16_variant(a)) return conv_stem16_launch(a, (hipStream_t)stream);
    if (a->dtype == SALT_F32) return launch_T<float>(pl, (hipStream_t)stream);
    if (a->dtype == SALT_BF16) return launch_T<bf16_t>(pl, (hipStream_t)stream);
    SALT_FAIL(SALT_E_BADARG, "conv: dtype %d", a->dtype);
}

extern "C" int salt_conv_stats_parts(const salt_conv_args* a) {
    // always conv_mfma_kernel's tile count: the per-tile partials protocols (SALT_BN_FIN=0: stats / bnb_partials) run on that kernel
    // only, and the engine sizes their workspaces BEFORE it sets those fields - conv_ws / conv_ls (shard protocols) never use the count
    Plan pl;
    if (make_plan(a, &pl)) return -1;
    return pl.parts;
}

extern "C" int salt_conv_kernel_id(const salt_conv_args* a) {
    Plan pl;
    if (make_plan(a, &pl)) return -1;
    if (a->in_scale || a->in_fin_acc) return pl.cfg.id;
    if (a->y_plane) return conv_ws_eligible(a) ? 9 : -1;
    if (a->x_plane) return conv_ls_variant(a) ? 10 : -1;
    return conv_ws_eligible(a) ? 9 : (conv_ls_variant(a) ? 10 : (conv1x1_ls_variant(a) ? 11 : (conv_thin_variant(a) ? 12 : (conv_stem16_variant(a) ? 13 : pl.cfg.id))));
}

extern "C" int salt_conv_tile_shape(const salt_conv_args* a) {
    Plan pl;
    if (make_plan(a, &pl)) return -1;
    return pl.kp.tw | (pl.kp.th << 8) | (pl.kp.nb << 16) | (pl.kp.gen << 24);
}

extern "C" int64_t salt_packed_weight_elems(int dtype, int ntaps, int n, int c) {
    const int KCE = dtype == SALT_F32 ? 16 : 32;
    return (int64_t)cdiv(c, KCE) * ntaps * n * KCE;
}

extern "C" int salt_pack_conv_weight(const salt_pack_conv_weight_args* a, void* stream) {
    if (!a || !a->w || !a->wp || a->ntaps < 1 || a->ntaps > SALT_MAX_TAPS) SALT_FAIL(SALT_E_BADARG, "pack: bad args");
    PackKP p;
    p.w = a->w; p.wp = a->wp; p.D0 = a->D0; p.D1 = a->D1; p.KH = a->KH; p.KW = a->KW; p.ntaps = a->ntaps; p.transpose = a->transpose;
    const int S1 = a->d1_cnt > 0 ? a->d1_cnt : a->D1;
    if (a->d1_cnt < 0 || a->d1_cnt > a->D1 || a->n_off < 0 || a->n_total < 0 || a->chunk_off < 0) SALT_FAIL(SALT_E_BADARG, "pack: sub-block");
    p.N = a->transpose ? S1 : a->D0; p.C = a->transpose ? a->D0 : S1;
    p.n_off = a->n_off; p.n_total = a->n_total > 0 ? a->n_total : p.N; p.chunk_off = a->chunk_off;
    if (p.n_off + p.N > p.n_total) SALT_FAIL(SALT_E_BADARG, "pack: n_off + rows > n_total");
    const int KCE = a->dtype == SALT_F32 ? 16 : 32;
    p.nchunk = cdiv(p.C, KCE);
    for (int t = 0; t < a->ntaps; ++t) {
        if (a->tap_kh[t] >= a->KH || a->tap_kw[t] < 0 || a->tap_kw[t] >= a->KW) SALT_FAIL(SALT_E_BADARG, "pack: tap");      // tap_kh < 0: zero tap
        p.tap_kh[t] = a->tap_kh[t]; p.tap_kw[t] = a->tap_kw[t];
    }
    const int64_t total = (int64_t)p.nchunk * p.ntaps * p.N * KCE;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (a->dtype == SALT_F32) hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else if (a->dtype == SALT_BF16) hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    else SALT_FAIL(SALT_E_BADARG, "pack: dtype");
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_pack_job_blocks(const salt_pack_conv_weight_args* a) {
    if (!a) return -1;
    const int KCE = a->dtype == SALT_F32 ? 16 : 32;
    const int S1 = a->d1_cnt > 0 ? a->d1_cnt : a->D1;
    const int N = a->transpose ? S1 : a->D0, C = a->transpose ? a->D0 : S1;
    if (pack_vec_ok(*a, a->dtype)) return (int)(((int64_t)N * (C / 32) * 4 + 255) / 256);      // one thread per (chunk, n, 8 channels)
    if (!a->transpose) return (int)(((int64_t)N * cdiv(C, KCE) * KCE + 255) / 256);      // one thread per (n, padded channel) pair
    const int64_t total = (int64_t)cdiv(C, KCE) * a->ntaps * N * KCE;
    return (int)((total + PACK_EPB - 1) / PACK_EPB);
}

extern "C" int salt_pack_job_is_vec(const salt_pack_conv_weight_args* a) {      // (salt_adam_pack fuses the 3x3 layers; the few 1x1 vector packs stay with salt_pack_batched)
    return a && !a->transpose && a->KH * a->KW == 9 && pack_vec_ok(*a, a->dtype) ? 1 : 0;
}

extern "C" int salt_adam_pack(const salt_adam_pack_args* a, void* stream) {
    if (!a || !a->param || !a->grad || !a->exp_avg || !a->exp_avg_sq || !a->hyper || a->n < 0 || a->njobs < 0 || a->nrest < 0 || a->pack_blocks < 0 ||
        a->rest_blocks < 0 || (a->njobs && (!a->jobs || !a->job_block0)) || (a->nrest && (!a->rest || !a->rest_block0)))
        SALT_FAIL(SALT_E_BADARG, "adam_pack: bad args");
    if ((reinterpret_cast<uintptr_t>(a->param) | reinterpret_cast<uintptr_t>(a->grad) | reinterpret_cast<uintptr_t>(a->exp_avg) | reinterpret_cast<uintptr_t>(a->exp_avg_sq)) & 15)
        SALT_FAIL(SALT_E_BADARG, "adam_pack: buffers must be 16-byte aligned");
    const int64_t blocks = (int64_t)a->pack_blocks + a->rest_blocks;
    if (blocks == 0) return SALT_OK;
    if (blocks >= (1LL << 31)) SALT_FAIL(SALT_E_UNSUPPORTED, "adam_pack: too many blocks");
    AdamPackKP k;
    k.p = a->param; k.g = a->grad; k.m = a->exp_avg; k.v = a->exp_avg_sq; k.hyper = a->hyper;
    k.jobs = reinterpret_cast<const salt_pack_conv_weight_args*>(a->jobs); k.job_block0 = a->job_block0; k.njobs = a->njobs; k.pack_blocks = a->pack_blocks;
    k.rest = a->rest; k.rest_block0 = a->rest_block0; k.nrest = a->nrest;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(adam_pack_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(adam_pack_kernel, dim3((unsigned)blocks), dim3(256), AP_SEG * AP_SEGF * sizeof(float), (hipStream_t)stream, k);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_pack_batched(const salt_pack_batched_args* a, void* stream) {
    if (!a || !a->jobs || !a->job_block0 || a->njobs < 1 || a->total_blocks < 1) SALT_FAIL(SALT_E_BADARG, "pack_batched: bad args");
    const auto* jobs = reinterpret_cast<const salt_pack_conv_weight_args*>(a->jobs);
    if (a->dtype == SALT_F32) hipLaunchKernelGGL(pack_batched_kernel<float>, dim3(a->total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, a->job_block0, a->njobs);
    else if (a->dtype == SALT_BF16) hipLaunchKernelGGL(pack_batched_kernel<bf16_t>, dim3(a->total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, a->job_block0, a->njobs);
    else SALT_FAIL(SALT_E_BADARG, "pack_batched: dtype");
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_conv_wgrad_nsplit(const salt_conv_wgrad_args* a) {
    WgradKP k; int ns = 0;
    if (wgrad_plan(a, &k, &ns)) return -1;
    const int th = conv_wgrad_thin(a, false, nullptr, nullptr);     // fp32, 16 / 32 channels on both sides: one slab per persistent workgroup
    if (th > 0) return th;
    const int ls = conv_wgrad_ls(a, false, nullptr, nullptr);       // the loader-specialised row-streaming kernel has its own split rule
    return ls > 0 ? ls : ns;
}

// salt_conv_wgrad_kernel_id: when set, launch_wgrad records which kernel family it would run (3 fast, 4 fast32, 5 generic) and returns
static thread_local int* g_wgrad_probe = nullptr;
#define SALT_WGRAD_PROBE(ID) if (g_wgrad_probe) { *g_wgrad_probe = (ID); return SALT_OK; }

template <typename T>
static int launch_wgrad(const WgradKP& k, hipStream_t st) {
    constexpr int ROWB = (sizeof(T) == 2) ? 192 : 256;
    const size_t lds = (size_t)(k.bmp + k.nb * k.hh * k.hw) * ROWB;
    if (lds > 160 * 1024) SALT_FAIL(SALT_E_LDS, "wgrad: needs %zu bytes of LDS", lds);
    const dim3 grid((unsigned)(k.a_blocks * k.b_blocks * k.nsplit));
    if constexpr (sizeof(T) == 2) {
        // fast path: whole aligned 16-byte pieces, 3x3, 128-pixel K tiles (conv_wgrad_fast_kernel)
        static const bool generic = getenv("SALT_WGRAD_GENERIC") != nullptr;
        // (5..8 taps - the two 8-tap halves of the 4x4 space-to-depth stem - run the 9-tap instance: the padding tap re-reads tap 0's rows
        //  and is never stored)
        const bool fast = !generic && k.ntaps >= 5 && k.ntaps <= 9 && k.bmp == 128 && k.p_cs % 8 == 0 && k.q_cs % 8 == 0 && k.Ca % 8 == 0 && k.Cb % 8 == 0 &&
                          ((reinterpret_cast<uintptr_t>(k.P) | reinterpret_cast<uintptr_t>(k.Q) | reinterpret_cast<uintptr_t>(k.partials)) & 15) == 0 &&
                          k.nb * k.hh * k.hw * 8 <= 10 * 256;
        if (fast) {
            SALT_WGRAD_PROBE(3)
            bool row16 = k.ntaps == 9 && k.tw_log2 == 4 && k.th_log2 == 3 && k.nb == 1 && k.q_step == 1 && k.hw == 18;
            for (int t = 0; t < 9; ++t) row16 = row16 && k.tap_off[t] == (t / 3) * 18 + t % 3;       // raster tap order
            static const bool four_waves = getenv("SALT_WGRAD_W4") != nullptr;
            if (row16 && !four_waves) {
                auto kern8 = k.pad_mode ? conv_wgrad_fast8_kernel<true> : conv_wgrad_fast8_kernel<false>;
                if (lds > 64 * 1024) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern8), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e)); }
                hipLaunchKernelGGL(kern8, grid, dim3(512), lds, st, k);
                SALT_CHECK_LAUNCH();
                return SALT_OK;
            }
            auto kern = k.pad_mode ? (row16 ? conv_wgrad_fast_kernel<9, 8, true, true> : conv_wgrad_fast_kernel<9, 8, true, false>)
                                   : (row16 ? conv_wgrad_fast_kernel<9, 8, false, true> : conv_wgrad_fast_kernel<9, 8, false, false>);
            if (lds > 64 * 1024) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e)); }
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, k);
            SALT_CHECK_LAUNCH();
            return SALT_OK;
        }
        // round 3: the stride-2 3x3 layers (ResNet layer2-4 conv1): their halo (17 x 17 pixels for an 8 x 8 tile) only leaves room for
        // 64-pixel K tiles, which used to send them to the generic kernel (51 us at 94 TFLOP/s); the fast kernel with 4 k-steps per tile
        static const bool no_fast64 = getenv("SALT_WGRAD_NO_FAST64") != nullptr;
        const bool fast64 = !generic && !no_fast64 && k.ntaps == 9 && k.bmp == 64 && k.p_cs % 8 == 0 && k.q_cs % 8 == 0 && k.Ca % 8 == 0 && k.Cb % 8 == 0 &&
                            ((reinterpret_cast<uintptr_t>(k.P) | reinterpret_cast<uintptr_t>(k.Q) | reinterpret_cast<uintptr_t>(k.partials)) & 15) == 0 &&
                            k.nb * k.hh * k.hw * 8 <= 10 * 256;
        // ... and the 1x1 stride-2 projection shortcuts (one tap; 33 us at 16 TFLOP/s on the generic kernel)
        const bool fast1 = !generic && !no_fast64 && k.ntaps == 1 && !k.pad_mode && (k.bmp == 64 || k.bmp == 128) && k.p_cs % 8 == 0 && k.q_cs % 8 == 0 &&
                           k.Ca % 8 == 0 && k.Cb % 8 == 0 &&
                           ((reinterpret_cast<uintptr_t>(k.P) | reinterpret_cast<uintptr_t>(k.Q) | reinterpret_cast<uintptr_t>(k.partials)) & 15) == 0 &&
                           k.nb * k.hh * k.hw * 8 <= 10 * 256;
        if (fast1) {
            SALT_WGRAD_PROBE(3)
            auto kern = k.bmp == 64 ? conv_wgrad_fast_kernel<1, 4, false, false> : conv_wgrad_fast_kernel<1, 8, false, false>;
            if (lds > 64 * 1024) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e)); }
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, k);
            SALT_CHECK_LAUNCH();
            return SALT_OK;
        }
        if (fast64) {
            SALT_WGRAD_PROBE(3)
            auto kern = k.pad_mode ? conv_wgrad_fast_kernel<9, 4, true, false> : conv_wgrad_fast_kernel<9, 4, false, false>;
            if (lds > 64 * 1024) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e)); }
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, k);
            SALT_CHECK_LAUNCH();
            return SALT_OK;
        }
    }
    if constexpr (sizeof(T) == 4) {
        static const bool generic = getenv("SALT_WGRAD_GENERIC") != nullptr;
        bool fast = !generic && k.ntaps == 9 && k.bmp == 128 && k.p_cs % 4 == 0 && k.q_cs % 4 == 0 && k.Ca % 4 == 0 && k.Cb % 4 == 0 &&
                    ((reinterpret_cast<uintptr_t>(k.P) | reinterpret_cast<uintptr_t>(k.Q) | reinterpret_cast<uintptr_t>(k.partials)) & 15) == 0 &&
                    k.tw_log2 == 4 && k.th_log2 == 3 && k.nb == 1 && k.q_step == 1 && k.hw == 18 && k.hh == 10;
        for (int t = 0; t < 9 && fast; ++t) fast = k.tap_off[t] == (t / 3) * 18 + t % 3;
        if (fast) {
            SALT_WGRAD_PROBE(4)
            static const bool nosplit = getenv("SALT_WGRAD32_NOSPLIT") != nullptr;
            const bool a32 = !nosplit && k.Ca <= 32, b32 = !nosplit && k.Cb <= 32;
            auto kern = k.pad_mode ? conv_wgrad_fast32_kernel<true> : conv_wgrad_fast32_kernel<false>;
            if (a32 && b32) kern = k.pad_mode ? conv_wgrad_fast32_kernel<true, 4, 0> : conv_wgrad_fast32_kernel<false, 4, 0>;
            else if (a32) kern = k.pad_mode ? conv_wgrad_fast32_kernel<true, 2, 0> : conv_wgrad_fast32_kernel<false, 2, 0>;
            else if (b32) kern = k.pad_mode ? conv_wgrad_fast32_kernel<true, 2, 1> : conv_wgrad_fast32_kernel<false, 2, 1>;
            if (lds > 64 * 1024) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e)); }
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, k);
            SALT_CHECK_LAUNCH();
            return SALT_OK;
        }
    }
    SALT_WGRAD_PROBE(5)
    WgradKP kg = k;
    if constexpr (sizeof(T) == 4) {
        static const bool no_ks = getenv("SALT_WGRAD32_NOSPLIT") != nullptr;
        if (!no_ks && !k.atomic && lds >= 4 * 16 * 64 * sizeof(float)) {
            const bool a32 = k.Ca <= 32, b32 = k.Cb <= 32;
            kg.ksplit = (a32 && b32) ? 4 : (a32 ? 2 : (b32 ? 3 : 0));
        }
    }
#define SALT_WG(NT) { auto kern = conv_wgrad_kernel<T, NT>; \
        if (lds > 64 * 1024) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            if (e != hipSuccess) SALT_FAIL((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e)); } \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, kg); }
    if (k.ntaps == 1) SALT_WG(1)
    else if (k.ntaps <= 3) SALT_WG(3)
    else if (k.ntaps == 4) SALT_WG(4)
    else SALT_WG(9)
#undef SALT_WG
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}

extern "C" int salt_conv_wgrad_kernel_id(const salt_conv_wgrad_args* a) {
    WgradKP k; int ns = 0;
    if (wgrad_plan(a, &k, &ns)) return -1;
    if (conv_wgrad_thin(a, false, nullptr, nullptr) > 0) return 2;
    if (conv_wgrad_ls(a, false, nullptr, nullptr) > 0) return 1;
    int id = 0;
    g_wgrad_probe = &id;
    const int rc = a->dtype == SALT_F32 ? launch_wgrad<float>(k, nullptr) : launch_wgrad<bf16_t>(k, nullptr);
    g_wgrad_probe = nullptr;
    return rc ? -1 : id;
}

extern "C" int salt_conv_wgrad(const salt_conv_wgrad_args* a, void* stream) {
    WgradKP k; int ns = 0;
    int rc = wgrad_plan(a, &k, &ns);
    if (rc) return rc;
    if (!a->partials) SALT_FAIL(SALT_E_BADARG, "wgrad: partials workspace missing");
    { int lrc = SALT_OK; if (conv_wgrad_thin(a, true, (hipStream_t)stream, &lrc) > 0) return lrc; }
    { int lrc = SALT_OK; if (conv_wgrad_ls(a, true, (hipStream_t)stream, &lrc) > 0) return lrc; }
    if (a->nsplit != ns) SALT_FAIL(SALT_E_BADARG, "wgrad: nsplit %d, expected %d", a->nsplit, ns);
    // A/B switch: the splits add into ONE slab with global_atomic_add_f32 (zeroed here, in stream order) instead of
    // writing nsplit slabs that salt_wgrad_reduce sums.  Not bit-reproducible; off by default.
    static const bool atomic = getenv("SALT_WGRAD_ATOMIC") != nullptr;
    k.atomic = atomic && ns > 1;
    if (k.atomic) {
        hipError_t e = hipMemsetAsync(a->partials, 0, (size_t)a->ntaps * k.Ca * k.Cb * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) SALT_FAIL((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    if (a->dtype == SALT_F32) return launch_wgrad<float>(k, (hipStream_t)stream);
    if (a->dtype == SALT_BF16) return launch_wgrad<bf16_t>(k, (hipStream_t)stream);
    SALT_FAIL(SALT_E_BADARG, "wgrad: dtype");
}

extern "C" int salt_wgrad_reduce(const salt_wgrad_reduce_args* a, void* stream) {
    if (!a || !a->partials || !a->grad || a->ntaps < 1 || a->ntaps > SALT_MAX_TAPS) SALT_FAIL(SALT_E_BADARG, "wgrad_reduce: bad args");
    ReduceKP p;
    p.partials = a->partials; p.grad = a->grad; p.nsplit = a->nsplit; p.ntaps = a->ntaps; p.Ca = a->Ca; p.Cb = a->Cb;
    p.KH = a->KH; p.KW = a->KW; p.accumulate = a->accumulate;
    p.ldb = a->ldb > 0 ? a->ldb : a->Cb; p.a_mod = a->a_mod;
    int nt_tab = a->ntaps;
    if (a->a_mod) {
        if (a->a_mod < 0 || a->ntaps != 1 || a->Ca % a->a_mod || a->Ca / a->a_mod > SALT_MAX_TAPS) SALT_FAIL(SALT_E_BADARG, "wgrad_reduce: tap-GEMM slab");
        nt_tab = a->Ca / a->a_mod;
    }
    if (a->ldb && a->ldb < a->Cb) SALT_FAIL(SALT_E_BADARG, "wgrad_reduce: ldb < Cb");
    for (int t = 0; t < nt_tab; ++t) {
        if (a->tap_kh[t] < 0 || a->tap_kh[t] >= a->KH || a->tap_kw[t] < 0 || a->tap_kw[t] >= a->KW) SALT_FAIL(SALT_E_BADARG, "wgrad_reduce: tap");
        p.tap_kh[t] = a->tap_kh[t]; p.tap_kw[t] = a->tap_kw[t];
    }
    const int64_t slab = (int64_t)a->ntaps * a->Ca * a->Cb;
    static const bool atomic = getenv("SALT_WGRAD_ATOMIC") != nullptr;
    if (atomic && p.nsplit > 1) p.nsplit = 1;                      // salt_conv_wgrad added every split into slab 0
    static const bool rows_reduce = getenv("SALT_WGRAD_REDUCE_ROWS") != nullptr;
    bool raster9 = a->ntaps == 9 && a->KH == 3 && a->KW == 3 && !a->a_mod && a->nsplit <= 8;
    for (int t = 0; t < 9 && raster9; ++t) raster9 = a->tap_kh[t] * 3 + a->tap_kw[t] == t;
    // opt-in (SALT_WGRAD_REDUCE9=1; per call, the parity test switches it): measured NOT faster than the thread-per-element kernel (the
    // 512 x 512 layers 14.0 against 12.8 us, the class 0.56 against 0.55 ms per step - the 36-byte-stride stores are merged in L2)
    const char* r9_env = getenv("SALT_WGRAD_REDUCE9");
    const bool use_r9 = r9_env && atoi(r9_env) == 1;
    if (!rows_reduce && raster9 && use_r9)
        hipLaunchKernelGGL(wgrad_reduce9_kernel, dim3((unsigned)(((int64_t)a->Ca * a->Cb + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    else if (!rows_reduce && a->nsplit <= 8) hipLaunchKernelGGL(wgrad_reduce8_kernel, dim3((unsigned)((slab + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((slab + 63) / 64)), dim3(256), 0, (hipStream_t)stream, p);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}
