/* saltnet.h — C-ABI of the MI355X-native U-Net hot path (libsaltnet_hip.so, gfx950).
 *
 * The reference (neptune-ai/open-solution-salt-identification) is pure Python on PyTorch; the
 * arithmetic this library replaces is reached from it through torch operator calls inside the
 * nn.Module classes cited next to each entry point (paths relative to the reference's
 * common_blocks/).  A maintainer binds these symbols with ctypes; there are no
 * torch types in any signature: plain pointers to device memory, sizes and a HIP stream.
 *
 * Conventions
 *   - Every operator is `int salt_<op>(const salt_<op>_args*, void* stream)`; 0 = success,
 *     otherwise a SALT_E_* code or a hipError_t (> 0).  `stream` is a hipStream_t (NULL = default).
 *   - The library never allocates or frees device memory; workspaces are passed in.
 *   - Activations are NHWC, element type given by `dtype` (SALT_F32 / SALT_BF16).  A `salt_view`
 *     describes a [B,H,W,C] tensor whose pixels are `cs` elements apart (cs >= C), so a channel
 *     slice of a wider buffer (a "concat-free" skip connection) is a view, not a copy.
 *   - Parameters, gradients, statistics and losses are fp32.
 *   - This header is parsed by the Python host (salt_amd/_abi.py) to build the ctypes structures:
 *     keep one field per line, only the scalar/array/view forms used below.
 */
#ifndef SALTNET_H
#define SALTNET_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SALT_F32 0
#define SALT_BF16 1

#define SALT_OK 0
#define SALT_E_BADARG -1
#define SALT_E_UNSUPPORTED -2
#define SALT_E_LDS -3

#define SALT_MAX_TAPS 16

typedef struct {
    void* p;      /* device pointer to element [0,0,0,0] of the view */
    int B;
    int H;
    int W;
    int C;        /* channels in the view */
    int cs;       /* pixel stride in elements */
} salt_view;

/* ------------------------------------------------------------------ library / device info */
int salt_abi_version(void);                 /* bumps when any struct below changes */
int salt_device_info(int* cu_count, int* lds_bytes, char* arch_name, int arch_name_len);
int salt_abi_struct_sizes(int* out, int n); /* sizeof of every struct below, declaration order */
const char* salt_last_error(void);          /* text for the last non-zero return on this thread */

/* ------------------------------------------------------------------ implicit-GEMM convolution
 * One kernel family serves nn.Conv2d 3x3/1x1 (stride 1|2, zero pad: unet_models.py:24, torchvision
 * BasicBlock/Bottleneck; replicate pad top/right: architectures/base.py:21-27), one output-parity
 * phase of nn.ConvTranspose2d k3/k4 s2 (unet_models.py:44,60; base.py:48-49), and their
 * data-gradients (same kernel, transposed packed weights, mirrored taps).
 *   out[b, oy*out_step+out_oy, ox*out_step+out_ox, n] (+)= epi( sum_t sum_c
 *        X[b, pad(oy*in_step + tap_dy[t]), pad(ox*in_step + tap_dx[t]), c] * Wp[t][n][c] )
 *   epi(v) = relu?( (v + bias[n]) * scale[n] + shift[n] )        (each part optional; + res before the ReLU: see `res` below)
 * Wp is the packed layout produced by salt_pack_conv_weight: [ceil(Cin/KC)][ntaps][Cout][KC],
 * KC = 64 bytes of channels (16 f32 / 32 bf16), zero padded.
 * stats (optional): per output-tile-wave partial (sum, M2 about the partial's own mean, count) of
 * the stored value per channel, for train-mode BatchNorm (deterministic: no atomics).
 */
typedef struct {
    int dtype;
    salt_view x;
    const void* w;            /* packed weights */
    int ntaps;
    int tap_dy[SALT_MAX_TAPS];
    int tap_dx[SALT_MAX_TAPS];
    int in_step;              /* 1 | 2 */
    int pad_mode;             /* 0 zero, 1 clamp (replicate) */
    salt_view y;              /* FULL output buffer view [B,OHf,OWf,Cout] */
    int OH;                   /* logical output grid of this launch */
    int OW;
    int out_step;             /* 1 | 2 */
    int out_oy;
    int out_ox;
    const float* bias;        /* [Cout] or NULL */
    const float* scale;       /* [Cout] or NULL */
    const float* shift;       /* [Cout] or NULL */
    int relu;
    int accumulate;           /* y += result */
    float* stats;             /* [nparts][2][Cout] or NULL */
    float* stats_cnt;         /* [nparts] */
    int stats_part0;          /* first partial index written by this launch (ConvT phases) */
    int cfg;                  /* low byte: 0 = auto, else tile-config / kernel id (tests / tuning; salt_conv_kernel_id); (cfg >> 8) & 0xff:
                               * cap on the workgroups per XCD of conv_ws_kernel / conv_ls_kernel (0 = one per CU), whatever the low byte */
    /* Fold mode (data-gradient of a replicate-padded convolution, architectures/base.py:21-27): the launch computes the gradient on
     * the extended grid OH = y.H + fold_top + fold_bottom, OW = y.W + fold_left + fold_right; interior pixels go straight to y
     * (the UNPADDED tensor, (+)= per `accumulate`), the pad ring goes to `strip` ([B][salt_fold_strip_pixels][strip_cs], always
     * overwritten) and salt_pad_fold_strip adds the ring onto the edge pixels of y.
     * strip == NULL with fold_top > 0 or fold_right > 0 (fold_bottom = fold_left = 0): FUSED fold - same extended grid, but the launch
     * lays its tiles out so that every ring pixel shares a tile with the edge pixel it folds onto and the epilogue adds them before the
     * store: no strip, no second pass (needs out_step 1, no stats, 16-byte aligned y with C and cs multiples of 8 (bf16) / 4 (f32)).
     * strip == NULL and all pads 0: normal mode. */
    void* strip;
    int strip_cs;
    int fold_top;
    int fold_bottom;
    int fold_left;
    int fold_right;
    /* BatchNorm-backward sums fused into a data-gradient launch (the launch that completes g = dL/da of a = relu?(bn(yc)), yc the
     * producer convolution's output): each workgroup also reduces, over its pixel tile and per channel, sum(g m) and
     * sum(g m xhat) with m = [yc*scale + shift > 0] (or [a > 0] / 1, see bnb_a / bnb_relu), xhat = (yc - mean) invstd, of the value it STORES, into
     * bnb_partials[tile][2][Cout], tile < salt_conv_stats_parts(args).  salt_bn_bwd then runs with partials_ready = 1 and skips
     * its own reduction pass over g and yc.  bnb_partials == NULL: off.  Needs out_step 1 on the full grid, no strip, no stats,
     * 16-byte aligned views with Cout a multiple of 8 (bf16) / 4 (f32). */
    salt_view bnb_y;          /* yc, same shape as y */
    salt_view bnb_a;          /* residual layers, a = relu(bn(yc) + res): the forward output the mask comes from (m = [a > 0]); p == NULL: mask from yc */
    const float* bnb_mean;
    const float* bnb_invstd;
    const float* bnb_gamma;
    const float* bnb_beta;
    float* bnb_partials;
    int bnb_relu;
    /* Phase-fused stride-2 launch (nphase == 4): the four output-parity phases of a transposed convolution / stride-2 data gradient in
     * ONE launch.  Every phase runs the same `ntaps` tap offsets over x and writes y at (2 oy + ay, 2 ox + ax), phase = 2 ay + ax
     * (out_oy / out_ox are ignored, out_step must be 2, OH x OW is the per-phase grid); phase p reads the packed weights at
     * w + p * w_phase_elems elements (taps a phase does not have are packed as zeros: tap_kh < 0 in salt_pack_conv_weight).  BN
     * statistics partials are numbered over all phases.  nphase <= 1: a plain launch. */
    int nphase;
    int64_t w_phase_elems;
    /* In-launch BatchNorm finalize.  fin != NULL (a const salt_bn_finalize_args*, declared below; its stats / stats_cnt / nparts are
     * ignored): the launch accumulates its per-tile (sum, sum of squares, count) into fin_acc with fp64 atomics - 8 shards
     * [8][2 * Cout + 1] doubles, shard = workgroup id % 8 (one per XCD), summed in shard order - takes a ticket on fin_ticket, and the
     * workgroup that arrives last computes what salt_bn_finalize would have (mean / invstd / scale / shift, running statistics) before
     * the launch ends: no partials round trip, no separate launch.  `stats` must be NULL.  fin_acc and fin_ticket must be zero before
     * the first launch; every launch leaves them zero.  One launch per BatchNorm layer (no stats_part0 chaining).  The fp64 sums are
     * order dependent in their last bits only (the fp32 results are reproducible in practice, not by construction).
     * bnb_fin != NULL (a const salt_bn_bwd_args*): the same for the BatchNorm-backward sums of bnb_*: the last workgroup writes
     * dgamma / dbeta / coef, and salt_bn_bwd then runs with partials_ready = 2 (apply pass only).  bnb_partials is ignored (may be NULL).
     * fin_ticket == NULL with fin_acc != NULL (bnb_ticket == NULL with bnb_acc != NULL): the launch only ADDS to the shards - no wait,
     * no ticket, fin / bnb_fin unused - and the consumer finalizes them (salt_affine_act_args.fin_acc; salt_bn_bwd with
     * partials_ready = 3).  The shards must then be cleared by the caller before the next launch (one salt_zero over all layers). */
    const void* fin;
    double* fin_acc;
    uint32_t* fin_ticket;
    const void* bnb_fin;
    double* bnb_acc;          /* [8][2 * Cout] */
    uint32_t* bnb_ticket;
    /* Residual epilogue (eval-mode torchvision BasicBlock / Bottleneck, out = relu(bn(conv(a)) + identity), architectures/encoders.py:6-45):
     * res.p != NULL: y = relu?( round_to_dtype((v + bias) * scale + shift) + res ), res a [B,OH,OW,Cout] view - the values a separate
     * salt_affine_act(y, res, relu) pass over the stored convolution output would produce, bit for bit, without that pass.  Plain
     * full-grid launches only (out_step 1, no fold / strip / statistics / accumulate). */
    salt_view res;
    /* Input transform: the PRODUCER's BatchNorm apply + ReLU folded into this launch's loader (the measured alternative to the
     * separate salt_affine_act pass between a convolution and its single consumer, architectures/base.py:29-37, unet_models.py:21-30;
     * DESIGN 10 has the numbers).  in_scale != NULL or in_fin_acc != NULL: every in-range input element becomes
     * x' = round_to_dtype(relu?(x * in_scale[c] + in_shift[c])) - the value salt_affine_act would have stored - between the global
     * load and the LDS store; zero padding stays zero.  in_fin_acc: the [8][2 Cin + 1] fp64 shards of the producer's statistics
     * (salt_conv_args.fin_acc without a ticket) are finalized in every workgroup's prologue with in_fin (the producer layer's const
     * salt_bn_finalize_args*), workgroup 0 stores mean / invstd / scale / shift / running statistics: in_scale / in_shift are ignored.
     * bf16, 9 taps, whole aligned 16-byte pieces; runs on conv_mfma_kernel's register-staged loader (never the LDS-DMA kernels). */
    const float* in_scale;    /* [Cin] */
    const float* in_shift;
    int in_relu;
    const void* in_fin;
    const double* in_fin_acc;
    /* Planar y: y_plane != 0 (elements): y is stored as C / y.cs dense planes [B,H,W,y.cs] - channel c of pixel q at
     * y.p + (c / y.cs) * y_plane + q * y.cs + c % y.cs - instead of channel-interleaved rows (y.cs < y.C is legal only here).  The
     * gradient of the 320-channel hypercolumn (architectures/unet.py:101-107) is kept this way: the five consumers (the up-sampling
     * adjoints, the scSE backward of dec1) each read ONE dense 64-channel plane instead of 128-byte pieces at a 640-byte pitch.
     * conv_ws_kernel only (y.cs == 64 = its channel block); salt_conv fails for any other launch, salt_conv_kernel_id returns -1. */
    int64_t y_plane;
    /* Planar x, the same layout on the input side: channel c of pixel q at x.p + (c / x.cs) * x_plane + q * x.cs + c % x.cs.  The
     * forward convolution over the hypercolumn reads its 32-channel chunks from the five level planes.  conv_ls_kernel only
     * (x.cs == 64: two chunks per plane). */
    int64_t x_plane;
} salt_conv_args;
int salt_conv(const salt_conv_args*, void* stream);
/* number of stats partials a launch with these args writes (host sizes the workspace with it) */
int salt_conv_stats_parts(const salt_conv_args*);
/* which kernel salt_conv runs these arguments on: 1..5 conv_mfma_kernel tile configs, 6..8 conv_glds_kernel, 9 conv_ws_kernel (the
 * weight-stationary multi-tile kernel of the 3x3 layers with <= 64 channels: bf16, Cin in {32, 64}, Cout in {32, 64}, output grid a
 * multiple of 16 x 16), 10 conv_ls_kernel (the loader-specialised streaming kernel of the deeper 3x3 layers: bf16, Cin >= 64 and Cout
 * multiples of 32, same grid rule), 11 conv1x1_ls_kernel (the streaming kernel of the 1x1 convolutions with eval / plain epilogues
 * or train-mode statistics through fin_acc: bf16, Cin a multiple of 64, Cout of 32, B OH OW a multiple of 256; stride 1, or stride 2
 * on 16 x 16 output tiles), 12 conv_thin_kernel (the persistent weight-stationary kernel of the fp32 3x3 layers with 16 or 32
 * channels on BOTH sides, unit steps, output grid a multiple of 16 x 16; statistics / BatchNorm-backward sums through the fp64 shards
 * only), 13 conv_stem16_kernel (the ResNet stem after the 2 x 2 space-to-depth: bf16, 16 taps over 16 input channels -> 64, zero
 * padding, eval epilogue or train-mode statistics through fin_acc).  `cfg` & 0xff:
 * 0 = heuristic, 1..8 = that config, 9 .. 13 = that kernel wherever it applies (else heuristic); (cfg >> 8) & 0xff caps the
 * workgroups per XCD of kernels 9 - 11 (0 = one per CU) and, for 10 / 11 when asked for, (cfg >> 16) & 3 fixes the output channels
 * per item to 32 x that (0 = by size).  Asked-for variants of those two kernels (tests / A-B; round 5): kernel 11 - bit 18 the item-major
 * walk, bit 19 conv1x1_xs_kernel (input tile resident in LDS); kernel 10 - bit 20 two-tile items, bit 21 single tiles only. */
int salt_conv_kernel_id(const salt_conv_args*);
/* pixel tile of conv_mfma_kernel / conv_glds_kernel for these arguments (tests / tuning): tw | th << 8 | images << 16 | general << 24.
 * general = 1: a full-width strip of the extended grid of a fused-fold data gradient (th rows of tw = OW columns, or whole images)
 * instead of a power-of-two tile - the 10 / 18 / 34-wide grids of the decoder's replicate-padded layers fill 16-wide tiles to 40 - 60 %. */
int salt_conv_tile_shape(const salt_conv_args*);

/* weight gradient of the same family:
 *   dW[t][a][b] = sum_{pixels p of P} P[p, a] * Q[pad(p*q_step + tap[t]), b]
 * conv:  P = dY (a = cout), Q = X (b = cin);   convT: P = X (a = cin), Q = dY (b = cout).
 * Writes nsplit partial slabs [nsplit][ntaps][Ca][Cb] fp32 into `partials`;
 * salt_wgrad_reduce sums them in fixed order into the reference parameter layout. */
typedef struct {
    int dtype;
    salt_view p;
    salt_view q;
    int ntaps;
    int tap_dy[SALT_MAX_TAPS];
    int tap_dx[SALT_MAX_TAPS];
    int q_step;
    int pad_mode;
    float* partials;
    int nsplit;               /* as returned by salt_conv_wgrad_nsplit */
    int64_t q_plane;          /* != 0: q is planar (salt_conv_args.y_plane layout), q.cs == 64 = the kernels' b-block: block bb reads plane bb */
} salt_conv_wgrad_args;
int salt_conv_wgrad(const salt_conv_wgrad_args*, void* stream);
int salt_conv_wgrad_nsplit(const salt_conv_wgrad_args*);
/* which kernel family salt_conv_wgrad runs these arguments on (bench.py: roofline.kernel_symbol): 1 conv_wgrad_ls_kernel, 2
 * conv_wgrad_thin_kernel, 3 conv_wgrad_fast_kernel / fast8, 4 conv_wgrad_fast32_kernel, 5 conv_wgrad_kernel (generic); < 0: no plan */
int salt_conv_wgrad_kernel_id(const salt_conv_wgrad_args*);

typedef struct {
    const float* partials;    /* [nsplit][ntaps][Ca][Cb] */
    int nsplit;
    int ntaps;
    int Ca;
    int Cb;
    int KH;                   /* reference weight tensor is [Ca][Cb][KH][KW] */
    int KW;
    int tap_kh[SALT_MAX_TAPS];
    int tap_kw[SALT_MAX_TAPS];
    float* grad;              /* fp32, reference layout */
    int accumulate;
    int ldb;                  /* > 0: `grad` points at channel b = 0 of a SLICE of a wider [Ca][ldb][KH][KW] tensor (row stride of the Cb axis); 0: Cb */
    int a_mod;                /* > 0: "tap GEMM" slab (ntaps must be 1): slab row a = t * a_mod + a' is tap (tap_kh[t], tap_kw[t]) of reference row a';
                               * Ca / a_mod <= SALT_MAX_TAPS.  The factored hypercolumn (salt_hyper_stencil) computes the 3x3 weights of a level this way */
} salt_wgrad_reduce_args;
int salt_wgrad_reduce(const salt_wgrad_reduce_args*, void* stream);

/* round 6: the slab reductions of SEVERAL layers in one launch (the 53 reductions of the ResNet34 U-Net's backward pass were 53 launches
 * of 4 - 25 us on the weight-gradient queue).  `jobs` = DEVICE array of njobs salt_wgrad_reduce_args (each with its own slab),
 * `job_block0` = DEVICE prefix sums [njobs + 1] of salt_wgrad_reduce_job_blocks over the jobs; per element the same loads and the same
 * summation order as salt_wgrad_reduce: bit-identical.  salt_wgrad_reduce_job_blocks validates a job (< 0: bad arguments). */
typedef struct {
    const void* jobs;
    const int* job_block0;
    int njobs;
    int total_blocks;
} salt_wgrad_reduce_batched_args;
int salt_wgrad_reduce_batched(const salt_wgrad_reduce_batched_args*, void* stream);
int salt_wgrad_reduce_job_blocks(const salt_wgrad_reduce_args*);

/* fp32 master weight (reference layout) -> packed compute layout.
 * transpose=0: Wp[chunk][t][n=d0][c=d1]  from W[d0][d1][kh][kw]   (conv forward; convT dgrad)
 * transpose=1: Wp[chunk][t][n=d1][c=d0]                           (conv dgrad;   convT forward) */
typedef struct {
    int dtype;                /* of the packed copy */
    const float* w;           /* [D0][D1][KH][KW] */
    int D0;
    int D1;
    int KH;
    int KW;
    int ntaps;
    int tap_kh[SALT_MAX_TAPS];   /* < 0: a zero tap (phase-fused launches) */
    int tap_kw[SALT_MAX_TAPS];
    int transpose;
    void* wp;
    /* Sub-blocks (all 0: the whole tensor, as before).  d1_cnt > 0: only d1 in [0, d1_cnt) is packed - `w` points at the first channel of
     * a slice of the D1 axis, D1 stays the row stride.  n_off / n_total / chunk_off place the job inside a WIDER packed tensor: packed row
     * n + n_off of n_total rows (0: the job's own count), channel chunk + chunk_off.  The "tap GEMM" of the factored hypercolumn packs
     * the nine [Cout][Cin] tap matrices of a level as ONE 1x1 weight with 9 Cout output channels (forward: n_off = t Cout) and its
     * transpose with 9 Cout input channels (chunk_off = t Cout / KC). */
    int d1_cnt;
    int n_off;
    int n_total;
    int chunk_off;
} salt_pack_conv_weight_args;
int salt_pack_conv_weight(const salt_pack_conv_weight_args*, void* stream);
int64_t salt_packed_weight_elems(int dtype, int ntaps, int n, int c);

/* All pack jobs of a network in ONE launch: `jobs` is a DEVICE array of salt_pack_conv_weight_args (same dtype),
 * `job_block0` a DEVICE array [njobs+1] of prefix block counts (256 elements of the packed copy per block... see
 * salt_pack_job_blocks).  Replaces ~100 tiny launches per optimizer step by one. */
typedef struct {
    const void* jobs;
    const int* job_block0;
    int njobs;
    int total_blocks;
    int dtype;
} salt_pack_batched_args;
int salt_pack_batched(const salt_pack_batched_args*, void* stream);
int salt_pack_job_blocks(const salt_pack_conv_weight_args*);

/* ------------------------------------------------------------------ thin-channel direct convolutions
 * First layer (Cin <= 4: ResNet stem conv7x7 s2 p3, encoders.py:23-31 / torchvision; vanilla U-Net's
 * first 3x3) and heads with Cout <= 4 (final 1x1 unet.py:84-87, unet_models.py:138; sSE base.py:110).
 * These are not dense contractions; they run on the vector ALUs. */
typedef struct {
    int dtype;                /* dtype of y */
    const float* x;           /* input image batch, fp32 NCHW [B,Cin,H,W] (the reference batch contract) */
    int B;
    int Cin;
    int H;
    int W;
    const float* w;           /* fp32 master weight [Cout][Cin][K][K] */
    int K;
    int stride;
    int pad;                  /* zero padding */
    salt_view y;              /* [B,OH,OW,Cout] */
    const float* bias;
    const float* scale;
    const float* shift;
    int relu;
    float* stats;
    float* stats_cnt;
} salt_conv_first_args;
int salt_conv_first(const salt_conv_first_args*, void* stream);
int salt_conv_first_stats_parts(const salt_conv_first_args*);

typedef struct {
    int dtype;                /* dtype of dy */
    const float* x;
    int B;
    int Cin;
    int H;
    int W;
    int K;
    int stride;
    int pad;
    salt_view dy;
    float* partials;          /* [nparts][Cout*Cin*K*K] */
    int nparts;               /* as returned by salt_conv_first_wgrad_parts */
    float* grad;              /* [Cout][Cin][K][K] */
    int accumulate;
} salt_conv_first_wgrad_args;
int salt_conv_first_wgrad(const salt_conv_first_wgrad_args*, void* stream);
int salt_conv_first_wgrad_parts(const salt_conv_first_wgrad_args*);

/* ResNet stem on the matrix cores: conv KxK stride 2 (K odd, pad K/2; 7x7 s2 p3 in torchvision) over Cin <= 4 channels
 * == a ((K+1)/2)^2-tap stride-1 convolution over the 2x2 space-to-depth image z[b,Y,X,(ph*2+pw)*Cin+c] = x[b,c,2Y+ph,2X+pw]
 * (4*Cin channels zero-padded to 16).  These three helpers move data / weights / weight-gradients between the two forms;
 * the convolution itself and its weight gradient are salt_conv / salt_conv_wgrad on z. */
typedef struct {
    int dtype;
    const float* x;           /* fp32 NCHW [B,Cin,H,W], H and W even */
    int B;
    int Cin;
    int H;
    int W;
    salt_view z;              /* [B,H/2,W/2,16] */
} salt_s2d_args;
int salt_s2d(const salt_s2d_args*, void* stream);

typedef struct {
    int dtype;                /* of the packed copy */
    const float* w;           /* [Cout][Cin][K][K] fp32 master */
    int Cout;
    int Cin;
    int K;
    void* wp;                 /* packed [1][T*T][Cout][KC], T = (K+1)/2, tap t = (dh+T/2)*T + (dw+T/2) */
} salt_pack_stem_weight_args;
int salt_pack_stem_weight(const salt_pack_stem_weight_args*, void* stream);

typedef struct {
    const float* g16;         /* [Cout][16][T][T] fp32: weight gradient in the space-to-depth form */
    int Cout;
    int Cin;
    int K;
    float* grad;              /* [Cout][Cin][K][K] */
    int accumulate;
} salt_stem_grad_unfold_args;
int salt_stem_grad_unfold(const salt_stem_grad_unfold_args*, void* stream);

typedef struct {
    int dtype;
    salt_view x;
    const float* w;           /* [Cout][Cin] fp32, Cout <= 4 */
    const float* bias;        /* [Cout] or NULL */
    int Cout;
    float* y_nchw;            /* fp32 [B,Cout,H,W] or NULL */
    salt_view y;              /* NHWC output (used when y_nchw == NULL) */
} salt_head1x1_args;
int salt_head1x1(const salt_head1x1_args*, void* stream);

typedef struct {
    int dtype;
    salt_view x;
    const float* w;
    int Cout;
    const float* dy_nchw;     /* fp32 [B,Cout,H,W] */
    salt_view dx;             /* grad wrt x */
    int accumulate;
    float* partials;          /* [nparts][Cout*(Cin+1)] */
    int nparts;               /* as returned by salt_head1x1_bwd_parts */
    float* gw;                /* [Cout][Cin] */
    float* gb;                /* [Cout] or NULL */
} salt_head1x1_bwd_args;
int salt_head1x1_bwd(const salt_head1x1_bwd_args*, void* stream);
int salt_head1x1_bwd_parts(const salt_head1x1_bwd_args*);

/* ------------------------------------------------------------------ BatchNorm2d (+ReLU, +residual)
 * torch semantics (eps 1e-5, momentum 0.1, biased batch variance for normalisation, unbiased for the
 * running estimate): unet_models.py:25,45,61; base.py:23; torchvision BasicBlock/Bottleneck. */
typedef struct {
    const float* stats;       /* [nparts][2][C] */
    const float* stats_cnt;   /* [nparts] */
    int nparts;
    int C;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    int64_t* num_batches_tracked;   /* may be NULL */
    float momentum;
    float eps;
    float* mean;              /* out [C] */
    float* invstd;            /* out [C] */
    float* scale;             /* out [C]: gamma*invstd */
    float* shift;             /* out [C]: beta - mean*scale */
} salt_bn_finalize_args;
int salt_bn_finalize(const salt_bn_finalize_args*, void* stream);
/* floats the `stats` workspace must hold for nparts partials of C channels (partials + chunk heads) */
int64_t salt_bn_stats_floats(int nparts, int C);

typedef struct {              /* eval mode: scale/shift from running statistics */
    int C;
    const float* gamma;
    const float* beta;
    const float* running_mean;
    const float* running_var;
    float eps;
    float* scale;
    float* shift;
} salt_bn_fold_args;
int salt_bn_fold(const salt_bn_fold_args*, void* stream);

typedef struct {              /* a = relu?( y*scale + shift (+ res) ) */
    int dtype;
    salt_view y;
    const float* scale;       /* NULL = identity */
    const float* shift;
    salt_view res;            /* res.p == NULL: none */
    int relu;
    salt_view a;
    /* Consumer-side BatchNorm finalize: fin_acc != NULL (the [8][2 C + 1] fp64 shards a salt_conv launch with fin_acc and NO
     * fin_ticket added its statistics to; fin = the const salt_bn_finalize_args* of the layer): scale / shift are ignored - every
     * workgroup derives them from the shards, workgroup 0 stores mean / invstd / scale / shift and updates the running statistics.
     * The shards are NOT cleared (salt_zero them before the producer's next launch). */
    const void* fin;
    const double* fin_acc;
} salt_affine_act_args;
int salt_affine_act(const salt_affine_act_args*, void* stream);

typedef struct {              /* backward of a = relu?(bn(y) (+res)) in train mode */
    int dtype;
    salt_view da;             /* grad wrt a */
    salt_view a;              /* forward output (ReLU mask).  a.p == NULL: relu == 0, or (no residual) recompute the mask from y */
    salt_view y;              /* conv output (pre-BN) */
    int relu;
    const float* mean;
    const float* invstd;
    const float* gamma;
    const float* beta;        /* needed when the mask is recomputed from y */
    float* partials;          /* [nparts][2][C] workspace */
    int nparts;               /* as returned by salt_bn_bwd_parts */
    float* dgamma;            /* [C] */
    float* dbeta;             /* [C] */
    int accumulate_param_grads;
    float* coef;              /* [3][C] workspace: k, c1, c2 */
    salt_view dy;             /* out: grad wrt y */
    salt_view dres;           /* out: grad wrt residual (masked da); dres.p == NULL: none */
    int accumulate_dres;
    int partials_ready;       /* 1: `partials` ([nparts][2][C], any nparts >= 1) was filled by the producer of da (salt_conv_args.bnb_*) */
                              /* 2: the producer also finalized (salt_conv_args.bnb_fin): coef / dgamma / dbeta are ready, only the apply pass runs */
                              /* 3: the producer added the sums to fin_acc (salt_conv_args.bnb_acc without ticket): the apply pass finalizes them */
    double* fin_acc;          /* partials_ready == 0 and fin_acc != NULL: the reduction pass accumulates into [8][2][C] fp64 shards and its last block */
    const float* da_bias;     /* NULL, or [B][C]: a per-image, per-channel constant added to da wherever it is read (partials_ready 0, or 3 when the sums in fin_acc already include it) */
    uint32_t* fin_ticket;     /* finalizes (no partials, no finalize launch); both zero before the first call, left zero (see salt_conv_args.fin).
                               * fin_ticket == NULL: no in-launch finalize - the apply pass finalizes the shards; the caller clears them */
    /* round 6 - secondary sums (apply pass with fin_acc and no ticket, dres written without accumulate_dres): the residual branch is the
     * output of ANOTHER train-mode BatchNorm without ReLU (a ResNet projection shortcut, torchvision layout through
     * architectures/encoders.py:38-45) whose dL/da is exactly the dres this call stores - the pass also takes that layer's
     * BatchNorm-backward sums (sum dres, sum dres xhat_sec with xhat_sec from sec_y / sec_mean / sec_invstd) into the fp64 shards
     * sec_acc [8][2][C] (zero on entry), so the shortcut's own salt_bn_bwd runs with partials_ready 3 and no reduction pass */
    salt_view sec_y;
    const float* sec_mean;
    const float* sec_invstd;
    double* sec_acc;
} salt_bn_bwd_args;
int salt_bn_bwd(const salt_bn_bwd_args*, void* stream);
int salt_bn_bwd_parts(const salt_bn_bwd_args*);

/* ------------------------------------------------------------------ train-mode BatchNorm + ReLU + 1x1 logit head in ONE pass (round 6)
 * The reference's `final` Sequential (architectures/unet.py:84-87: Conv2dBnRelu -> nn.Conv2d(C, num_classes, 1)) in TRAINING: the
 * activation a = relu(bn(y)) has exactly one reader, the 1x1 head, and the head's input gradient exactly one reader, the BatchNorm
 * backward - so neither tensor needs to exist.  salt_head_bn finalizes the statistics shards of y's producer (consumer side, like
 * salt_affine_act with fin_acc), applies scale / shift / ReLU to y on the way in, rounds to the storage dtype exactly where
 * salt_affine_act would have stored, and writes the fp32 NCHW logits.  salt_head_bn_bwd is the matching backward in two passes
 * over y: (1) head weight / bias gradients and the BatchNorm-backward sums of da = W^T dlogits (recomputed per pixel: rank
 * num_classes) into fp64 shards, (2) dy = k (mask da - c1 - xhat c2) with the sums finalized in every workgroup's prologue (like
 * salt_bn_bwd with partials_ready 3).  Replaces salt_affine_act + salt_head1x1 and salt_head1x1_bwd + salt_bn_bwd for that layer:
 * 0.47 GB of traffic -> 0.21 GB at the C2 shape. */
typedef struct {
    int dtype;
    salt_view y;              /* RAW convolution output (pre-BatchNorm) [B,H,W,C]; C / (16-byte piece) a power of two <= 64 */
    const void* fin;          /* const salt_bn_finalize_args* of the layer */
    const double* fin_acc;    /* [8][2 C + 1] fp64 statistics shards the producer of y added to (not cleared here) */
    int relu;
    const float* w;           /* head weight [Cout][C] fp32, Cout <= 4 */
    const float* bias;        /* [Cout] or NULL */
    int Cout;
    float* y_nchw;            /* out: fp32 logits [B,Cout,H,W] */
} salt_head_bn_args;
int salt_head_bn(const salt_head_bn_args*, void* stream);

typedef struct {
    int dtype;
    salt_view y;              /* raw convolution output, as in the forward call */
    int relu;
    const float* mean;        /* [C] batch statistics the forward call stored */
    const float* invstd;
    const float* gamma;
    const float* beta;
    const float* w;           /* head weight [Cout][C] */
    int Cout;
    const float* dy_nchw;     /* d loss / d logits, fp32 [B,Cout,H,W] */
    float* partials;          /* workspace [nparts][Cout (C + 1)] */
    int nparts;               /* as returned by salt_head_bn_bwd_parts */
    float* gw;                /* out: head weight gradient [Cout][C] */
    float* gb;                /* out: head bias gradient [Cout] or NULL */
    double* fin_acc;          /* [8][2][C] fp64 shards of the BatchNorm-backward sums, zero on entry (salt_zero), left filled */
    float* dgamma;            /* out [C] */
    float* dbeta;             /* out [C] */
    float* coef;              /* out [3][C] (k, c1, c2) or NULL */
    salt_view dy;             /* out: grad wrt y */
} salt_head_bn_bwd_args;
int salt_head_bn_bwd(const salt_head_bn_bwd_args*, void* stream);
int salt_head_bn_bwd_parts(const salt_head_bn_bwd_args*);

typedef struct {              /* backward of a = relu(y (+res)) without BN, or plain masked copy */
    int dtype;
    salt_view da;
    salt_view a;
    salt_view dy;
    int accumulate;
} salt_relu_bwd_args;
int salt_relu_bwd(const salt_relu_bwd_args*, void* stream);

/* ------------------------------------------------------------------ pooling / resize / layout */
typedef struct {              /* nn.MaxPool2d(2,2) (unet_models.py:119) */
    int dtype;
    salt_view x;
    salt_view y;
} salt_maxpool2_args;
int salt_maxpool2(const salt_maxpool2_args*, void* stream);

typedef struct {              /* gradient to the first maximal element of each window (torch rule) */
    int dtype;
    salt_view x;
    salt_view dy;
    salt_view dx;
    int accumulate;
} salt_maxpool2_bwd_args;
int salt_maxpool2_bwd(const salt_maxpool2_bwd_args*, void* stream);
/* nn.MaxPool2d(3, stride 2, padding 1): the ResNet stem pool of ResNetEncoders(pool0=True) (architectures/encoders.py:23-27); same
 * argument structs, y / dy are ceil(H/2) x ceil(W/2); the gradient goes to the first maximum of every (overlapping) window */
int salt_maxpool3s2(const salt_maxpool2_args*, void* stream);
int salt_maxpool3s2_bwd(const salt_maxpool2_bwd_args*, void* stream);

typedef struct {              /* nn.AvgPool2d(2,2) (unet.py:62); bwd: dx = dy/4 */
    int dtype;
    salt_view x;
    salt_view y;
    int backward;             /* 0: y = avg(x);  1: x(+)= y/4 broadcast (x is the grad output) */
    int accumulate;
} salt_avgpool2_args;
int salt_avgpool2(const salt_avgpool2_args*, void* stream);

typedef struct {              /* bilinear xR (nn.Upsample / F.upsample(mode='bilinear'), base.py:70, unet.py:103-106) */
    int dtype;
    salt_view x;              /* low resolution  [B,H,W,C] */
    salt_view y;              /* high resolution [B,H*R,W*R,C] */
    int R;
    int backward;             /* 0: y = up(x);  1: x (+)= up^T(y) */
    int accumulate;
    void* tmp;                /* backward, R >= 4: workspace of B*(H*R)*W*roundup(C) elements for the separable adjoint, or NULL */
    int align_corners;        /* 0: src = (dst + 0.5) / R - 0.5 clamped at 0 (torch >= 0.4 default: what the oracle / goldens executed);
                               * 1: src = dst * (H - 1) / (R H - 1) - how torch 0.3.1, the reference's pinned version (environment.yml:17),
                               * evaluated the same call: use it for checkpoints trained in the reference's own environment */
} salt_bilinear_args;
int salt_bilinear(const salt_bilinear_args*, void* stream);

/* Hypercolumn rows (architectures/unet.py:101-107: torch.cat of the decoder maps up-sampled x2 / x4 / x8 / x16): ONE pass writes the
 * nlev up-sampled levels of every output pixel next to each other - channels [c0 + k*C, c0 + (k+1)*C) of y's pixel row from x[k] at
 * factor R[k] - instead of one salt_bilinear launch per level into a channel slice (128-byte pieces at the row pitch; at the C4 size
 * that streams a 2.7 GB buffer four times at 0.6 TB/s).  Same arithmetic per value as salt_bilinear: bit-identical. */
typedef struct {
    int dtype;
    int nlev;                 /* 1..4 */
    salt_view x[4];           /* [B, H / R[k], W / R[k], C] */
    int R[4];
    salt_view y;              /* [B, H, W, *] view whose channels [c0, c0 + nlev*C) are written (y.C = nlev*C, y.p at channel c0) */
    int align_corners;
} salt_hyper_rows_args;
int salt_hyper_rows(const salt_hyper_rows_args*, void* stream);

/* Factored hypercolumn (architectures/unet.py:101-109 + architectures/base.py:21-37).  The reference builds
 *   hyper = cat([dec1, up2(dec2), up4(dec3), up8(dec4), up16(dec5)])  and runs  Conv2dBnRelu(5 C, C): replicate pad (top 2, right 2), 3x3.
 * A 1x1 contraction over channels commutes with bilinear up-sampling and with the tap shift, so for an up-sampled level k
 *   conv3x3(pad(up_R(x_k)))[o, Y, X] = sum_{t = (kh, kw)} up_R(z_k[t])[o, max(Y + kh - 2, 0), min(X + kw, W - 1)],   z_k[t] = W[:, level k, kh, kw] x_k
 * with z_k computed at LOW resolution by one 1x1 convolution (Cin -> 9 Cout; salt_conv + the tap-GEMM pack above).  This operator is the
 * remaining stencil: forward   y = y_in + sum_k sum_t shift_t(up_R[k](z[k][t]))   (+ train-mode BatchNorm statistics of y through
 * fin_acc, or the eval epilogue relu?(y scale + shift)); backward (backward = 1): z[k] (gradients, overwritten) = adjoint of that sum
 * applied to y (the gradient of the convolution output).  The up-sampled level, its 9-tap convolution (R^2 times the MACs of the 1x1
 * at low resolution) and its weight-gradient slab never exist.  z[k] is [B, H / R[k], W / R[k], 9 C], channel t C + o = tap t = 3 kh + kw.
 * Views 16-byte aligned, C and every pixel stride multiples of 8; backward: W <= 256. */
typedef struct {
    int dtype;
    int nlev;                 /* 1..4 */
    salt_view z[4];
    int R[4];                 /* H / z[k].H, powers of two in 4 .. 32 (a x2 level gains nothing: its z is 9/4 of the up-sampled plane) */
    salt_view y_in;           /* forward: [B,H,W,C] partial sum (the convolution over the full-resolution operands); p == NULL: zero */
    salt_view y;              /* forward: out [B,H,W,C] (may alias y_in);  backward: the gradient read */
    int backward;
    int align_corners;        /* as salt_bilinear_args.align_corners */
    const float* scale;       /* forward, eval: y = relu?(y scale[c] + shift[c]); NULL: none */
    const float* shift;
    int relu;
    double* fin_acc;          /* forward, train: [8][2 C + 1] fp64 shards (sum, sum of squares, count) of y, as salt_conv_args.fin_acc without a ticket */
    /* forward, eval: the logit head nn.Conv2d(C, head_cout, 1) (architectures/unet.py:84-87: final = Sequential(Conv2dBnRelu, Conv2d 1x1),
     * its only consumer) applied to the epilogue's values as they would be STORED (rounded to dtype) - salt_head1x1's arithmetic and
     * summation order, bit-identical to running it on y.  With head_y_nchw set, y.p may be NULL: y (2 x 2.1 GB of traffic at
     * [64,256,256,256]) is then never written or read.  C = 64, 128, 256 (bf16: also 512); head_cout 1..2. */
    const float* head_w;      /* [head_cout][C] fp32 or NULL */
    const float* head_b;      /* [head_cout] or NULL */
    float* head_y_nchw;       /* fp32 [B, head_cout, H, W] */
    int head_cout;
    float* head_ws;           /* C > 64: fp32 workspace of B (C / 64) head_cout H W elements (per-channel-block values, joined in salt_head1x1's tree order) */
} salt_hyper_stencil_args;
int salt_hyper_stencil(const salt_hyper_stencil_args*, void* stream);

typedef struct {              /* adjoint of replicate padding: fold an extended grad back (base.py:21-27) */
    int dtype;
    salt_view xp;             /* [B,H+top+bottom,W+left+right,C] */
    int top;
    int bottom;
    int left;
    int right;
    salt_view x;              /* [B,H,W,C] */
    int accumulate;
} salt_pad_fold_args;
int salt_pad_fold(const salt_pad_fold_args*, void* stream);

typedef struct {              /* second half of the fused fold: y edge pixels += the pad ring a fold-mode salt_conv left in `strip` */
    int dtype;
    const void* strip;
    int strip_cs;
    int top;
    int bottom;
    int left;
    int right;
    salt_view x;              /* [B,H,W,C]: the unpadded gradient the convolution wrote */
} salt_pad_fold_strip_args;
int salt_pad_fold_strip(const salt_pad_fold_strip_args*, void* stream);
/* pixels per image of the ring: (top+bottom)*(W+left+right) + H*(left+right) */
int64_t salt_fold_strip_pixels(int H, int W, int top, int bottom, int left, int right);

typedef struct {              /* y = a + b (or y += a when b.p == NULL); also plain copy/cast between views */
    int dtype;
    salt_view a;
    salt_view b;
    salt_view y;
    int accumulate;
} salt_add_args;
int salt_add(const salt_add_args*, void* stream);

typedef struct {              /* fp32 NCHW [B,C,H,W] <-> NHWC view (dtype) */
    int dtype;
    float* nchw;
    salt_view nhwc;
    int to_nhwc;              /* 1: nchw -> nhwc, 0: nhwc -> nchw */
} salt_layout_args;
int salt_layout(const salt_layout_args*, void* stream);

/* ------------------------------------------------------------------ squeeze-excitation (base.py:82-117)
 * out = relu( x*cSE(x) + x*sSE(x) ),  cSE = sigmoid(W2 relu(W1 gap(x)+b1)+b2),  sSE = sigmoid(w.x+b) */
typedef struct {
    int dtype;
    salt_view x;
    const float* w1;          /* [R][C] */
    const float* b1;          /* [R] */
    const float* w2;          /* [C][R] */
    const float* b2;          /* [C] */
    int R;
    const float* ws;          /* [C] (sSE 1x1 conv weight) */
    const float* bs;          /* [1] */
    float* gap_partials;      /* [B][nparts][C] workspace */
    int nparts;               /* as returned by salt_scse_parts */
    float* gap;               /* [B][C]  (saved for backward) */
    float* hidden;            /* [B][R]  (saved, post-ReLU) */
    float* gate_c;            /* [B][C]  (saved) */
    float* gate_s;            /* [B][H*W] (saved) */
    salt_view y;
    double* gap_acc;          /* NULL, or [B][C] fp64 ZEROED by the caller: the pooling pass adds its channel sums there (atomics) and the
                                 apply pass derives the gates of its image in its prologue - two launches instead of three; gap_partials unused */
    /* round 6 - x is the RAW output of a train-mode convolution whose BatchNorm + ReLU (base.Conv2dBnRelu, architectures/base.py:29-36)
     * this launch applies on the way in: in_fin = the layer's const salt_bn_finalize_args*, in_fin_acc = the [8][2 C + 1] fp64 shards
     * its producer added to (gap_acc path only).  The pooling pass finalizes the statistics (workgroup 0 stores mean / invstd / scale /
     * shift and the running statistics), both passes evaluate a = round(relu(y scale + shift)) per element - the activation tensor
     * and the salt_affine_act launch that would have written it do not exist.  salt_scse_bwd gets the same transform (in_scale ..). */
    const void* in_fin;
    const double* in_fin_acc;
    int in_relu;
} salt_scse_args;
int salt_scse(const salt_scse_args*, void* stream);
int salt_scse_parts(const salt_scse_args*);

typedef struct {
    int dtype;
    salt_view x;
    salt_view y;              /* forward output (ReLU mask) */
    salt_view dy;
    const float* w1;
    const float* w2;
    int R;
    const float* ws;
    const float* gap;
    const float* hidden;
    const float* gate_c;
    const float* gate_s;
    float* partials;          /* [B][nparts][2C+1] workspace */
    int nparts;
    float* g_w1;
    float* g_b1;
    float* g_w2;
    float* g_b2;
    float* g_ws;
    float* g_bs;
    float* dgap;              /* [B][C] workspace */
    salt_view dx;
    int accumulate;
    double* acc;              /* NULL, or [B][2C+1] fp64 ZEROED by the caller: the first pass adds its per-part sums there (atomics) and the
                                 FC backward reads them - no parts-reduction launch; partials unused */
    int skip_bcast;           /* 1: dx is left WITHOUT the channel-SE term dgap[b][c]; the consumer of dx adds it on the fly
                                 (salt_bn_bwd_args.da_bias = dgap) - one full pass over dx less */
    const float* in_scale;    /* != NULL: x is the raw convolution output of salt_scse_args.in_fin; the forward call stored scale / shift */
    const float* in_shift;
    int in_relu;
    /* bnb_acc != NULL (needs acc, in_scale, skip_bcast): dx is dL/da of the Conv-BN-ReLU layer that produced x (minus dgap, which that
     * layer's salt_bn_bwd adds through da_bias) - the first pass also takes that layer's BatchNorm-backward sums per image and the
     * FC backward writes (sum, sum xhat) INCLUDING the dgap term to shard 0 of bnb_acc ([8][2][C] fp64, zero on entry): the layer's
     * salt_bn_bwd then runs with partials_ready 3 + da_bias and no reduction pass.  acc must then hold [B][6 C + 1] doubles. */
    const float* bn_mean;
    const float* bn_invstd;
    double* bnb_acc;
    /* 1 (needs acc): salt_scse_bwd leaves the parameter gradients g_w1 .. g_bs to salt_scse_fc_grads (same arguments, any stream ordered
     * behind this call - the weight-gradient queue) and computes dgap (+ the bnb_acc sums, spread over the shards by image) with one
     * workgroup per image instead of one workgroup for the batch: the parameter gradients are nobody's input before the optimizer */
    int defer_param_grads;
} salt_scse_bwd_args;
int salt_scse_bwd(const salt_scse_bwd_args*, void* stream);
int salt_scse_fc_grads(const salt_scse_bwd_args*, void* stream);

/* ------------------------------------------------------------------ losses
 * Lovasz hinge, per image, both channels flattened together, F.elu variant
 * (lovasz_losses.py:81-115,21-33; models.py:326-328).  One workgroup per image: LSD radix sort of
 * the hinge errors (stable; ties keep flat-index order), label scan, Jaccard gradient, dot. */
typedef struct {
    const float* logits;      /* fp32 NCHW [B,C,H,W] */
    const float* target;      /* fp32 NCHW [B,C,H,W] in {0,1} */
    int B;
    int P;                    /* elements per image = C*H*W */
    uint32_t* ws_keys;        /* [2][B][P] workspace */
    uint32_t* ws_vals;        /* [2][B][P] workspace */
    float* loss_per_image;    /* [B] */
    float* loss;              /* [1]: mean over images * loss_scale */
    float* dlogits;           /* fp32 NCHW or NULL (forward only) */
    float loss_scale;         /* loss weight (models.py:194) and 1/world for data parallel */
    uint32_t* ws_split;       /* NULL, or [B * salt_lovasz_split_words(P)] words: images of >= 4096 elements are then sorted by several
                                 workgroups each (segments of 2048 positions, 1 + 4 + 1 launches); same positions, ties and gradients */
} salt_lovasz_args;
int salt_lovasz_hinge(const salt_lovasz_args*, void* stream);
/* words per image of salt_lovasz_args.ws_split (0: the split form does not apply to this size) */
int64_t salt_lovasz_split_words(int P);

/* 0.2*mean_c dice(sigmoid) + 0.9*BCEWithLogits (models.py:315-340,361-388), sums over the whole batch */
typedef struct {
    const float* logits;
    const float* target;
    int B;
    int C;
    int HW;
    float dice_weight;
    float bce_weight;
    float* partials;          /* [nparts][4] workspace */
    int nparts;               /* as returned by salt_bce_dice_parts */
    float* sums;              /* [3C+1] workspace (kept for backward) */
    float* loss;              /* [1] */
    float* dlogits;           /* or NULL */
    float loss_scale;
} salt_bce_dice_args;
int salt_bce_dice(const salt_bce_dice_args*, void* stream);
int salt_bce_dice_parts(const salt_bce_dice_args*);

/* ------------------------------------------------------------------ optimizer (models.py:74-75,289-297)
 * torch.optim.Adam with L2-in-gradient weight decay over one flat fp32 parameter buffer. */
typedef struct {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    const float* hyper;       /* device [8]: lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, grad_scale */
} salt_adam_args;
int salt_adam(const salt_adam_args*, void* stream);

/* Adam + L2 AND the bf16 forward weight packs in one pass (round 5).  The forward packs of the 3x3 / 1x1 layers (97 % of a ResNet's
 * parameters) are re-derived from the fp32 masters after every optimizer step (salt_pack_batched: 405 MB of traffic, the first launch of
 * the next step on the critical queue).  Here the thread that updates 8 input channels x all taps of one output channel - 8 KK contiguous
 * floats of the master - also stores their KK 16-byte packed pieces: the same values salt_pack_batched would write, bit for bit.
 * `jobs` are the salt_pack_conv_weight_args of those layers (salt_pack_job_is_vec(job) != 0, each weight at most once, every `w` inside
 * [param, param + n)); `rest` lists the ranges of the flat buffer no job covers (everything else takes the plain update of salt_adam). */
typedef struct {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
    const float* hyper;       /* as salt_adam_args.hyper */
    const void* jobs;         /* DEVICE array of salt_pack_conv_weight_args */
    const int* job_block0;    /* DEVICE [njobs + 1]: prefix of salt_pack_job_blocks */
    int njobs;
    int pack_blocks;
    const int64_t* rest;      /* DEVICE [nrest][2]: (first element, element count), both multiples of 4 */
    const int* rest_block0;   /* DEVICE [nrest + 1]: prefix of ceil(count / 1024) */
    int nrest;
    int rest_blocks;
} salt_adam_pack_args;
int salt_adam_pack(const salt_adam_pack_args*, void* stream);
/* != 0 for the jobs salt_adam_pack can write on the way: salt_pack_batched's vector path, 3x3 layers (KH KW = 9) */
int salt_pack_job_is_vec(const salt_pack_conv_weight_args*);

typedef struct {              /* advance step counter and bias corrections on device (graph-replay safe) */
    float* hyper;             /* as above */
    int64_t* step;            /* device [1] */
} salt_adam_tick_args;
int salt_adam_tick(const salt_adam_tick_args*, void* stream);

typedef struct {
    void* p;
    int64_t bytes;
} salt_zero_args;
int salt_zero(const salt_zero_args*, void* stream);

/* ------------------------------------------------------------------ inference epilogue
 * sigmoid -> inverse flip (TTA) -> mean over variants (models.py:143-144, augmentation.py:156-163,
 * loaders.py:722-760);  flips are index permutations. */
typedef struct {
    const float* logits;      /* fp32 NCHW [V*B, C, H, W]: variant-major */
    int V;
    int B;
    int C;
    int H;
    int W;
    const int* flip_ud;       /* host arrays [V] */
    const int* flip_lr;
    float* prob;              /* fp32 NCHW [B,C,H,W] */
    const int* rot;           /* host array [V] or NULL: quarter turns k of the variant's forward np.rot90 (augmentation.py:143-153; H == W);
                               * the inverse applied here is rot90(-k), then fliplr, then flipud (augmentation.py:156-163) */
    int method;               /* aggregation over the variants (loaders.py:727-735): 0 mean, 1 max, 2 min, 3 gmean = exp(mean(log p)) */
} salt_tta_mean_args;
int salt_tta_mean(const salt_tta_mean_args*, void* stream);

typedef struct {              /* x_out[b] = flip(x[b]) for NCHW fp32 batches (TTA forward transform) */
    const float* x;
    int B;
    int C;
    int H;
    int W;
    int flip_ud;
    int flip_lr;
    float* y;
} salt_flip_args;
int salt_flip(const salt_flip_args*, void* stream);

/* ------------------------------------------------------------------ on-device input pipeline
 * gray tile -> [resize: cubic | bilinear] -> edge pad -> Normalize -> AddDepthChannels; mask -> [resize] -> pad -> one-hot
 * (loaders.py:603-612,763-769; augmentation.py:79-96,247-284; utils.py:494-500).  Normalisation divides by std exactly as
 * torchvision ((g - mean) / std is evaluated as (g - mean) * (1 / std), agreement 1 ulp). */
typedef struct {
    const void* img;          /* [B,h,w] uint8 (img_is_u8) or fp32 in [0,1] */
    int img_is_u8;
    const uint8_t* mask;      /* [B,h,w] {0,1} or NULL */
    int B;
    int h;
    int w;
    int resize_h;             /* 0 = no resize */
    int resize_w;
    int top;                  /* rows / columns of edge padding before the tile; the rest of [H,W] is padded after it */
    int left;
    int H;
    int W;
    int channels;             /* 1 (gray only) | 3 (gray, depth ramp, gray*ramp) */
    float mean[3];
    float std[3];
    float* x;                 /* out fp32 NCHW [B,channels,H,W] */
    float* target;            /* out fp32 NCHW [B,2,H,W] one-hot {background, salt}; required when mask != NULL */
    int interpolation;        /* of the resize.  1: cubic - cv2.INTER_CUBIC as imgaug 0.2.5's iaa.Scale default (augmentation.py:79-85 passes no
                               * interpolation; environment.yml:15-16): Keys kernel a = -0.75, half-pixel centres, replicated border; a uint8
                               * tile is rounded back to the uint8 grid with saturation (cv2's uint8 output) and the {0,1} mask goes through the
                               * same resize, then round-half-up (the FLOAT form of the filter).  2 (uint8 tiles only): the same filter as
                               * opencv_python 3.4.0.12 evaluates it on CV_8U - per-axis coefficients rounded to 11-bit fixed point
                               * (cvRound(c * 2048), each on its own), integer horizontal / vertical sums, saturate((v + 2^21) >> 22);
                               * differs from 1 by one LSB on a few percent of the pixels.  0: bilinear, half-pixel centres, nearest for
                               * the mask (rounds 1 - 2 default) */
} salt_preprocess_args;
int salt_preprocess(const salt_preprocess_args*, void* stream);

/* centre crop + binarize one class of a probability map (postprocessing.py:24-43, utils.py:308-313): mask = prob[cls] > threshold */
typedef struct {
    const float* prob;        /* fp32 NCHW [B,C,H,W] */
    int B;
    int C;
    int H;
    int W;
    int cls;
    int top;                  /* crop window [top, top+h) x [left, left+w) */
    int left;
    int h;
    int w;
    float threshold;
    uint8_t* mask;            /* out [B,h,w] in {0,1} */
} salt_crop_threshold_args;
int salt_crop_threshold(const salt_crop_threshold_args*, void* stream);

/* validation metric counts for a whole threshold sweep in one pass (callbacks.py:503-513, metrics.py:21-66): per image and
 * threshold t: |pred_t| and |pred_t & gt| with pred_t = (double)prob[cls] > t on the cropped window; plus |gt| per image.
 * IoU / IOUT (with the reference's empty-mask conventions) follow on the host from these integers. */
#define SALT_MAX_THRESHOLDS 32
typedef struct {
    const float* prob;        /* fp32 NCHW [B,C,H,W] */
    int B;
    int C;
    int H;
    int W;
    int cls;
    int top;
    int left;
    int h;
    int w;
    const uint8_t* gt;        /* [B,h,w] in {0,1} */
    int T;
    const double* thresholds; /* HOST array [T], T <= SALT_MAX_THRESHOLDS */
    int* inter;               /* out [B][T] */
    int* pred;                /* out [B][T] */
    int* gt_count;            /* out [B] */
} salt_iou_sweep_args;
int salt_iou_sweep(const salt_iou_sweep_args*, void* stream);

/* ------------------------------------------------------------------ program executor
 * A training/inference step is a static list of the operators above.  The host builds it once per
 * (model, batch shape); running it is one call.  capture/replay wraps it in a hipGraph. */
typedef int (*salt_op_fn)(const void* args, void* stream);
typedef struct {
    salt_op_fn fn;
    const void* args;
    int stream;               /* 0 = main, 1 = side, 2 = main after joining side work of this range, 3 = main after joining the side
                                 stream unconditionally (work enqueued there before the call) - salt_program_run_streams;
                                 4, 5 = auxiliary stream (salt_set_aux_stream), side stream without one */
    int reserved;
} salt_program_entry;
int salt_program_run(const salt_program_entry* entries, int n, void* stream);
int salt_program_run_range(const salt_program_entry* entries, int begin, int end, void* stream);
/* as run_range, with a HIP event pair around every entry on `stream`; ms_out[i] = elapsed ms of entry begin+i */
int salt_program_run_timed(const salt_program_entry* entries, int begin, int end, void* stream, float* ms_out);
/* Two-stream execution: entries with stream == 1 (weight-gradient kernels) are enqueued on `side`, ordered after every
 * main-stream entry that precedes them in the list (event record/wait), and `main` re-joins `side` at the end of the range.
 * A main-stream entry that depends on side entries inside the range is tagged stream == 2: main waits for the side stream
 * right before it (forward: independent branches such as the hypercolumn up-samplings; backward: nothing - side entries only
 * produce parameter gradients).  side == NULL or side == main degenerates to salt_program_run_range. */
int salt_program_run_streams(const salt_program_entry* entries, int begin, int end, void* main_stream, void* side_stream);
/* join_at_end == 0: the caller orders its consumers after BOTH streams itself (bucketed all-reduce between backward segments) */
int salt_program_run_streams_ex(const salt_program_entry* entries, int begin, int end, void* main_stream, void* side_stream, int join_at_end);
/* ... with MARKS: before entry marks[m] (ascending, begin <= marks[m] <= end) is issued, ev_main[m] is recorded on the main stream and
 * ev_side[m] on the side stream (hipEvent_t handles from salt_event_create).  The data-parallel backward (one process per GPU, replaces
 * nn.DataParallel of models.py:81-85) learns this way that an all-reduce bucket's gradients are final WITHOUT cutting the program into
 * one call per bucket (each cut cost a flush of the pending side-stream entries and the completion-signal fork hand-off). */
int salt_program_run_streams_marks(const salt_program_entry* entries, int begin, int end, void* main_stream, void* side_stream, int join_at_end,
                                   const int* marks, int nmarks, void* const* ev_main, void* const* ev_side);
/* stream-tag 4 entries (the weight-gradient slab reductions; behind the side entry in front of them, two alternating slab buffers) and
 * stream-tag 5 entries (optimizer updates of parameter ranges whose gradients are final; behind both queues) run on this stream in the
 * plain eager two-stream run, joined into the main stream at the end of the range; NULL (default): on the side stream.  Per host thread. */
int salt_set_aux_stream(void* stream);
int salt_event_create(void** event_out);            /* a hipEvent_t without timing */
int salt_event_destroy(void* event);
int salt_stream_wait_event(void* stream, void* event);
int salt_graph_capture(const salt_program_entry* entries, int n, void* stream, void** graph_exec_out);
/* whole-step capture: every salt_program_run* call between begin and end on `stream` (and on the side stream the two-stream executor
 * forks to) is recorded instead of executed; `stream` must not be the default stream */
int salt_graph_begin(void* stream);
int salt_graph_end(void* stream, void** graph_exec_out);
int salt_graph_launch(void* graph_exec, void* stream);
int salt_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* SALTNET_H */
