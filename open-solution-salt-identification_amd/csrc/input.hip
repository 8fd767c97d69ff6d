// input.hip — on-device input pipeline.
//
// What the reference does per image on the CPU with PIL / imgaug / torchvision (loaders.py:603-612, augmentation.py:79-96,
// 247-284, utils.py:494-500, loaders.py:763-769) is one pass here:
//   gray tile [h, w] (uint8 or float in [0,1])
//     -> optional resize to [rh, rw]: cubic (cv2.INTER_CUBIC, imgaug's iaa.Scale default) or bilinear     train: 101 -> 102
//        uint8 tiles: cv2's own 11-bit FIXED-POINT evaluation of the separable cubic (interpolation 2, cubic_fixed_u8 below);
//        float tiles / interpolation 1: the float form of the same filter
//     -> edge (replicate) pad: `top` rows / `left` columns, rest to [H, W]                                train: 13; inference 13/14
//     -> Grayscale(3) + ToTensor + Normalize(mean, std) per channel
//     -> AddDepthChannels: ch1 := linspace(0, 1, H)[row], ch2 := ch0 * ch1                                (3-channel mode)
//   mask tile [h, w] -> nearest resize -> same pad -> one-hot {1 - m, m} target [2, H, W]
#include "common.h"

namespace {

struct PreKP {
    const void* img; const unsigned char* mask; float* x; float* target;
    int img_is_u8, B, h, w, rh, rw, top, left, H, W, channels, cubic;
    float mean[3], inv_std[3];
};

__device__ __forceinline__ float src_index(int dst, float scale) {      // torch area_pixel_compute_source_index, align_corners=False
    const float s = scale * ((float)dst + 0.5f) - 0.5f;
    return s < 0.f ? 0.f : s;
}

// cv2.INTER_CUBIC / imgaug 0.2.5 iaa.Scale default: Keys cubic convolution, a = -0.75, taps at floor(s) - 1 .. floor(s) + 2 of the
// half-pixel-centred source coordinate s = (dst + 0.5) * in / out - 0.5, indices clamped to the image (BORDER_REPLICATE)
__device__ __forceinline__ void cubic_w(float t, float (&w)[4]) {
    const float A = -0.75f;
    w[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
    w[3] = 1.f - w[0] - w[1] - w[2];
}

// cv2.resize(..., INTER_CUBIC) on CV_8U as opencv_python 3.4.0.12 (environment.yml:16) evaluates it (modules/imgproc/src/resize.cpp:
// resizeGeneric_<HResizeCubic<uchar, int, short>, VResizeCubic<uchar, int, short, FixedPtCast<int, uchar, 22>, ...>>), restated:
//   per axis   fx = (float)((d + 0.5) * scale - 0.5) with scale = 1 / ((double)out / in);  s = floor(fx);  t = fx - s   (float)
//              coefficients interpolateCubic(t) in float32 (A = -0.75f, the four expressions below, no FMA contraction), each
//              rounded on its own to a short: cvRound(c * 2048)   (their sum is 2047 .. 2049)
//   horizontal int sums of uchar x short over taps s - 1 .. s + 2 (indices clamped = BORDER_REPLICATE)
//   vertical   int sum of those x short, then saturate_cast<uchar>((v + (1 << 21)) >> 22)
// (the SSE2 build of that release runs the vertical pass of whole 8-pixel groups in float32 with round-to-nearest-even: the same
//  value except where v / 2^22 sits within float rounding of a tie - the integer form is the documented one and is what is restated.)
__device__ __forceinline__ void cubic_coef_fixed(int d, double scale, int& s0, int (&c)[4]) {
    const float fx = (float)(((double)d + 0.5) * scale - 0.5);
    const float sf = floorf(fx);
    const float t = __fsub_rn(fx, sf);
    s0 = (int)sf - 1;
    const float A = -0.75f;
    const float x1 = __fadd_rn(t, 1.f);
    const float c0 = __fsub_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fsub_rn(__fmul_rn(A, x1), __fmul_rn(5.f, A)), x1), __fmul_rn(8.f, A)), x1), __fmul_rn(4.f, A));
    const float c1 = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(A, 2.f), t), __fadd_rn(A, 3.f)), t), t), 1.f);
    const float u = __fsub_rn(1.f, t);
    const float c2 = __fadd_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(A, 2.f), u), __fadd_rn(A, 3.f)), u), u), 1.f);
    const float c3 = __fsub_rn(__fsub_rn(__fsub_rn(1.f, c0), c1), c2);
    c[0] = __float2int_rn(__fmul_rn(c0, 2048.f)); c[1] = __float2int_rn(__fmul_rn(c1, 2048.f));
    c[2] = __float2int_rn(__fmul_rn(c2, 2048.f)); c[3] = __float2int_rn(__fmul_rn(c3, 2048.f));
}

__global__ void preprocess_kernel(PreKP p) {
    const int64_t n = (int64_t)p.B * p.H * p.W;
    const float sy = (float)p.h / (float)p.rh, sx = (float)p.w / (float)p.rw;
    for (int64_t i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
        const int X = (int)(i % p.W); int64_t r = i / p.W; const int Y = (int)(r % p.H); const int b = (int)(r / p.H);
        const int ry = min(max(Y - p.top, 0), p.rh - 1), rx = min(max(X - p.left, 0), p.rw - 1);      // edge pad = clamp
        auto px = [&](int yy, int xx) -> float {
            const int64_t o = ((int64_t)b * p.h + yy) * p.w + xx;
            return p.img_is_u8 ? (float)reinterpret_cast<const unsigned char*>(p.img)[o] * (1.f / 255.f) : reinterpret_cast<const float*>(p.img)[o];
        };
        float g, mres = -1.f;                                 // mres >= 0: the mask value out of the cubic resize
        if (p.rh == p.h && p.rw == p.w) {
            g = px(ry, rx);
        } else if (p.cubic == 2) {
            const unsigned char* im8 = reinterpret_cast<const unsigned char*>(p.img) + (int64_t)b * p.h * p.w;
            const unsigned char* mk8 = p.mask ? p.mask + (int64_t)b * p.h * p.w : nullptr;
            int y0, x0, cy[4], cx[4];
            cubic_coef_fixed(ry, 1.0 / ((double)p.rh / (double)p.h), y0, cy);
            cubic_coef_fixed(rx, 1.0 / ((double)p.rw / (double)p.w), x0, cx);
            int acc = 0, macc = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int yy = min(max(y0 + i, 0), p.h - 1);
                int row = 0, mrow = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int xx = min(max(x0 + j, 0), p.w - 1);
                    row += cx[j] * (int)im8[yy * p.w + xx];
                    if (mk8) mrow += cx[j] * (int)(mk8[yy * p.w + xx] != 0);  // the uint8 mask goes through the same augment_image call, binarised first like the float path / oracle.inputs (a 0 / 255 mask must not dilate)
                }
                acc += cy[i] * row; macc += cy[i] * mrow;
            }
            g = (float)min(max((acc + (1 << 21)) >> 22, 0), 255) * (1.f / 255.f);
            mres = (float)min(max((macc + (1 << 21)) >> 22, 0), 255);
            mres = mres > 0.5f ? 1.f : 0.f;
        } else if (p.cubic) {
            const float fy = ((float)ry + 0.5f) * sy - 0.5f, fx = ((float)rx + 0.5f) * sx - 0.5f;
            const float y0f = floorf(fy), x0f = floorf(fx);
            float wy[4], wx[4];
            cubic_w(fy - y0f, wy); cubic_w(fx - x0f, wx);
            const int y0 = (int)y0f - 1, x0 = (int)x0f - 1;
            float acc = 0.f, macc = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int yy = min(max(y0 + i, 0), p.h - 1);
                float row = 0.f, mrow = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int xx = min(max(x0 + j, 0), p.w - 1);
                    row += wx[j] * px(yy, xx);
                    if (p.mask) mrow += wx[j] * (p.mask[((int64_t)b * p.h + yy) * p.w + xx] ? 1.f : 0.f);
                }
                acc += wy[i] * row; macc += wy[i] * mrow;
            }
            // the reference resizes uint8 images: cv2 writes saturate_cast<uchar>(round(value)) - back onto the uint8 grid
            g = p.img_is_u8 ? fminf(fmaxf(floorf(acc * 255.f + 0.5f), 0.f), 255.f) * (1.f / 255.f) : acc;
            mres = fminf(fmaxf(floorf(macc + 0.5f), 0.f), 1.f);
        } else {
            const float fy = src_index(ry, sy), fx = src_index(rx, sx);
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < p.h - 1 ? 1 : 0), x1 = x0 + (x0 < p.w - 1 ? 1 : 0);
            const float ly = fy - (float)y0, lx = fx - (float)x0;
            const float hy = 1.f - ly, hx = 1.f - lx;
            g = hy * (hx * px(y0, x0) + lx * px(y0, x1)) + ly * (hx * px(y1, x0) + lx * px(y1, x1));
        }
        const int64_t hw = (int64_t)p.H * p.W;
        float* xo = p.x + (int64_t)b * p.channels * hw + (int64_t)Y * p.W + X;
        const float c0 = (g - p.mean[0]) * p.inv_std[0];
        xo[0] = c0;
        if (p.channels == 3) {
            const float depth = p.H > 1 ? (float)((double)Y / (double)(p.H - 1)) : 0.f;              // np.linspace(0, 1, H)[Y]
            xo[hw] = depth;
            xo[2 * hw] = c0 * depth;
        }
        if (p.mask) {
            // nearest: torch 'nearest' picks floor(dst * in / out)
            const int my = p.rh == p.h ? ry : min((int)floorf((float)ry * sy), p.h - 1);
            const int mx = p.rw == p.w ? rx : min((int)floorf((float)rx * sx), p.w - 1);
            const float m = mres >= 0.f ? mres : (p.mask[((int64_t)b * p.h + my) * p.w + mx] ? 1.f : 0.f);
            float* to = p.target + (int64_t)b * 2 * hw + (int64_t)Y * p.W + X;
            to[0] = 1.f - m;
            to[hw] = m;
        }
    }
}

}  // namespace

extern "C" int salt_preprocess(const salt_preprocess_args* a, void* stream) {
    if (!a || !a->img || !a->x || a->B < 1 || a->h < 1 || a->w < 1 || a->H < 1 || a->W < 1 || (a->channels != 1 && a->channels != 3) ||
        a->top < 0 || a->left < 0 || (a->mask && !a->target))
        SALT_FAIL(SALT_E_BADARG, "preprocess: bad args");
    PreKP p;
    p.img = a->img; p.mask = a->mask; p.x = a->x; p.target = a->target;
    p.img_is_u8 = a->img_is_u8; p.B = a->B; p.h = a->h; p.w = a->w;
    p.rh = a->resize_h > 0 ? a->resize_h : a->h; p.rw = a->resize_w > 0 ? a->resize_w : a->w;
    p.top = a->top; p.left = a->left; p.H = a->H; p.W = a->W; p.channels = a->channels; p.cubic = a->interpolation;
    if (a->interpolation < 0 || a->interpolation > 2) SALT_FAIL(SALT_E_BADARG, "preprocess: interpolation %d (0 bilinear | 1 cubic, float | 2 cubic, cv2 fixed point)", a->interpolation);
    if (a->interpolation == 2 && !a->img_is_u8) SALT_FAIL(SALT_E_BADARG, "preprocess: the fixed-point cubic resize is cv2's uint8 path: uint8 tiles only");
    if (a->interpolation == 2 && (int64_t)a->h * a->w >= (1ll << 31)) SALT_FAIL(SALT_E_BADARG, "preprocess: tile too large");
    if (p.top + p.rh > p.H || p.left + p.rw > p.W) SALT_FAIL(SALT_E_BADARG, "preprocess: resized tile + pad offset exceeds the output");
    for (int c = 0; c < 3; ++c) {
        if (a->std[c] <= 0.f) SALT_FAIL(SALT_E_BADARG, "preprocess: std must be positive");
        p.mean[c] = a->mean[c]; p.inv_std[c] = 1.f / a->std[c];
    }
    const int64_t n = (int64_t)a->B * a->H * a->W;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(preprocess_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    SALT_CHECK_LAUNCH();
    return SALT_OK;
}
