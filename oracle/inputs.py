"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's input preparation (never imported by the product path).

Restates, with plain torch CPU ops:
  * ImageSegmentationLoader.image_transform (loaders.py:603-612): Grayscale(3) -> ToTensor -> Normalize(ImageNet) -> AddDepthChannels
    (utils.py:494-500);  mask_transform / one-hot target (loaders.py:763-769, 186-190)
  * resize_pad_seq (augmentation.py:79-85, neptune.yaml:22-26): resize 101 -> 102, edge-pad 13 -> 128            (train)
  * pad_to_fit_net / InferencePad (augmentation.py:93-96, 247-284) with get_crop_pad_sequence (utils.py:308-313)  (inference)

Parity note (the 101 -> 102 resize only; pad geometry, normalisation and depth channels are pinned by golden F10 / the reference source):
`augmentation.py:79-85` calls `iaa.Scale({'height': ..., 'width': ...})` with no interpolation argument; the pinned imgaug==0.2.5
(environment.yml:15) declares `Scale(size, interpolation="cubic", ...)` and resizes with `cv2.resize(..., interpolation=cv2.INTER_CUBIC)`
(opencv_python==3.4.0.12), on the uint8 tile and - through the same `augment_image` call, loaders.py:137-139 + utils.py:343-344 - on the
uint8 {0,1} mask.  imgaug / cv2 are absent from this image, so that kernel cannot be executed here: PARITY UNPINNED, restated from the
public definition of INTER_CUBIC (Keys cubic convolution with a = -0.75, half-pixel centres, BORDER_REPLICATE taps, uint8 output =
saturate_cast(round(v))).  `torch.nn.functional.interpolate(mode='bicubic', align_corners=False)` implements the same definition
(A = -0.75, clamped taps) and serves as the executable restatement; `cubic_weights` below is the hand-checkable form
(KAT: t = 0.5 -> [-0.09375, 0.59375, 0.59375, -0.09375], tests/test_oracle_golden.py).
Round 4: cv2's uint8 path does NOT evaluate that filter in float - `resize_cubic_u8_fixed` below restates what opencv_python 3.4.0.12
(environment.yml:16; modules/imgproc/src/resize.cpp: HResizeCubic<uchar, int, short> + VResizeCubic<..., FixedPtCast<int, uchar, 22>>)
computes on CV_8U: float32 coefficients rounded one by one to 11-bit fixed point, integer sums, saturating shift.  It differs from
the float form by one LSB on ~4 % of the pixels of a noise image (KATs that tell the two apart: tests/test_oracle_golden.py).  Still
UNPINNED by the reference's own outputs (cv2 absent); one known residue: that release's SSE2 build runs the vertical pass of whole
8-pixel groups in float32 (round-to-nearest-even of v * 2^-22), which can differ from the integer form where v / 2^22 sits within
float rounding of a tie.  uint8 tiles (`uint8_grid=True`) take the fixed-point form, float tiles the float form;
interpolation='cubic_float' keeps round 3's float form for uint8 tiles; 'bilinear' rounds 1-2's restatement (bilinear / nearest).
"""
import numpy as np
import torch
import torch.nn.functional as F

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def crop_pad_sequence(vertical, horizontal):                  # utils.py:308-313 -> (top, right, bottom, left)
    top = int(vertical / 2)
    right = int(horizontal / 2)
    return top, right, vertical - top, horizontal - right


def cubic_weights(t, a=-0.75):
    """The four Keys cubic-convolution tap weights (taps floor(s) - 1 .. floor(s) + 2) at fraction t, as cv2.INTER_CUBIC defines them."""
    w0 = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
    w1 = ((a + 2) * t - (a + 3)) * t * t + 1
    w2 = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1
    return [w0, w1, w2, 1 - w0 - w1 - w2]


def cv_cubic_coeffs_fixed(t):
    """interpolateCubic(t) of resize.cpp in float32 (A = -0.75f, this operation order, no contraction), each coefficient rounded on its
    own to the 11-bit fixed point of the uint8 path: cvRound(c * INTER_RESIZE_COEF_SCALE), INTER_RESIZE_COEF_BITS = 11."""
    f = np.float32
    t = f(t); A = f(-0.75); one = f(1)
    x1 = f(t + one)
    c0 = f(f(f(f(f(f(A * x1) - f(f(5) * A)) * x1) + f(f(8) * A)) * x1) - f(f(4) * A))
    c1 = f(f(f(f(f(f(A + f(2)) * t) - f(A + f(3))) * t) * t) + one)
    u = f(one - t)
    c2 = f(f(f(f(f(f(A + f(2)) * u) - f(A + f(3))) * u) * u) + one)
    c3 = f(f(f(one - c0) - c1) - c2)
    return [int(np.rint(f(c * f(2048)))) for c in (c0, c1, c2, c3)]      # np.rint: round half to even = cvRound


def cv_axis_tables(n_in, n_out):
    """Per output index: first tap (floor(fx) - 1) and the four fixed-point coefficients, fx = (float)((d + 0.5) * scale - 0.5) with
    scale = 1 / ((double)n_out / n_in) as cv::resize derives it from dsize."""
    scale = 1.0 / (float(n_out) / float(n_in))
    first, coef = [], []
    for d in range(n_out):
        fx = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(fx))
        first.append(s - 1)
        coef.append(cv_cubic_coeffs_fixed(np.float32(fx - np.float32(s))))
    return first, coef


def resize_cubic_u8_fixed(img, oh, ow):
    """cv2.resize(img, (ow, oh), interpolation=cv2.INTER_CUBIC) for ONE uint8 [h, w] image, integer arithmetic as in resize.cpp."""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 2
    h, w = img.shape
    y0, yc = cv_axis_tables(h, oh)
    x0, xc = cv_axis_tables(w, ow)
    S = img.astype(np.int64)
    hs = np.zeros((h, ow), np.int64)                       # HResizeCubic: int sums of uchar x short, taps clamped to the row
    for ox in range(ow):
        for k in range(4):
            hs[:, ox] += S[:, min(max(x0[ox] + k, 0), w - 1)] * xc[ox][k]
    out = np.zeros((oh, ow), np.int64)
    for oy in range(oh):
        for k in range(4):
            out[oy] += hs[min(max(y0[oy] + k, 0), h - 1)] * yc[oy][k]
    assert np.abs(out).max() < 2 ** 31                     # resize.cpp sums in `int`
    return np.clip((out + (1 << 21)) >> 22, 0, 255).astype(np.uint8)      # FixedPtCast<int, uchar, 22>: saturate_cast<uchar>((v + 2^21) >> 22)


def resize_cubic_numpy(img, oh, ow):
    """Direct (loop) evaluation of the cubic resize of ONE [h, w] float image - the slow, hand-checkable twin of the torch call below."""
    img = np.asarray(img, dtype=np.float64)
    h, w = img.shape
    out = np.zeros((oh, ow))
    for oy in range(oh):
        fy = (oy + 0.5) * h / oh - 0.5
        y0 = int(np.floor(fy)); wy = cubic_weights(fy - y0)
        for ox in range(ow):
            fx = (ox + 0.5) * w / ow - 0.5
            x0 = int(np.floor(fx)); wx = cubic_weights(fx - x0)
            v = 0.0
            for i in range(4):
                yy = min(max(y0 - 1 + i, 0), h - 1)
                for j in range(4):
                    xx = min(max(x0 - 1 + j, 0), w - 1)
                    v += wy[i] * wx[j] * img[yy, xx]
            out[oy, ox] = v
    return out


def preprocess(img, mask, train, channels, resize=102, pad=13, divisor=64, interpolation='cubic', uint8_grid=True):
    """img: float [B,h,w] in [0,1]; mask: {0,1} [B,h,w] or None -> (X [B,channels,H,W], target [B,2,H,W] | None).
    uint8_grid: the tile is a uint8 image / 255 (the reference's case): the cubic resize rounds back onto that grid like cv2's uint8 output."""
    x = torch.as_tensor(img, dtype=torch.float32)[:, None]
    m = None if mask is None else torch.as_tensor(mask, dtype=torch.float32)[:, None]
    if train and interpolation == 'cubic' and uint8_grid:
        u8 = torch.clamp(torch.floor(x * 255.0 + 0.5), 0, 255).to(torch.uint8).numpy()          # the uint8 tile the reference's loader holds
        x = torch.from_numpy(np.stack([resize_cubic_u8_fixed(t[0], resize, resize) for t in u8]).astype(np.float32) / 255.0)[:, None]
        if m is not None:
            m8 = (m > 0.5).to(torch.uint8).numpy()
            m = torch.from_numpy(np.stack([resize_cubic_u8_fixed(t[0], resize, resize) for t in m8]).astype(np.float32))[:, None]
            m = (m > 0.5).float()
    elif train and interpolation in ('cubic', 'cubic_float'):
        x = F.interpolate(x, size=(resize, resize), mode='bicubic', align_corners=False)
        if uint8_grid:
            x = torch.clamp(torch.floor(x * 255.0 + 0.5), 0, 255) / 255.0
        if m is not None:
            m = torch.clamp(torch.floor(F.interpolate(m, size=(resize, resize), mode='bicubic', align_corners=False) + 0.5), 0, 1)
    elif train:
        x = F.interpolate(x, size=(resize, resize), mode='bilinear', align_corners=False)
        if m is not None:
            m = (F.interpolate(m, size=(resize, resize), mode='nearest') > 0.5).float()
    if train:
        pads = (pad, pad, pad, pad)                           # (left, right, top, bottom)
    else:
        h, w = x.shape[2:]
        pv = 0 if h % divisor == 0 else divisor - h % divisor
        ph = 0 if w % divisor == 0 else divisor - w % divisor
        top, right, bottom, left = crop_pad_sequence(pv, ph)
        pads = (left, right, top, bottom)
    x = F.pad(x, pads, mode='replicate')
    if m is not None:
        m = F.pad(m, pads, mode='replicate')
    H = x.shape[2]
    if channels == 1:
        X = (x - MEAN[0]) / STD[0]
    else:
        X = torch.cat([(x - MEAN[c]) / STD[c] for c in range(3)], 1)
        ramp = torch.from_numpy(np.linspace(0, 1, H)).float()  # AddDepthChannels: row constant, then ch2 = ch0 * ch1
        X[:, 1] = ramp[None, :, None]
        X[:, 2] = X[:, 0] * X[:, 1]
    T = None if m is None else torch.cat([1 - m, m], 1)
    return X, T
