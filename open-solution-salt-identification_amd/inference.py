"""GPU inference epilogue and validation metric (SURVEY.md §8 a18/a19 and the "next" rows f-2 / f-3).

What the reference does on the host with numpy (after copying the full fp32 logits of every batch to the CPU,
models.py:167) happens here on the device:

  test-time augmentation   loaders.py:662-682 (variants), augmentation.py:143-163 (transform / inverse),
                           loaders.py:722-760 (mean aggregation)            -> salt_flip, salt_tta_mean
  post-processing          postprocessing.py:24-43 + utils.py:308-313 (centre crop 128 -> 101, binarize)
                                                                             -> salt_crop_threshold
  validation metric        callbacks.py:503-513 (threshold sweep linspace(0.5, 0.3, 21) with early stop),
                           metrics.py:21-66 (IoU / IOUT with the empty-mask conventions)
                                                                             -> salt_iou_sweep (+ a few host floats)

Everything here needs the HIP library; there is no CPU path.
"""
import ctypes
import itertools

import numpy as np
import torch

from ._abi import OP_FUNCS, SaltError, check, fill

IOUT_THRESHOLDS = (0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95)        # metrics.py:37-50


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _run(op, **fields):
    fn, S = OP_FUNCS['salt_' + op]
    s = S()
    fill(s, **fields)
    check(fn(ctypes.byref(s), _stream()), op)


def _f32c(t):
    if not t.is_cuda:
        raise SaltError('inference epilogue: tensors must live on the GPU (there is no CPU path)')
    return t.contiguous().float()


# ----------------------------------------------------------------------------- test-time augmentation
def tta_variants(flip_ud=True, flip_lr=True):
    """[(ud, lr)] in the order of the reference generator (loaders.py:662-682 via itertools.product): identity first."""
    out = [(False, False)]
    for ud, lr in itertools.product([True, False] if flip_ud else [False], [True, False] if flip_lr else [False]):
        if ud or lr:
            out.append((ud, lr))
    return out


def flip(x, ud, lr, out=None):
    """augmentation.py:143-153 on an NCHW batch: flipud / fliplr of every image (``out``: a contiguous fp32 tensor of the same shape,
    e.g. a slice of the batch that holds all TTA variants)."""
    x = _f32c(x)
    y = torch.empty_like(x) if out is None else out
    if y.shape != x.shape or y.dtype != torch.float32 or not y.is_contiguous() or not y.is_cuda:
        raise SaltError('flip: out must be a contiguous fp32 device tensor of the input shape')
    B, C, H, W = x.shape
    _run('flip', x=x.data_ptr(), B=B, C=C, H=H, W=W, flip_ud=int(ud), flip_lr=int(lr), y=y.data_ptr())
    return y


TTA_METHODS = {'mean': 0, 'max': 1, 'min': 2, 'gmean': 3}


def tta_mean(logits, variants, batch, method='mean'):
    """sigmoid -> inverse transform -> aggregation over the variants in ONE kernel; ``logits`` is variant-major [V*B, C, H, W];
    ``variants``: (ud, lr) or (ud, lr, k) with k the quarter turns of the variant's np.rot90 (square maps); ``method``: mean (default,
    neptune.yaml:80) | max | min | gmean (loaders.py:727-735)."""
    logits = _f32c(logits)
    VB, C, H, W = logits.shape
    V = len(variants)
    if VB != V * batch:
        raise SaltError('tta_mean: %d logits for %d variants x batch %d' % (VB, V, batch))
    if method not in TTA_METHODS:
        raise SaltError('TTA aggregation %r (mean | max | min | gmean, loaders.py:727-735)' % (method,))
    ud = (ctypes.c_int * V)(*[int(v[0]) for v in variants])
    lr = (ctypes.c_int * V)(*[int(v[1]) for v in variants])
    rot = (ctypes.c_int * V)(*[int(v[2]) if len(v) > 2 else 0 for v in variants])
    prob = torch.empty((batch, C, H, W), dtype=torch.float32, device=logits.device)
    _run('tta_mean', logits=logits.data_ptr(), V=V, B=batch, C=C, H=H, W=W, flip_ud=ctypes.cast(ud, ctypes.c_void_p).value,
         flip_lr=ctypes.cast(lr, ctypes.c_void_p).value, prob=prob.data_ptr(), rot=ctypes.cast(rot, ctypes.c_void_p).value,
         method=TTA_METHODS[method])
    return prob


def _flip_input(X, ud, lr, depth_channels, out=None):
    """One TTA variant of a preprocessed batch.  The reference flips the RAW tile and only then normalises and adds the depth channels
    (loaders.py:401-423 -> 603-612, utils.py:494-500), so channel 1 (the row ramp) is NOT flipped and channel 2 is
    flipped(gray) * ramp; flipping all three channels of the network input would hand the network an inverted ramp."""
    if not (ud or lr):
        if out is None:
            return X
        out.copy_(X)
        return out
    Y = flip(X, ud, lr, out=out)
    if depth_channels and ud and X.shape[1] == 3:
        Y[:, 1] = X[:, 1]
        Y[:, 2] = Y[:, 0] * X[:, 1]
    return Y


def _looks_like_depth_channels(X):
    """True when channels 1 / 2 of a 3-channel batch are the reference's depth channels (utils.py:494-500): channel 1 the row ramp
    linspace(0, 1, H) shared by every image and column, channel 2 = channel 0 * channel 1.  A 3-channel input that merely has a flat
    channel 1 (blank tiles, striped data) is NOT one and must be flipped channel by channel.  One host sync; pass
    depth_channels=True / False to predict_tta to skip the test."""
    if X.dim() != 4 or X.shape[1] != 3:
        return False
    H = X.shape[2]
    # compared in float32 with tolerances scaled by the input's own precision (a bf16 / fp16 ramp is ~4e-3 / 5e-4 off the fp32 one)
    eps = torch.finfo(X.dtype).eps if X.dtype.is_floating_point else 0.0
    Xf = X.float()
    ramp = torch.linspace(0, 1, H, device=X.device, dtype=torch.float32).view(1, H, 1)
    ok = ((Xf[:, 1] - ramp).abs().max() <= max(1e-5, 2 * eps)) & \
         ((Xf[:, 2] - Xf[:, 0] * Xf[:, 1]).abs().max() <= max(1e-4, 4 * eps) * (1 + Xf[:, 0].abs().max()))
    return bool(ok.item())


def predict_tta(net, X, flip_ud=False, flip_lr=True, variants_per_pass=None, depth_channels=None, method='mean'):
    """Probabilities [B, C, H, W] of an eval-mode HipNetwork aggregated over the flip variants (reference default main.py:282-285:
    left-right only; BASELINE C4's "4-flip" is flip_ud=True, flip_lr=True).  ``depth_channels``: the input is the reference's
    3-channel [gray, depth ramp, gray*ramp] batch, whose channels 1 / 2 an up-down flip must rebuild rather than flip (see
    _flip_input); None (default) = decide from the data (channel 1 = linspace(0, 1, H) AND channel 2 = channel 0 * channel 1; one host
    sync), True raises if it is not.
    ``method``: 'mean' (default, neptune.yaml:80; one fused kernel) or 'max' / 'min' / 'gmean' (loaders.py:727-735).

    The variants are forwarded ``variants_per_pass`` at a time as one larger batch (default: all of them).  For bit-faithful
    handling of the asymmetric 13/14 edge pad use :func:`predict_tta_tiles`, which flips the raw tiles like the reference."""
    if net.training:
        raise SaltError('predict_tta: call net.eval() first')
    X = _f32c(X)
    B = X.shape[0]
    variants = tta_variants(flip_ud, flip_lr)
    if flip_ud and depth_channels is not False:          # only an up-down flip touches the depth channels
        is_ramp = _looks_like_depth_channels(X)
        if depth_channels and not is_ramp:
            raise SaltError('predict_tta(depth_channels=True): channel 1 of the batch is not a row ramp (utils.py:494-500); pass '
                            'depth_channels=False for an ordinary 3-channel input')
        depth_channels = is_ramp
    per = len(variants) if not variants_per_pass else int(variants_per_pass)
    V = len(variants)
    logits = None
    with torch.no_grad():
        for i in range(0, V, per):
            group = variants[i:i + per]
            if len(group) > 1:                              # the group's variants are written side by side into ONE batch (no torch.cat)
                xb = torch.empty((len(group) * B,) + tuple(X.shape[1:]), dtype=torch.float32, device=X.device)
                for j, (ud, lr) in enumerate(group):
                    _flip_input(X, ud, lr, depth_channels, out=xb[j * B:(j + 1) * B])
            else:
                xb = _flip_input(X, group[0][0], group[0][1], depth_channels)
            lg = net(xb)
            if len(group) == V:
                logits = lg.float()
            else:
                if logits is None:
                    logits = torch.empty((V * B,) + tuple(lg.shape[1:]), dtype=torch.float32, device=X.device)
                logits[i * B:(i + len(group)) * B].copy_(lg)
    return tta_mean(logits, variants, B, method)


def predict_tta_tiles(net, preprocessor, images, flip_ud=False, flip_lr=True, rotation=False, method='mean'):
    """TTA exactly in the reference's order of operations: transform the RAW [B,h,w] tiles (augmentation.py:143-153: flipud, fliplr,
    rot90), preprocess every variant (inference pad + Normalize + AddDepthChannels), forward, inverse-transform the probability
    maps (augmentation.py:156-163) and aggregate (loaders.py:722-760).  ``rotation`` needs square tiles."""
    if net.training:
        raise SaltError('predict_tta_tiles: call net.eval() first')
    if not images.is_cuda:
        raise SaltError('predict_tta_tiles: tensors must live on the GPU (there is no CPU path)')
    B = images.shape[0]
    specs = [(False, False, 0)]
    for ud, lr, k in itertools.product([True, False] if flip_ud else [False], [True, False] if flip_lr else [False],
                                       [0, 1, 2, 3] if rotation else [0]):
        if ud or lr or k:
            specs.append((ud, lr, k))
    if method not in TTA_METHODS:
        raise SaltError('TTA aggregation %r (mean | max | min | gmean, loaders.py:727-735)' % (method,))
    logits = None
    with torch.no_grad():
        for v, (ud, lr, k) in enumerate(specs):
            t = images                                    # the RAW tiles are transformed as the reference's loader does (host-side
            if ud:                                        # numpy there; torch index ops on the small [B,h,w] tiles here)
                t = torch.flip(t, dims=(1,))
            if lr:
                t = torch.flip(t, dims=(2,))
            if k:
                t = torch.rot90(t, k, dims=(1, 2))
            x, _ = preprocessor(t.contiguous())
            lg = net(x)
            if logits is None:
                logits = torch.empty((len(specs) * B,) + tuple(lg.shape[1:]), dtype=torch.float32, device=lg.device)
            logits[v * B:(v + 1) * B].copy_(lg)
    # sigmoid, inverse transform (rot90(-k), fliplr, flipud: augmentation.py:156-163) and aggregation of all variants: one kernel
    return tta_mean(logits, specs, B, method)


# ----------------------------------------------------------------------------- post-processing
def crop_window(H, W, target_size):
    """utils.py:308-313 get_crop_pad_sequence: (top, left) of the centre crop; 128 -> 101 gives rows 13:114, cols 14:115."""
    vertical, horizontal = H - target_size[0], W - target_size[1]
    top, right = int(vertical / 2), int(horizontal / 2)
    return top, horizontal - right


def crop_threshold(prob, target_size=(101, 101), threshold=0.5, cls=1):
    """postprocessing.py crop_image + binarize on the device: uint8 masks [B, h, w] of ``prob[:, cls] > threshold``."""
    prob = _f32c(prob)
    B, C, H, W = prob.shape
    h, w = target_size
    top, left = crop_window(H, W, target_size)
    mask = torch.empty((B, h, w), dtype=torch.uint8, device=prob.device)
    _run('crop_threshold', prob=prob.data_ptr(), B=B, C=C, H=H, W=W, cls=cls, top=top, left=left, h=h, w=w,
         threshold=float(threshold), mask=mask.data_ptr())
    return mask


def run_length_encoding(mask):
    """Submission encoding of one binary mask (utils.py:99-111): runs over the column-major pixel order as a flat list
    [start_1-based, length, start, length, ...].  Host side (the mask is a 10 KB array that leaves the device anyway)."""
    m = mask.detach().cpu().numpy() if torch.is_tensor(mask) else np.asarray(mask)
    flat = np.concatenate([[0], (m.T.reshape(-1) != 0).astype(np.int8), [0]])
    edges = np.flatnonzero(np.diff(flat))                  # run starts at even positions, run ends (exclusive) at odd ones
    starts, ends = edges[0::2], edges[1::2]
    out = np.empty(2 * len(starts), dtype=np.int64)
    out[0::2] = starts + 1
    out[1::2] = ends - starts
    return out.tolist()


def run_length_decoding(mask_rle, shape):
    """Inverse (utils.py:114-133); ``mask_rle``: 'start length start length ...' or the list run_length_encoding returns."""
    vals = [int(v) for v in (mask_rle.split() if isinstance(mask_rle, str) else mask_rle)]
    img = np.zeros(shape[0] * shape[1], dtype=np.uint8)
    for lo, n in zip(vals[0::2], vals[1::2]):
        img[lo - 1:lo - 1 + n] = 1
    return img.reshape((shape[1], shape[0])).T


# ----------------------------------------------------------------------------- validation metric
def iou_counts(prob, gt, thresholds, cls=1):
    """Per image and threshold: |pred & gt|, |pred|; per image |gt|.  ``gt`` is uint8 [B, h, w]; prob is cropped to it."""
    prob = _f32c(prob)
    if not gt.is_cuda:
        raise SaltError('iou_counts: gt must live on the GPU')
    gt = gt.contiguous().to(torch.uint8)
    B, C, H, W = prob.shape
    h, w = gt.shape[1:]
    top, left = crop_window(H, W, (h, w))
    T = len(thresholds)
    th = (ctypes.c_double * T)(*[float(t) for t in thresholds])
    inter = torch.empty((B, T), dtype=torch.int32, device=prob.device)
    pred = torch.empty((B, T), dtype=torch.int32, device=prob.device)
    gtc = torch.empty((B,), dtype=torch.int32, device=prob.device)
    _run('iou_sweep', prob=prob.data_ptr(), B=B, C=C, H=H, W=W, cls=cls, top=top, left=left, h=h, w=w, gt=gt.data_ptr(), T=T,
         thresholds=ctypes.cast(th, ctypes.c_void_p).value, inter=inter.data_ptr(), pred=pred.data_ptr(), gt_count=gtc.data_ptr())
    return inter.cpu().numpy().astype(np.int64), pred.cpu().numpy().astype(np.int64), gtc.cpu().numpy().astype(np.int64)


def scores_from_counts(inter, pred, gt):
    """(iou[T], iout[T]): dataset means per threshold, metrics.py:21-66 semantics (both empty -> 1, exactly one empty -> 0)."""
    inter, pred, gt = np.asarray(inter, np.float64), np.asarray(pred, np.float64), np.asarray(gt, np.float64)[:, None]
    union = pred + gt - inter
    both_empty = (pred == 0) & (gt == 0)
    one_empty = (pred == 0) != (gt == 0)
    with np.errstate(divide='ignore', invalid='ignore'):
        iou = np.where(both_empty, 1.0, np.where(one_empty, 0.0, inter / np.where(union > 0, union, 1.0)))
    iout = np.mean(np.stack([(iou >= t).astype(np.float64) for t in IOUT_THRESHOLDS], 0), 0)
    return iou.mean(0), iout.mean(0)


def select_threshold(counts_batches, thresholds=None):
    """callbacks.py:503-513: walk ``np.linspace(0.5, 0.3, 21)`` while IOUT improves, keep the best.

    ``counts_batches``: list of ``iou_counts`` results (one per validation batch, all with the same thresholds).
    Returns (threshold_best, iou, iout)."""
    thresholds = np.linspace(0.5, 0.3, 21) if thresholds is None else np.asarray(thresholds)
    inter = np.concatenate([c[0] for c in counts_batches], 0)
    pred = np.concatenate([c[1] for c in counts_batches], 0)
    gt = np.concatenate([c[2] for c in counts_batches], 0)
    iou, iout = scores_from_counts(inter, pred, gt)
    best, t_best = 0.0, 0.5
    k_best = None
    for k, t in enumerate(thresholds):
        if iout[k] > best:
            best, t_best, k_best = iout[k], float(t), k
        else:
            break
    if k_best is None:                       # no threshold beat 0.0: the reference scores its default 0.5
        k_best = int(np.argmin(np.abs(thresholds - 0.5)))
    return t_best, float(iou[k_best]), float(iout[k_best])
