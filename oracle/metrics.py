"""Oracle: prediction post-processing, TTA and validation metric in numpy (TEST INFRASTRUCTURE ONLY).

Reference anchors (relative to /root/reference/common_blocks):
  utils.py:173-174          sigmoid                          -> sigmoid
  utils.py:308-313          get_crop_pad_sequence            -> crop_pad_sequence
  postprocessing.py:24-38   crop_image                       -> crop_image
  postprocessing.py:41-43   binarize                         -> binarize
  metrics.py:21-34,53-59    compute_ious / intersection_over_union (pycocotools RLE IoU; un-vendored)
  metrics.py:37-50,62-66    compute_eval_metric / intersection_over_union_thresholds
  augmentation.py:143-187   TTA transform / inverse          -> tta_transform / tta_inverse
  loaders.py:662-682        TTA parameter product            -> tta_specs
  loaders.py:722-760        aggregator (mean over variants)  -> tta_aggregate
  utils.py:494-500          AddDepthChannels                 -> add_depth_channels
  augmentation.py:247-284   InferencePad geometry            -> inference_pad

pycocotools is absent here and on the GPU box: for a *binary* mask ``get_segmentations`` yields at
most one segment (labels 1..max), so the IoU matrix is 1x1 and the metric collapses to the closed
form below.
"""
import itertools

import numpy as np


def sigmoid(x):
    return 1. / (1 + np.exp(-x))


def crop_pad_sequence(vertical, horizontal):
    top = int(vertical / 2)
    right = int(horizontal / 2)
    return top, right, vertical - top, horizontal - right      # (top, right, bottom, left)


def crop_image(image, target_size):
    """image (C,H,W) -> centre crop to target (H,W); 128->101 gives rows 13:114, cols 14:115."""
    top, right, bottom, left = crop_pad_sequence(image.shape[1] - target_size[0], image.shape[2] - target_size[1])
    return image[:, top:image.shape[1] - bottom, left:image.shape[2] - right]


def inference_pad(image_hw, divisor=64, mode='edge'):
    """Pad (H,W[,C]) so both dims are multiples of ``divisor`` with the reference's asymmetric split."""
    h, w = image_hw.shape[:2]
    pv = 0 if h % divisor == 0 else divisor - h % divisor
    ph = 0 if w % divisor == 0 else divisor - w % divisor
    top, right, bottom, left = crop_pad_sequence(pv, ph)
    pads = [(top, bottom), (left, right)] + [(0, 0)] * (image_hw.ndim - 2)
    return np.pad(image_hw, pads, mode=mode)


def binarize(image, threshold=0.5):
    return (image[1, :, :] > threshold).astype(np.uint8)


def iou_single(gt, pred):
    """IoU of two binary masks with the reference's empty-mask conventions."""
    g = gt > 0
    p = pred > 0
    if not g.any() and not p.any():
        return 1.0
    if g.any() != p.any():
        return 0.0
    return float(np.logical_and(g, p).sum()) / float(np.logical_or(g, p).sum())


THRESHOLDS = (0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95)


def iout_single(gt, pred):
    v = iou_single(gt, pred)
    return sum(1.0 if v >= t else 0.0 for t in THRESHOLDS) / len(THRESHOLDS)


def intersection_over_union(y_true, y_pred):
    return float(np.mean([iou_single(t, p) for t, p in zip(y_true, y_pred)]))


def intersection_over_union_thresholds(y_true, y_pred):
    return float(np.mean([iout_single(t, p) for t, p in zip(y_true, y_pred)]))


def add_depth_channels(x_chw):
    """In place: ch1 := linspace(0,1,H) per row, ch2 := ch0*ch1."""
    _, h, _ = x_chw.shape
    x_chw[1] = np.linspace(0, 1, h, dtype=np.float64)[:, None].astype(x_chw.dtype)
    x_chw[2] = x_chw[0] * x_chw[1]
    return x_chw


def tta_specs(flip_ud, flip_lr, rotation=False):
    base = {'ud_flip': False, 'lr_flip': False, 'rotation': 0}
    out = [base]
    for ud, lr, rot in itertools.product([True, False] if flip_ud else [False],
                                         [True, False] if flip_lr else [False],
                                         [0, 90, 180, 270] if rotation else [0]):
        if not ud and not lr and rot == 0:
            continue
        out.append({'ud_flip': ud, 'lr_flip': lr, 'rotation': rot})
    return out


def tta_transform(image_hwc, spec):
    if spec['ud_flip']:
        image_hwc = np.flipud(image_hwc)
    if spec['lr_flip']:
        image_hwc = np.fliplr(image_hwc)
    return np.rot90(image_hwc, spec['rotation'] // 90, axes=(0, 1))


def tta_inverse(pred_chw, spec):
    x = np.rot90(pred_chw, -spec['rotation'] // 90, axes=(1, 2))
    if spec['lr_flip']:
        x = x[:, :, ::-1]
    if spec['ud_flip']:
        x = x[:, ::-1, :]
    return np.ascontiguousarray(x)


def tta_aggregate(preds_chw, specs, method='mean'):
    stack = np.stack([tta_inverse(p, s) for p, s in zip(preds_chw, specs)], axis=-1)
    if method == 'gmean':                              # scipy.stats.gmean (loaders.py:727-735): exp(mean(log x))
        return np.exp(np.mean(np.log(stack), axis=-1))
    return {'mean': np.mean, 'max': np.max, 'min': np.min}[method](stack, axis=-1)
