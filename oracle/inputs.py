"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's input preparation (never imported by the product path).

Restates, with plain torch CPU ops:
  * ImageSegmentationLoader.image_transform (loaders.py:603-612): Grayscale(3) -> ToTensor -> Normalize(ImageNet) -> AddDepthChannels
    (utils.py:494-500);  mask_transform / one-hot target (loaders.py:763-769, 186-190)
  * resize_pad_seq (augmentation.py:79-85, neptune.yaml:22-26): resize 101 -> 102, edge-pad 13 -> 128            (train)
  * pad_to_fit_net / InferencePad (augmentation.py:93-96, 247-284) with get_crop_pad_sequence (utils.py:308-313)  (inference)

Parity note: imgaug / cv2 are absent from the image, so the interpolation kernel of `iaa.Scale` cannot be executed here;
the restatement uses bilinear with half-pixel centres for the tile and nearest for the mask ("parity unpinned" for the
101 -> 102 resize only; the pad geometry, normalisation and depth channels are pinned by golden F10 / the reference source).
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


def preprocess(img, mask, train, channels, resize=102, pad=13, divisor=64):
    """img: float [B,h,w] in [0,1]; mask: {0,1} [B,h,w] or None -> (X [B,channels,H,W], target [B,2,H,W] | None)."""
    x = torch.as_tensor(img, dtype=torch.float32)[:, None]
    m = None if mask is None else torch.as_tensor(mask, dtype=torch.float32)[:, None]
    if train:
        x = F.interpolate(x, size=(resize, resize), mode='bilinear', align_corners=False)
        if m is not None:
            m = (F.interpolate(m, size=(resize, resize), mode='nearest') > 0.5).float()
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
