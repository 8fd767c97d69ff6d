"""On-device input pipeline (SURVEY.md §8 "next" row f-1).

The reference prepares every tile on the CPU with PIL / imgaug / torchvision and copies fp32 tensors to the GPU
(loaders.py:603-612, augmentation.py:79-96,247-284, utils.py:494-500, loaders.py:763-769).  Here the raw 101x101 gray tiles
(uint8, 10 KB each) are copied once and one kernel produces the network's input batch and the one-hot target:

    train      resize 101 -> 102 (cubic: imgaug 0.2.5's iaa.Scale default; or bilinear) + edge-pad 13 on every side -> 128   (neptune.yaml:22-26)
    inference  edge-pad to the next multiple of 64 with the reference's split: top 13 / bottom 14, left 14 / right 13
    both       Grayscale(3) + ToTensor + Normalize(ImageNet) + AddDepthChannels; mask -> {background, salt} one-hot

Geometric augmentation (imgaug affine / intensity sequences) stays out of scope.
"""
import ctypes

import torch

from ._abi import OP_FUNCS, SaltError, check, fill

MEAN = (0.485, 0.456, 0.406)          # ImageNet statistics (neptune.yaml / loaders.py dataset_params)
STD = (0.229, 0.224, 0.225)


def pad_split(size, divisor=64):
    """(before, after) edge padding of one dimension for InferencePad (augmentation.py:262-277 + utils.py:308-313).
    Vertical: before = top = int(pad / 2).  Horizontal: the sequence is (top, right, bottom, left) with right = int(pad / 2),
    so the *left* side gets the larger half."""
    pad = 0 if size % divisor == 0 else divisor - size % divisor
    return int(pad / 2), pad - int(pad / 2)


class DevicePreprocessor:
    """Callable: (images [B,h,w] uint8|float, masks [B,h,w] or None) on the GPU -> (X [B,C,H,W], target [B,2,H,W] | None)."""

    def __init__(self, train, channels=3, resize=102, pad=13, divisor=64, mean=MEAN, std=STD, interpolation='cubic'):
        """``interpolation`` of the train-branch resize: 'cubic' (default) is what the reference executes - augmentation.py:79-85 calls
        ``iaa.Scale({...})`` without an interpolation argument and imgaug 0.2.5 (environment.yml:15) defaults to 'cubic' =
        cv2.INTER_CUBIC, applied to the uint8 tile AND the uint8 {0,1} mask.  uint8 tiles take cv2's own evaluation of that filter
        (11-bit fixed-point coefficients, integer sums, saturating shift: opencv_python 3.4.0.12, environment.yml:16); float tiles -
        and 'cubic_float' for uint8 ones - the float form of the same filter (differs by 1 LSB on a few percent of the pixels);
        'bilinear' is rounds 1-2 of this build."""
        if interpolation not in ('cubic', 'cubic_float', 'bilinear'):
            raise SaltError('DevicePreprocessor: interpolation %r (cubic | cubic_float | bilinear)' % (interpolation,))
        self.interpolation = interpolation
        self.train, self.channels, self.resize, self.pad, self.divisor = bool(train), int(channels), resize, pad, divisor
        self.mean, self.std = tuple(mean), tuple(std)

    def geometry(self, h, w):
        """(resize_h, resize_w, top, left, H, W)"""
        if self.train:
            r = self.resize
            return r, r, self.pad, self.pad, r + 2 * self.pad, r + 2 * self.pad
        top, bottom = pad_split(h, self.divisor)
        right, left = pad_split(w, self.divisor)          # horizontal: int(pad/2) goes to the right
        return 0, 0, top, left, h + top + bottom, w + left + right

    def __call__(self, images, masks=None):
        if not images.is_cuda or (masks is not None and not masks.is_cuda):
            raise SaltError('DevicePreprocessor: tensors must live on the GPU (there is no CPU path)')
        if images.dtype not in (torch.uint8, torch.float32):
            raise SaltError('DevicePreprocessor: images must be uint8 or float32 in [0, 1]')
        images = images.contiguous()
        B, h, w = images.shape
        rh, rw, top, left, H, W = self.geometry(h, w)
        x = torch.empty((B, self.channels, H, W), dtype=torch.float32, device=images.device)
        target, mptr = None, None
        if masks is not None:
            masks = masks.contiguous().to(torch.uint8)
            target = torch.empty((B, 2, H, W), dtype=torch.float32, device=images.device)
            mptr = masks.data_ptr()
        fn, S = OP_FUNCS['salt_preprocess']
        s = S()
        fill(s, img=images.data_ptr(), img_is_u8=int(images.dtype == torch.uint8), mask=mptr, B=B, h=h, w=w, resize_h=rh, resize_w=rw,
             top=top, left=left, H=H, W=W, channels=self.channels, mean=list(self.mean), std=list(self.std), x=x.data_ptr(),
             target=target.data_ptr() if target is not None else None,
             interpolation=0 if self.interpolation == 'bilinear' else (2 if self.interpolation == 'cubic' and images.dtype == torch.uint8 else 1))
        check(fn(ctypes.byref(s), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'preprocess')
        return x, target
