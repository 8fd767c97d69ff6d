"""Loss functions of the reference's training loop as hand-written HIP kernels.

  lovasz_loss          <- models.py:326-328  (lovasz_losses.py:81-115, per-image, F.elu variant)
  mixed_dice_bce_loss  <- models.py:331-340  (defaults dice 0.2 / bce 0.9, sigmoid dice, batch-wide sums)

Both accept what the reference passes (fp32 logits [B,C,H,W], float one-hot target [B,C,H,W]) and return
a differentiable 0-dim tensor.  They run only on GPU tensors; there is no CPU fallback."""
import ctypes

import torch

from ._abi import SaltError, STRUCTS, fill, lib, check

_ws = {}


def _workspace(key, n, dtype, device):
    t = _ws.get(key)
    if t is None or t.numel() < n or t.device != device:
        t = torch.empty(n, dtype=dtype, device=device)
        _ws[key] = t
    return t


def native_loss(logits, target, kind, want_grad=True, loss_scale=1.0):
    if logits.device.type != 'cuda':
        raise SaltError('native losses run on the GPU only (got %s)' % logits.device)
    logits = logits.contiguous().float()
    target = target.contiguous().float()
    if target.shape[1] != logits.shape[1]:
        target = target[:, :logits.shape[1]].contiguous()
    B, C, H, W = logits.shape
    dev = logits.device
    loss = torch.zeros(1, dtype=torch.float32, device=dev)
    dl = torch.empty_like(logits) if want_grad else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if kind == 'lovasz':
        P = C * H * W
        # sort scratch is cached per (device, stream): kernels of ONE stream run in order, so two losses may share it; losses
        # enqueued on different streams (two models on one device) get their own
        sid = torch.cuda.current_stream().cuda_stream
        wk = _workspace(('k', dev, sid), 2 * B * P, torch.int32, dev)
        wv = _workspace(('v', dev, sid), 2 * B * P, torch.int32, dev)
        lpi = torch.empty(B, dtype=torch.float32, device=dev)
        sw = int(lib.salt_lovasz_split_words(P))                 # > 0: several workgroups per image (segments of the sort)
        ws = _workspace(('s', dev, sid), B * sw, torch.int32, dev) if sw else None
        a = STRUCTS['salt_lovasz_args']()
        fill(a, logits=logits.data_ptr(), target=target.data_ptr(), B=B, P=P, ws_keys=wk.data_ptr(), ws_vals=wv.data_ptr(),
             loss_per_image=lpi.data_ptr(), loss=loss.data_ptr(), dlogits=dl.data_ptr() if want_grad else None, loss_scale=loss_scale,
             ws_split=ws.data_ptr() if ws is not None else None)
        check(lib.salt_lovasz_hinge(ctypes.byref(a), st), 'lovasz_hinge')
    elif kind == 'bce_dice':
        a = STRUCTS['salt_bce_dice_args']()
        fill(a, B=B, C=C, HW=H * W)
        nparts = lib.salt_bce_dice_parts(ctypes.byref(a))
        parts = torch.empty(nparts * 4, dtype=torch.float32, device=dev)
        sums = torch.empty(3 * C + 1, dtype=torch.float32, device=dev)
        fill(a, logits=logits.data_ptr(), target=target.data_ptr(), dice_weight=0.2, bce_weight=0.9, partials=parts.data_ptr(), nparts=nparts,
             sums=sums.data_ptr(), loss=loss.data_ptr(), dlogits=dl.data_ptr() if want_grad else None, loss_scale=loss_scale)
        check(lib.salt_bce_dice(ctypes.byref(a), st), 'bce_dice')
    else:
        raise SaltError('unknown loss %r' % kind)
    return loss[0], dl


def lovasz_loss(output, target):
    from .autograd import _NativeLoss
    return _NativeLoss.apply(output, target, 'lovasz')


def mixed_dice_bce_loss(output, target, dice_weight=0.2, dice_loss=None, bce_weight=0.9, bce_loss=None, smooth=0, dice_activation='sigmoid'):
    if dice_weight != 0.2 or bce_weight != 0.9 or smooth != 0 or dice_activation != 'sigmoid' or dice_loss is not None or bce_loss is not None:
        raise NotImplementedError('only the reference defaults (0.2 dice + 0.9 BCE, sigmoid) have a HIP kernel')
    from .autograd import _NativeLoss
    return _NativeLoss.apply(output, target, 'bce_dice')


lovasz_loss.native_kind = 'lovasz'
mixed_dice_bce_loss.native_kind = 'bce_dice'
