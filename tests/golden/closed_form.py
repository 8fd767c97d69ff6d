"""Closed-form, name-keyed weights and inputs shared by the golden generator and the tests.

No weight files are committed: every tensor is a deterministic function of its state-dict key and
shape (numpy RandomState seeded with crc32(key)), so the reference modules (in the build
container), the oracle and the HIP path can all be driven with identical parameters by key name.
"""
import zlib
from collections import OrderedDict

import numpy as np
import torch


def _rs(key, salt=0):
    return np.random.RandomState((zlib.crc32(key.encode()) + salt) % (2 ** 32))


def tensor_for(key, shape, salt=0):
    """Deterministic fp32 tensor for a state-dict entry, scaled so activations stay O(1)."""
    shape = tuple(shape)
    r = _rs(key, salt)
    leaf = key.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return torch.zeros((), dtype=torch.long)
    if leaf == 'running_var':
        return torch.from_numpy((0.6 + 0.8 * r.rand(*shape)).astype(np.float32))
    if leaf == 'running_mean':
        return torch.from_numpy((0.1 * r.randn(*shape)).astype(np.float32))
    if len(shape) == 1 and leaf == 'weight':        # BatchNorm gamma
        return torch.from_numpy((1.0 + 0.2 * r.randn(*shape)).astype(np.float32))
    if len(shape) == 1:                             # any bias / BN beta
        return torch.from_numpy((0.1 * r.randn(*shape)).astype(np.float32))
    fan_in = int(np.prod(shape[1:]))
    if len(shape) == 4 and 'deconv' in key and 'conv.' not in key.rsplit('deconv', 1)[-1]:
        fan_in = shape[0] * shape[2] * shape[3] // 4   # ConvTranspose [Cin,Cout,k,k], stride 2
    w = r.randn(*shape) * np.sqrt(1.6 / max(fan_in, 1))
    return torch.from_numpy(w.astype(np.float32))


def state_for(named_shapes, salt=0):
    """named_shapes: iterable of (key, shape).  Aliased keys must be resolved by the caller."""
    return OrderedDict((k, tensor_for(k, s, salt)) for k, s in named_shapes)


def fill_module(module, salt=0, canonical=None):
    """Load closed-form values into a torch module by its own state_dict keys.
    ``canonical(key) -> key`` maps alias spellings to the canonical spelling used as the seed."""
    sd = module.state_dict()
    new = OrderedDict()
    for k, v in sd.items():
        ck = canonical(k) if canonical else k
        new[k] = tensor_for(ck, v.shape, salt).to(v.dtype)
    module.load_state_dict(new)
    return module


def input_for(tag, shape, scale=1.0):
    r = _rs('input:' + tag)
    return torch.from_numpy((scale * r.randn(*shape)).astype(np.float32))


def mask_for(tag, shape, p=0.4):
    """Blocky binary mask [B,H,W] -> one-hot float target [B,2,H,W] (background, salt)."""
    r = _rs('mask:' + tag)
    B, H, W = shape
    m = np.zeros(shape, np.float32)
    for b in range(B):
        if r.rand() < 0.25:
            continue                                  # empty mask (dataset trait)
        cy, cx = r.randint(0, H), r.randint(0, W)
        ry, rx = r.randint(H // 6 + 1, H // 2 + 2), r.randint(W // 6 + 1, W // 2 + 2)
        yy, xx = np.mgrid[0:H, 0:W]
        m[b] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0).astype(np.float32)
    t = np.stack([1.0 - m, m], axis=1)
    return torch.from_numpy(t)
