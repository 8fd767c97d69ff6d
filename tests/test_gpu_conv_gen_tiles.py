"""General pixel tiles of conv_mfma_kernel / conv_glds_kernel (csrc/conv_mfma.hip, tile_pix): the fused-fold data gradients of the
replicate-padded decoder layers (architectures/base.py:21-27 pads top / right by 2, so the gradient lives on an (H + 2) x (W + 2)
grid) run on full-width strips of that grid instead of 16-wide power-of-two tiles.  Two stacked conv + BN + ReLU layers with the
reference's padding against torch CPU fp32 - forward, both data gradients (the second layer's carries the first layer's
BatchNorm-backward sums), weight and BatchNorm gradients - with the tile configuration forced per case, and the tile shape the
launch plan picked read back through salt_conv_tile_shape."""
import ctypes

import pytest
import torch
import torch.nn.functional as F
from torch import nn

from helpers import assert_close
from test_gpu_conv_ws import _force_cfg, _rand

pytestmark = pytest.mark.gpu
TOL32, TOLBF = 2e-4, 4e-2

# B, Cin, H, W, Cmid: extended grids 10 x 10 (one whole image per 128-pixel tile), 18 x 18 (7-row strips), 34 x 34 (3-row strips),
# 6 x 6 (three whole images per tile), 14 x 22 (5-row strips, ragged last strip), 7 x 9 (two images per tile, odd sizes)
CASES = [(3, 64, 8, 8, 96), (2, 64, 16, 16, 64), (2, 32, 32, 32, 64), (16, 32, 4, 4, 32), (2, 32, 12, 20, 64), (3, 32, 5, 7, 32)]


def _tile_shapes(prog):
    from salt_amd._abi import lib
    out = []
    for name, _, s in prog.ops:
        if name == 'conv' and (s.fold_top or s.fold_right):
            v = lib.salt_conv_tile_shape(ctypes.byref(s))
            out.append((v & 255, (v >> 8) & 255, (v >> 16) & 255, v >> 24, lib.salt_conv_kernel_id(ctypes.byref(s))))
    return out


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype,cfg', [('bf16', 4), ('bf16', 1), ('bf16', 5), ('bf16', 8), ('bf16', 7), ('bf16', 6), ('f32', 4), ('f32', 1), ('f32', 0)])
def test_fold_dgrad_on_general_tiles_vs_torch(case, dtype, cfg):
    from gpu_harness import BlockRun
    B, Cin, H, W, Cmid = case
    if cfg >= 6 and H * W <= 16:
        pytest.skip('a forced conv_glds_kernel configuration does not take 4 x 4 maps (8 images and their halos per 128-pixel tile)')
    Cout = Cmid + 32
    c1, b1, c2, b2 = nn.Conv2d(Cin, Cmid, 3, 1, 0, bias=False), nn.BatchNorm2d(Cmid), nn.Conv2d(Cmid, Cout, 3, 1, 0, bias=False), nn.BatchNorm2d(Cout)
    mod = nn.Sequential(c1, b1, c2, b2)
    with torch.no_grad():
        c1.weight.copy_(_rand(c1.weight.shape, 11, (2.0 / (Cin * 9)) ** 0.5)); c2.weight.copy_(_rand(c2.weight.shape, 12, (2.0 / (Cmid * 9)) ** 0.5))
        for i, b in enumerate((b1, b2)):
            b.weight.copy_(1 + 0.1 * _rand(b.weight.shape, 13 + i)); b.bias.copy_(0.1 * _rand(b.bias.shape, 15 + i))
    x = _rand((B, Cin, H, W), 17)
    if dtype == 'bf16':
        x = x.bfloat16().float()
    ref = nn.Sequential(nn.Conv2d(Cin, Cmid, 3, 1, 0, bias=False), nn.BatchNorm2d(Cmid), nn.Conv2d(Cmid, Cout, 3, 1, 0, bias=False), nn.BatchNorm2d(Cout))
    ref.load_state_dict({k: v.clone() for k, v in mod.state_dict().items()})
    if dtype == 'bf16':
        with torch.no_grad():
            ref[0].weight.copy_(ref[0].weight.bfloat16().float()); ref[2].weight.copy_(ref[2].weight.bfloat16().float())

    def emit(g, a):
        if cfg:
            _force_cfg(g, cfg)
        h = g.conv(a, c1, b1, relu=True, replicate=True)
        return g.conv(h, c2, b2, relu=True, replicate=True)

    mod.train()
    run = BlockRun(mod, [x], emit, train=True, dtype=dtype)
    shapes = _tile_shapes(run.g.bwd)
    assert len(shapes) == 2, shapes
    for tw, th, nb, gen, kid in shapes:
        if kid in (9, 10):
            continue                                            # (conv_ws_kernel has its own fused fold on 16 x 16 tiles)
        if cfg in (0, 1, 4, 7, 8):                                # 128-pixel tiles: every case here takes the general tile
            assert gen == 1, (shapes, case)
        if gen:                                                  # a full-width strip with more rows than the pad, or whole images
            assert tw == W + 2 and th > 2 and (th == H + 2 or nb == 1) and nb * th * tw <= 256, (shapes, case)
    y = run.forward()
    xr = x.clone().requires_grad_(True)
    pad = lambda t: F.pad(t, (0, 2, 2, 0), mode='replicate')      # l, r, t, b
    hr = F.relu(ref[1](ref[0](pad(xr))))
    yr = F.relu(ref[3](ref[2](pad(hr))))
    tol = TOL32 if dtype == 'f32' else TOLBF
    assert_close(y, yr, tol * (1 if dtype == 'f32' else 2), 'y')
    gy = _rand(tuple(yr.shape), 18)
    yr.backward(gy)
    gx, grads = run.backward(gy.to('cuda:0'))
    pairs = [('dgrad', gx[0], xr.grad)] + [(k, grads[k], dict(ref.named_parameters())[k].grad) for k in grads]
    for name, got, want in pairs:
        if dtype == 'f32':
            assert_close(got, want, tol * 4, name)
        else:
            l2 = float((got.double() - want.double()).norm() / want.double().norm())
            assert l2 <= 2 * tol, '%s: rel-L2 %.3e' % (name, l2)
