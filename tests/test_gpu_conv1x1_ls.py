"""conv1x1_ls_kernel (round 4): the streaming kernel of the eval-mode Bottleneck 1x1 convolutions (architectures/encoders.py:13-19 through
torchvision's Bottleneck: conv1 / conv3 with folded BatchNorm, ReLU, residual) through the C-ABI, asked for per launch (cfg = 11)
on tensors small enough for the CPU: against torch in fp32 on the same bf16-rounded operands, and against conv_mfma_kernel (cfg = 1)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    # B, H, W, Cin, Cout, residual, affine + relu, accumulate, view slack (pixel stride > channels), cap (workgroups per XCD; 0 = auto)
    (2, 16, 16, 64, 256, True, True, False, 0, 0),        # 64 -> 256 expansion + residual (layer1 conv3)
    (2, 16, 16, 256, 64, False, True, False, 0, 0),       # 256 -> 64 reduction (conv1)
    (1, 16, 32, 128, 96, False, False, False, 0, 0),      # Cout % 64 != 0: 32-channel items
    (4, 8, 8, 512, 128, False, True, True, 0, 0),         # accumulate, 8 chunks of 64 channels (ring wraps several times)
    (2, 16, 16, 128, 512, True, True, False, 64, 0),      # strided views
    (8, 16, 16, 256, 128, True, True, False, 0, 2),       # 2 workgroups per XCD: several items per workgroup
    (8, 16, 16, 64, 64, False, False, False, 0, 1),
]
CASES_S2 = [
    # B, OH, OW, Cin, Cout, affine, view slack, cap: nn.Conv2d(Cin, Cout, 1, stride 2) of the ResNet projection shortcuts (input 2 OH x 2 OW)
    (2, 16, 16, 256, 512, True, 0, 0),
    (4, 16, 32, 64, 96, False, 0, 0),
    (3, 32, 16, 128, 256, True, 128, 2),
]


def _conv1x1(case, cfg):
    import salt_amd  # noqa: F401
    from salt_amd._abi import STRUCTS, lib, fill, check
    from salt_amd.engine import shaped_view
    B, H, W, Cin, Cout, res, aff, acc, slack, cap = case
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, W, Cin + slack, generator=g).bfloat16().cuda()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5).cuda()
    r = torch.randn(B, H, W, Cout + slack, generator=g).bfloat16().cuda()
    y = (torch.randn(B, H, W, Cout + slack, generator=g).bfloat16() if acc else torch.zeros(B, H, W, Cout + slack).bfloat16()).cuda()
    y0 = y.clone()
    bias = torch.randn(Cout, generator=g).cuda(); scale = (1 + 0.1 * torch.randn(Cout, generator=g)).cuda(); shift = (0.1 * torch.randn(Cout, generator=g)).cuda()
    n = lib.salt_packed_weight_elems(1, 1, Cout, Cin)
    wp = torch.zeros(n, dtype=torch.bfloat16, device='cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    check(lib.salt_pack_conv_weight(ctypes.byref(fill(STRUCTS['salt_pack_conv_weight_args'](), dtype=1, w=w.data_ptr(), D0=Cout, D1=Cin, KH=1, KW=1, ntaps=1,
                                                      tap_kh=[0], tap_kw=[0], transpose=0, wp=wp.data_ptr())), st))
    S = fill(STRUCTS['salt_conv_args'](), dtype=1, x=shaped_view(x.data_ptr(), B, H, W, Cin, Cin + slack), w=wp.data_ptr(), ntaps=1, tap_dy=[0], tap_dx=[0],
             in_step=1, pad_mode=0, y=shaped_view(y.data_ptr(), B, H, W, Cout, Cout + slack), OH=H, OW=W, out_step=1,
             bias=bias.data_ptr() if aff else None, scale=scale.data_ptr() if aff else None, shift=shift.data_ptr() if aff else None,
             relu=int(aff), accumulate=int(acc), cfg=cfg | (cap << 8))
    if res:
        S.res = shaped_view(r.data_ptr(), B, H, W, Cout, Cout + slack)
    kid = lib.salt_conv_kernel_id(ctypes.byref(S))
    check(lib.salt_conv(ctypes.byref(S), st), 'salt_conv')
    torch.cuda.synchronize()
    # fp32 reference on the bf16-rounded operands, with the kernel's rounding points (bf16 before the residual add)
    xf, wf = x[..., :Cin].float().cpu(), w.bfloat16().float().cpu().view(Cout, Cin)
    v = xf.reshape(-1, Cin) @ wf.t()
    if aff:
        v = (v + bias.cpu()) * scale.cpu() + shift.cpu()
        if not res:
            v = v.clamp_min(0)
    v = v.view(B, H, W, Cout)
    if res:
        v = v.bfloat16().float() + r[..., :Cout].float().cpu()
        if aff:
            v = v.clamp_min(0)
    if acc:
        v = v.bfloat16().float() + y0[..., :Cout].float().cpu()
    return y.float().cpu(), v, kid, (y0.float().cpu() if slack else None)


@pytest.mark.parametrize('case', CASES)
def test_conv1x1_ls_vs_torch_and_conv_mfma(case):
    y, ref, kid, y0 = _conv1x1(case, 11)
    assert kid == 11, kid
    Cout, slack = case[4], case[8]
    got = y[..., :Cout]
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= 1e-2, err                                                 # one bf16 rounding of the stored value
    if slack:
        assert torch.equal(y[..., Cout:], y0[..., Cout:])                   # the channels beside the view are untouched
    y1, _, kid1, _ = _conv1x1(case, 1)
    assert kid1 == 1
    d = float((y1[..., :Cout] - got).abs().max() / ref.abs().max())
    assert d <= 8e-3, d                                                     # same products, same bf16 rounding points: at most one ulp apart


CASES_ITEM_MAJOR = [
    # (case, extra cfg bits): more channel blocks than workgroups per XCD - the hypercolumn's tap GEMMs C -> 9 C (engine.hyper_level) -
    # walk (pixel tile, channel block) items tile-major; bit 18 asks for that walk on small tensors, bit 16 for 32-channel items
    ((2, 16, 16, 256, 2304, False, False, False, 0, 0), 0),               # 36 blocks of 64 channels > 32 workgroups per XCD
    ((2, 16, 16, 256, 2304, False, False, False, 0, 0), 1 << 16),         # 72 blocks of 32
    ((8, 16, 16, 64, 192, True, False, False, 0, 3), 1 << 18),            # residual, 3 workgroups per XCD
    ((4, 16, 16, 128, 576, False, False, True, 32, 5), (1 << 18) | (1 << 16)),   # (+)=, strided views, a ragged last round of items
    ((16, 16, 16, 64, 64, False, False, False, 0, 7), 1 << 18),           # ONE channel block, items = tiles
]


@pytest.mark.parametrize('case,bits', CASES_ITEM_MAJOR)
def test_conv1x1_ls_item_major_walk_vs_torch_and_conv_mfma(case, bits):
    y, ref, kid, y0 = _conv1x1(case, 11 | bits)
    assert kid == 11, kid
    Cout, slack = case[4], case[8]
    got = y[..., :Cout]
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= 1e-2, err
    if slack:
        assert torch.equal(y[..., Cout:], y0[..., Cout:])
    y1, _, kid1, _ = _conv1x1(case, 1)
    assert kid1 == 1
    d = float((y1[..., :Cout] - got).abs().max() / ref.abs().max())
    assert d <= 8e-3, d


CASES_XS = [
    # conv1x1_xs_kernel (cfg bit 19): the input tile stays in LDS, the workgroup walks every 64-channel block of it
    (2, 16, 16, 256, 2304, False, False, False, 0, 0),       # the C4 tap GEMM's channel counts: 4 chunks, 36 blocks, one tile per workgroup
    (16, 16, 16, 128, 576, False, False, False, 0, 1),       # 8 workgroups, two tiles each: the tile buffer is fenced and refilled
    (24, 16, 16, 64, 192, False, False, False, 0, 1),        # three tiles each, one chunk
    (4, 16, 16, 192, 320, False, False, False, 64, 0),       # strided views, 3 chunks, 5 blocks
    # the eval epilogue (Bottleneck conv3: folded BatchNorm, + identity, ReLU; per-block constants in LDS, residual tile requested per block)
    (2, 16, 16, 64, 256, True, True, False, 0, 0),
    (8, 16, 16, 128, 512, True, True, False, 64, 1),         # strided views, one tile per workgroup of 8
    (16, 16, 16, 256, 1024, False, True, False, 0, 1),       # affine + ReLU without a residual, two tiles per workgroup
    (4, 8, 8, 64, 128, False, True, True, 0, 0),             # (+)=
]


@pytest.mark.parametrize('case', CASES_XS)
def test_conv1x1_xs_vs_torch_and_conv_mfma(case):
    y, ref, kid, y0 = _conv1x1(case, 11 | (1 << 19))
    assert kid == 11, kid
    Cout, slack = case[4], case[8]
    got = y[..., :Cout]
    assert torch.isfinite(got).all()
    err = float((got - ref).abs().max() / ref.abs().max())
    assert err <= 1e-2, err
    if slack:
        assert torch.equal(y[..., Cout:], y0[..., Cout:])
    y1, _, kid1, _ = _conv1x1(case, 1)
    assert kid1 == 1
    d = float((y1[..., :Cout] - got).abs().max() / ref.abs().max())
    assert d <= 8e-3, d
    # and the streaming kernel's result for the same launch, bit for bit (same MFMA sequence per output, same rounding)
    y2, _, _, _ = _conv1x1(case, 11)
    assert torch.equal(y2[..., :Cout], got)


def test_conv1x1_ls_is_picked_for_the_bottleneck_shapes():
    """Without a request the plan hands the big eval-mode 1x1 launches (enough items for the chip) to the streaming kernel and keeps
    the rest on conv_mfma_kernel."""
    big = (16, 64, 64, 64, 256, True, True, False, 0, 0)
    _, _, kid, _ = _conv1x1(big, 0)
    assert kid == 11
    small = (1, 16, 16, 64, 256, True, True, False, 0, 0)
    _, _, kid, _ = _conv1x1(small, 0)
    assert kid != 11


@pytest.mark.parametrize('case', CASES_S2)
def test_conv1x1_ls_stride2_vs_torch(case):
    import salt_amd  # noqa: F401
    from salt_amd._abi import STRUCTS, lib, fill, check
    from salt_amd.engine import shaped_view
    B, OH, OW, Cin, Cout, aff, slack, cap = case
    H, W = 2 * OH, 2 * OW
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, W, Cin + slack, generator=g).bfloat16().cuda()
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5).cuda()
    y = torch.zeros(B, OH, OW, Cout).bfloat16().cuda()
    scale = (1 + 0.1 * torch.randn(Cout, generator=g)).cuda(); shift = (0.1 * torch.randn(Cout, generator=g)).cuda()
    wp = torch.zeros(lib.salt_packed_weight_elems(1, 1, Cout, Cin), dtype=torch.bfloat16, device='cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    check(lib.salt_pack_conv_weight(ctypes.byref(fill(STRUCTS['salt_pack_conv_weight_args'](), dtype=1, w=w.data_ptr(), D0=Cout, D1=Cin, KH=1, KW=1, ntaps=1,
                                                      tap_kh=[0], tap_kw=[0], transpose=0, wp=wp.data_ptr())), st))
    S = fill(STRUCTS['salt_conv_args'](), dtype=1, x=shaped_view(x.data_ptr(), B, H, W, Cin, Cin + slack), w=wp.data_ptr(), ntaps=1, tap_dy=[0], tap_dx=[0],
             in_step=2, pad_mode=0, y=shaped_view(y.data_ptr(), B, OH, OW, Cout), OH=OH, OW=OW, out_step=1,
             scale=scale.data_ptr() if aff else None, shift=shift.data_ptr() if aff else None, relu=0, accumulate=0, cfg=11 | (cap << 8))
    assert lib.salt_conv_kernel_id(ctypes.byref(S)) == 11
    check(lib.salt_conv(ctypes.byref(S), st), 'salt_conv')
    torch.cuda.synchronize()
    xs = x[:, ::2, ::2, :Cin].float().cpu()
    v = xs.reshape(-1, Cin) @ w.bfloat16().float().cpu().view(Cout, Cin).t()
    if aff:
        v = v * scale.cpu() + shift.cpu()
    v = v.view(B, OH, OW, Cout)
    err = float((y.float().cpu() - v).abs().max() / v.abs().max())
    assert err <= 1e-2, err


@pytest.mark.parametrize('case', [(4, 64, 32, 32, 128, 2), (2, 128, 32, 32, 256, 2), (4, 64, 16, 16, 96, 1), (8, 256, 32, 32, 512, 2)])
def test_conv1x1_ls_train_statistics_vs_torch(case):
    """Train mode: the ResNet projection shortcut nn.Conv2d(Cin, Cout, 1, stride) + BatchNorm (torchvision's `downsample`, via
    architectures/encoders.py:6-45): conv1x1_ls_kernel's MODE 1 adds the per-channel sums of the stored values to the fp64 shards that the
    consumer finalizes.  Forward, running statistics, data / weight / BatchNorm gradients against torch CPU fp32."""
    import ctypes as C
    from torch import nn
    import torch.nn.functional as F
    from gpu_harness import BlockRun
    from helpers import assert_close
    from salt_amd._abi import lib
    B, Cin, H, W, Cout, stride = case
    conv, bn = nn.Conv2d(Cin, Cout, 1, stride, 0, bias=False), nn.BatchNorm2d(Cout)
    mod = nn.Sequential(conv, bn)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / Cin) ** 0.5)
        bn.weight.copy_(1 + 0.1 * torch.randn(Cout, generator=g)); bn.bias.copy_(0.1 * torch.randn(Cout, generator=g))
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    mod.train()
    from test_gpu_conv_ws import _force_cfg

    def emit(gr, a):
        _force_cfg(gr, 11)                                        # (small tensors: the kernel is not picked by itself below half a tile per CU)
        return gr.conv(a, conv, bn, relu=False)
    run = BlockRun(mod, [x], emit, train=True, dtype='bf16')
    ids = [lib.salt_conv_kernel_id(C.byref(s)) for name, _, s in run.g.fwd.ops if name == 'conv']
    assert ids == [11], ids
    y = run.forward()
    rc, rb = nn.Conv2d(Cin, Cout, 1, stride, 0, bias=False), nn.BatchNorm2d(Cout)
    with torch.no_grad():
        rc.weight.copy_(conv.weight.detach().cpu().bfloat16().float()); rb.weight.copy_(bn.weight.detach().cpu()); rb.bias.copy_(bn.bias.detach().cpu())
    xr = x.clone().requires_grad_(True)
    yr = rb(rc(xr))
    assert_close(y, yr, 4e-2, 'y')
    assert_close(bn.running_mean.cpu(), rb.running_mean, 4e-2, 'running_mean')
    assert_close(bn.running_var.cpu(), rb.running_var, 4e-2, 'running_var')
    gy = torch.randn(tuple(yr.shape), generator=g)
    yr.backward(gy)
    gx, grads = run.backward(gy.to('cuda:0'))
    for name, got, want in [('dgrad', gx[0], xr.grad), ('wgrad', grads['0.weight'], rc.weight.grad)]:
        l2 = float((got.double() - want.double()).norm() / want.double().norm())
        assert l2 <= 4e-2, '%s: rel-L2 %.3e' % (name, l2)
    assert_close(grads['1.weight'], rb.weight.grad, 0.12, 'dgamma')
    assert_close(grads['1.bias'], rb.bias.grad, 0.12, 'dbeta')
    y2 = run.forward()
    assert float((y2.float() - y.float()).abs().max()) <= 1e-6 * float(y.float().abs().max())
