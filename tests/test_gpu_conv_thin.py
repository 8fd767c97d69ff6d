"""conv_thin_kernel (csrc/conv_thin.hip): the persistent weight-stationary fp32 kernel of the 3x3 layers with 16 / 32 channels on both
sides (the outer levels of the vanilla U-Net, unet_models.py:21-30) through the C-ABI, asked for per launch (cfg = 12) on tensors
small enough for the CPU: two stacked conv + BN + ReLU layers against torch CPU fp32 - train mode (statistics through the fp64
shards, both data gradients, the second one carrying the first layer's BatchNorm-backward sums, weight / BN gradients) and eval mode
(folded BatchNorm + ReLU in the epilogue) - and against conv_mfma_kernel on the same launches."""
import ctypes

import pytest
import torch
import torch.nn.functional as F
from torch import nn

from helpers import assert_close
from test_gpu_conv_ws import _force_cfg, _kernel_ids, _rand

pytestmark = pytest.mark.gpu
TOL = 2e-4

# B, Cin, H, W, Cmid, Cout
CASES = [(2, 16, 32, 32, 16, 16), (1, 32, 48, 32, 32, 32), (2, 16, 16, 64, 32, 16), (3, 32, 32, 16, 16, 32), (5, 16, 16, 16, 16, 32)]


def _build(case):
    B, Cin, H, W, Cmid, Cout = case
    c1, b1, c2, b2 = nn.Conv2d(Cin, Cmid, 3, 1, 1, bias=False), nn.BatchNorm2d(Cmid), nn.Conv2d(Cmid, Cout, 3, 1, 1, bias=True), nn.BatchNorm2d(Cout)
    mod = nn.Sequential(c1, b1, c2, b2)
    with torch.no_grad():
        c1.weight.copy_(_rand(c1.weight.shape, 11, (2.0 / (Cin * 9)) ** 0.5)); c2.weight.copy_(_rand(c2.weight.shape, 12, (2.0 / (Cmid * 9)) ** 0.5))
        c2.bias.copy_(0.1 * _rand((Cout,), 19))
        for i, b in enumerate((b1, b2)):
            b.weight.copy_(1 + 0.1 * _rand(b.weight.shape, 13 + i)); b.bias.copy_(0.1 * _rand(b.bias.shape, 15 + i))
            b.running_mean.copy_(0.1 * _rand(b.bias.shape, 23 + i)); b.running_var.copy_(1 + 0.1 * _rand(b.bias.shape, 25 + i).abs())
    x = _rand((B, Cin, H, W), 17)
    ref = nn.Sequential(nn.Conv2d(Cin, Cmid, 3, 1, 1, bias=False), nn.BatchNorm2d(Cmid), nn.Conv2d(Cmid, Cout, 3, 1, 1, bias=True), nn.BatchNorm2d(Cout))
    ref.load_state_dict({k: v.clone() for k, v in mod.state_dict().items()})
    return mod, ref, x


@pytest.mark.parametrize('case', CASES)
def test_thin_conv_bn_relu_train_vs_torch_and_conv_mfma(case):
    from gpu_harness import BlockRun
    outs = {}
    for cfg in (12, 4):
        mod, ref, x = _build(case)
        c1, b1, c2, b2 = mod

        def emit(g, a):
            _force_cfg(g, cfg)
            h = g.conv(a, c1, b1, relu=True)
            return g.conv(h, c2, b2, relu=True)

        mod.train()
        run = BlockRun(mod, [x], emit, train=True, dtype='f32')
        ids_f, ids_b = _kernel_ids(run.g.fwd), _kernel_ids(run.g.bwd)
        if cfg == 12:
            assert ids_f == [12, 12] and ids_b == [12, 12], (ids_f, ids_b)
            ready = [int(st.partials_ready) for name, _, st in run.g.bwd.ops if name == 'bn_bwd']
            assert ready == [0, 3], ready                  # layer 1's BatchNorm-backward sums came from layer 2's data-gradient launch
        else:
            assert 12 not in ids_f + ids_b
        y = run.forward()
        gy = _rand(tuple(y.shape), 18)
        gx, grads = run.backward(gy.to('cuda:0'))
        outs[cfg] = (y, gx[0], grads, {k: v.detach().cpu().clone() for k, v in mod.state_dict().items()})
    mod, ref, x = _build(case)
    ref.train()
    xr = x.clone().requires_grad_(True)
    yr = F.relu(ref[3](ref[2](F.relu(ref[1](ref[0](xr))))))
    yr.backward(_rand(tuple(yr.shape), 18))
    y, gx, grads, sd = outs[12]
    assert_close(y, yr, TOL, 'y')
    assert_close(gx, xr.grad, 4 * TOL, 'dgrad')
    for k in grads:
        if k == '2.bias':                                  # a bias in front of a train-mode BatchNorm has no gradient (rounding noise only)
            assert float(grads[k].abs().max()) < 1e-4
            continue
        assert_close(grads[k], dict(ref.named_parameters())[k].grad, 4 * TOL, k)
    for k in ('1.running_mean', '1.running_var', '3.running_mean', '3.running_var'):
        assert_close(sd[k], ref.state_dict()[k], TOL, k)
    # the two kernels on the same launches: fp32 sums in different orders
    assert_close(outs[12][0], outs[4][0], TOL, 'forward thin vs mfma')
    assert_close(outs[12][1], outs[4][1], 4 * TOL, 'dgrad thin vs mfma')


@pytest.mark.parametrize('case', CASES[:3])
def test_thin_conv_eval_folded_bn_relu_vs_torch(case):
    from gpu_harness import BlockRun
    mod, ref, x = _build(case)
    c1, b1, c2, b2 = mod

    def emit(g, a):
        _force_cfg(g, 12)
        h = g.conv(a, c1, b1, relu=True)
        return g.conv(h, c2, b2, relu=False)

    mod.eval(); ref.eval()
    run = BlockRun(mod, [x], emit, train=False, dtype='f32')
    assert _kernel_ids(run.g.fwd) == [12, 12]
    y = run.forward()
    with torch.no_grad():
        yr = ref[3](ref[2](F.relu(ref[1](ref[0](x)))))
    assert_close(y, yr, TOL, 'eval y')


@pytest.mark.parametrize('case', [(16, 16, 0), (32, 16, 0), (16, 32, 1), (32, 32, 0), (32, 32, 1)])
def test_thin_wgrad_vs_torch(case):
    """conv_wgrad_thin_kernel through salt_conv_wgrad + salt_wgrad_reduce (one slab per persistent workgroup) against
    torch.nn.grad.conv2d_weight on the CPU: zero padding (taps -1 .. 1) and the reference's replicate top / right padding (rows -2 .. 0,
    columns 0 .. 2, clamped), enough tiles (8 x 64 x 64) for the launch plan to pick the kernel by itself."""
    import salt_amd  # noqa: F401
    from salt_amd._abi import STRUCTS, lib, fill, check
    from salt_amd.engine import shaped_view
    Ca, Cb, rep = case
    B, H, W = 8, 64, 64
    g = torch.Generator().manual_seed(7)
    P = torch.randn(B, H, W, Ca, generator=g)
    Q = torch.randn(B, H, W, Cb, generator=g)
    taps = [(dy - 2, dx) for dy in range(3) for dx in range(3)] if rep else [(dy - 1, dx - 1) for dy in range(3) for dx in range(3)]
    Pd, Qd = P.cuda(), Q.cuda()
    S = fill(STRUCTS['salt_conv_wgrad_args'](), dtype=0, p=shaped_view(Pd.data_ptr(), B, H, W, Ca), q=shaped_view(Qd.data_ptr(), B, H, W, Cb),
             ntaps=9, tap_dy=[t[0] for t in taps], tap_dx=[t[1] for t in taps], q_step=1, pad_mode=rep, q_plane=0)
    ns = lib.salt_conv_wgrad_nsplit(ctypes.byref(S))
    assert ns == 128, ns                                   # 8 x 4 x 4 tiles, one per workgroup: the thin kernel's plan (the general kernels split differently)
    part = torch.full((ns, 9, Ca, Cb), float('nan'), device='cuda:0')
    grad = torch.empty(Ca, Cb, 3, 3, device='cuda:0')
    S.partials = part.data_ptr(); S.nsplit = ns
    R = fill(STRUCTS['salt_wgrad_reduce_args'](), partials=part.data_ptr(), nsplit=ns, ntaps=9, Ca=Ca, Cb=Cb, KH=3, KW=3,
             tap_kh=[t // 3 for t in range(9)], tap_kw=[t % 3 for t in range(9)], grad=grad.data_ptr(), accumulate=0)
    st = torch.cuda.current_stream().cuda_stream
    check(lib.salt_conv_wgrad(ctypes.byref(S), st)); check(lib.salt_wgrad_reduce(ctypes.byref(R), st))
    torch.cuda.synchronize()
    X, dY = Q.permute(0, 3, 1, 2).contiguous(), P.permute(0, 3, 1, 2).contiguous()
    if rep:
        ref = torch.nn.grad.conv2d_weight(F.pad(X, (0, 2, 2, 0), mode='replicate'), (Ca, Cb, 3, 3), dY, padding=0)
    else:
        ref = torch.nn.grad.conv2d_weight(X, (Ca, Cb, 3, 3), dY, padding=1)
    assert_close(grad.cpu(), ref, 1e-4, 'dW')


@pytest.mark.parametrize('case', [(2, 32, 16, 16, 16), (3, 16, 32, 16, 32), (2, 32, 16, 32, 32)])
@pytest.mark.parametrize('train', [True, False])
def test_thin_transposed_conv_phases_vs_torch(case, train):
    """nn.ConvTranspose2d(Cin, Cout, 3, stride 2, padding 1, output_padding 1) + BatchNorm + ReLU (unet_models.py:38-50,
    base.py:40-57: the up-sampling step of the vanilla / Ternaus decoders) as ONE phase-fused launch on conv_thin_kernel<..., 4, 4>
    (a tile's halo serves its four output-parity phases): forward (train statistics / folded eval), data and weight gradients vs torch."""
    from gpu_harness import BlockRun
    B, Cin, H, W, Cout = case
    dc, bn = nn.ConvTranspose2d(Cin, Cout, 3, 2, 1, 1, bias=True), nn.BatchNorm2d(Cout)
    mod = nn.Sequential(dc, bn)
    with torch.no_grad():
        dc.weight.copy_(_rand(dc.weight.shape, 31, (2.0 / (Cin * 9)) ** 0.5)); dc.bias.copy_(0.1 * _rand((Cout,), 32))
        bn.weight.copy_(1 + 0.1 * _rand((Cout,), 33)); bn.bias.copy_(0.1 * _rand((Cout,), 34))
        bn.running_mean.copy_(0.1 * _rand((Cout,), 35)); bn.running_var.copy_(1 + 0.1 * _rand((Cout,), 36).abs())
    ref_dc, ref_bn = nn.ConvTranspose2d(Cin, Cout, 3, 2, 1, 1, bias=True), nn.BatchNorm2d(Cout)
    ref_dc.load_state_dict(dc.state_dict()); ref_bn.load_state_dict(bn.state_dict())
    x = _rand((B, Cin, H, W), 37)

    def emit(g, a):
        _force_cfg(g, 12)
        return g.conv_transpose(a, dc, bn, relu=True)

    mod.train(train); ref_dc.train(train); ref_bn.train(train)
    run = BlockRun(mod, [x], emit, train=train, dtype='f32')
    assert _kernel_ids(run.g.fwd) == [12], _kernel_ids(run.g.fwd)
    y = run.forward()
    xr = x.clone().requires_grad_(True)
    yr = F.relu(ref_bn(ref_dc(xr)))
    assert_close(y, yr.detach(), TOL, 'y')
    if train:
        assert_close(bn.running_mean.cpu(), ref_bn.running_mean, TOL, 'running_mean')
        assert_close(bn.running_var.cpu(), ref_bn.running_var, TOL, 'running_var')
        gy = _rand(tuple(yr.shape), 38)
        yr.backward(gy)
        gx, grads = run.backward(gy.to('cuda:0'))
        assert_close(gx[0], xr.grad, 4 * TOL, 'dgrad')
        assert_close(grads['0.weight'], ref_dc.weight.grad, 4 * TOL, 'wgrad')
        assert_close(grads['1.weight'], ref_bn.weight.grad, 4 * TOL, 'dgamma')
        assert_close(grads['1.bias'], ref_bn.bias.grad, 4 * TOL, 'dbeta')


@pytest.mark.parametrize('case', [(32, 16), (16, 16)])
def test_thin_wgrad_stride2_operand_vs_torch(case):
    """The weight gradient of nn.ConvTranspose2d(Ca, Cb, 3, 2, 1, 1) (P = x, Q = dY read at 2 p + tap: salt_conv_wgrad_args.q_step 2) on
    conv_wgrad_thin_kernel<.., .., 2> (8 x 16-pixel tiles, 17 x 33 halo) against torch autograd on the CPU."""
    import salt_amd  # noqa: F401
    from salt_amd._abi import STRUCTS, lib, fill, check
    from salt_amd.engine import shaped_view
    Ca, Cb = case
    B, H, W = 8, 32, 64                                    # 8 x 4 x 4 tiles of 8 x 16
    g = torch.Generator().manual_seed(9)
    X = torch.randn(B, Ca, H, W, generator=g)
    dY = torch.randn(B, Cb, 2 * H, 2 * W, generator=g)
    taps = [(u - 1, v - 1) for u in range(3) for v in range(3)]
    Pd, Qd = X.permute(0, 2, 3, 1).contiguous().cuda(), dY.permute(0, 2, 3, 1).contiguous().cuda()
    S = fill(STRUCTS['salt_conv_wgrad_args'](), dtype=0, p=shaped_view(Pd.data_ptr(), B, H, W, Ca), q=shaped_view(Qd.data_ptr(), B, 2 * H, 2 * W, Cb),
             ntaps=9, tap_dy=[t[0] for t in taps], tap_dx=[t[1] for t in taps], q_step=2, pad_mode=0, q_plane=0)
    ns = lib.salt_conv_wgrad_nsplit(ctypes.byref(S))
    assert ns == 128, ns
    part = torch.full((ns, 9, Ca, Cb), float('nan'), device='cuda:0')
    grad = torch.empty(Ca, Cb, 3, 3, device='cuda:0')
    S.partials = part.data_ptr(); S.nsplit = ns
    R = fill(STRUCTS['salt_wgrad_reduce_args'](), partials=part.data_ptr(), nsplit=ns, ntaps=9, Ca=Ca, Cb=Cb, KH=3, KW=3,
             tap_kh=[t // 3 for t in range(9)], tap_kw=[t % 3 for t in range(9)], grad=grad.data_ptr(), accumulate=0)
    st = torch.cuda.current_stream().cuda_stream
    check(lib.salt_conv_wgrad(ctypes.byref(S), st)); check(lib.salt_wgrad_reduce(ctypes.byref(R), st))
    torch.cuda.synchronize()
    dc = nn.ConvTranspose2d(Ca, Cb, 3, 2, 1, 1, bias=False)
    dc(X).backward(dY)
    assert_close(grad.cpu(), dc.weight.grad, 1e-4, 'dW')
