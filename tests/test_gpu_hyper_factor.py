"""GPU parity of the FACTORED hypercolumn (round 5): the final Conv2dBnRelu of architectures.unet.UNetResNet
(/root/reference/common_blocks/architectures/unet.py:84-87,101-109; base.py:21-37) with the up-sampled levels taken out of the
full-resolution convolution - z_k = [W_tap] dec_k at low resolution + salt_hyper_stencil.

Yardstick: plain torch on the CPU doing what the reference does - F.interpolate(bilinear) -> torch.cat -> ReplicationPad2d((0, 2, 2, 0))
-> conv2d (-> train-mode BatchNorm -> ReLU) - and its autograd.  fp32 <= 5e-5 of the tensor's largest magnitude, forward and every
gradient; bf16 against the same pipeline on bf16-rounded operands."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from helpers import assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _abi():
    from salt_amd import _abi
    return _abi


def _view(t, C=None):
    """salt_view of a contiguous NHWC tensor [B,H,W,Ct] (first C channels)"""
    from salt_amd.engine import shaped_view
    B, H, W, Ct = t.shape
    return shaped_view(t.data_ptr(), B, H, W, Ct if C is None else C, Ct)


def _stencil_ref(zs, Rs, y_in, ac):
    """sum_k sum_tap shift_tap(up_R(z_k[tap])) + y_in; zs[k]: [B, 9, C, h, w] (tap-major), y_in [B, C, H, W]"""
    out = y_in.clone()
    H, W = y_in.shape[-2:]
    pad = nn.ReplicationPad2d((0, 2, 2, 0))
    for z, R in zip(zs, Rs):
        for t in range(9):
            kh, kw = divmod(t, 3)
            u = pad(F.interpolate(z[:, t], scale_factor=R, mode='bilinear', align_corners=bool(ac)))
            out = out + u[:, :, kh:kh + H, kw:kw + W]
    return out


def _nhwc_taps(z):
    """[B, 9, C, h, w] -> NHWC [B, h, w, 9 C] with channel t C + o"""
    B, T9, C, h, w = z.shape
    return z.permute(0, 3, 4, 1, 2).reshape(B, h, w, T9 * C).contiguous()


HEAD_CASES = [
    # B, C, H, W, levels, classes
    (2, 64, 32, 48, (4, 8, 16), 2),       # the ResNet34 model's width: one channel block
    (1, 256, 40, 36, (4,), 2),            # ResNet101 / 152: four channel blocks (three butterfly steps across them), partial pixel tiles
    (1, 128, 64, 32, (16, 4), 1),         # two blocks, one class
    (1, 512, 16, 16, (4, 8), 2),          # eight blocks
]


@pytest.mark.parametrize('case', HEAD_CASES)
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_stencil_fused_logit_head_is_bit_identical_to_stencil_then_head1x1(case, dtype):
    """Eval: final = Sequential(Conv2dBnRelu, Conv2d(C, classes, 1)) (architectures/unet.py:84-87).  salt_hyper_stencil's epilogue applies
    the 1x1 head to the values it would store; the logits must equal salt_head1x1 on the stored activation bit for bit (same products,
    same summation tree), with and without y being written."""
    abi = _abi()
    B, C, H, W, Rs, CO = case
    if dtype == 'f32' and C > 256:
        pytest.skip('fp32: salt_head1x1 leaves its vector kernels above 64 lanes per pixel')
    tdt = torch.float32 if dtype == 'f32' else torch.bfloat16
    dt = 0 if dtype == 'f32' else 1
    g = torch.Generator().manual_seed(7 + C)
    zd = [_nhwc_taps(torch.randn(B, 9, C, H // R, W // R, generator=g)).to(DEV, tdt) for R in Rs]
    yi = torch.randn(B, H, W, C, generator=g).to(DEV, tdt)
    sc, sh = (torch.rand(C, generator=g) + 0.5).to(DEV), torch.randn(C, generator=g).to(DEV)
    hw, hb = (torch.randn(CO, C, generator=g) * C ** -0.5).to(DEV), torch.randn(CO, generator=g).to(DEV)
    # unfused: stencil (eval epilogue) -> y, then salt_head1x1
    y = torch.zeros_like(yi)
    S = abi.STRUCTS['salt_hyper_stencil_args']()
    abi.fill(S, dtype=dt, nlev=len(Rs), z=[_view(z) for z in zd], R=list(Rs), y_in=_view(yi), y=_view(y), backward=0, align_corners=0,
             scale=sc.data_ptr(), shift=sh.data_ptr(), relu=1)
    abi.check(abi.lib.salt_hyper_stencil(ctypes.byref(S), None), 'hyper_stencil')
    ref = torch.full((B, CO, H, W), float('nan'), device=DEV)
    Hd = abi.STRUCTS['salt_head1x1_args']()
    from salt_amd.engine import null_view
    abi.fill(Hd, dtype=dt, x=_view(y), w=hw.data_ptr(), bias=hb.data_ptr(), Cout=CO, y_nchw=ref.data_ptr(), y=null_view())
    abi.check(abi.lib.salt_head1x1(ctypes.byref(Hd), None), 'head1x1')
    # fused, y written too
    y2 = torch.zeros_like(yi)
    got = torch.full((B, CO, H, W), float('nan'), device=DEV)
    ws = torch.full((B * (C // 64) * CO * H * W,), float('nan'), device=DEV)
    abi.fill(S, y=_view(y2), head_w=hw.data_ptr(), head_b=hb.data_ptr(), head_y_nchw=got.data_ptr(), head_cout=CO, head_ws=ws.data_ptr() if C > 64 else None)
    abi.check(abi.lib.salt_hyper_stencil(ctypes.byref(S), None), 'hyper_stencil + head')
    torch.cuda.synchronize()
    assert torch.equal(y2, y)
    assert torch.isfinite(ref).all()
    assert torch.equal(got, ref), float((got - ref).abs().max())
    # fused, y dropped (shape-only view)
    got2 = torch.full((B, CO, H, W), float('nan'), device=DEV)
    yv = _view(y2)
    yv.p = None
    abi.fill(S, y=yv, head_y_nchw=got2.data_ptr())
    abi.check(abi.lib.salt_hyper_stencil(ctypes.byref(S), None), 'hyper_stencil + head, no y')
    torch.cuda.synchronize()
    assert torch.equal(got2, ref)
    # and against torch on the stored activation
    want = torch.einsum('bhwc,oc->bohw', y.float().cpu(), hw.cpu()) + hb.cpu().view(1, CO, 1, 1)
    assert_close(got.cpu(), want, 2e-5, 'fused head vs torch')


def test_stencil_fused_head_refuses_what_it_cannot_do():
    abi = _abi()
    y = torch.zeros(1, 16, 16, 192, device=DEV)          # three channel blocks: the butterfly across blocks needs a power of two
    z = torch.zeros(1, 4, 4, 9 * 192, device=DEV)
    lg = torch.zeros(1, 2, 16, 16, device=DEV)
    w = torch.zeros(2, 192, device=DEV)
    S = abi.STRUCTS['salt_hyper_stencil_args']()
    ws = torch.zeros(3 * 2 * 256, device=DEV)
    abi.fill(S, dtype=0, nlev=1, z=[_view(z)], R=[4], y_in=_view(y), y=_view(y), backward=0, align_corners=0, head_w=w.data_ptr(),
             head_y_nchw=lg.data_ptr(), head_cout=2, head_ws=ws.data_ptr())
    assert abi.lib.salt_hyper_stencil(ctypes.byref(S), None) != 0
    y = torch.zeros(1, 16, 16, 64, device=DEV)
    z = torch.zeros(1, 4, 4, 9 * 64, device=DEV)
    acc = torch.zeros(8 * 129, dtype=torch.float64, device=DEV)
    abi.fill(S, z=[_view(z)], y_in=_view(y), y=_view(y), fin_acc=acc.data_ptr())       # train-mode statistics + head: no
    assert abi.lib.salt_hyper_stencil(ctypes.byref(S), None) != 0


STENCIL_CASES = [
    # B, C, H, W, levels
    (2, 64, 32, 48, (4, 8, 16)),
    (1, 32, 20, 28, (4,)),            # partial pixel tiles and a partial channel block
    (2, 24, 16, 16, (4, 8)),
    (1, 72, 64, 32, (16, 4)),         # a second, partial channel block
    (1, 64, 64, 256, (4, 16)),        # two column slots in the adjoint
    (1, 64, 128, 128, (4, 8, 16)),    # the C2 geometry
    (1, 32, 96, 160, (32, 8)),        # non-square, R = 32 (with align_corners the adjoint's row span is (H - 1) / (h - 1) > R per step;
                                      #  the host now sizes the row table from the kernel's own range function)
]


@pytest.mark.parametrize('case', STENCIL_CASES)
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('ac', [0, 1])
def test_stencil_forward_and_adjoint_vs_torch(case, dtype, ac):
    abi = _abi()
    B, C, H, W, Rs = case
    tdt = torch.float32 if dtype == 'f32' else torch.bfloat16
    g = torch.Generator().manual_seed(hash(case) % 1000 + ac)
    zs = [torch.randn(B, 9, C, H // R, W // R, generator=g).to(tdt).float() for R in Rs]
    y_in = torch.randn(B, C, H, W, generator=g).to(tdt).float()
    gy = torch.randn(B, C, H, W, generator=g).to(tdt).float()
    zr = [z.double().requires_grad_(True) for z in zs]
    ref = _stencil_ref(zr, Rs, y_in.double(), ac)
    ref.backward(gy.double())
    tol = 2e-5 if dtype == 'f32' else 6e-3
    dt = 0 if dtype == 'f32' else 1

    zd = [_nhwc_taps(z).to(DEV, tdt) for z in zs]
    yi = y_in.permute(0, 2, 3, 1).contiguous().to(DEV, tdt)
    yo = torch.zeros_like(yi)
    S = abi.STRUCTS['salt_hyper_stencil_args']()
    acc = torch.zeros(8 * (2 * C + 1), dtype=torch.float64, device=DEV)
    abi.fill(S, dtype=dt, nlev=len(Rs), z=[_view(z) for z in zd], R=list(Rs), y_in=_view(yi), y=_view(yo), backward=0, align_corners=ac,
             fin_acc=acc.data_ptr())
    abi.check(abi.lib.salt_hyper_stencil(ctypes.byref(S), None), 'hyper_stencil')
    torch.cuda.synchronize()
    got = yo.float().cpu().permute(0, 3, 1, 2)
    assert_close(got, ref.detach().float(), tol, 'stencil forward')
    # statistics shards: sum / sum of squares / count of the fp32 values
    a = acc.cpu().view(8, 2 * C + 1).sum(0)
    r64 = ref.detach()
    assert a[2 * C].item() == B * H * W
    stol = 1e-5 if dtype == 'f32' else 1e-5          # the sums are taken BEFORE the rounding to the storage type
    np.testing.assert_allclose(a[:C].numpy(), r64.sum((0, 2, 3)).numpy(), rtol=0, atol=stol * float(r64.abs().sum((0, 2, 3)).max()))
    np.testing.assert_allclose(a[C:2 * C].numpy(), (r64 * r64).sum((0, 2, 3)).numpy(), rtol=stol * 10, atol=0)

    # eval epilogue: y = relu(y scale + shift), in place over y_in
    sc = torch.rand(C, generator=g) + 0.5
    sh = torch.randn(C, generator=g)
    y2 = yi.clone()
    scd, shd = sc.to(DEV), sh.to(DEV)
    abi.fill(S, y_in=_view(y2), y=_view(y2), fin_acc=None, scale=scd.data_ptr(), shift=shd.data_ptr(), relu=1)
    abi.check(abi.lib.salt_hyper_stencil(ctypes.byref(S), None), 'hyper_stencil eval')
    torch.cuda.synchronize()
    ref2 = torch.relu(ref.detach().float() * sc.view(1, C, 1, 1) + sh.view(1, C, 1, 1))
    assert_close(y2.float().cpu().permute(0, 3, 1, 2), ref2, tol * 2, 'stencil eval epilogue')

    if W > 256:
        return
    # adjoint
    gyd = gy.permute(0, 2, 3, 1).contiguous().to(DEV, tdt)
    dz = [torch.full_like(z, float('nan')) for z in zd]
    Sb = abi.STRUCTS['salt_hyper_stencil_args']()
    abi.fill(Sb, dtype=dt, nlev=len(Rs), z=[_view(z) for z in dz], R=list(Rs), y=_view(gyd), backward=1, align_corners=ac)
    abi.check(abi.lib.salt_hyper_stencil(ctypes.byref(Sb), None), 'hyper_stencil adjoint')
    torch.cuda.synchronize()
    for k, R in enumerate(Rs):
        want = _nhwc_taps(zr[k].grad.float())
        assert_close(dz[k].float().cpu(), want, tol, 'stencil adjoint level R=%d' % R)


class _Final(nn.Module):
    def __init__(self, d):
        super().__init__()
        from salt_amd import architectures as A
        self.block = A.Conv2dBnRelu(5 * d, d)


def _final_ref(m, xs, train, dt64=True):
    """what the reference does: unet.py:101-109 + base.py:29-37"""
    conv, bn = m.block.conv, m.block.batch_norm
    cat = torch.cat([xs[0]] + [F.interpolate(x, scale_factor=R, mode='bilinear', align_corners=False) for x, R in zip(xs[1:], (2, 4, 8, 16))], 1)
    y = F.conv2d(nn.ReplicationPad2d((0, 2, 2, 0))(cat), conv.weight, conv.bias)
    y = F.batch_norm(y, bn.running_mean.clone(), bn.running_var.clone(), bn.weight, bn.bias, train, bn.momentum, bn.eps)
    return torch.relu(y)


@pytest.mark.parametrize('shape', [(2, 32, 32, 48), (2, 64, 64, 64)])
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('mode', ['train', 'eval'])
@pytest.mark.parametrize('fmin', [4, 8])
def test_factored_final_block_vs_torch(shape, dtype, mode, fmin):
    """The whole factored block through the engine (tap-GEMM packs, 1x1 launches, stencil, BatchNorm, every gradient incl. the
    scattered 3x3 weight gradient) against torch's interpolate -> cat -> pad -> conv -> BN -> ReLU."""
    from gpu_harness import BlockRun
    B, d, H, W = shape
    torch.manual_seed(3)
    m = _Final(d)
    with torch.no_grad():
        m.block.batch_norm.weight.uniform_(0.5, 1.5)
        m.block.batch_norm.bias.normal_(0, 0.3)
        m.block.batch_norm.running_mean.normal_(0, 0.2)
        m.block.batch_norm.running_var.uniform_(0.5, 2.0)
        if dtype == 'bf16':
            m.block.conv.weight.copy_(m.block.conv.weight.bfloat16().float())
    train = mode == 'train'
    m.train(train)
    g = torch.Generator().manual_seed(11)
    Rs_all = (1, 2, 4, 8, 16)
    rnd = (lambda t: t.bfloat16().float()) if dtype == 'bf16' else (lambda t: t)
    xs = [rnd(torch.randn(B, d, H // R, W // R, generator=g)) for R in Rs_all]
    gy = torch.randn(B, d, H, W, generator=g)
    ref_m = _Final(d)
    ref_m.load_state_dict(m.state_dict())
    ref_m.train(train).double()
    xr = [x.double().requires_grad_(True) for x in xs]
    yr = _final_ref(ref_m, xr, train)
    if train:
        yr.backward(gy.double())

    from helpers import rel_err

    def run(fact):
        """forward (+ backward) of the block with the levels of `fact` factored out ([] = the materialised hypercolumn of round 4)"""
        nfull = 5 - len(fact)

        def emit(g_, *acts):
            conv, bn = m.block.conv, m.block.batch_norm
            planes = d if (nfull > 1 and g_.planar_ok(B, H, W, nfull * d, d, conv)) else 0
            hyper = g_.new_act(B, H, W, nfull * d, 'hypercolumn', planes=planes)
            g_.copy(acts[0], hyper.slice(0, d))
            for k in range(1, nfull):
                g_.upsample(acts[k], Rs_all[k], out=hyper.slice(k * d, d))
            if not fact:
                return m.block.emit(g_, hyper)
            assert g_.hyper_factor_ok(B, H, W, d, [R for R, _ in fact])
            zs = [g_.hyper_level(acts[k], conv, k * d) for k in range(nfull, 5)]
            return g_.conv_hyper(hyper, zs, [Rs_all[k] for k in range(nfull, 5)], conv, bn, relu=True)

        with torch.no_grad():                      # every run starts from the same running statistics
            m.block.batch_norm.running_mean.copy_(ref_m.block.batch_norm.running_mean.float())
            m.block.batch_norm.running_var.copy_(ref_m.block.batch_norm.running_var.float())
        r = BlockRun(m, xs, emit, train=train, dtype=dtype)
        y_ = r.forward()
        if not train:
            return y_, None, None
        gx_, grads_ = r.backward(gy.to(DEV))
        return y_, gx_, grads_

    fact = [(R, k) for R, k in ((16, 4), (8, 3), (4, 2)) if R >= fmin]
    y, gx, grads = run(fact)
    tol = 5e-5 if dtype == 'f32' else 2e-2
    assert_close(y, yr.detach().float(), tol, 'y')
    if not train:
        return
    refg = {n: p.grad.float() for n, p in ref_m.named_parameters()}
    if dtype == 'f32':
        gtol = 1e-4
        for k in range(5):
            assert_close(gx[k], xr[k].grad.float(), gtol, 'dL/d(level %d)' % k)
        for n, gv in grads.items():
            if n.endswith('conv.bias'):
                assert float(gv.abs().max()) < 1e-4 * float(refg['block.conv.weight'].abs().max())
                continue
            assert_close(gv, refg[n], gtol * 2, 'g:' + n)
    else:
        # bf16 storage through a train-mode BatchNorm backward: the yardstick is the SAME block on the materialised hypercolumn (round
        # 4's path, itself checked against the bf16-storage oracle): the factored form may sit at most 1.5x as far from the fp64 gradients
        yu, gxu, gradsu = run([])
        for k in range(5):
            ef, eu = rel_err(gx[k], xr[k].grad.float()), rel_err(gxu[k], xr[k].grad.float())
            assert ef <= 1.5 * eu + 5e-3, ('dL/d(level %d)' % k, ef, eu)
        for n, gv in grads.items():
            if n.endswith('conv.bias'):
                continue
            ef, eu = rel_err(gv, refg[n]), rel_err(gradsu[n], refg[n])
            assert ef <= 1.5 * eu + 5e-3, (n, ef, eu)
        y, gx, grads = run(fact)                    # (leave the factored run's running statistics in the module for the check below)
    sd = m.state_dict()
    rs = ref_m.block.batch_norm
    # (the reference function above ran batch_norm on clones: redo the running-statistics update it would have made)
    with torch.no_grad():
        cat = torch.cat([xr[0]] + [F.interpolate(x, scale_factor=R, mode='bilinear', align_corners=False) for x, R in zip(xr[1:], (2, 4, 8, 16))], 1)
        yy = F.conv2d(nn.ReplicationPad2d((0, 2, 2, 0))(cat), ref_m.block.conv.weight, ref_m.block.conv.bias)
        mean = yy.mean((0, 2, 3)); var = yy.var((0, 2, 3), unbiased=True)
        want_mean = 0.9 * rs.running_mean + 0.1 * mean
        want_var = 0.9 * rs.running_var + 0.1 * var
    assert_close(sd['block.batch_norm.running_mean'].cpu(), want_mean.float(), tol * 4, 'running_mean')
    assert_close(sd['block.batch_norm.running_var'].cpu(), want_var.float(), tol * 4, 'running_var')


def test_tapgemm_pack_and_sliced_reduce_layouts():
    """salt_pack_conv_weight sub-block fields (d1_cnt / n_off / n_total / chunk_off) and salt_wgrad_reduce ldb / a_mod against numpy."""
    abi = _abi()
    D0, D1, c0, cn = 32, 80, 16, 32
    w = torch.randn(D0, D1, 3, 3)
    wd = w.to(DEV)
    for dt, tdt, kce in ((0, torch.float32, 16), (1, torch.bfloat16, 32)):
        # forward tap GEMM: [chunk][1][9 D0][kce], row t D0 + o = W[o, c0 + c, t]
        n = 9 * D0
        out = torch.zeros(((cn + kce - 1) // kce) * n * kce, dtype=tdt, device=DEV)
        for t in range(9):
            S = abi.STRUCTS['salt_pack_conv_weight_args']()
            abi.fill(S, dtype=dt, w=wd.data_ptr() + 4 * c0 * 9, D0=D0, D1=D1, KH=3, KW=3, ntaps=1, tap_kh=[t // 3], tap_kw=[t % 3], transpose=0,
                     wp=out.data_ptr(), d1_cnt=cn, n_off=t * D0, n_total=n)
            abi.check(abi.lib.salt_pack_conv_weight(ctypes.byref(S), None), 'pack')
        torch.cuda.synchronize()
        got = out.float().cpu().view(-1, n, kce)
        want = w[:, c0:c0 + cn].reshape(D0, cn, 9).permute(2, 0, 1).reshape(n, cn)          # [(t, o), c]
        want = want.to(tdt).float().view(n, -1, kce).permute(1, 0, 2)
        assert torch.equal(got, want)
        # transposed: [chunk over 9 D0][1][cn][kce], row c, channel t D0 + o
        outT = torch.zeros((9 * D0 // kce) * cn * kce, dtype=tdt, device=DEV)
        for t in range(9):
            S = abi.STRUCTS['salt_pack_conv_weight_args']()
            abi.fill(S, dtype=dt, w=wd.data_ptr() + 4 * c0 * 9, D0=D0, D1=D1, KH=3, KW=3, ntaps=1, tap_kh=[t // 3], tap_kw=[t % 3], transpose=1,
                     wp=outT.data_ptr(), d1_cnt=cn, n_total=cn, chunk_off=t * D0 // kce)
            abi.check(abi.lib.salt_pack_conv_weight(ctypes.byref(S), None), 'packT')
        torch.cuda.synchronize()
        gotT = outT.float().cpu().view(-1, cn, kce)
        wantT = w[:, c0:c0 + cn].reshape(D0, cn, 9).permute(1, 2, 0).reshape(cn, 9 * D0).to(tdt).float()     # [c, (t, o)]
        wantT = wantT.view(cn, -1, kce).permute(1, 0, 2)
        assert torch.equal(gotT, wantT)
    # reduce: slab [ns][1][9 D0][cn] -> grad[o, c0 + c, t]
    ns = 3
    slab = torch.randn(ns, 1, 9 * D0, cn)
    grad = torch.zeros(D0, D1, 3, 3, device=DEV)
    S = abi.STRUCTS['salt_wgrad_reduce_args']()
    sd = slab.to(DEV)
    abi.fill(S, partials=sd.data_ptr(), nsplit=ns, ntaps=1, Ca=9 * D0, Cb=cn, KH=3, KW=3, tap_kh=[t // 3 for t in range(9)], tap_kw=[t % 3 for t in range(9)],
             grad=grad.data_ptr() + 4 * c0 * 9, accumulate=0, ldb=D1, a_mod=D0)
    abi.check(abi.lib.salt_wgrad_reduce(ctypes.byref(S), None), 'reduce')
    torch.cuda.synchronize()
    want = torch.zeros(D0, D1, 3, 3)
    want[:, c0:c0 + cn] = slab.sum(0)[0].view(9, D0, cn).permute(1, 2, 0).reshape(D0, cn, 3, 3)
    assert_close(grad.cpu(), want, 1e-6, 'tap-GEMM reduce')


@pytest.mark.parametrize('shape', [(64, 64, 3), (512, 512, 2), (96, 40, 8), (64, 320, 1)])
def test_wgrad_reduce9_is_bit_identical_to_the_per_element_kernel(shape, monkeypatch):
    """wgrad_reduce9_kernel (a thread per (a, b) pair writes its nine taps as 36 contiguous bytes) against wgrad_reduce8_kernel (a thread per
    slab element): the SAME slabs presented with the taps in reverse order take the per-element kernel (the launch is routed by the raster
    order of tap_kh / tap_kw) and must produce the same bits, plain and accumulating, full tensor and channel slice (ldb)."""
    abi = _abi()
    monkeypatch.setenv('SALT_WGRAD_REDUCE9', '1')          # opt-in kernel (measured no faster: DESIGN 10)
    Ca, Cb, ns = shape
    g = torch.Generator().manual_seed(Ca + Cb)
    slab = torch.randn(ns, 9, Ca, Cb, generator=g)
    ldb = Cb + 24
    res = {}
    for order in ('raster', 'reverse'):
        taps = list(range(9)) if order == 'raster' else list(range(8, -1, -1))
        sd = slab[:, taps].contiguous().to(DEV)
        grad = torch.full((Ca, ldb, 3, 3), 0.5, device=DEV)
        for acc in (0, 1):
            S = abi.STRUCTS['salt_wgrad_reduce_args']()
            abi.fill(S, partials=sd.data_ptr(), nsplit=ns, ntaps=9, Ca=Ca, Cb=Cb, KH=3, KW=3, tap_kh=[t // 3 for t in taps], tap_kw=[t % 3 for t in taps],
                     grad=grad.data_ptr() + 4 * 8 * 9, accumulate=acc, ldb=ldb)
            abi.check(abi.lib.salt_wgrad_reduce(ctypes.byref(S), None), 'reduce')
        torch.cuda.synchronize()
        res[order] = grad.cpu()
    assert torch.equal(res['raster'], res['reverse'])
    want = torch.full((Ca, ldb, 3, 3), 0.5)
    want[:, 8:8 + Cb] = 2 * slab.sum(0).permute(1, 2, 0).reshape(Ca, Cb, 3, 3)
    assert_close(res['raster'], want, 1e-6, 'reduce9 vs torch')
