"""salt_head_bn / salt_head_bn_bwd (round 6): the reference's `final` Sequential in TRAINING (architectures/unet.py:84-87: base.Conv2dBnRelu ->
nn.Conv2d(C, num_classes, 1)) as one pass over the raw convolution output - BatchNorm finalize + apply + ReLU + 1x1 head forward, and
head gradients + BatchNorm backward without the two tensors between them.

(1) the two entry points alone, through the C-ABI, against torch autograd of  head(relu(batch_norm(y)))  on seeded tensors incl. ragged
    pixel counts, non-power-of-two images, 1 - 4 classes, strided views, both dtypes;
(2) the whole ResNet34 hypercolumn U-Net training step with the fusion on (default) against the same step with SALT_HEAD_BN=0 (the
    separate salt_affine_act + salt_head1x1 / salt_head1x1_bwd + salt_bn_bwd launches): same rounding points, other summation order.
The reference-generated goldens (F8 whole-network fixtures: eval logits, one training step) run on the fused default in
tests/test_gpu_models.py."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

CASES = [
    # B, H, W, C, Cout, relu, slack (pixel stride - C)
    (2, 16, 16, 64, 2, 1, 0),
    (3, 7, 9, 64, 2, 1, 0),          # ragged: 189 pixels, H W not a power of two
    (1, 5, 3, 32, 1, 1, 0),
    (2, 12, 20, 128, 3, 1, 64),      # strided view
    (2, 8, 8, 256, 4, 0, 0),         # no ReLU, 4 classes
    (5, 32, 32, 64, 2, 1, 0),        # several blocks per pass
]


def _run_case(case, dtype):
    import salt_amd  # noqa: F401
    from salt_amd._abi import STRUCTS, lib, fill, check
    from salt_amd.engine import shaped_view
    B, H, W, C, Cout, relu, slack = case
    tdt = torch.float32 if dtype == 'f32' else torch.bfloat16
    dt = 0 if dtype == 'f32' else 1
    g = torch.Generator().manual_seed(B * 131 + C)
    y = (torch.randn(B, H, W, C + slack, generator=g) * 1.5 + 0.3).to(tdt).to(DEV)
    gamma = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV); beta = (0.2 * torch.randn(C, generator=g)).to(DEV)
    w = (torch.randn(Cout, C, generator=g) * C ** -0.5).to(DEV); bias = (0.1 * torch.randn(Cout, generator=g)).to(DEV)
    dl = torch.randn(B, Cout, H, W, generator=g).to(DEV)
    rm, rv, nbt = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
    mean, invstd, scale, shift = (torch.zeros(C, device=DEV) for _ in range(4))
    # the producer's part: (sum, sum of squares, count) of the STORED y in fp64, spread over the 8 shards
    yv = y[..., :C].double().reshape(-1, C)
    M = yv.shape[0]
    acc = torch.zeros(8, 2 * C + 1, dtype=torch.float64, device=DEV)
    for s in range(8):
        part = yv[s::8]
        acc[s, :C] = part.sum(0); acc[s, C:2 * C] = (part * part).sum(0); acc[s, 2 * C] = part.shape[0]
    Fa = fill(STRUCTS['salt_bn_finalize_args'](), C=C, gamma=gamma.data_ptr(), beta=beta.data_ptr(), running_mean=rm.data_ptr(), running_var=rv.data_ptr(),
              num_batches_tracked=nbt.data_ptr(), momentum=0.1, eps=1e-5, mean=mean.data_ptr(), invstd=invstd.data_ptr(), scale=scale.data_ptr(), shift=shift.data_ptr())
    logits = torch.zeros(B, Cout, H, W, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    yview = shaped_view(y.data_ptr(), B, H, W, C, C + slack)
    check(lib.salt_head_bn(ctypes.byref(fill(STRUCTS['salt_head_bn_args'](), dtype=dt, y=yview, fin=ctypes.addressof(Fa), fin_acc=acc.data_ptr(), relu=relu,
                                             w=w.data_ptr(), bias=bias.data_ptr(), Cout=Cout, y_nchw=logits.data_ptr())), st), 'head_bn')
    S = fill(STRUCTS['salt_head_bn_bwd_args'](), y=yview)
    nparts = lib.salt_head_bn_bwd_parts(ctypes.byref(S))
    partials = torch.zeros(nparts * Cout * (C + 1), device=DEV)
    gw, gb, dgamma, dbeta, coef = torch.zeros(Cout, C, device=DEV), torch.zeros(Cout, device=DEV), torch.zeros(C, device=DEV), torch.zeros(C, device=DEV), torch.zeros(3 * C, device=DEV)
    acc_b = torch.zeros(8, 2, C, dtype=torch.float64, device=DEV)
    dy = torch.zeros(B, H, W, C + slack, dtype=tdt, device=DEV)
    check(lib.salt_head_bn_bwd(ctypes.byref(fill(STRUCTS['salt_head_bn_bwd_args'](), dtype=dt, y=yview, relu=relu, mean=mean.data_ptr(), invstd=invstd.data_ptr(),
                                                 gamma=gamma.data_ptr(), beta=beta.data_ptr(), w=w.data_ptr(), Cout=Cout, dy_nchw=dl.data_ptr(),
                                                 partials=partials.data_ptr(), nparts=nparts, gw=gw.data_ptr(), gb=gb.data_ptr(), fin_acc=acc_b.data_ptr(),
                                                 dgamma=dgamma.data_ptr(), dbeta=dbeta.data_ptr(), coef=coef.data_ptr(),
                                                 dy=shaped_view(dy.data_ptr(), B, H, W, C, C + slack))), st), 'head_bn_bwd')
    torch.cuda.synchronize()
    # torch reference on the stored operand (fp32 math; the bf16 path rounds a and dL/da to bf16 - emulated with straight-through casts)
    yr = y[..., :C].float().cpu().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    gam, bet, wr, br = gamma.cpu().clone().requires_grad_(True), beta.cpu().clone().requires_grad_(True), w.cpu().clone().requires_grad_(True), bias.cpu().clone().requires_grad_(True)
    rm_r, rv_r = torch.zeros(C), torch.ones(C)
    a = F.batch_norm(yr, rm_r, rv_r, gam, bet, True, 0.1, 1e-5)
    if relu:
        a = F.relu(a)
    if dtype == 'bf16':
        class Rnd(torch.autograd.Function):
            @staticmethod
            def forward(ctx, t):
                return t.bfloat16().float()

            @staticmethod
            def backward(ctx, gdy):
                return gdy.bfloat16().float()
        a = Rnd.apply(a)
    out = F.conv2d(a, wr.view(Cout, C, 1, 1), br)
    out.backward(dl.cpu())
    res = dict(logits=(logits.cpu(), out.detach()), gw=(gw.cpu(), wr.grad), gb=(gb.cpu(), br.grad), dgamma=(dgamma.cpu(), gam.grad), dbeta=(dbeta.cpu(), bet.grad),
               dy=(dy[..., :C].float().cpu(), yr.grad.permute(0, 2, 3, 1)), running_mean=(rm.cpu(), rm_r), running_var=(rv.cpu(), rv_r))
    return res, int(nbt.item()), (dy[..., C:].float().abs().max().item() if slack else 0.0)


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_head_bn_ops_vs_torch(case, dtype):
    res, nbt, slack_max = _run_case(case, dtype)
    assert nbt == 1 and slack_max == 0.0                                    # statistics updated once; nothing written outside the view's channels
    tol = {'f32': dict(logits=2e-5, gw=5e-5, gb=2e-5, dgamma=1e-4, dbeta=1e-4, dy=1e-4, running_mean=1e-5, running_var=1e-5),
           'bf16': dict(logits=3e-3, gw=3e-3, gb=2e-5, dgamma=3e-3, dbeta=3e-3, dy=1.5e-2, running_mean=1e-5, running_var=1e-5)}[dtype]
    for k, (got, want) in res.items():
        assert torch.isfinite(got).all(), k
        err = float((got.double() - want.double()).abs().max() / (want.double().abs().max() + 1e-12))
        assert err <= tol[k], '%s: max-rel error %.3e > %.1e' % (k, err, tol[k])


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_fused_final_block_equals_separate_launches(dtype, monkeypatch):
    """The shipped training step (SegmentationModel._fit_loop) with the fused final block against SALT_HEAD_BN=0: same operands, same
    rounding points, other summation partitions - logits, loss, every parameter after two steps."""
    from test_gpu_fused_step import _segmentation_model
    results = {}
    for mode in ('fused', 'separate'):
        if mode == 'separate':
            monkeypatch.setenv('SALT_HEAD_BN', '0')
        torch.manual_seed(21)
        m = _segmentation_model('UNetResNet', 'lovasz', dtype=dtype, lr=1e-3)
        m._to_device()
        m.model.train()
        g = torch.Generator().manual_seed(5)
        X = torch.randn(4, 3, 64, 64, generator=g).to(DEV)
        M = (torch.rand(4, 1, 64, 64, generator=g) > 0.6).float()
        Tt = torch.cat([1 - M, M], 1).to(DEV)
        l0 = float(m._fit_loop([X, Tt])['sum'])
        torch.cuda.synchronize()
        eng = m.model.engine()
        net = eng.net((4, 3, 64, 64), True)
        names = [n for n, _, _ in net.fwd.ops] + [n for n, _, _ in net.bwd.ops]
        assert ('head_bn' in names and 'head_bn_bwd' in names) == (mode == 'fused'), names[-8:]
        assert ('head1x1' in names) == (mode == 'separate')
        logits0, grads0 = net.logits.cpu().clone(), eng.grads.cpu().clone()
        l1 = float(m._fit_loop([X, Tt])['sum'])
        torch.cuda.synchronize()
        results[mode] = (l0, l1, logits0, grads0, eng.flat.cpu().clone(), len(names))
    a, b = results['fused'], results['separate']
    assert a[5] == b[5] - 2                                                   # one launch fewer in each direction

    def rel(x, y):
        return float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30))
    tol = 2e-5 if dtype == 'f32' else 4e-3
    assert rel(a[2], b[2]) <= tol, rel(a[2], b[2])                             # first-step logits
    assert abs(a[0] - b[0]) <= tol * max(1.0, abs(b[0])) and abs(a[1] - b[1]) <= 10 * tol * max(1.0, abs(b[1])), (a[:2], b[:2])
    assert rel(a[3], b[3]) <= (2e-4 if dtype == 'f32' else 3e-2), rel(a[3], b[3])      # whole flat gradient of step 1
    # parameters after two Adam steps.  Adam's first updates are sign-like (|update| <= lr whatever the gradient's size), so a last-bit
    # difference in a near-zero gradient becomes a difference of up to 2 lr per step: the bound that holds by construction is
    # 2 steps x 2 lr per element; the relative L2 is what is typical (fp32: few elements flip; bf16: 2e-3 .. 8e-3 depending on which
    # rounding the summation order of the day realises - it moved from 1.9e-3 to 7.7e-3 when conv_glds_kernel's lane order changed)
    assert float((a[4] - b[4]).abs().max()) <= 4 * 1e-3 * 1.01, float((a[4] - b[4]).abs().max())
    assert rel(a[4], b[4]) <= (2e-3 if dtype == 'f32' else 2e-2), rel(a[4], b[4])


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_scse_applies_the_producer_batchnorm_itself(dtype, monkeypatch):
    """base.DecoderBlock (architectures/base.py:60-85): conv2's BatchNorm + ReLU output has one reader, the scSE block.  Round 6 drops its
    salt_affine_act launch - the scSE kernels transform the raw convolution output on the way in (salt_scse_args.in_fin), forward and
    backward.  Against SALT_SE_IN_BN=0 (separate launch) on the reference-generated DecoderBlock fixture: one forward operator fewer,
    same outputs / gradients / running statistics up to the contraction of one multiply-add; against the golden itself in
    tests/test_gpu_blocks.py (F4_decoderblock_*, which runs the fused default)."""
    from gpu_harness import BlockRun, load_into
    from helpers import golden, T
    from test_gpu_blocks import BLOCKS, _fixture_state, _mods
    A = _mods()
    fx = golden('F4_decoderblock_skip_train')
    make, emit = BLOCKS['F4_decoderblock_skip']
    res = {}
    for mode in ('fused', 'fc_main', 'no_bnb', 'separate'):
        monkeypatch.setenv('SALT_SE_FC_SIDE', '0' if mode == 'fc_main' else '1')   # fc_main: parameter gradients + dgap in the single-workgroup kernel
        if mode == 'no_bnb':
            monkeypatch.setenv('SALT_SE_BNB', '0')       # input transform on, but the layer's bn_bwd keeps its own reduction pass
        if mode == 'separate':
            monkeypatch.setenv('SALT_SE_IN_BN', '0')
        m = make(A)
        load_into(m, _fixture_state(fx, m))
        m.train(True)
        run = BlockRun(m, [T(fx['x']), T(fx['e0'])], emit(m), train=True, dtype=dtype)
        y = run.forward()
        gx, grads = run.backward(T(fx['gy']).to(DEV))
        names = [n for n, _, _ in run.g.fwd.ops]
        sc = [s_ for n, _, s_ in run.g.fwd.ops if n == 'scse'][0]
        assert bool(sc.in_fin) == (mode != 'separate')
        sb = [s_ for n, _, s_ in run.g.bwd.ops if n == 'scse_bwd'][0]
        bnb = [s_ for n, _, s_ in run.g.bwd.ops if n == 'bn_bwd']
        # backward order: scse_bwd, then conv2's bn_bwd: apply-only (partials_ready 3, da_bias) when the scSE pass carried its sums
        assert bool(sb.bnb_acc) == (mode in ('fused', 'fc_main')) and int(bnb[0].partials_ready) == (3 if mode in ('fused', 'fc_main') else 0) and bool(bnb[0].da_bias)
        assert ('scse_fc_grads' in [n for n, _, _ in run.g.bwd.ops]) == (mode != 'fc_main') and int(sb.defer_param_grads) == int(mode != 'fc_main')
        res[mode] = (y, gx, grads, names.count('affine_act'), {k: v.clone().cpu() for k, v in m.state_dict().items() if 'running' in k})
    a, b = res['fused'], res['separate']
    assert a[3] == b[3] - 1 and res['no_bnb'][3] == a[3]
    tol = 1e-5 if dtype == 'f32' else 1e-2

    def close(u, v, what, t=tol):
        e = float((u.double() - v.double()).abs().max() / (v.double().abs().max() + 1e-12))
        assert e <= t, '%s: %.3e' % (what, e)
    close(a[0], b[0], 'y')
    for u, v in zip(a[1], b[1]):
        close(u, v, 'dx', tol * 4)
    for k in a[2]:
        if float(b[2][k].abs().max()) > 0:
            close(a[2][k], b[2][k], k, tol * 4)
    for k in a[4]:
        close(a[4][k], b[4][k], k, 1e-6)
    for u, v in zip(a[1], res['fc_main'][1]):
        close(u, v, 'dx (per-image dgap kernel vs single-workgroup FC backward)', tol * 4)
    for k in a[2]:
        if float(res['fc_main'][2][k].abs().max()) > 0:
            close(a[2][k], res['fc_main'][2][k], k + ' (FC gradients on the weight-gradient queue)', tol * 4)
    c = res['no_bnb']
    for u, v in zip(a[1], c[1]):
        close(u, v, 'dx (sums carried by the scSE pass vs bn_bwd reduction pass)', tol * 4)
    for k in a[2]:
        if float(c[2][k].abs().max()) > 0:
            close(a[2][k], c[2][k], k + ' (carried sums)', tol * 4)


@pytest.mark.parametrize('arch,dtype', [('VanillaUNet', 'f32'), ('VanillaUNet', 'bf16')])
def test_conv_bn_relu_final_blocks_fuse_through_graph_head(arch, dtype, monkeypatch):
    """Graph.head takes over the BatchNorm apply + ReLU of a Conv-BN-ReLU block that feeds the 1x1 logit head whenever that block's statistics
    travel through the fp64 shards (the vanilla U-Net of BASELINE C0 / C1; the k4 transposed convolutions in front of unet_models' heads
    keep the per-tile partials protocol and the separate launches): one training step with the fusion against SALT_HEAD_BN=0 - same
    loss, same gradients to summation order, one operator fewer in each direction."""
    from test_gpu_fused_step import _segmentation_model
    res = {}
    for mode in ('fused', 'separate'):
        monkeypatch.setenv('SALT_HEAD_BN', '1' if mode == 'fused' else '0')
        torch.manual_seed(31)
        m = _segmentation_model(arch, 'bce_dice', dtype=dtype, lr=1e-3)
        m._to_device()
        m.model.train()
        ch = 1 if arch == 'VanillaUNet' else 3
        g = torch.Generator().manual_seed(6)
        X = torch.randn(4, ch, 64, 64, generator=g).to(DEV)
        M = (torch.rand(4, 1, 64, 64, generator=g) > 0.6).float()
        Tt = torch.cat([1 - M, M], 1).to(DEV)
        loss = float(m._fit_loop([X, Tt])['sum'])
        torch.cuda.synchronize()
        eng = m.model.engine()
        net = eng.net((4, ch, 64, 64), True)
        names = [n for n, _, _ in net.fwd.ops] + [n for n, _, _ in net.bwd.ops]
        assert ('head_bn' in names and 'head_bn_bwd' in names and 'head1x1' not in names) == (mode == 'fused'), [n for n in names if 'head' in n]
        res[mode] = (loss, net.logits.cpu().clone(), eng.grads.cpu().clone(), len(names))
    a, b = res['fused'], res['separate']
    assert a[3] == b[3] - 2
    rel = lambda u, v: float((u.double() - v.double()).norm() / (v.double().norm() + 1e-30))
    tol = 2e-5 if dtype == 'f32' else 4e-3
    assert abs(a[0] - b[0]) <= tol * max(1.0, abs(b[0])) and rel(a[1], b[1]) <= tol, (a[0], b[0], rel(a[1], b[1]))
    assert rel(a[2], b[2]) <= (2e-4 if dtype == 'f32' else 3e-2), rel(a[2], b[2])
