"""conv_ws_kernel (csrc/conv_ws.hip), the weight-stationary multi-tile kernel of the 3x3 layers with <= 64 channels - torchvision
BasicBlock convs of ResNet34 layer1 (architectures/encoders.py:6-45), Conv2dBnRelu of the shallow DecoderBlocks
(architectures/base.py:7-37) and their data gradients - through the C-ABI against torch CPU fp32 and against conv_mfma_kernel on
the same launch.  cfg = 9 | cap << 8 asks for the kernel wherever it applies and caps the workgroups per XCD, so that small tensors
walk the whole pipeline: several tiles per workgroup, both wave groups, the halo DMA issued from the epilogue, the aliased
transposition slices of the 64 -> 64 variant."""
import ctypes

import pytest
import torch
import torch.nn.functional as F
from torch import nn

from helpers import assert_close

pytestmark = pytest.mark.gpu
TOLBF = 4e-2


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _force_cfg(g, cfg):
    orig = g._conv_launch

    def launch(*args, **kw):
        kw.setdefault('cfg', cfg)
        return orig(*args, **kw)
    g._conv_launch = launch
    orig_parts = g._conv_parts
    g._conv_parts = lambda *a_, **k_: orig_parts(*a_, cfg=cfg, **k_)


def _kernel_ids(prog):
    from salt_amd._abi import lib
    return [lib.salt_conv_kernel_id(ctypes.byref(s)) for name, _, s in prog.ops if name == 'conv']


def _ls(cap, ni):
    """cfg word that asks for conv_ls_kernel (id 10) with `cap` workgroups per XCD and 32 * ni output channels per item"""
    return 10 | (cap << 8) | (ni << 16)


# conv_ls_kernel (loader-specialised streaming kernel, K = 9 Cin too large for resident weights): B, Cin, H, W, Cout, cfg
LS_CASES = [
    (1, 128, 32, 32, 128, _ls(4, 1)),    # 4 tiles x 4 channel blocks, one item per workgroup, 4 chunks through the 4-deep ring
    (2, 128, 64, 32, 64, _ls(2, 1)),     # 16 tiles, 2 channel blocks, 2 items per workgroup: the chunk stream runs across items
    (2, 128, 32, 64, 128, _ls(2, 2)),    # 64 channels per item (2-deep ring), 2 items per workgroup
    (1, 256, 16, 16, 64, _ls(2, 1)),     # 8 chunks per item, a single tile: most XCDs have no item
    (1, 64, 16, 32, 32, _ls(1, 1)),      # fewer chunks (2) than the ring's prefetch depth (3)
    (3, 96, 32, 32, 96, _ls(3, 1)),      # 3 chunks, 3 channel blocks
    (2, 320, 32, 16, 64, _ls(1, 2)),     # 10 chunks, 64 channels per item (the final convolution's shape, scaled down)
]

# B, Cin, H, W, Cout, workgroups per XCD (0: one per tile, up to one per CU)
WS_CASES = [
    (2, 64, 32, 32, 64, 0),      # 8 tiles, 8 workgroups: one tile each (group 1 idles)
    (1, 64, 64, 64, 64, 1),      # 16 tiles over 8 workgroups: 2-tile pipelines, both groups, no mid-kernel DMA (the C2 layer1 regime)
    (3, 64, 64, 32, 64, 1),      # 24 tiles: 3 per workgroup - the epilogue issues the next halo into its own transposition slices
    (5, 64, 32, 64, 32, 1),      # 64 -> 32 (DecoderBlock dec1.conv1), 5 tiles per workgroup; its data gradient is the 32 -> 64 variant
    (5, 32, 32, 64, 64, 1),      # 32 -> 64 (dec1.conv2), data gradient 64 -> 32
    (2, 64, 48, 80, 64, 2),      # 30 tiles, 16 workgroups: uneven tile counts (2 and 1) inside one XCD range
    (2, 64, 32, 32, 128, 2),     # two channel blocks of 64 (each workgroup keeps ONE block's weights): the shape of dec2.conv1's data gradient
    (1, 64, 32, 64, 320, 5),     # five channel blocks: the final convolution's data gradient (64 -> 320), scaled down
]


@pytest.mark.parametrize('case', [('ws',) + c[:5] + (9 | (c[5] << 8),) for c in WS_CASES] + [('ls',) + c for c in LS_CASES])
@pytest.mark.parametrize('replicate', [False, True])
def test_ws_conv_bn_relu_train_vs_torch(case, replicate):
    """conv (+ bias) -> train-mode BN -> ReLU, bf16: forward (statistics through the fp64 shards), data gradient, weight / BN gradients.
    replicate = the reference's top/right replicate padding (base.py:21-27): the forward launch clamps its halo rows."""
    from gpu_harness import BlockRun
    kind, B, Cin, H, W, Cout, cfg = case
    kid = 9 if kind == 'ws' else 10
    conv = nn.Conv2d(Cin, Cout, 3, 1, 0 if replicate else 1, bias=True)
    bn = nn.BatchNorm2d(Cout)
    mod = nn.Sequential(conv, bn)
    with torch.no_grad():
        conv.weight.copy_(_rand(conv.weight.shape, 1, (2.0 / (Cin * 9)) ** 0.5)); conv.bias.copy_(0.1 * _rand((Cout,), 6))
        bn.weight.copy_(1 + 0.1 * _rand((Cout,), 2)); bn.bias.copy_(0.1 * _rand((Cout,), 3))
    x = _rand((B, Cin, H, W), 4).bfloat16().float()

    def emit(g, a):
        _force_cfg(g, cfg)
        return g.conv(a, conv, bn, relu=True, replicate=replicate)

    mod.train()
    run = BlockRun(mod, [x], emit, train=True, dtype='bf16')
    assert _kernel_ids(run.g.fwd) == [kid]
    # the data gradient swaps the channel counts.  conv_ws also runs the FUSED-fold data gradient of the replicate-padded layer (extended
    # 34 x 34 ... grids: ragged tiles, pad ring folded onto the edge pixels by lane permutation); conv_ls only plain ones with >= 64
    # input channels (= Cout here)
    ids = _kernel_ids(run.g.bwd)
    if kind == 'ws':
        assert ids == [9] or Cout > 64, ids
    elif not replicate:
        assert ids == [10] or Cout < 64, ids
    y = run.forward()
    ref_conv, ref_bn = nn.Conv2d(Cin, Cout, 3, 1, 0 if replicate else 1, bias=True), nn.BatchNorm2d(Cout)
    with torch.no_grad():
        ref_conv.weight.copy_(conv.weight.detach().cpu().bfloat16().float()); ref_conv.bias.copy_(conv.bias.detach().cpu())
        ref_bn.weight.copy_(bn.weight.detach().cpu()); ref_bn.bias.copy_(bn.bias.detach().cpu())
    xr = x.clone().requires_grad_(True)
    xin = F.pad(xr, (0, 2, 2, 0), mode='replicate') if replicate else xr
    yr = F.relu(ref_bn(ref_conv(xin)))
    assert_close(y, yr, TOLBF, 'y')
    assert_close(bn.running_mean.cpu(), ref_bn.running_mean, TOLBF, 'running_mean')
    assert_close(bn.running_var.cpu(), ref_bn.running_var, TOLBF, 'running_var')
    gy = _rand(tuple(yr.shape), 5)
    yr.backward(gy)
    gx, grads = run.backward(gy.to('cuda:0'))
    for name, got, want in [('dgrad', gx[0], xr.grad), ('wgrad', grads['0.weight'], ref_conv.weight.grad)]:
        l2 = float((got.double() - want.double()).norm() / want.double().norm())
        assert l2 <= TOLBF, '%s: rel-L2 %.3e' % (name, l2)
    assert_close(grads['1.weight'], ref_bn.weight.grad, 3 * TOLBF, 'dgamma')
    assert_close(grads['1.bias'], ref_bn.bias.grad, 3 * TOLBF, 'dbeta')
    y2 = run.forward()                                             # the program is re-runnable (shards cleared by its head)
    assert float((y2.float() - y.float()).abs().max()) <= 1e-6 * float(y.float().abs().max())


@pytest.mark.parametrize('case', [(1, 64, 32, 32, 64, 64, 9 | (1 << 8)), (3, 64, 64, 32, 64, 64, 9 | (1 << 8)), (2, 32, 64, 64, 64, 32, 9 | (1 << 8)),
                                  (2, 64, 32, 64, 32, 64, 9 | (1 << 8)), (2, 128, 32, 32, 128, 128, _ls(2, 1)), (1, 128, 32, 64, 128, 128, _ls(1, 2)),
                                  (2, 256, 16, 32, 128, 256, _ls(4, 1))])
def test_ws_equals_conv_mfma_kernel_on_a_residual_block(case):
    """A BasicBlock-shaped chain (conv-BN-ReLU, conv-BN, + identity, ReLU) in train mode, once on conv_ws_kernel and once on
    conv_mfma_kernel: forward statistics, (+)= data gradients, the BatchNorm-backward sums carried by the second layer's data gradient
    (with the residual's ReLU mask read from the block output) - results of the two kernels on the same launches agree to bf16
    rounding of single values (they sum in different orders), and both match torch."""
    from gpu_harness import BlockRun
    B, Cin, H, W, Cmid, Cout, cfg_new = case
    kid = cfg_new & 0xff
    res_ok = Cin == Cout
    outs = {}
    for tag, cfg in (('ws', cfg_new), ('mfma', 2)):
        c1, b1, c2, b2 = nn.Conv2d(Cin, Cmid, 3, 1, 1, bias=False), nn.BatchNorm2d(Cmid), nn.Conv2d(Cmid, Cout, 3, 1, 1, bias=False), nn.BatchNorm2d(Cout)
        mod = nn.Sequential(c1, b1, c2, b2)
        with torch.no_grad():
            c1.weight.copy_(_rand(c1.weight.shape, 11, (2.0 / (Cin * 9)) ** 0.5)); c2.weight.copy_(_rand(c2.weight.shape, 12, (2.0 / (Cmid * 9)) ** 0.5))
            for i, b in enumerate((b1, b2)):
                b.weight.copy_(1 + 0.1 * _rand(b.weight.shape, 13 + i)); b.bias.copy_(0.1 * _rand(b.bias.shape, 15 + i))
        x = _rand((B, Cin, H, W), 17).bfloat16().float()

        def emit(g, a):
            _force_cfg(g, cfg)
            # a leading 1x1-free identity consumer keeps `a` a plain activation; the block: a -> conv1 -> conv2 (+ a)
            h = g.conv(a, c1, b1, relu=True)
            return g.conv(h, c2, b2, relu=True, res=a if res_ok else None)

        mod.train()
        run = BlockRun(mod, [x], emit, train=True, dtype='bf16')
        ids_f, ids_b = _kernel_ids(run.g.fwd), _kernel_ids(run.g.bwd)
        if tag == 'ws':
            assert ids_f == [kid, kid] and ids_b == [kid, kid], (ids_f, ids_b)
            ready = [int(st.partials_ready) for name, _, st in run.g.bwd.ops if name == 'bn_bwd']
            assert ready == [0, 3], ready          # layer 1's BatchNorm-backward sums came from layer 2's data-gradient launch
        else:
            assert 9 not in ids_f + ids_b and 10 not in ids_f + ids_b
        y = run.forward()
        gy = _rand(tuple(y.shape), 18)
        gx, grads = run.backward(gy.to('cuda:0'))
        outs[tag] = (y, gx[0], grads, {k: v.detach().cpu().clone() for k, v in mod.state_dict().items()})
    # the two kernels against each other
    y_w, y_m = outs['ws'][0], outs['mfma'][0]
    assert_close(y_w, y_m, 1e-2, 'forward ws vs mfma')
    l2 = float((outs['ws'][1].double() - outs['mfma'][1].double()).norm() / outs['mfma'][1].double().norm())
    assert l2 < 1e-2, 'dgrad ws vs mfma rel-L2 %.3e' % l2
    for k in outs['ws'][2]:
        a, b = outs['ws'][2][k].double(), outs['mfma'][2][k].double()
        assert float((a - b).norm() / (b.norm() + 1e-12)) < 2e-2, k
    for k in outs['ws'][3]:
        if 'running' in k:
            assert_close(outs['ws'][3][k], outs['mfma'][3][k], 1e-3, k)
    # and against torch CPU fp32 (weights rounded to bf16 as the kernels see them)
    c1r, b1r, c2r, b2r = nn.Conv2d(Cin, Cmid, 3, 1, 1, bias=False), nn.BatchNorm2d(Cmid), nn.Conv2d(Cmid, Cout, 3, 1, 1, bias=False), nn.BatchNorm2d(Cout)
    with torch.no_grad():
        c1r.weight.copy_(_rand(c1r.weight.shape, 11, (2.0 / (Cin * 9)) ** 0.5).bfloat16().float())
        c2r.weight.copy_(_rand(c2r.weight.shape, 12, (2.0 / (Cmid * 9)) ** 0.5).bfloat16().float())
        for i, b in enumerate((b1r, b2r)):
            b.weight.copy_(1 + 0.1 * _rand(b.weight.shape, 13 + i)); b.bias.copy_(0.1 * _rand(b.bias.shape, 15 + i))
    xr = _rand((B, Cin, H, W), 17).bfloat16().float().requires_grad_(True)
    hr = F.relu(b1r(c1r(xr)))
    yr = b2r(c2r(hr))
    yr = F.relu(yr + xr) if res_ok else F.relu(yr)
    assert_close(y_w, yr, 2 * TOLBF, 'forward vs torch')
    yr.backward(_rand(tuple(yr.shape), 18))
    l2 = float((outs['ws'][1].double() - xr.grad.double()).norm() / xr.grad.double().norm())
    assert l2 < 2 * TOLBF, 'dgrad vs torch rel-L2 %.3e' % l2


@pytest.mark.parametrize('case', [(2, 64, 32, 32, 64, 9 | (1 << 8)), (4, 64, 32, 32, 32, 9 | (1 << 8)), (4, 32, 32, 32, 64, 9 | (1 << 8)),
                                  (2, 128, 32, 32, 96, _ls(3, 1)), (2, 192, 32, 32, 128, _ls(2, 2)),
                                  # two-tile items (cfg bit 20): one tile per workgroup (half an item), 4 tiles, 3 tiles across images
                                  (2, 192, 32, 32, 128, _ls(2, 2) | (1 << 20)), (4, 256, 64, 32, 64, _ls(1, 2) | (1 << 20)), (3, 128, 48, 32, 128, _ls(2, 2) | (1 << 20)),
                                  (3, 64, 40, 24, 64, 9 | (1 << 8)), (2, 64, 101, 101, 64, 9 | (2 << 8)), (2, 32, 20, 50, 128, 9 | (2 << 8))])   # ragged grids
def test_ws_eval_folded_bn_relu_vs_torch(case):
    """eval mode: bias + folded BatchNorm + ReLU in the epilogue (salt_conv_args.bias / scale / shift / relu)."""
    from gpu_harness import BlockRun
    B, Cin, H, W, Cout, cfg = case
    conv, bn = nn.Conv2d(Cin, Cout, 3, 1, 1, bias=True), nn.BatchNorm2d(Cout)
    mod = nn.Sequential(conv, bn)
    with torch.no_grad():
        conv.weight.copy_(_rand(conv.weight.shape, 1, (2.0 / (Cin * 9)) ** 0.5)); conv.bias.copy_(0.1 * _rand((Cout,), 6))
        bn.weight.copy_(1 + 0.1 * _rand((Cout,), 2)); bn.bias.copy_(0.1 * _rand((Cout,), 3))
        bn.running_mean.copy_(0.2 * _rand((Cout,), 7)); bn.running_var.copy_(1 + 0.3 * _rand((Cout,), 8).abs())
    x = _rand((B, Cin, H, W), 4).bfloat16().float()

    def emit(g, a):
        _force_cfg(g, cfg)
        return g.conv(a, conv, bn, relu=True)

    mod.eval()
    run = BlockRun(mod, [x], emit, train=False, dtype='bf16')
    assert _kernel_ids(run.g.fwd) == [cfg & 0xff]
    y = run.forward()
    rc, rb = nn.Conv2d(Cin, Cout, 3, 1, 1, bias=True), nn.BatchNorm2d(Cout)
    rc.load_state_dict({k: v.detach().cpu() for k, v in conv.state_dict().items()}); rb.load_state_dict({k: v.detach().cpu() for k, v in bn.state_dict().items()})
    with torch.no_grad():
        rc.weight.copy_(rc.weight.bfloat16().float())
    rb.eval()
    with torch.no_grad():
        yr = F.relu(rb(rc(x)))
    assert_close(y, yr, TOLBF, 'eval y')


@pytest.mark.parametrize('case', [(4, 256, 64, 32, 64, 1), (3, 128, 48, 32, 128, 2), (2, 320, 32, 48, 64, 1), (5, 128, 16, 16, 192, 3)])
@pytest.mark.parametrize('replicate', [False, True])
def test_ls_two_tile_items_are_bit_identical_to_single_tile_items(case, replicate):
    """conv_ls_kernel MT = 2 (two pixel tiles per stream of weight chunks; cfg bit 20) against MT = 1 (bit 21): the same MFMA sequence
    per output value, so the eval-mode results must be equal bit for bit - zero padding and the reference's replicate padding
    (architectures/base.py:21-27), whole and half items, tiles of different images in one item."""
    from gpu_harness import BlockRun
    B, Cin, H, W, Cout, cap = case
    conv, bn = nn.Conv2d(Cin, Cout, 3, 1, 0 if replicate else 1, bias=True), nn.BatchNorm2d(Cout)
    mod = nn.Sequential(nn.ReplicationPad2d((0, 2, 2, 0)), conv, bn) if replicate else nn.Sequential(conv, bn)
    with torch.no_grad():
        conv.weight.copy_(_rand(conv.weight.shape, 1, (2.0 / (Cin * 9)) ** 0.5)); conv.bias.copy_(0.1 * _rand((Cout,), 6))
        bn.weight.copy_(1 + 0.1 * _rand((Cout,), 2)); bn.bias.copy_(0.1 * _rand((Cout,), 3))
        bn.running_mean.copy_(0.2 * _rand((Cout,), 7)); bn.running_var.copy_(1 + 0.3 * _rand((Cout,), 8).abs())
    x = _rand((B, Cin, H, W), 4).bfloat16().float()
    mod.eval()
    ys = []
    for bit in (20, 21):
        def emit(g, a, bit=bit):
            _force_cfg(g, _ls(cap, 2) | (1 << bit))
            return g.conv(a, conv, bn, relu=True, replicate=replicate) if replicate else g.conv(a, conv, bn, relu=True)
        run = BlockRun(mod, [x], emit, train=False, dtype='bf16')
        assert _kernel_ids(run.g.fwd) == [10]
        ys.append(run.forward())
    assert torch.isfinite(ys[0]).all()
    assert torch.equal(ys[0], ys[1])
    with torch.no_grad():
        rc = nn.Conv2d(Cin, Cout, 3, 1, 0 if replicate else 1, bias=True); rc.load_state_dict({k: v.detach().cpu() for k, v in conv.state_dict().items()})
        rc.weight.copy_(rc.weight.bfloat16().float())
        rb = nn.BatchNorm2d(Cout); rb.load_state_dict({k: v.detach().cpu() for k, v in bn.state_dict().items()}); rb.eval()
        xin = nn.ReplicationPad2d((0, 2, 2, 0))(x) if replicate else x
        yr = F.relu(rb(rc(xin)))
    assert_close(ys[0], yr, TOLBF, 'two-tile items vs torch')


@pytest.mark.parametrize('case', [(4, 128, 64, 32, 1), (3, 192, 48, 32, 2)])
def test_ls_two_tile_items_with_the_eval_residual_epilogue_are_bit_identical(case):
    """eval-mode BasicBlock tail - conv, folded BatchNorm, + identity, ReLU in the convolution's epilogue (salt_conv_args.res) - on two-tile
    items (cfg bit 20) against single-tile items (bit 21): the residual pieces are read at each tile's own store addresses."""
    from gpu_harness import BlockRun
    B, C, H, W, cap = case
    conv, bn = nn.Conv2d(C, C, 3, 1, 1, bias=False), nn.BatchNorm2d(C)
    mod = nn.Sequential(conv, bn)
    with torch.no_grad():
        conv.weight.copy_(_rand(conv.weight.shape, 1, (2.0 / (C * 9)) ** 0.5))
        bn.weight.copy_(1 + 0.1 * _rand((C,), 2)); bn.bias.copy_(0.1 * _rand((C,), 3))
        bn.running_mean.copy_(0.2 * _rand((C,), 7)); bn.running_var.copy_(1 + 0.3 * _rand((C,), 8).abs())
    x = _rand((B, C, H, W), 4).bfloat16().float()
    mod.eval()
    ys = []
    for bit in (20, 21):
        def emit(g, a, bit=bit):
            _force_cfg(g, _ls(cap, 2) | (1 << bit))
            return g.conv(a, conv, bn, relu=True, res=a)
        run = BlockRun(mod, [x], emit, train=False, dtype='bf16')
        assert _kernel_ids(run.g.fwd) == [10]
        ys.append(run.forward())
    assert torch.equal(ys[0], ys[1])
    with torch.no_grad():
        rc = nn.Conv2d(C, C, 3, 1, 1, bias=False); rc.weight.copy_(conv.weight.detach().cpu().bfloat16().float())
        rb = nn.BatchNorm2d(C); rb.load_state_dict({k: v.detach().cpu() for k, v in bn.state_dict().items()}); rb.eval()
        yr = F.relu(rb(rc(x)) + x)
    assert_close(ys[0], yr, 2 * TOLBF, 'residual epilogue vs torch')


@pytest.mark.parametrize('shape', [(2, 64, 32, 32, 64), (2, 32, 32, 48, 96), (1, 128, 16, 16, 128)])
def test_bn_apply_in_consumer_loader_forward_is_bit_identical(shape, monkeypatch):
    """salt_conv_args.in_fin / in_fin_acc / in_relu (the SALT_FWD_BN_FOLD switch: forward-only graphs): conv -> BN -> ReLU -> conv ->
    BN -> ReLU (DecoderBlock, architectures/base.py:29-37) with the first BatchNorm + ReLU applied by the SECOND convolution's loader
    instead of a salt_affine_act pass.  The forward values (block output, both layers' running statistics) must equal the separate-pass
    program bit for bit; the affine_act of layer 1 must be gone from the program."""
    from gpu_harness import BlockRun
    B, Cin, H, W, C = shape
    c1, b1, c2, b2 = nn.Conv2d(Cin, C, 3, 1, 1, bias=False), nn.BatchNorm2d(C), nn.Conv2d(C, C, 3, 1, 1, bias=False), nn.BatchNorm2d(C)
    mod = nn.Sequential(c1, b1, c2, b2)
    with torch.no_grad():
        c1.weight.copy_(_rand(c1.weight.shape, 1, (2.0 / (Cin * 9)) ** 0.5)); c2.weight.copy_(_rand(c2.weight.shape, 2, (2.0 / (C * 9)) ** 0.5))
        b1.weight.copy_(1 + 0.1 * _rand((C,), 3)); b1.bias.copy_(0.1 * _rand((C,), 4))
    x = _rand((B, Cin, H, W), 5).bfloat16().float()
    outs = {}
    for fold in ('', '1'):
        if fold:
            monkeypatch.setenv('SALT_FWD_BN_FOLD', '1')
        else:
            monkeypatch.delenv('SALT_FWD_BN_FOLD', raising=False)
        for m in (b1, b2):
            m.reset_running_stats()
        mod.train()
        def emit(g, a):
            _force_cfg(g, 1)                   # the same conv_mfma_kernel tiles (128 pixels x 64 channels) in both programs
            return g.conv(g.conv(a, c1, b1, relu=True), c2, b2, relu=True)
        if fold:                                  # a folded graph has no backward pass: building one fails loudly
            from salt_amd._abi import SaltError
            with pytest.raises(SaltError, match='forward-only'):
                BlockRun(mod, [x], emit, train=True, dtype='bf16')
        run = BlockRun(mod, [x], emit, train=True, dtype='bf16', backward=not fold)
        assert _kernel_ids(run.g.fwd) == [1, 1]
        n_aff = sum(1 for name, _, _ in run.g.fwd.ops if name == 'affine_act')
        assert n_aff == (1 if fold else 2), n_aff
        assert getattr(run.g, 'n_folded', 0) == (1 if fold else 0)
        y = run.forward()
        outs[fold] = (y.clone(), b1.running_mean.clone(), b1.running_var.clone(), b2.running_mean.clone(), b2.running_var.clone())
    for a, b in zip(outs[''], outs['1']):
        assert torch.equal(a.cpu(), b.cpu())


@pytest.mark.parametrize('replicate', [True, False])
def test_planar_operand_buffers_are_bit_identical_to_interleaved(replicate):
    """salt_conv_args.x_plane (conv_ls_kernel reads its 32-channel chunks from dense 64-channel planes), y_plane (conv_ws_kernel writes
    channel block b of the data gradient to plane b) and salt_conv_wgrad_args.q_plane (weight-gradient b-block bb reads plane bb) - the
    planar hypercolumn's three launches (architectures/unet.py:101-109) - on a 128 -> 64 3x3 convolution whose input lives in two planes,
    against the same program on an interleaved input: output, input gradient and weight gradient bit for bit (no BatchNorm: no atomics)."""
    from gpu_harness import BlockRun
    B, Cin, H, W, Cout = 2, 128, 128, 128, 64
    conv = nn.Conv2d(Cin, Cout, 3, 1, 0 if replicate else 1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(_rand(conv.weight.shape, 1, (2.0 / (Cin * 9)) ** 0.5))
    x = _rand((B, Cin, H, W), 4).bfloat16().float()
    gy = _rand((B, Cout, H, W), 9)
    res = {}
    for planar in (False, True):
        def emit(g, a):
            if planar:
                xp = g.new_act(a.B, a.H, a.W, a.C, 'xp', planes=64)
                for k in range(a.C // 64):
                    g.copy(a.slice(64 * k, 64), xp.slice(64 * k, 64))
                a = xp
            return g.conv(a, conv, None, relu=False, replicate=replicate)
        run = BlockRun(conv, [x], emit, train=True, dtype='bf16')
        f = [s for n, _, s in run.g.fwd.ops if n == 'conv']
        b = [s for n, _, s in run.g.bwd.ops if n == 'conv']
        w = [s for n, _, s in run.g.bwd.ops if n == 'conv_wgrad']
        assert _kernel_ids(run.g.fwd) == [10] and _kernel_ids(run.g.bwd) == [9]
        assert (bool(f[0].x_plane), bool(b[0].y_plane), bool(w[0].q_plane)) == (planar, planar, planar)
        y = run.forward()
        gx, gw = run.backward(gy.to('cuda:0'))
        res[planar] = (y, gx[0], gw['weight'])
    yr = F.conv2d(F.pad(x, (0, 2, 2, 0), mode='replicate') if replicate else x, conv.weight.detach().cpu().bfloat16().float(), padding=0 if replicate else 1)
    assert_close(res[True][0], yr, TOLBF, 'y vs torch')
    for a, b_, what in zip(res[True], res[False], ('y', 'dx', 'dw')):
        assert torch.equal(a, b_), what


@pytest.mark.parametrize('case', [(4, 64, 64), (2, 96, 64), (3, 32, 128)])
@pytest.mark.parametrize('train', [True, False])
def test_stem16_kernel_vs_torch_and_conv_mfma(case, train):
    """conv_stem16_kernel (the ResNet stem nn.Conv2d(3, 64, 7, 2, 3) after the 2 x 2 space-to-depth: 16 taps over 16 channels, bf16,
    persistent workgroups with the weights of all taps in LDS) asked for per launch (cfg 13): forward with train-mode statistics /
    folded eval BatchNorm + ReLU against torch on the bf16-rounded operands and against conv_mfma_kernel (cfg 2) on the same launch;
    in train mode also the running statistics and the weight / BatchNorm gradients behind it."""
    from gpu_harness import DEV
    from salt_amd.engine import Graph
    from salt_amd.runtime import Engine
    B, H, W = case
    outs = {}
    for cfg in (13, 2):
        conv, bn = nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64)
        with torch.no_grad():
            conv.weight.copy_(_rand(conv.weight.shape, 41, (2.0 / 147) ** 0.5))
            bn.weight.copy_(1 + 0.1 * _rand((64,), 42)); bn.bias.copy_(0.1 * _rand((64,), 43))
            bn.running_mean.copy_(0.1 * _rand((64,), 44)); bn.running_var.copy_(1 + 0.1 * _rand((64,), 45).abs())
        mod = nn.Sequential(conv, bn).to(DEV)
        mod.train(train)
        eng = Engine(mod, torch.device(DEV), 'bf16')
        g = Graph(eng, train)
        _force_cfg(g, cfg)
        x = _rand((B, 3, H, W), 46)
        xd = g.alloc(tuple(x.shape), torch.float32); xd.copy_(x)
        a = g.conv_first(xd, conv, bn, relu=True)
        out = g.alloc((a.B, a.C, a.H, a.W), torch.float32)
        g.to_nchw(a, out)
        if train:
            g.build_backward()
        g.finalize()
        ids = _kernel_ids(g.fwd)
        assert ids == [cfg], ids
        eng.refresh(train); g.fwd.run()
        res = [out.cpu().clone()]
        if train:
            gy = _rand(tuple(out.shape), 47)
            g.dlogits.copy_(gy); g.bwd.run(); torch.cuda.synchronize()
            off, n = eng.grad_range(conv.weight)
            res += [eng.grads[off:off + n].view(conv.weight.shape).cpu().clone(), bn.running_mean.cpu().clone(), bn.running_var.cpu().clone()]
        outs[cfg] = res
    rc, rb = nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64)
    with torch.no_grad():
        rc.weight.copy_(_rand(rc.weight.shape, 41, (2.0 / 147) ** 0.5).bfloat16().float())
        rb.weight.copy_(1 + 0.1 * _rand((64,), 42)); rb.bias.copy_(0.1 * _rand((64,), 43))
        rb.running_mean.copy_(0.1 * _rand((64,), 44)); rb.running_var.copy_(1 + 0.1 * _rand((64,), 45).abs())
    rc.train(train); rb.train(train)
    x = _rand((B, 3, H, W), 46).bfloat16().float()
    yr = F.relu(rb(rc(x)))
    assert_close(outs[13][0], yr.detach(), TOLBF, 'y vs torch')
    assert_close(outs[13][0], outs[2][0], 2e-2, 'y: stem16 vs conv_mfma')
    if train:
        yr.backward(_rand(tuple(yr.shape), 47))
        l2 = float((outs[13][1].double() - rc.weight.grad.double()).norm() / rc.weight.grad.double().norm())
        assert l2 <= TOLBF, 'wgrad rel-L2 %.3e' % l2
        assert_close(outs[13][2], rb.running_mean, TOLBF, 'running_mean')
        assert_close(outs[13][3], rb.running_var, TOLBF, 'running_var')
        assert_close(outs[13][2], outs[2][2], 1e-3, 'running_mean: stem16 vs conv_mfma')
