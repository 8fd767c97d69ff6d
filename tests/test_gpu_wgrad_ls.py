"""conv_wgrad_ls_kernel (loader-specialised, row-streaming bf16 weight gradient) through the C-ABI:
salt_conv_wgrad + salt_wgrad_reduce against torch's fp32 weight gradient of the same bf16-rounded operands, and against the
previous kernels (SALT_WGRAD_LS=0 in a child process is not needed: the generic path is reached with a shape the new kernel
declines).  Replaces autograd of common_blocks/models.py:133 over architectures/base.py:7-37 / unet_models.py:21-30."""
import ctypes
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _wgrad_hip(p_nhwc, q_nhwc, taps, pad_mode, nsplit=None, q_planar=False, Ca=None, Cb=None, q_step=1):
    """p_nhwc [B,H,W,Csa] / q_nhwc [B,QH,QW,Csb] bf16 device tensors (views of Ca / Cb leading channels); returns dW [Ca][Cb][3][3] fp32."""
    import salt_amd  # noqa: F401
    from salt_amd._abi import STRUCTS, lib, fill, check
    from salt_amd.engine import shaped_view
    B, H, W, csa = p_nhwc.shape
    _, QH, QW, csb = q_nhwc.shape
    csa, csb = p_nhwc.stride(2), q_nhwc.stride(2)                 # pixel stride of a channel slice of a wider buffer
    Ca = csa if Ca is None else Ca
    Cb = csb if Cb is None else Cb
    pv = shaped_view(p_nhwc.data_ptr(), B, H, W, Ca, csa)
    if q_planar:
        # q_nhwc is [planes][B,QH,QW,64]: the view names one plane's pixel stride, q_plane the plane stride
        qv = shaped_view(q_nhwc.data_ptr(), B, QH, QW, Cb, 64)
        qp = B * QH * QW * 64
    else:
        qv = shaped_view(q_nhwc.data_ptr(), B, QH, QW, Cb, csb)
        qp = 0
    S = fill(STRUCTS['salt_conv_wgrad_args'](), dtype=1, p=pv, q=qv, ntaps=9, tap_dy=[t[0] for t in taps], tap_dx=[t[1] for t in taps],
             q_step=q_step, pad_mode=pad_mode, q_plane=qp)
    ns = lib.salt_conv_wgrad_nsplit(ctypes.byref(S))
    assert ns >= 1, lib.salt_last_error()
    if nsplit is not None:
        ns = nsplit
    part = torch.full((ns, 9, Ca, Cb), float('nan'), device='cuda:0')
    S.partials = part.data_ptr()
    S.nsplit = ns
    st = torch.cuda.current_stream().cuda_stream
    check(lib.salt_conv_wgrad(ctypes.byref(S), st), 'salt_conv_wgrad')
    grad = torch.full((Ca, Cb, 3, 3), float('nan'), device='cuda:0')
    R = fill(STRUCTS['salt_wgrad_reduce_args'](), partials=part.data_ptr(), nsplit=ns, ntaps=9, Ca=Ca, Cb=Cb, KH=3, KW=3,
             tap_kh=[t // 3 for t in range(9)], tap_kw=[t % 3 for t in range(9)], grad=grad.data_ptr(), accumulate=0)
    check(lib.salt_wgrad_reduce(ctypes.byref(R), st), 'salt_wgrad_reduce')
    torch.cuda.synchronize()
    return grad.cpu(), ns


def _ref(p, q, replicate):
    """p [B,Ca,H,W], q [B,Cb,H,W] fp32 (bf16-representable) -> dW [Ca][Cb][3][3] of the zero-pad-1 conv, or of the reference's
    ReplicationPad2d((0, 2, 2, 0)) + valid conv (architectures/base.py:21-27)."""
    w = torch.zeros(p.shape[1], q.shape[1], 3, 3, dtype=torch.float64, requires_grad=True)
    qq = q.double()
    if replicate:
        y = F.conv2d(F.pad(qq, (0, 2, 2, 0), mode='replicate'), w)
    else:
        y = F.conv2d(qq, w, padding=1)
    (y * p.double()).sum().backward()
    return w.grad.float()


def _mk(B, C, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, C, H, W, generator=g).bfloat16().float()


CASES = [
    # B, H, W, Ca, Cb
    (2, 16, 16, 64, 64),
    (1, 64, 64, 64, 64),       # one 64-row strip per column block
    (3, 33, 17, 72, 40),       # ragged rows / columns / channel blocks
    (2, 32, 32, 128, 192),     # several channel blocks: the XCD triple map
    (4, 8, 8, 128, 64),        # two images side by side in a k-step
    (5, 8, 8, 160, 72),        # ... odd batch, ragged channels
    (4, 4, 4, 64, 128),        # maps narrower than 8
    (3, 2, 2, 256, 64),
    (2, 128, 16, 32, 64),      # half-empty a-block, long strips
    (1, 9, 40, 64, 32),
]


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('replicate', [False, True])
@pytest.mark.parametrize('ku', [4, 8])
def test_wgrad_ls_vs_torch(case, replicate, ku, monkeypatch):
    monkeypatch.setenv('SALT_WL_KU', str(ku))
    B, H, W, Ca, Cb = case
    p, q = _mk(B, Ca, H, W, 1), _mk(B, Cb, H, W, 2)
    taps = [(dy - 2, dx) for dy in range(3) for dx in range(3)] if replicate else [(dy - 1, dx - 1) for dy in range(3) for dx in range(3)]
    pd = p.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    qd = q.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    ref = _ref(p, q, replicate)
    got, ns = _wgrad_hip(pd, qd, taps, 1 if replicate else 0)
    err = float((got.double() - ref.double()).norm() / ref.double().norm())
    assert torch.isfinite(got).all()
    assert err <= 2e-5, 'rel-L2 %.3e (nsplit %d)' % (err, ns)          # exact products, fp32 accumulation in another order


@pytest.mark.parametrize('nsplit', [1, 3, 7, 40])
@pytest.mark.parametrize('ku', [4, 8])
def test_wgrad_ls_split_starts_mid_strip(nsplit, ku, monkeypatch):
    """Any split count: splits that begin inside a strip open with their own two-halo-row entry; splits beyond the last unit write
    zero slabs."""
    monkeypatch.setenv('SALT_WL_KU', str(ku))
    B, H, W, Ca, Cb = 2, 40, 24, 64, 64
    p, q = _mk(B, Ca, H, W, 3), _mk(B, Cb, H, W, 4)
    taps = [(dy - 1, dx - 1) for dy in range(3) for dx in range(3)]
    pd = p.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    qd = q.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    ref = _ref(p, q, False)
    got, _ = _wgrad_hip(pd, qd, taps, 0, nsplit=nsplit)
    err = float((got.double() - ref.double()).norm() / ref.double().norm())
    assert err <= 2e-5, 'rel-L2 %.3e' % err


@pytest.mark.parametrize('replicate', [False, True])
def test_wgrad_ls_strided_views_and_planar_q(replicate):
    """Channel slices of wider buffers (concat-free skips: pixel stride > channels) and the planar hypercolumn as Q."""
    B, H, W = 2, 32, 32
    Ca, Cb = 64, 128
    p, q = _mk(B, Ca, H, W, 5), _mk(B, Cb, H, W, 6)
    taps = [(dy - 2, dx) for dy in range(3) for dx in range(3)] if replicate else [(dy - 1, dx - 1) for dy in range(3) for dx in range(3)]
    ref = _ref(p, q, replicate)
    # strided: P lives in channels [16, 80) of a 96-channel buffer, Q in channels [8, 136) of a 160-channel buffer
    pbuf = torch.randn(B, H, W, 96).bfloat16().cuda()
    qbuf = torch.randn(B, H, W, 160).bfloat16().cuda()
    pbuf[..., 16:80] = p.permute(0, 2, 3, 1).bfloat16().cuda()
    qbuf[..., 8:136] = q.permute(0, 2, 3, 1).bfloat16().cuda()
    got, _ = _wgrad_hip(pbuf[..., 16:], qbuf[..., 8:], taps, 1 if replicate else 0, Ca=Ca, Cb=Cb)
    err = float((got.double() - ref.double()).norm() / ref.double().norm())
    assert err <= 2e-5, 'strided: rel-L2 %.3e' % err
    # planar: Q as two dense [B,H,W,64] planes
    planes = torch.stack([q[:, 0:64].permute(0, 2, 3, 1), q[:, 64:128].permute(0, 2, 3, 1)]).contiguous().bfloat16().cuda()
    pd = p.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
    got, _ = _wgrad_hip(pd, planes.view(2 * B, H, W, 64)[:B], taps, 1 if replicate else 0, q_planar=True, Cb=Cb)
    err = float((got.double() - ref.double()).norm() / ref.double().norm())
    assert err <= 2e-5, 'planar: rel-L2 %.3e' % err


CASES_S2 = [
    # B, QH, QW, Ca (= cout), Cb (= cin): nn.Conv2d(Cb, Ca, 3, stride 2, padding 1) - ResNet layer2-4 conv1
    (2, 32, 32, 128, 64),
    (2, 16, 16, 256, 128),     # P is 8 x 8: two images side by side in a k-step
    (3, 8, 8, 512, 256),       # P is 4 x 4
    (3, 17, 9, 72, 40),        # odd input sizes, ragged channel blocks
    (1, 64, 64, 128, 64),
    (5, 15, 13, 48, 32),
]


@pytest.mark.parametrize('case', CASES_S2)
@pytest.mark.parametrize('nsplit', [None, 3])
def test_wgrad_ls_stride2_vs_torch(case, nsplit):
    """The stride-2 variant (Q de-interleaved into even / odd column planes by the loader): dW of nn.Conv2d(k3, s2, p1)."""
    B, QH, QW, Ca, Cb = case
    PH, PW = (QH + 2 - 3) // 2 + 1, (QW + 2 - 3) // 2 + 1
    p, q = _mk(B, Ca, PH, PW, 11), _mk(B, Cb, QH, QW, 12)
    w = torch.zeros(Ca, Cb, 3, 3, dtype=torch.float64, requires_grad=True)
    (F.conv2d(q.double(), w, stride=2, padding=1) * p.double()).sum().backward()
    ref = w.grad.float()
    taps = [(dy - 1, dx - 1) for dy in range(3) for dx in range(3)]
    got, ns = _wgrad_hip(p.permute(0, 2, 3, 1).contiguous().bfloat16().cuda(), q.permute(0, 2, 3, 1).contiguous().bfloat16().cuda(), taps, 0,
                         nsplit=nsplit, q_step=2)
    err = float((got.double() - ref.double()).norm() / ref.double().norm())
    assert torch.isfinite(got).all()
    assert err <= 2e-5, 'rel-L2 %.3e (nsplit %d)' % (err, ns)


def test_wgrad_ls_is_the_kernel_that_runs_and_matches_previous_kernels():
    """The plan hands the ResNet34 layer shapes to the new kernel (salt_conv_wgrad_nsplit follows ITS split rule), and the result
    equals the previous kernels' (child process with SALT_WGRAD_LS=0) to fp32 summation-order noise."""
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_wgrad_ls as t
p, q = t._mk(2, 64, 32, 32, 7), t._mk(2, 64, 32, 32, 8)
taps = [(dy - 1, dx - 1) for dy in range(3) for dx in range(3)]
got, ns = t._wgrad_hip(p.permute(0, 2, 3, 1).contiguous().bfloat16().cuda(), q.permute(0, 2, 3, 1).contiguous().bfloat16().cuda(), taps, 0)
torch.save(got, sys.argv[1])
''' % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import tempfile
    outs = []
    for ls in ('1', '0'):
        with tempfile.NamedTemporaryFile(suffix='.pt') as f:
            env = dict(os.environ, SALT_WGRAD_LS=ls)
            subprocess.run([sys.executable, '-c', code, f.name], check=True, env=env, timeout=600)
            outs.append(torch.load(f.name))
    d = float((outs[0].double() - outs[1].double()).norm() / outs[1].double().norm())
    assert d <= 2e-5, d


def test_batched_slab_reduction_is_bit_identical_to_the_per_layer_launches():
    """salt_wgrad_reduce_batched (round 6): several layers' slab reductions in ONE launch - per element the same loads and summation order as
    salt_wgrad_reduce, for nsplit <= 8 (thread per element), nsplit > 8 (4 split rows x 64 elements), a weight slice (ldb), a tap-GEMM slab
    (a_mod), and accumulate - compared bit for bit."""
    import numpy as np
    import salt_amd  # noqa: F401
    from salt_amd._abi import STRUCTS, lib, fill, check
    g = torch.Generator().manual_seed(9)
    jobs_spec = [  # nsplit, ntaps, Ca, Cb, KH, KW, ldb, a_mod, accumulate
        (2, 9, 64, 64, 3, 3, 0, 0, 0), (8, 9, 40, 24, 3, 3, 0, 0, 1), (32, 9, 64, 32, 3, 3, 0, 0, 0), (13, 1, 96, 16, 1, 1, 0, 0, 0),
        (4, 9, 32, 64, 3, 3, 160, 0, 0), (2, 1, 9 * 16, 24, 3, 3, 0, 16, 0), (128, 4, 16, 16, 4, 4, 0, 0, 1)]
    st = torch.cuda.current_stream().cuda_stream
    structs, outs_single, outs_batched = [], [], []
    for k, (ns, nt, Ca, Cb, KH, KW, ldb, a_mod, acc) in enumerate(jobs_spec):
        part = torch.randn(ns * nt * Ca * Cb, generator=g).cuda()
        rows = a_mod if a_mod else Ca
        width = ldb if ldb else Cb
        base = torch.randn(rows * width * KH * KW, generator=g).cuda()
        taps = [(t // KW, t % KW) for t in range(KH * KW)][:(Ca // a_mod if a_mod else nt)]
        g1, g2 = base.clone(), base.clone()
        mk = lambda gr: fill(STRUCTS['salt_wgrad_reduce_args'](), partials=part.data_ptr(), nsplit=ns, ntaps=nt, Ca=Ca, Cb=Cb, KH=KH, KW=KW,
                             tap_kh=[t[0] for t in taps], tap_kw=[t[1] for t in taps], grad=gr.data_ptr(), accumulate=acc, ldb=ldb, a_mod=a_mod)
        check(lib.salt_wgrad_reduce(ctypes.byref(mk(g1)), st), 'wgrad_reduce')
        structs.append((mk(g2), part))
        outs_single.append(g1); outs_batched.append(g2)
    blocks = [lib.salt_wgrad_reduce_job_blocks(ctypes.byref(s_)) for s_, _ in structs]
    assert min(blocks) > 0
    table = torch.frombuffer(bytearray(b''.join(bytes(s_) for s_, _ in structs)), dtype=torch.uint8).cuda()
    pref = torch.from_numpy(np.concatenate([[0], np.cumsum(blocks)]).astype(np.int32)).cuda()
    check(lib.salt_wgrad_reduce_batched(ctypes.byref(fill(STRUCTS['salt_wgrad_reduce_batched_args'](), jobs=table.data_ptr(), job_block0=pref.data_ptr(),
                                                           njobs=len(structs), total_blocks=int(sum(blocks)))), st), 'wgrad_reduce_batched')
    torch.cuda.synchronize()
    for k, (a_, b_) in enumerate(zip(outs_single, outs_batched)):
        assert torch.equal(a_, b_), k
    bad = fill(STRUCTS['salt_wgrad_reduce_args'](), partials=1, nsplit=1, ntaps=1, Ca=4, Cb=4, KH=1, KW=1, tap_kh=[3], tap_kw=[0], grad=1)
    assert lib.salt_wgrad_reduce_job_blocks(ctypes.byref(bad)) < 0


def test_training_step_with_batched_reductions_equals_per_layer_reductions(monkeypatch, deterministic_sums):
    """The step with its slab reductions batched (SALT_WGRAD_BATCH_MB=12: a launch per ~12 MB of gradient, every layer its own slab - opt-in,
    measured slower: engine.Graph._reduce_batching) against the default (a launch per layer, one shared slab workspace): the flat
    gradient buffer and the parameters after two steps are bit-identical; the backward program is shorter."""
    import sys
    import os as _os
    sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))
    from test_gpu_fused_step import _segmentation_model
    res = {}
    for mode in ('batched', 'per_layer'):
        monkeypatch.setenv('SALT_WGRAD_BATCH_MB', '12' if mode == 'batched' else '0')
        torch.manual_seed(4)
        m = _segmentation_model('UNetResNet', 'lovasz', dtype='bf16', lr=1e-3)
        m._to_device(); m.model.train()
        gg = torch.Generator().manual_seed(5)
        X = torch.randn(4, 3, 128, 128, generator=gg).cuda()
        M = (torch.rand(4, 1, 128, 128, generator=gg) > 0.6).float()
        Tt = torch.cat([1 - M, M], 1).cuda()
        ls = [float(m._fit_loop([X, Tt])['sum']) for _ in range(2)]
        torch.cuda.synchronize()
        eng = m.model.engine()
        net = eng.net((4, 3, 128, 128), True)
        names = [n for n, _, _ in net.bwd.ops]
        res[mode] = (ls, eng.grads.clone(), eng.flat.clone(), names.count('wgrad_reduce'), names.count('wgrad_reduce_batched'), len(net.g.grad_ready))
    a, b = res['batched'], res['per_layer']
    assert a[4] >= 2 and a[4] <= 14 and b[4] == 0 and b[3] >= 50 and a[3] <= 2, (a[3:], b[3:])     # (the stem keeps its two own reductions)
    assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert a[5] == b[5]                                                          # every parameter still has its gradient-ready position
