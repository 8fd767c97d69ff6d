"""Row-structured bilinear kernels (round 4): nn.Upsample / F.upsample(mode='bilinear') of base.py:70, unet.py:103-106 and its adjoint
through the C-ABI against torch (fp32 operands rounded to bf16 where the dtype is bf16), and bit for bit against the unit-per-thread
kernels they replace (child process with SALT_BILINEAR_ROWS=0)."""
import ctypes
import os
import subprocess
import sys
import tempfile

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
CASES = [  # B, H, W, C, R, dtype
    (2, 64, 64, 64, 2, 'bf16'), (2, 32, 32, 64, 4, 'bf16'), (2, 16, 16, 64, 8, 'bf16'), (2, 8, 8, 64, 16, 'bf16'),
    (3, 5, 7, 32, 2, 'bf16'), (1, 1, 1, 64, 2, 'bf16'), (2, 3, 2, 256, 4, 'bf16'), (2, 9, 6, 16, 2, 'f32'), (1, 4, 4, 64, 8, 'f32'),
    (2, 8, 8, 8, 2, 'bf16'), (2, 6, 10, 24, 2, 'bf16'),      # 24 channels: 3 pieces per pixel do not divide 256 -> the unit-per-thread kernels
]


def _run_case(B, H, W, C, R, dtype, seed=0):
    """-> (up(x) as fp32 NHWC, up^T(g) as fp32 NHWC, accumulate variant) computed by the library on cuda:0."""
    import salt_amd  # noqa: F401
    from salt_amd._abi import STRUCTS, lib, fill, check
    from salt_amd.engine import shaped_view
    td = torch.bfloat16 if dtype == 'bf16' else torch.float32
    g_ = torch.Generator().manual_seed(seed)
    x = torch.randn(B, H, W, C, generator=g_).to(td).cuda()
    g = torch.randn(B, H * R, W * R, C, generator=g_).to(td).cuda()
    y = torch.zeros(B, H * R, W * R, C, dtype=td, device='cuda:0')
    dx = torch.full((B, H, W, C), float('nan'), dtype=td, device='cuda:0')
    dx2 = torch.ones(B, H, W, C, dtype=td, device='cuda:0')
    tmp = torch.empty(B * H * R * W * C, dtype=td, device='cuda:0')
    st = torch.cuda.current_stream().cuda_stream
    code = 1 if dtype == 'bf16' else 0
    S = STRUCTS['salt_bilinear_args']
    check(lib.salt_bilinear(ctypes.byref(fill(S(), dtype=code, x=shaped_view(x.data_ptr(), B, H, W, C), y=shaped_view(y.data_ptr(), B, H * R, W * R, C),
                                               R=R, backward=0, accumulate=0, tmp=None, align_corners=0)), st))
    for out, acc in ((dx, 0), (dx2, 1)):
        check(lib.salt_bilinear(ctypes.byref(fill(S(), dtype=code, x=shaped_view(out.data_ptr(), B, H, W, C), y=shaped_view(g.data_ptr(), B, H * R, W * R, C),
                                                   R=R, backward=1, accumulate=acc, tmp=tmp.data_ptr(), align_corners=0)), st))
    torch.cuda.synchronize()
    return x.float().cpu(), g.float().cpu(), y.float().cpu(), dx.float().cpu(), dx2.float().cpu()


@pytest.mark.parametrize('case', CASES)
def test_bilinear_rows_vs_torch(case):
    B, H, W, C, R, dtype = case
    x, g, y, dx, dx2 = _run_case(*case)
    xr = x.permute(0, 3, 1, 2).clone().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=R, mode='bilinear', align_corners=False)
    yr.backward(g.permute(0, 3, 1, 2))
    tol = 1e-2 if dtype == 'bf16' else 1e-5
    assert float((y.permute(0, 3, 1, 2) - yr.detach()).abs().max()) <= tol * max(1.0, float(yr.abs().max()))
    ref = xr.grad
    assert float((dx.permute(0, 3, 1, 2) - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max())) * (2 if R >= 4 and dtype == 'bf16' else 1)
    assert float((dx2.permute(0, 3, 1, 2) - (ref + 1)).abs().max()) <= tol * max(1.0, float(ref.abs().max())) * 3


def test_bilinear_rows_bit_identical_to_unit_kernels():
    code = r'''
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_bilinear_rows as t
out = [t._run_case(*c)[2:] for c in t.CASES]
torch.save(out, sys.argv[1])
''' % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    res = []
    for rows in ('1', '0'):
        with tempfile.NamedTemporaryFile(suffix='.pt') as f:
            subprocess.run([sys.executable, '-c', code, f.name], check=True, env=dict(os.environ, SALT_BILINEAR_ROWS=rows), timeout=600)
            res.append(torch.load(f.name))
    for c, a, b in zip(CASES, res[0], res[1]):
        for name, u, v in zip(('up', 'adjoint', 'adjoint accumulate'), a, b):
            assert torch.equal(u, v), (c, name, float((u - v).abs().max()))
