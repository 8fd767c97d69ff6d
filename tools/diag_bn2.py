import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch, ctypes
import salt_amd
from salt_amd.engine import Program, shaped_view, null_view
from salt_amd._abi import lib, STRUCTS, fill
torch.manual_seed(0)
dev = 'cuda:0'
B, H, W, C = 2, 12, 12, 16
y = torch.randn(B, H, W, C, device=dev)
da = torch.randn(B, H, W, C, device=dev)
gamma = (1 + 0.1 * torch.randn(C, device=dev)); beta = 0.1 * torch.randn(C, device=dev)
yt = y.permute(0, 3, 1, 2).contiguous().requires_grad_(True)
bn = torch.nn.BatchNorm2d(C).to(dev); bn.weight.data.copy_(gamma); bn.bias.data.copy_(beta)
out = torch.relu(bn(yt)); out.backward(da.permute(0, 3, 1, 2).contiguous())
ref = yt.grad.permute(0, 2, 3, 1).contiguous()
mean = y.mean((0, 1, 2)); var = y.var((0, 1, 2), unbiased=False); invstd = 1 / torch.sqrt(var + 1e-5)
a = out.detach().permute(0, 2, 3, 1).contiguous()
for use_a in (True, False):
    dy = torch.zeros_like(y)
    S = STRUCTS['salt_bn_bwd_args'](); fill(S, y=shaped_view(y.data_ptr(), B, H, W, C))
    nparts = lib.salt_bn_bwd_parts(ctypes.byref(S))
    partials = torch.zeros(nparts * 2 * C, device=dev); coef = torch.zeros(3 * C, device=dev); dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev)
    p = Program('t')
    p.add('bn_bwd', dtype=0, da=shaped_view(da.data_ptr(), B, H, W, C), a=shaped_view(a.data_ptr(), B, H, W, C) if use_a else null_view(),
          y=shaped_view(y.data_ptr(), B, H, W, C), relu=1, mean=mean.data_ptr(), invstd=invstd.data_ptr(), gamma=gamma.data_ptr(), beta=beta.data_ptr(),
          partials=partials.data_ptr(), nparts=nparts, dgamma=dg.data_ptr(), dbeta=db.data_ptr(), accumulate_param_grads=0, coef=coef.data_ptr(),
          dy=shaped_view(dy.data_ptr(), B, H, W, C), dres=null_view(), accumulate_dres=0)
    p.run(); torch.cuda.synchronize()
    print("use_a", use_a, 'dy err', float((dy - ref).abs().max() / ref.abs().max()), 'dgamma err', float((dg - bn.weight.grad).abs().max()), 'coef', coef[:3].tolist(), coef[C:C+3].tolist())
    print(' dy', dy[0,0,0,:6].tolist()); print(' rf', ref[0,0,0,:6].tolist())
    xh = (y - mean) * invstd
    mask = (a > 0).float(); gg = da * mask
    k = gamma * invstd; c1 = gg.mean((0,1,2)); c2 = (gg * xh).mean((0,1,2))
    man = k * (gg - c1 - xh * c2)
    print(' manual vs ref', float((man - ref).abs().max() / ref.abs().max()), 'manual vs dy', float((man - dy).abs().max() / ref.abs().max()))
    print(' coef c2', coef[2*C:2*C+3].tolist(), c2[:3].tolist(), 'c1', c1[:3].tolist())
    sc = gamma * invstd; sh = beta - mean * sc
    v = y * sc + sh
    print(' a[..3]', a[0,0,0,:6].tolist(), 'v', v[0,0,0,:6].tolist(), 'da', da[0,0,0,:6].tolist())
    bad = ((dy - ref).abs() > 1e-4 * ref.abs().max())
    print(' nbad', int(bad.sum()), 'of', bad.numel(), 'bad where a>0:', int((bad & (a > 0)).sum()), 'bad where a==0:', int((bad & (a == 0)).sum()))
    idx = bad.nonzero()[:8]; print(idx.tolist())
