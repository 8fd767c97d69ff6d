import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + '/tests', R + '/tests/golden'):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
from torch import nn
import salt_amd
from salt_amd import architectures as A
from gpu_harness import BlockRun
torch.manual_seed(0)
for dtype in ('f32', 'bf16'):
    m = A.ConvBnRelu(16, 16)
    x = torch.randn(2, 16, 12, 12)
    import copy
    cpu = copy.deepcopy(m)
    xr = x.clone().requires_grad_(True)
    yr = F.relu(cpu.conv[1](cpu.conv[0](xr)))
    gy = torch.randn(yr.shape); yr.backward(gy)
    m.train()
    r = BlockRun(m, [x], lambda g, a: m.emit(g, a), train=True, dtype=dtype)
    y = r.forward()
    gx, grads = r.backward(gy.to('cuda:0'))
    def e(a, b): return float((a - b).abs().max() / (b.abs().max() + 1e-30))
    print(dtype, 'y', e(y, yr.detach()), 'gx', e(gx[0], xr.grad))
    ref = dict(cpu.named_parameters())
    for k, g in grads.items():
        print('   ', k, e(g, ref[k].grad), float(g.abs().max()), float(ref[k].grad.abs().max()))
