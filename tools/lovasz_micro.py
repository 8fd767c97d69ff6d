import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import salt_amd
from salt_amd import losses
B, H = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 128
g = torch.Generator().manual_seed(0)
z = torch.randn(B, 2, H, H, generator=g).cuda(); m = (torch.rand(B, 1, H, H, generator=g) > 0.6).float().cuda(); t = torch.cat([1 - m, m], 1)
for _ in range(3): losses.native_loss(z, t, 'lovasz')
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): l, d = losses.native_loss(z, t, 'lovasz')
e1.record(); torch.cuda.synchronize()
print('lovasz B%d %dx%d: %.1f us  loss %.6f' % (B, H, H, e0.elapsed_time(e1) * 1e3 / 20, float(l)))
