import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import salt_amd
from salt_amd.models import SegmentationModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
arch = {'model_params': {'architecture': 'UNetResNet', 'out_channels': 2, 'activation': 'sigmoid', 'compute_dtype': 'bf16'},
        'optimizer_params': {'lr': 1e-4}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
m = SegmentationModel(arch, {'epochs': 1}, {}); m._to_device(); m.model.train()
X = torch.randn(B, 3, 128, 128, device='cuda'); T = (torch.rand(B, 1, 128, 128, device='cuda') < 0.3).float(); T = torch.cat([1 - T, T], 1)
for _ in range(5): m._fit_loop([X, T])
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): m._fit_loop([X, T])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('B=%d host enqueue %.3f ms/step, total %.3f ms/step' % (B, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
eng = m.model.engine(); net = eng.net((B, 3, 128, 128), True)
print('ops: fwd %d bwd %d pack %d' % (len(net.fwd), len(net.bwd), len(eng._pack_ops)))
# pure program enqueue cost (no python per op)
torch.cuda.synchronize(); t0 = time.perf_counter(); net.fwd.run(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print('fwd program: enqueue %.3f ms, done %.3f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
