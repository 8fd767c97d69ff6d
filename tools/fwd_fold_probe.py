#!/usr/bin/env python
"""Forward-only probe: the train-mode forward program of the headline network with / without SALT_FWD_BN_FOLD (BatchNorm apply + ReLU of
single-consumer activations in the consumer convolution's loader), timed alone.  usage: SALT_FWD_BN_FOLD=0|1 python tools/fwd_fold_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import salt_amd
from salt_amd import architectures as A
from salt_amd.engine import Graph
from salt_amd.runtime import Engine

dev = torch.device('cuda', 0)
torch.manual_seed(0)
m = A.UNetResNet(34, 2, use_hypercolumn=True).to(dev)
m.train()
eng = Engine(m, dev, 'bf16')
g = Graph(eng, True)
x = g.alloc((32, 3, 128, 128), torch.float32); x.normal_()
logits = g.alloc((32, 2, 128, 128), torch.float32)
m.emit(g, x, logits)
g._resolve_lazies()
g.finalize()
eng.refresh(True)
names = [n for n, _, _ in g.fwd.ops]
for _ in range(10):
    g.fwd.run(side=eng.side_stream)
torch.cuda.synchronize()
res = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(50):
        g.fwd.run(side=eng.side_stream)
    torch.cuda.synchronize()
    res.append(round((time.perf_counter() - t0) / 50 * 1e3, 4))
import ctypes
from salt_amd._abi import lib
kid = {}
for n, _, s in g.fwd.ops:
    if n == 'conv' and s.in_fin:
        k = int(lib.salt_conv_kernel_id(ctypes.byref(s))); kid[k] = kid.get(k, 0) + 1
print('SALT_FWD_BN_FOLD=%s folded=%d ops=%d affine_act=%d fwd_ms=%s folded_kernel_ids=%s' % (os.environ.get('SALT_FWD_BN_FOLD', '0'), getattr(g, 'n_folded', 0), len(names), names.count('affine_act'), res, kid))
