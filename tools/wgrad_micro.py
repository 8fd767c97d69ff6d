#!/usr/bin/env python
"""Micro-benchmark of the weight-gradient kernel of one conv shape (isolated, one stream).
usage: python tools/wgrad_micro.py B Cin H W Cout [k] [dtype] [reps]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + '/tests'):
    sys.path.insert(0, p)
import torch
from torch import nn
import salt_amd
from gpu_harness import BlockRun
B, Cin, H, W, Cout = map(int, sys.argv[1:6])
k = int(sys.argv[6]) if len(sys.argv) > 6 else 3
dtype = sys.argv[7] if len(sys.argv) > 7 else 'bf16'
reps = int(sys.argv[8]) if len(sys.argv) > 8 else 50
conv = nn.Conv2d(Cin, Cout, k, 1, k // 2, bias=False)
bn = nn.BatchNorm2d(Cout)
mod = nn.Sequential(conv, bn)
x = torch.randn(B, Cin, H, W)
run = BlockRun(mod, [x], lambda g, a: g.conv(a, conv, bn, relu=True), train=True, dtype=dtype)
run.forward()
run.backward(torch.randn(B, Cout, H, W, device='cuda:0'))
ops = run.g.bwd.ops
for name in ('conv_wgrad', 'wgrad_reduce', 'conv'):
    idx = [i for i, o in enumerate(ops) if o[0] == name]
    for i in idx:
        for _ in range(3):
            run.g.bwd.run(begin=i, end=i + 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run.g.bwd.run(begin=i, end=i + 1)
        e1.record()
        torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) / reps * 1e-3
        fl = 2.0 * B * H * W * Cout * Cin * k * k
        print('%-12s B%d %dx%dx%d -> %d k%d %s: %.1f us  %.1f TF/s' % (name, B, H, W, Cin, Cout, k, dtype, dt * 1e6, fl / dt / 1e12))
