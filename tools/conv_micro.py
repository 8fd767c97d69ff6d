#!/usr/bin/env python
"""Micro-benchmark of one dense conv shape through the engine (forward only, eval-mode epilogue off).
usage: python tools/conv_micro.py B Cin H W Cout [k] [stride] [dtype] [reps] [cfg]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + '/tests'):
    sys.path.insert(0, p)
import torch
from torch import nn
import salt_amd
from salt_amd.engine import Graph
from salt_amd.runtime import Engine
B, Cin, H, W, Cout = map(int, sys.argv[1:6])
k = int(sys.argv[6]) if len(sys.argv) > 6 else 3
s = int(sys.argv[7]) if len(sys.argv) > 7 else 1
dtype = sys.argv[8] if len(sys.argv) > 8 else 'bf16'
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 20
cfg = int(sys.argv[10]) if len(sys.argv) > 10 else 0
conv = nn.Conv2d(Cin, Cout, k, s, k // 2, bias=False)
mod = nn.Sequential(conv).to('cuda:0')
eng = Engine(mod, torch.device('cuda:0'), dtype)
g = Graph(eng, False)
x = g.new_act(B, H, W, Cin, 'x')
x.buf.t.normal_()
orig = g._conv_launch
g._conv_launch = lambda *a, **kw: orig(*a, **{**kw, 'cfg': cfg})
y = g.conv(x, conv, None, relu=False)
g.finalize(); eng.refresh(False)
for _ in range(3):
    g.fwd.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    g.fwd.run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
fl = 2.0 * B * y.H * y.W * Cout * Cin * k * k
print('conv B%d %dx%dx%d -> %d k%d s%d %s cfg%d: %.1f us  %.1f TF/s' % (B, H, W, Cin, Cout, k, s, dtype, cfg, dt * 1e6, fl / dt / 1e12))
