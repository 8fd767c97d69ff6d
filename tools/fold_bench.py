#!/usr/bin/env python
"""Isolated cost of folding BatchNorm-apply + ReLU into the consumer convolution's loader (salt_conv_args.in_*, SALT_EXP_BN_FOLD) versus
the separate salt_affine_act pass, for one conv -> BN -> ReLU -> conv -> BN -> ReLU block (architectures/base.py:29-37) in train mode.
Each forward operator of both programs is timed alone (50 repeats between two events, as tools/wgrad_micro.py does).
usage: python tools/fold_bench.py B C H W [reps]      -> one line per operator and the two totals that matter"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + '/tests'):
    sys.path.insert(0, p)
import torch
from torch import nn
import salt_amd  # noqa: F401
from gpu_harness import BlockRun

B, C, H, W = map(int, sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 50
c1, b1, c2, b2 = nn.Conv2d(C, C, 3, 1, 1, bias=False), nn.BatchNorm2d(C), nn.Conv2d(C, C, 3, 1, 1, bias=False), nn.BatchNorm2d(C)
mod = nn.Sequential(c1, b1, c2, b2)
x = torch.randn(B, C, H, W)
res = {}
for fold in ('', '1'):
    if fold:
        os.environ['SALT_EXP_BN_FOLD'] = '1'
        os.environ['SALT_TIMING_ONLY'] = '1'
    else:
        os.environ.pop('SALT_EXP_BN_FOLD', None)
    mod.train()
    run = BlockRun(mod, [x], lambda g, a: g.conv(g.conv(a, c1, b1, relu=True), c2, b2, relu=True), train=True, dtype='bf16')
    run.forward()
    from salt_amd._abi import lib
    import ctypes
    rows = []
    for i, (name, _, s) in enumerate(run.g.fwd.ops):
        for _ in range(3):
            run.g.fwd.run(begin=i, end=i + 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run.g.fwd.run(begin=i, end=i + 1)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        kid = lib.salt_conv_kernel_id(ctypes.byref(s)) if name == 'conv' else ''
        rows.append((name, kid, us))
        print('%-6s %-12s kernel %-3s %7.1f us' % ('fold' if fold else 'plain', name, kid, us))
    res[fold] = rows
# plain: ... conv1, affine_act1, conv2, affine_act2 ...; fold: ... conv1, conv2 (transform in the loader), affine_act2 ...
def pick(rows, name):
    return [r for r in rows if r[0] == name]
p, f = res[''], res['1']
plain = pick(p, 'affine_act')[0][2] + pick(p, 'conv')[1][2]
folded = pick(f, 'conv')[1][2]
print('B%d C%d %dx%d: affine_act + conv (kernel %s) = %.1f us   |   conv with the transform in its loader (kernel %s) = %.1f us   |   fold saves %.1f us'
      % (B, C, H, W, pick(p, 'conv')[1][1], plain, pick(f, 'conv')[1][1], folded, plain - folded))
