#!/usr/bin/env python
"""In-kernel clock breakdown of conv_ws_kernel (a library built with -DSALT_WS_CLK=1: tools/build_variant.sh wsclk -DSALT_WS_CLK=1, run with
SALT_LIB=.../libsaltnet_hip.wsclk.so).  Every workgroup stamps s_memtime at: kernel entry, after each phase barrier, after each phase's
work (MFMA tile or epilogue), kernel end - once per wave group.  Prints, per stamp interval, the median / max over workgroups.
usage: SALT_LIB=... python tools/ws_clocks.py B,Cin,H,W,Cout [train]"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + '/tests'):
    sys.path.insert(0, p)
import numpy as np
import torch
from torch import nn
import salt_amd
from salt_amd._abi import lib
from salt_amd.engine import Graph
from salt_amd.runtime import Engine

B, Cin, H, W, Cout = [int(v) for v in sys.argv[1].split(',')]
mode = sys.argv[2] if len(sys.argv) > 2 else 'eval'
train = mode in ('train', 'dgrad_rep')
conv = nn.Conv2d(Cin, Cout, 3, 1, 0 if mode == 'dgrad_rep' else 1, bias=False)
bn = nn.BatchNorm2d(Cout)
mod = nn.Sequential(conv, bn).to('cuda:0')
eng = Engine(mod, torch.device('cuda:0'), 'bf16')
g = Graph(eng, train)
x = g.new_act(B, H, W, Cin, 'x')
x.buf.t.normal_()
y = g.conv(x, conv, bn if train else None, relu=False, replicate=(mode == 'dgrad_rep'))
if train:
    g.build_backward()
g.finalize(); eng.refresh(train)
for _ in range(3):
    g.fwd.run(side=eng.side_stream)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.fwd.run(side=eng.side_stream); e1.record()
torch.cuda.synchronize()
print('program wall %.1f us (%d ops)' % (e0.elapsed_time(e1) * 1e3, len(g.fwd.ops)))
if mode == 'dgrad_rep':                     # the data gradient of the replicate-padded layer (fused fold) is the last conv_ws launch
    y.buf.grad().normal_()
    for name, st, ms in g.bwd.run_timed():
        print('  bwd %-14s %8.1f us' % (name, ms * 1e3))
    torch.cuda.synchronize()
NS = 24
buf = (ctypes.c_ulonglong * (256 * 2 * NS))()
lib.salt_debug_ws_clk.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = lib.salt_debug_ws_clk(buf, 256 * 2 * NS)
assert rc == 0, 'library was not built with -DSALT_WS_CLK=1'
a = np.array(buf[:], dtype=np.uint64).reshape(256, 2, NS).astype(np.float64)
for grp in (0, 1):
    st = a[:, grp, :]
    valid = st[:, 0] > 0
    st = st[valid]
    n = int((st[0, :22] > 0).sum())
    tot = st[:, 23] - st[:, 0]
    print('  kernel entry -> end: median %.0f max %.0f cycles; tiles per workgroup %d..%d' % (np.median(tot), tot.max(), st[:, 22].min(), st[:, 22].max()))
    print('group %d: %d workgroups, %d stamps' % (grp, st.shape[0], n))
    t0 = st[:, 0].min()
    print('  stamps in program order: entry, then per phase k: after the barrier, [epilogue role: before its halo DMA issue], after the role\'s work; end')
    for i in range(n):
        rel = st[:, i] - st[:, 0]
        d = (st[:, i] - st[:, i - 1]) if i else rel
        print('  stamp %2d at median %8.0f cycles after entry (max %8.0f);  delta median %7.0f  max %7.0f' % (i, np.median(rel), rel.max(), np.median(d), d.max()))
