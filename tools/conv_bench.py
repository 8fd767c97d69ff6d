#!/usr/bin/env python
"""Device-side micro-benchmark of dense conv launches: each (shape, cfg[, dbg]) is emitted REP times into one program, captured into
a hipGraph (no host launch cost between the kernels) and timed with an event pair around the replay.
usage: python tools/conv_bench.py "B,Cin,H,W,Cout[,k]:cfg[:dbg]" ...      (cfg 0 = heuristic; set SALT_CONV_V2=0 for the old kernel)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + '/tests'):
    sys.path.insert(0, p)
import torch
from torch import nn
import salt_amd
from salt_amd.engine import Graph
from salt_amd.runtime import Engine
REP = 20
st = torch.cuda.Stream()
for spec in sys.argv[1:]:
    parts = spec.split(':')
    dims = [int(v) for v in parts[0].split(',')]
    B, Cin, H, W, Cout = dims[:5]
    k = dims[5] if len(dims) > 5 else 3
    cfg = int(parts[1]) if len(parts) > 1 else 0
    dbg = parts[2] if len(parts) > 2 else None
    v2 = parts[3] if len(parts) > 3 else None
    if dbg is not None:
        os.environ['SALT_CONV_DBG'] = dbg
    conv = nn.Conv2d(Cin, Cout, k, 1, k // 2, bias=False)
    mod = nn.Sequential(conv).to('cuda:0')
    eng = Engine(mod, torch.device('cuda:0'), 'bf16')
    g = Graph(eng, False)
    x = g.new_act(B, H, W, Cin, 'x')
    x.buf.t.normal_()
    orig = g._conv_launch
    g._conv_launch = lambda *a, **kw: orig(*a, **{**kw, 'cfg': cfg})
    y = g.new_act(B, H, W, Cout, 'y')
    for _ in range(REP):
        g.conv(x, conv, None, relu=False, out=y)
    g.finalize(); eng.refresh(False)
    torch.cuda.synchronize()
    with torch.cuda.stream(st):
        g.fwd.run(stream=st)
        torch.cuda.synchronize()
        g.fwd.capture(st)
        g.fwd.replay(st)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); g.fwd.replay(st); e1.record(st)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / REP)
    fl = 2.0 * B * H * W * Cout * Cin * k * k
    print('conv B%d %dx%dx%d -> %d k%d cfg%d dbg%s: %7.2f us  %7.1f TF/s' % (B, H, W, Cin, Cout, k, cfg, dbg, best, fl / best / 1e6))
    os.environ.pop('SALT_CONV_DBG', None)
    g.fwd.release_graph()
