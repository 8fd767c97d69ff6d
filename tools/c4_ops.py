#!/usr/bin/env python
"""Per-operator timing of ONE eval forward of BASELINE C4's network (ResNet152 hypercolumn U-Net, [64,3,256,256] = 16 images x 4 TTA
variants, bf16): HIP event pairs around every native op, with the kernel id salt_conv picks for each convolution.
usage: python tools/c4_ops.py [--top 40] [--depth 152] [--batch 64] [--size 256]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import salt_amd  # noqa: F401
from salt_amd import architectures as A
from salt_amd._abi import lib
from bench import op_flops

ap = argparse.ArgumentParser()
ap.add_argument('--top', type=int, default=40); ap.add_argument('--depth', type=int, default=152)
ap.add_argument('--batch', type=int, default=64); ap.add_argument('--size', type=int, default=256)
args = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = A.UNetResNet(args.depth, 2, use_hypercolumn=True, dropout_2d=0.0, pretrained=False)
net.set_compute_dtype('bf16')
net.to(dev).eval()
X = torch.randn(args.batch, 3, args.size, args.size, device=dev)
with torch.no_grad():
    for _ in range(2):
        net(X)
eng = net.engine()
cn = eng.net((args.batch, 3, args.size, args.size), False)
rows = {}
for r in range(3):
    for i, (name, s, ms) in enumerate(cn.fwd.run_timed()):
        if i not in rows:
            d = ''
            if name == 'conv':
                d = 'x[%d,%d,%d,%d|cs%d] -> [%d,%d,%d|cs%d] taps%d s%d o%d kernel %d%s' % (
                    s.x.B, s.x.H, s.x.W, s.x.C, s.x.cs, s.OH, s.OW, s.y.C, s.y.cs, s.ntaps, s.in_step, s.out_step, lib.salt_conv_kernel_id(ctypes.byref(s)),
                    ' +res' if s.res.p else '')
            elif hasattr(s, 'y') and hasattr(s.y, 'C'):
                d = '[%d,%d,%d,%d]' % (s.y.B, s.y.H, s.y.W, s.y.C)
            rows[i] = [name, d, op_flops(name, s), []]
        rows[i][3].append(ms)
tab = sorted(((sorted(v[3])[1], i, v[0], v[1], v[2]) for i, v in rows.items()), reverse=True)
tot = sum(t[0] for t in tab)
print('total %.3f ms over %d ops' % (tot, len(tab)))
for ms, i, name, d, fl in tab[:args.top]:
    print('%4d %-12s %-78s %9.1f us %8.1f TF/s' % (i, name, d, ms * 1e3, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0))
agg = {}
for ms, i, name, d, fl in tab:
    key = name + ((' k%s' % d.split('kernel ')[1].split()[0]) if 'kernel' in d else '')
    a = agg.setdefault(key, [0, 0, 0]); a[0] += ms; a[1] += fl; a[2] += 1
print('--- by operator / kernel id')
for name, (ms, fl, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print('%-18s %4d ops %8.3f ms %8.1f TF/s' % (name, n, ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0))
