#!/usr/bin/env python
"""Per-operator timing table of one training step (HIP event pairs around every native op).
usage: python tools/op_profile.py [--dtype bf16] [--workload r34_hyper] [--batch 32] [--top 60]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import salt_amd
from salt_amd.models import SegmentationModel
from salt_amd._abi import lib
from bench import op_flops

ap = argparse.ArgumentParser()
ap.add_argument('--dtype', default='bf16'); ap.add_argument('--workload', default='r34_hyper'); ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--top', type=int, default=70); ap.add_argument('--size', type=int, default=128)
args = ap.parse_args()
arch_name = {'r34_hyper': 'UNetResNet', 'ternaus34': 'TernausUNetResNet', 'vanilla': 'VanillaUNet'}[args.workload]
ch = 1 if args.workload == 'vanilla' else 3
arch = {'model_params': {'architecture': arch_name, 'out_channels': 2, 'activation': 'sigmoid', 'compute_dtype': args.dtype},
        'optimizer_params': {'lr': 1e-4}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
m = SegmentationModel(arch, {'epochs': 1}, {}); m._to_device(); m.model.train()
B = args.batch
X = torch.randn(B, ch, args.size, args.size, device='cuda'); T = (torch.rand(B, 1, args.size, args.size, device='cuda') < 0.3).float(); T = torch.cat([1 - T, T], 1)
for _ in range(3):
    m._fit_loop([X, T])
eng = m.model.engine(); net = eng.net((B, ch, args.size, args.size), True)
rows = {}
reps = 5
for r in range(reps):
    eng.refresh(True)
    for pname, prog in (('fwd', net.fwd), ('loss', net.loss_program('lovasz', 1.0)), ('bwd', net.bwd)):
        for i, (name, s, ms) in enumerate(prog.run_timed()):
            key = (pname, i)
            if key not in rows:
                d = ''
                if name == 'conv':
                    d = 'x[%d,%d,%d,%d] -> [%d,%d,%d] taps%d s%d o%d k%d%s%s%s' % (s.x.B, s.x.H, s.x.W, s.x.C, s.OH, s.OW, s.y.C, s.ntaps, s.in_step, s.out_step,
                                                                              lib.salt_conv_kernel_id(ctypes.byref(s)), ' fold' if s.fold_top or s.fold_right else '',
                                                                              ' bnb' if s.bnb_acc else '', ' +=' if s.accumulate else '')
                elif name == 'conv_wgrad':
                    d = 'P[%d,%d,%d,%d] Q[..%d,%d,%d] taps%d s%d split%d' % (s.p.B, s.p.H, s.p.W, s.p.C, s.q.H, s.q.W, s.q.C, s.ntaps, s.q_step, s.nsplit)
                elif hasattr(s, 'y') and hasattr(s.y, 'C'):
                    d = '[%d,%d,%d,%d]' % (s.y.B, s.y.H, s.y.W, s.y.C)
                elif hasattr(s, 'x') and hasattr(s.x, 'C'):
                    d = '[%d,%d,%d,%d]' % (s.x.B, s.x.H, s.x.W, s.x.C)
                rows[key] = [name, d, op_flops(name, s), []]
            rows[key][3].append(ms)
tot = 0
tab = []
for key, (name, d, fl, mss) in rows.items():
    ms = sorted(mss)[len(mss) // 2]
    tot += ms
    tab.append((ms, key, name, d, fl))
tab.sort(reverse=True)
print('total %.3f ms over %d ops' % (tot, len(tab)))
for ms, key, name, d, fl in tab[:args.top]:
    print('%-4s %4d %-16s %-62s %8.1f us %8.1f TF/s' % (key[0], key[1], name, d, ms * 1e3, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0))
agg = {}
for ms, key, name, d, fl in tab:
    a = agg.setdefault(name, [0, 0, 0]); a[0] += ms; a[1] += fl; a[2] += 1
print('--- by operator')
for name, (ms, fl, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print('%-18s %4d ops %8.3f ms %8.1f TF/s' % (name, n, ms, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0))
