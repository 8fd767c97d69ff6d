#!/usr/bin/env python
"""Control for profiles/rNN_convergence.json: how far apart do TWO runs of the CPU oracle itself drift when only the fp32 summation
order changes (torch thread count 1 vs N: other partitioning of the convolution / BatchNorm reductions)?  Same init, same batches,
same step as tools/convergence_parity.py (a).  usage: python tools/convergence_control.py --steps 20 --threads 1 8"""
import argparse, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
import bench
from oracle import nets as ON, specs as OS, losses as OL

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=20); ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--lr', type=float, default=1e-4); ap.add_argument('--threads', type=int, nargs=2, default=[1, 8])
args = ap.parse_args()
B, K = args.batch, args.steps
img, msk = bench.synth_tiles(B * 60, seed=4321)             # the first K batches of the 60-step run
X, T = bench.preprocess(img, msk, True, 3)
spec = OS.SPECS['UNetResNet'](with_fc=True)
sd0 = OS.init_state(spec, seed=7)
curves = {}
for nt in args.threads:
    torch.set_num_threads(nt)
    sd = {k: v.detach().clone() for k, v in sd0.items()}
    keys = [k for k in OS.trainable_keys(spec) if k not in ('encoders.encoder.fc.weight', 'encoders.encoder.fc.bias')]
    for k in keys:
        sd[k].requires_grad_(True)
    params = [sd[k] for k in keys]
    m_ = [torch.zeros_like(p) for p in params]; v_ = [torch.zeros_like(p) for p in params]
    losses = []
    for it in range(K):
        for p in params:
            p.grad = None
        l = OL.LOSSES['lovasz'](ON.unet_resnet(sd, X[it * B:(it + 1) * B], True), T[it * B:(it + 1) * B])
        l.backward()
        with torch.no_grad():
            OL.adam_l2_step([p.data for p in params], [p.grad for p in params], m_, v_, it + 1, lr=args.lr)
        losses.append(float(l))
        sys.stderr.write('threads %d step %d loss %.6f\n' % (nt, it + 1, losses[-1]))
    curves[str(nt)] = losses
a, b = (np.array(curves[str(t)]) for t in args.threads)
rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-6)
print(json.dumps({'what': 'CPU oracle vs CPU oracle, torch threads %d vs %d, R34 hypercolumn B=%d 128x128 Lovasz + Adam lr %g' % (args.threads[0], args.threads[1], B, args.lr),
                  'loss': curves, 'rel_dloss': [round(float(x), 6) for x in rel], 'max_rel_dloss': round(float(rel.max()), 6),
                  'first_step_above_1e-3': int(np.argmax(rel > 1e-3)) + 1 if (rel > 1e-3).any() else None}, indent=1))
