#!/usr/bin/env python
"""Convergence-level parity of the headline dtype: the SAME initial weights (oracle.specs.init_state) and the SAME
minibatches go through
   (a) the CPU oracle (plain PyTorch fp32 restatement of the reference step, common_blocks/models.py:105-136),
   (b) the HIP path in fp32 (exact-f32 MFMA),
   (c) the HIP path in bf16 (bf16 storage, fp32 accumulation / master weights / statistics / loss) - the dtype bench.py's value is quoted on,
for --steps Lovasz + Adam(lr, L2 1e-4) steps of the ResNet34 hypercolumn U-Net at 128x128, batch --batch.  Per step: the loss of all
three.  At every --eval-every steps: mean IoU (metrics.py:21-34,53-59 conventions, crop 128 -> 101, logit[1] > 0) of all three on the
same 128 held-out synthetic tiles.  The numbers go to stdout as one JSON document (committed as profiles/rNN_convergence.json).

Test infrastructure: this script (like bench.py's cpu_baseline leg) may import oracle/; the product never does.
usage (GPU box): python tools/convergence_parity.py --steps 60 --batch 8 > gpurun_out/r03_convergence.json"""
import argparse
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch

import bench

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=60)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--eval-every', type=int, default=20)
ap.add_argument('--lr', type=float, default=1e-4)
ap.add_argument('--threads', type=int, default=32)
ap.add_argument('--val', type=int, default=128)
ap.add_argument('--skip-cpu', action='store_true')
args = ap.parse_args()

torch.set_num_threads(args.threads)
from oracle import nets as ON, specs as OS, losses as OL

B, K = args.batch, args.steps
img, msk = bench.synth_tiles(B * K, seed=4321)
X, T = bench.preprocess(img, msk, True, 3)
vi, vm = bench.synth_tiles(args.val, seed=999)
Xv, _ = bench.preprocess(vi, vm, False, 3)
gt = vm > 0.5


def val_iou(forward_eval):
    preds = []
    for i in range(0, args.val, 32):
        lg = forward_eval(Xv[i:i + 32])
        preds.append((lg[:, 1, 13:114, 14:115] > 0).numpy())
    return bench.iou_metric(np.concatenate(preds), gt)


spec = OS.SPECS['UNetResNet'](with_fc=True)
sd0 = OS.init_state(spec, seed=7)
out = {'config': {'model': 'architectures.unet.UNetResNet(34, hypercolumn)', 'image': [128, 128], 'batch': B, 'steps': K, 'lr': args.lr,
                  'loss': 'lovasz_hinge', 'optimizer': 'Adam + L2 1e-4', 'init': 'oracle.specs.init_state(seed=7)', 'train_tiles': B * K,
                  'val_tiles': args.val, 'data': 'synthetic (bench.synth_tiles seeds 4321 / 999)'},
       'loss': {}, 'val_iou': {}, 'seconds': {}}

# ------------------------------------------------------------------ (a) CPU oracle fp32
if not args.skip_cpu:
    t0 = time.perf_counter()
    sd = {k: v.detach().clone() for k, v in sd0.items()}
    dead = {'encoders.encoder.fc.weight', 'encoders.encoder.fc.bias'}
    keys = [k for k in OS.trainable_keys(spec) if k not in dead]
    for k in keys:
        sd[k].requires_grad_(True)
    params = [sd[k] for k in keys]
    m_ = [torch.zeros_like(p) for p in params]
    v_ = [torch.zeros_like(p) for p in params]
    losses, ious = [], {}
    for it in range(K):
        for p in params:
            p.grad = None
        o = ON.unet_resnet(sd, X[it * B:(it + 1) * B], True)
        l = OL.LOSSES['lovasz'](o, T[it * B:(it + 1) * B])
        l.backward()
        with torch.no_grad():
            OL.adam_l2_step([p.data for p in params], [p.grad for p in params], m_, v_, it + 1, lr=args.lr)
        losses.append(float(l))
        if (it + 1) % args.eval_every == 0:
            with torch.no_grad():
                ious[str(it + 1)] = round(val_iou(lambda x: ON.unet_resnet(sd, x, False)), 4)
        sys.stderr.write('cpu step %d loss %.5f\n' % (it + 1, losses[-1]))
    out['loss']['cpu_oracle_f32'] = [round(v, 6) for v in losses]
    out['val_iou']['cpu_oracle_f32'] = ious
    out['seconds']['cpu_oracle_f32'] = round(time.perf_counter() - t0, 1)
    out['config']['cpu_threads'] = torch.get_num_threads()

# ------------------------------------------------------------------ (b), (c) HIP fp32 / bf16
if torch.cuda.is_available():
    import salt_amd  # noqa: F401
    from salt_amd.models import SegmentationModel
    dev = torch.device('cuda', 0)
    for dtype in ('f32', 'bf16'):
        t0 = time.perf_counter()
        arch = {'model_params': {'architecture': 'UNetResNet', 'out_channels': 2, 'activation': 'sigmoid', 'loss': 'lovasz', 'compute_dtype': dtype},
                'optimizer_params': {'lr': args.lr}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
        m = SegmentationModel(arch, {'epochs': 1}, {})
        m.model.load_state_dict({k: sd0[k] for k in m.model.state_dict() if k in sd0}, strict=False)
        m._to_device()
        m.model.train()
        losses, ious = [], {}
        for it in range(K):
            r = m._fit_loop([X[it * B:(it + 1) * B], T[it * B:(it + 1) * B]])
            losses.append(float(r['sum']))
            if (it + 1) % args.eval_every == 0:
                m.model.eval()
                with torch.no_grad():
                    ious[str(it + 1)] = round(val_iou(lambda x: m.model(x.to(dev)).float().cpu()), 4)
                m.model.train()
        out['loss']['hip_' + dtype] = [round(v, 6) for v in losses]
        out['val_iou']['hip_' + dtype] = ious
        out['seconds']['hip_' + dtype] = round(time.perf_counter() - t0, 1)
        del m
        torch.cuda.empty_cache()

# ------------------------------------------------------------------ summary
L = out['loss']
summ = {}
if 'cpu_oracle_f32' in L:
    a = np.array(L['cpu_oracle_f32'])
    for tag in ('hip_f32', 'hip_bf16'):
        if tag in L:
            d = np.abs(np.array(L[tag]) - a)
            rel = d / np.maximum(np.abs(a), 1e-6)
            summ[tag + '_vs_cpu'] = {'max_abs_dloss': round(float(d.max()), 6), 'max_abs_dloss_first20': round(float(d[:20].max()), 6),
                                     'max_rel_dloss_first20': round(float(rel[:20].max()), 6), 'mean_abs_dloss': round(float(d.mean()), 6),
                                     'iou_delta_at_end': (round(out['val_iou'][tag][str(K)] - out['val_iou']['cpu_oracle_f32'][str(K)], 4)
                                                          if str(K) in out['val_iou'].get(tag, {}) and str(K) in out['val_iou']['cpu_oracle_f32'] else None)}
if 'hip_f32' in L and 'hip_bf16' in L:
    d = np.abs(np.array(L['hip_bf16']) - np.array(L['hip_f32']))
    summ['hip_bf16_vs_hip_f32'] = {'max_abs_dloss': round(float(d.max()), 6), 'mean_abs_dloss': round(float(d.mean()), 6)}
out['summary'] = summ
print(json.dumps(out, indent=1))
