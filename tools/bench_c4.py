#!/usr/bin/env python
"""BASELINE C4: inference images/s of the ResNet152 hypercolumn U-Net at 256x256, batch 16 per GPU, 4-flip TTA
(flip -> forward of the 4 variants as one batch of 64 -> sigmoid -> inverse flip -> mean -> centre crop -> threshold), bf16.
usage: python tools/bench_c4.py [--steps K] [--warmup W] [--dtype bf16|f32] [--batch 16] [--depth 152]"""
import argparse
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import salt_amd
from salt_amd import architectures as A, inference as I

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--warmup', type=int, default=3)
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--batch', type=int, default=16)
ap.add_argument('--depth', type=int, default=152)
ap.add_argument('--size', type=int, default=256)
args = ap.parse_args()
dev = torch.device('cuda:0')
torch.manual_seed(0)
net = A.UNetResNet(args.depth, 2, use_hypercolumn=True, dropout_2d=0.0, pretrained=False)
net.set_compute_dtype(args.dtype)
net.to(dev).eval()
X = torch.randn(args.batch, 3, args.size, args.size, device=dev)
crop = (202, 202) if args.size == 256 else (101, 101)


def step():
    prob = I.predict_tta(net, X, True, True, depth_channels=False)
    return I.crop_threshold(prob, crop, 0.5, cls=1)


for _ in range(args.warmup):
    m = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    m = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
macs = {152: 384.7e9, 34: None}.get(args.depth) if args.size == 256 else None
out = {'metric': 'inference images/s incl. 4-flip TTA, U-Net ResNet%d %dx%d bs%d/GPU' % (args.depth, args.size, args.size, args.batch),
       'value': round(args.batch / dt, 2), 'unit': 'images/s', 'ms_per_batch': round(dt * 1e3, 2), 'dtype': args.dtype,
       'forward_images_per_s': round(4 * args.batch / dt, 1), 'mask_sum': int(m.sum())}
if macs:
    out['tflops'] = round(2 * macs * 4 * args.batch / dt / 1e12, 1)
print(json.dumps(out))
