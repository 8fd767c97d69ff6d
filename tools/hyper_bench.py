#!/usr/bin/env python
"""salt_hyper_stencil alone: forward / adjoint launch time per level set (C2 geometry [32,128,128,64] by default; --c4: [64,256,256,256]).
--eval: the eval epilogue (scale / shift / ReLU), then the same with the fused logit head and y dropped.
usage: python tools/hyper_bench.py [--c4] [--dtype bf16|f32] [--reps 20] [--eval]"""
import argparse, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import salt_amd
from salt_amd import _abi as abi
from salt_amd.engine import shaped_view

ap = argparse.ArgumentParser()
ap.add_argument('--c4', action='store_true'); ap.add_argument('--dtype', default='bf16'); ap.add_argument('--reps', type=int, default=20); ap.add_argument('--eval', action='store_true')
args = ap.parse_args()
B, H, W, C = (64, 256, 256, 256) if args.c4 else (32, 128, 128, 64)
tdt = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
dt = 1 if args.dtype == 'bf16' else 0
es = 2 if dt else 4
dev = 'cuda:0'


def view(t):
    b, h, w, c = t.shape
    return shaped_view(t.data_ptr(), b, h, w, c, c)


def run(Rs, backward, mode=''):
    zs = [torch.randn(B, H // R, W // R, 9 * C, device=dev).to(tdt) for R in Rs]
    y = torch.randn(B, H, W, C, device=dev).to(tdt)
    yo = torch.empty_like(y)
    S = abi.STRUCTS['salt_hyper_stencil_args']()
    if backward:
        abi.fill(S, dtype=dt, nlev=len(Rs), z=[view(z) for z in zs], R=list(Rs), y=view(y), backward=1)
    else:
        abi.fill(S, dtype=dt, nlev=len(Rs), z=[view(z) for z in zs], R=list(Rs), y_in=view(y), y=view(yo), backward=0)
        if mode:
            sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            abi.fill(S, scale=sc.data_ptr(), shift=sh.data_ptr(), relu=1)
        if mode == 'head':
            hw, hb, lg = torch.randn(2, C, device=dev), torch.zeros(2, device=dev), torch.zeros(B, 2, H, W, device=dev)
            ws = torch.zeros(B * (C // 64) * 2 * H * W, device=dev)
            yv = view(yo); yv.p = None
            abi.fill(S, y=yv, head_w=hw.data_ptr(), head_b=hb.data_ptr(), head_y_nchw=lg.data_ptr(), head_cout=2, head_ws=ws.data_ptr() if C > 64 else None)
    for _ in range(3):
        abi.check(abi.lib.salt_hyper_stencil(ctypes.byref(S), None), 'stencil')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        abi.lib.salt_hyper_stencil(ctypes.byref(S), None)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / args.reps
    zb = sum(z.numel() for z in zs) * es
    yb = y.numel() * es
    by = zb + yb * (1 if backward else 2)
    print('%-8s %-5s levels %-12s %8.1f us   %6.2f TB/s algorithmic (%d MB)' % ('adjoint' if backward else 'forward', mode, str(Rs), us, by / us / 1e6, by >> 20))


if args.eval:
    for mode in ('eval', 'head', 'eval', 'head'):
        run((4, 8, 16), False, mode)
    sys.exit(0)
for bw in (False, True):
    if bw and W > 256:
        continue
    for Rs in ((4, 8, 16), (4,), (8,), (16,)):
        run(Rs, bw)
