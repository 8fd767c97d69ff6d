#!/usr/bin/env python
"""Isolated timing of the weight-gradient launches of the C2 step through the C-ABI (salt_conv_wgrad [+ salt_wgrad_reduce]),
`reps` back-to-back launches between two events.  A/B by environment: SALT_WGRAD_LS=0 (previous kernels), SALT_WL_KU=4|8,
SALT_WGRAD_TPW=N, SALT_LIB=<variant .so>.
usage: python tools/wgrad_ls_bench.py [reps]"""
import ctypes
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (R, R + '/tests'):
    sys.path.insert(0, p_)
import torch
import salt_amd  # noqa: F401
from salt_amd._abi import STRUCTS, lib, fill, check
from salt_amd.engine import shaped_view

SHAPES = [
    # B, H, W, Ca, Cb, replicate        (P = dL/dy with Ca channels, Q = x with Cb channels)
    (32, 64, 64, 64, 64, 0),      # ResNet34 layer1 (6 launches)
    (32, 32, 32, 128, 128, 0),    # layer2 (7)
    (32, 16, 16, 256, 256, 0),    # layer3 (11)
    (32, 8, 8, 512, 512, 0),      # layer4 (5) + center
    (32, 8, 8, 512, 768, 1),      # dec5 conv1
    (32, 16, 16, 256, 320, 1),    # dec4 conv1
    (32, 32, 32, 128, 192, 1),    # dec3 conv1
    (32, 64, 64, 64, 128, 1),     # dec2 conv1
    (32, 128, 128, 32, 64, 1),    # dec1 conv1
    (32, 128, 128, 64, 32, 1),    # dec1 conv2
    (32, 128, 128, 64, 320, 1),   # final conv over the hypercolumn
    (32, 32, 32, 128, 64, 2),     # layer2.0.conv1, stride 2 (rep = 2: Q is twice P's size)
    (32, 16, 16, 256, 128, 2),    # layer3.0.conv1
    (32, 8, 8, 512, 256, 2),      # layer4.0.conv1
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    st = torch.cuda.current_stream().cuda_stream
    tot = [0.0, 0.0]
    for (B, H, W, Ca, Cb, rep) in SHAPES:
        s2 = rep == 2
        P = torch.randn(B, H, W, Ca, device='cuda:0').bfloat16()
        Q = torch.randn(B, H * (2 if s2 else 1), W * (2 if s2 else 1), Cb, device='cuda:0').bfloat16()
        taps = [(dy - 2, dx) for dy in range(3) for dx in range(3)] if rep == 1 else [(dy - 1, dx - 1) for dy in range(3) for dx in range(3)]
        S = fill(STRUCTS['salt_conv_wgrad_args'](), dtype=1, p=shaped_view(P.data_ptr(), B, H, W, Ca),
                 q=shaped_view(Q.data_ptr(), B, Q.shape[1], Q.shape[2], Cb),
                 ntaps=9, tap_dy=[t[0] for t in taps], tap_dx=[t[1] for t in taps], q_step=2 if s2 else 1, pad_mode=1 if rep == 1 else 0, q_plane=0)
        ns = lib.salt_conv_wgrad_nsplit(ctypes.byref(S))
        part = torch.empty(ns, 9, Ca, Cb, device='cuda:0')
        grad = torch.empty(Ca, Cb, 3, 3, device='cuda:0')
        S.partials = part.data_ptr(); S.nsplit = ns
        Rr = fill(STRUCTS['salt_wgrad_reduce_args'](), partials=part.data_ptr(), nsplit=ns, ntaps=9, Ca=Ca, Cb=Cb, KH=3, KW=3,
                  tap_kh=[t // 3 for t in range(9)], tap_kw=[t % 3 for t in range(9)], grad=grad.data_ptr(), accumulate=0)
        res = []
        for which in (0, 1):
            def go():
                if which == 0:
                    check(lib.salt_conv_wgrad(ctypes.byref(S), st))
                else:
                    check(lib.salt_wgrad_reduce(ctypes.byref(Rr), st))
            for _ in range(3):
                go()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                go()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / reps * 1e3)
            tot[which] += res[-1]
        fl = 2.0 * B * H * W * Ca * Cb * 9
        print('P[%d,%d,%d,%d] Q[..%d] rep%d split%-4d wgrad %7.1f us %7.1f TF/s   reduce %6.1f us' % (B, H, W, Ca, Cb, rep, ns, res[0], fl / res[0] / 1e6, res[1]))
    print('sum wgrad %.1f us, reduce %.1f us' % (tot[0], tot[1]))


if __name__ == '__main__':
    main()
