#!/usr/bin/env python
"""In-kernel clock breakdown of conv_wgrad_ls_kernel (library built with -DSALT_WL_CLK=1:
SRC=conv_wgrad_ls tools/build_variant.sh wlclk -DSALT_WL_CLK=1, run with SALT_LIB=.../libsaltnet_hip.wlclk.so).
Per workgroup: MFMA wave 0 stamps entry / first entry landed / loop end / end and sums its barrier waits; loader wave 4 stamps entry /
ring primed / end and sums its vmcnt waits, barrier waits and issue time.  Prints medians / maxima over workgroups (shader cycles).
usage: SALT_LIB=... python tools/wl_clocks.py B,H,W,Ca,Cb[,replicate]"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (R, R + '/tests'):
    sys.path.insert(0, p_)
import numpy as np
import torch
import salt_amd  # noqa: F401
from salt_amd._abi import STRUCTS, lib, fill, check
from salt_amd.engine import shaped_view

v = [int(x) for x in sys.argv[1].split(',')]
B, H, W, Ca, Cb = v[:5]
rep = v[5] if len(v) > 5 else 0
P = torch.randn(B, H, W, Ca, device='cuda:0').bfloat16()
Q = torch.randn(B, H, W, Cb, device='cuda:0').bfloat16()
taps = [(dy - 2, dx) for dy in range(3) for dx in range(3)] if rep else [(dy - 1, dx - 1) for dy in range(3) for dx in range(3)]
S = fill(STRUCTS['salt_conv_wgrad_args'](), dtype=1, p=shaped_view(P.data_ptr(), B, H, W, Ca), q=shaped_view(Q.data_ptr(), B, H, W, Cb),
         ntaps=9, tap_dy=[t[0] for t in taps], tap_dx=[t[1] for t in taps], q_step=1, pad_mode=rep, q_plane=0)
ns = lib.salt_conv_wgrad_nsplit(ctypes.byref(S))
part = torch.empty(ns, 9, Ca, Cb, device='cuda:0')
S.partials = part.data_ptr(); S.nsplit = ns
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    check(lib.salt_conv_wgrad(ctypes.byref(S), st))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); check(lib.salt_conv_wgrad(ctypes.byref(S), st)); e1.record()
torch.cuda.synchronize()
print('P[%d,%d,%d,%d] Q[..%d] rep%d nsplit %d: launch %.1f us' % (B, H, W, Ca, Cb, rep, ns, e0.elapsed_time(e1) * 1e3))
N = 1024 * 16
buf = (ctypes.c_ulonglong * N)()
lib.salt_debug_wl_clk.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.salt_debug_wl_clk(buf, N) == 0, 'library was not built with -DSALT_WL_CLK=1'
a = np.array(buf[:], dtype=np.uint64).reshape(1024, 16).astype(np.float64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
def row(name, x):
    print('  %-46s median %8.0f  min %8.0f  max %8.0f' % (name, np.median(x), x.min(), x.max()))
print('%d workgroups, %d..%d entries each' % (a.shape[0], a[:, 5].min(), a[:, 5].max()))
row('MFMA wave: entry after the first workgroup', a[:, 0] - t0)
row('MFMA wave: entry -> first entry landed', a[:, 1] - a[:, 0])
row('MFMA wave: first landed -> loop end', a[:, 3] - a[:, 1])
row('MFMA wave:   of which at barriers (after the first)', a[:, 2] - (a[:, 1] - a[:, 0]))
row('MFMA wave: slab stores (incl. vmcnt(0))', a[:, 4] - a[:, 3])
row('MFMA wave: entry -> end', a[:, 4] - a[:, 0])
row('loader: prime the ring (NS - 2 entries issued)', a[:, 9] - a[:, 8])
row('loader: sum of vmcnt waits', a[:, 10])
row('loader: sum of barrier waits', a[:, 11])
row('loader: sum of issue time', a[:, 12])
row('loader: entry -> end', a[:, 13] - a[:, 8])
print('  last workgroup ends %.0f cycles after the first one starts' % (a[:, 4].max() - t0))
