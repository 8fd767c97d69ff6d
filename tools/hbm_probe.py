#!/usr/bin/env python
"""HBM write / copy / read rates of this MI355X through torch's own kernels (fill, copy, sum) on 2 GB tensors: the yardstick for the
write-dominated launches of C4 (1x1 expansion convolutions: 80 % of their traffic is stores; up-sampling: all of it)."""
import torch
dev = torch.device('cuda:0')
n = 1 << 30                      # 2 GB of bf16
a = torch.empty(n, dtype=torch.bfloat16, device=dev)
b = torch.empty(n, dtype=torch.bfloat16, device=dev)
def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
gb = n * 2 / 1e9
t = timed(lambda: a.fill_(1.0)); print('fill  (write only)      %.2f TB/s' % (gb / t / 1e3))
t = timed(lambda: a.zero_()); print('zero  (memset)          %.2f TB/s' % (gb / t / 1e3))
t = timed(lambda: b.copy_(a)); print('copy  (read + write)    %.2f TB/s total' % (2 * gb / t / 1e3))
t = timed(lambda: a.view(torch.int16).sum()); print('sum   (read only)       %.2f TB/s' % (gb / t / 1e3))
t = timed(lambda: torch.add(a, b, out=b)); print('add   (2 reads + write) %.2f TB/s total' % (3 * gb / t / 1e3))

# ---- the library's own streaming kernels on the same 2 GB: salt_add as a copy (NHWC view [64,256,256,256] bf16, 16 bytes per lane,
# grid-stride over <= SALT_EW_BLOCKS workgroups) and salt_affine_act (y -> a with scale / shift)
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import salt_amd  # noqa: F401
from salt_amd._abi import STRUCTS, OP_FUNCS, fill, check, CONSTS
def view(t, B, H, W, C):
    v = STRUCTS['salt_view'](); v.p = t.data_ptr(); v.B, v.H, v.W, v.C, v.cs = B, H, W, C, C
    return v
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, H, W, C = 64, 256, 256, 256
fn, S = OP_FUNCS['salt_add']
s = fill(S(), dtype=CONSTS['SALT_BF16'], a=view(a, B, H, W, C), b=STRUCTS['salt_view'](), y=view(b, B, H, W, C), accumulate=0)
t = timed(lambda: check(fn(ctypes.byref(s), st), 'add')); print('salt_add copy           %.2f TB/s total (EW_BLOCKS=%s)' % (2 * gb / t / 1e3, os.environ.get('SALT_EW_BLOCKS', '1024')))
sc = torch.ones(C, device=dev); sh = torch.zeros(C, device=dev)
fn2, S2 = OP_FUNCS['salt_affine_act']
s2 = fill(S2(), dtype=CONSTS['SALT_BF16'], y=view(a, B, H, W, C), scale=sc.data_ptr(), shift=sh.data_ptr(), res=STRUCTS['salt_view'](), relu=1, a=view(b, B, H, W, C))
t = timed(lambda: check(fn2(ctypes.byref(s2), st), 'affine')); print('salt_affine_act         %.2f TB/s total' % (2 * gb / t / 1e3))
fn3, S3 = OP_FUNCS['salt_zero']
s3 = fill(S3(), p=a.data_ptr(), bytes=n * 2)
t = timed(lambda: check(fn3(ctypes.byref(s3), st), 'zero')); print('salt_zero               %.2f TB/s' % (gb / t / 1e3))
