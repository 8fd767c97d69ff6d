import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (R, R + '/tests', R + '/tests/golden'):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
from torch import nn
import salt_amd
from salt_amd import architectures as A
from gpu_harness import BlockRun
torch.manual_seed(0)

def report(tag, mine, ref):
    e = float((mine - ref).abs().max() / (ref.abs().max() + 1e-30))
    print('%-34s relerr %.3e' % (tag, e))

def run(tag, mod, inputs, emit, ref_fn):
    mod.train()
    ref_mod_sd = {k: v.clone() for k, v in mod.state_dict().items()}
    xs = [x.clone().requires_grad_(True) for x in inputs]
    import copy
    cpu = copy.deepcopy(mod)
    yr = ref_fn(cpu, *xs)
    gy = torch.randn(yr.shape)
    yr.backward(gy)
    r = BlockRun(mod, inputs, emit, train=True, dtype='f32')
    y = r.forward()
    report(tag + ' y', y, yr.detach())
    gx, grads = r.backward(gy.to('cuda:0'))
    for i, g in enumerate(gx):
        report(tag + ' gx%d' % i, g, xs[i].grad)
    ref_grads = dict(cpu.named_parameters())
    for k, g in grads.items():
        rg = ref_grads[k].grad
        if rg is not None and float(rg.abs().max()) > 1e-6:
            report(tag + ' g:' + k, g, rg)

# (A) standalone convT block at realistic size
class RefDeconv(nn.Module):
    pass
def ref_deconv(m, x):
    return F.relu(m.batch_norm(m.deconv(x)))
m = A.DeconvConv2dBnRelu(32, 16)
run('A deconv 16x16', m, [torch.randn(2, 32, 16, 16)], lambda g, a: m.emit(g, a), ref_deconv)

# (B) decoder stage: [up(x) | skip] -> ConvBnRelu, skip produced by a ConvBnRelu writing into the slice
class Stage(nn.Module):
    def __init__(s):
        super().__init__()
        s.up = A.DeconvConv2dBnRelu(32, 16); s.enc = A.ConvBnRelu(8, 16); s.dec = A.ConvBnRelu(32, 16)
def ref_stage(m, x, e):
    def cbr(b, t): return F.relu(b.conv[1](b.conv[0](t)))
    u = F.relu(m.up.batch_norm(m.up.deconv(x)))
    sk = cbr(m.enc, e)
    return cbr(m.dec, torch.cat([u, sk], 1)) 
st = Stage()
def emit_stage(g, x, e):
    cat = g.new_act(x.B, 2 * x.H, 2 * x.W, 32, 'cat')
    st.enc.emit(g, e, out=cat.slice(16, 16))
    st.up.emit(g, x, out=cat.slice(0, 16))
    return st.dec.emit(g, cat)
run('B stage', st, [torch.randn(2, 32, 16, 16), torch.randn(2, 8, 32, 32)], emit_stage, ref_stage)

# (C) same but pooled skip consumer too (accumulate into the slice gradient)
def ref_stage_c(m, x, e):
    def cbr(b, t): return F.relu(b.conv[1](b.conv[0](t)))
    u = F.relu(m.up.batch_norm(m.up.deconv(x)))
    sk = cbr(m.enc, e)
    y = cbr(m.dec, torch.cat([u, sk], 1))
    return y + F.interpolate(F.max_pool2d(sk, 2, 2), scale_factor=2, mode='bilinear', align_corners=False)
st2 = Stage()
def emit_stage_c(g, x, e):
    cat = g.new_act(x.B, 2 * x.H, 2 * x.W, 32, 'cat')
    sk = st2.enc.emit(g, e, out=cat.slice(16, 16))
    pooled = g.maxpool2(sk)
    st2.up.emit(g, x, out=cat.slice(0, 16))
    y = st2.dec.emit(g, cat)
    return g.add(y, g.upsample(pooled, 2))
run('C stage+pool', st2, [torch.randn(2, 32, 16, 16), torch.randn(2, 8, 32, 32)], emit_stage_c, ref_stage_c)
