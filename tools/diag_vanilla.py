import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
import torch
import closed_form as CF
import salt_amd
from salt_amd import architectures as A, losses
from oracle import nets as ON, specs as OS, losses as OL
DEV = 'cuda:0'
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 4
size = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
net = A.VanillaUNet(2, 1, 16, levels)
spec = OS.spec_vanilla_unet(levels=levels)
sd = CF.state_for((k, s) for k, (s, _) in spec.items())
if len(sys.argv) > 4 and sys.argv[4] == 'random':
    sd = OS.init_state(spec, seed=3)
    for k in sd:
        if k.endswith('running_var'): sd[k] = sd[k].clone()
net.load_state_dict(sd)
net.to(DEV).train()
x = CF.input_for('c1', (B, 1, size, size)); t = CF.mask_for('c1', (B, size, size))
for k in OS.trainable_keys(spec):
    sd[k].requires_grad_(True)
out_r = ON.vanilla_unet(sd, x, True, levels=levels)
loss_r = OL.mixed_dice_bce_loss(out_r, t); loss_r.backward()
out = net(x.to(DEV)); loss = losses.mixed_dice_bce_loss(out, t.to(DEV)); loss.backward()
print('loss', float(loss), float(loss_r), 'logit err', float((out.detach().cpu()-out_r.detach()).abs().max()/out_r.detach().abs().max()))
eng = net.engine()
sd64 = {k: (v.detach().double().requires_grad_(v.dtype.is_floating_point and not k.endswith(('running_mean','running_var'))) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
out64 = ON.vanilla_unet(sd64, x.double(), True, levels=levels)
l64 = OL.mixed_dice_bce_loss(out64, t.double()); l64.backward()
for k, p in net.named_parameters():
    gr = sd[k].grad
    off, n = eng.grad_range(p)
    mine = eng.grads[off:off+n].view(p.shape).cpu()
    e = float((mine-gr).abs().max()/(gr.abs().max()+1e-30))
    g64 = sd64[k].grad
    e_hip = float((mine.double()-g64).abs().max()/(g64.abs().max()+1e-30))
    e_t32 = float((gr.double()-g64).abs().max()/(g64.abs().max()+1e-30))
    print('%-28s |ref| %.3e  hip-vs-torch32 %.3e   hip-vs-f64 %.3e   torch32-vs-f64 %.3e' % (k, float(gr.abs().max()), e, e_hip, e_t32))
