import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import torch, numpy as np
import closed_form as CF
import salt_amd
from salt_amd import architectures as A, losses
from oracle import nets as ON, specs as OS, losses as OL
from helpers import golden, T
from test_gpu_models import _fill_closed_form
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
torch.manual_seed(5)
net = A.UNetResNet(34, 2, use_hypercolumn=True, pool0=True)
spec = OS.SPECS['UNetResNet'](with_fc=True)
sd = OS.init_state(spec, seed=7)
net.load_state_dict({k: sd[k] for k in net.state_dict() if k in sd}, strict=False)
sd = {k: v.detach().clone() for k, v in net.state_dict().items() if k in spec}
x = CF.input_for('r34', (4, 3, size, size)); t = CF.mask_for('r34', (4, size // 2, size // 2))
net.to('cuda:0').train()
dead = set(net.dead_parameter_names())
keys = [k for k in OS.trainable_keys(spec) if k not in dead]
for k in keys: sd[k].requires_grad_(True)
out_r = ON.unet_resnet(sd, x, True, pool0=True)
OL.mixed_dice_bce_loss(out_r, t).backward()
out = net(x.to('cuda:0')); losses.mixed_dice_bce_loss(out, t.to('cuda:0')).backward()
print('logits', float((out.detach().cpu() - out_r.detach()).abs().max() / out_r.detach().abs().max()))
eng = net.engine()
rows = []
for k, p in net.named_parameters():
    if k in dead or sd[k].grad is None: continue
    off, n = eng.grad_range(p)
    mine = eng.grads[off:off + n].view(p.shape).cpu().double(); g = sd[k].grad.double()
    if float(g.norm()) < 1e-7 or k.endswith("conv.bias"): continue
    rows.append((float((mine - g).norm() / g.norm()), k))
rows.sort(reverse=True)
print(rows[:8])
