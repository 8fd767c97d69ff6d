"""Per-tensor gradient error of the bf16 path vs the fp32 oracle (and of the f32 HIP path), forward parameter order.
usage: python tools/diag_bf16.py [B] [loss]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'golden')):
    sys.path.insert(0, p)
import torch
import closed_form as CF
import salt_amd
from salt_amd.models import SegmentationModel
from oracle import nets as ON, specs as OS, losses as OL
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
loss = sys.argv[2] if len(sys.argv) > 2 else 'bce_dice'
spec = OS.SPECS['UNetResNet'](with_fc=True)
sd0 = OS.init_state(spec, seed=7)
x = CF.input_for('c2', (B, 3, 128, 128)); t = CF.mask_for('c2', (B, 128, 128))
res = {}
for dtype in ('f32', 'bf16'):
    arch = {'model_params': {'architecture': 'UNetResNet', 'out_channels': 2, 'activation': 'sigmoid', 'loss': loss, 'compute_dtype': dtype},
            'optimizer_params': {'lr': 1e-4}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
    m = SegmentationModel(arch, {'epochs': 1}, {})
    m.model.load_state_dict({k: sd0[k] for k in m.model.state_dict() if k in sd0}, strict=False)
    m._to_device(); m.model.train()
    m._fit_loop([x, t]); torch.cuda.synchronize()
    eng = m.model.engine()
    res[dtype] = {k: eng.grads[eng.grad_range(p)[0]:eng.grad_range(p)[0] + eng.grad_range(p)[1]].view(p.shape).cpu().double()
                  for k, p in m.model.named_parameters() if id(p) in eng._off}
    res[dtype + '_logits'] = eng.net((B, 3, 128, 128), True).logits.cpu().double()
sd = {k: v.detach().clone() for k, v in sd0.items()}
dead = {'encoders.encoder.fc.weight', 'encoders.encoder.fc.bias'}
keys = [k for k in OS.trainable_keys(spec) if k not in dead]
for k in keys: sd[k].requires_grad_(True)
out = ON.unet_resnet(sd, x, True)
OL.LOSSES[loss](out, t).backward()
print('logits relL2: f32 %.3e  bf16 %.3e' % (float((res['f32_logits'] - out.detach().double()).norm() / out.detach().double().norm()),
                                            float((res['bf16_logits'] - out.detach().double()).norm() / out.detach().double().norm())))
for k in keys:
    g = sd[k].grad.double()
    if k not in res['bf16'] or float(g.norm()) == 0: continue
    e32 = float((res['f32'][k] - g).norm() / g.norm()); e16 = float((res['bf16'][k] - g).norm() / g.norm())
    cos = float((res['bf16'][k] * g).sum() / (res['bf16'][k].norm() * g.norm() + 1e-300))
    print('%-52s |g| %.3e  f32 relL2 %.2e   bf16 relL2 %.2e  cos %.4f' % (k, float(g.norm()), e32, e16, cos))
