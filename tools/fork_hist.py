"""Which main-stream operators precede a side-stream fork / consume a side-stream join in the backward program (GPU box)."""
import collections
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import salt_amd  # noqa
from salt_amd.models import SegmentationModel

arch = {'model_params': {'architecture': 'UNetResNet', 'out_channels': 2, 'activation': 'sigmoid', 'loss': 'lovasz', 'compute_dtype': 'bf16'},
        'optimizer_params': {'lr': 1e-4}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
model = SegmentationModel(arch, {'epochs': 1}, {})
model._to_device()
model.model.train()
eng = model.model.engine()
net = eng.net((32, 3, 128, 128), True)
for name in ('fwd', 'bwd'):
    prog = getattr(net, name, None)
    if prog is None:
        continue
    fork, join, runs = collections.Counter(), collections.Counter(), collections.Counter()
    st, ops = prog.streams, prog.ops
    dirty = True
    for i in range(len(ops)):
        if st[i] == 1:
            if dirty:
                fork[ops[i - 1][0] if i else '-'] += 1
                dirty = False
        else:
            dirty = True
            if st[i] in (2, 3):
                join[(ops[i][0], st[i])] += 1
    print(name, len(ops), 'forks after:', dict(fork), 'joins at:', dict(join))
