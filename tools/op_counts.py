"""Operator histogram of the fused training step's programs (which launches the engine emits): python tools/op_counts.py [workload] [dtype]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else 'C2'
dtype = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
wl = {'C2': 'r34_hyper', 'C1': 'vanilla'}.get(workload, workload)
dev = torch.device('cuda:0')
model = bench.train_config(wl, dtype, 32, "lovasz", 2, 1, dev)[0]
eng = model.model.engine()
net = eng.net((32, bench.WORKLOADS[wl][1], 128, 128), True)
for pname in ('fwd', 'bwd'):
    prog = getattr(net, pname)
    c = collections.Counter(name for name, _, _ in prog.ops)
    print(pname, len(prog.ops), dict(c.most_common()))
    if pname == 'bwd':
        print('  bn_bwd partials_ready:', dict(collections.Counter(int(st.partials_ready) for name, _, st in prog.ops if name == 'bn_bwd')))
