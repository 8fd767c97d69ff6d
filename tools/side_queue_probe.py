#!/usr/bin/env python
"""How much of the step does the weight-gradient queue decide?  Timing probe (WRONG gradients in the stripped modes - a tool, not a
switch of the product): the headline step with (a) everything, (b) the slab reductions removed from the backward program, (c) the
weight-gradient launches AND reductions removed, (d) scse_fc_grads removed.  If (c) is much faster than (a) the side queue is (close to)
critical and work on conv_wgrad / wgrad_reduce pays; if not, only the main queue matters.
usage (GPU box): python tools/side_queue_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
import salt_amd  # noqa: F401


def run(drop):
    model, batches, _, _ = bench.train_config('r34_hyper', 'bf16', 32, 'lovasz', 4, 6, dev)
    eng = model.model.engine()
    net = eng.net((32, 3, 128, 128), True)
    if drop:
        keep = [i for i, (n, _, _) in enumerate(net.bwd.ops) if n not in drop]
        net.bwd.ops = [net.bwd.ops[i] for i in keep]
        net.bwd.streams = [net.bwd.streams[i] for i in keep]
        net.bwd._entries = None
        net.bwd.finalize()
    for i in range(6):
        model._fit_loop(list(batches[i % 8]))
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(40):
            model._fit_loop(list(batches[i % 8]))
        torch.cuda.synchronize()
        res.append(round((time.perf_counter() - t0) / 40 * 1e3, 3))
    n = len(net.bwd.ops)
    del model, batches
    torch.cuda.empty_cache()
    return res, n


for tag, drop in (('all', ()), ('no_wgrad_reduce', ('wgrad_reduce',)), ('no_conv_wgrad_no_reduce', ('wgrad_reduce', 'conv_wgrad')), ('no_scse_fc_grads', ('scse_fc_grads',)),
                  ('all_again', ())):
    r, n = run(drop)
    print('%-28s bwd ops %3d   ms/step %s' % (tag, n, r), flush=True)
