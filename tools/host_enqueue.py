"""Host enqueue time of one fused training step vs its GPU time (is the step launch-bound on this box?)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
dev = torch.device('cuda', 0)
import contextlib
ctx = torch.cuda.stream(torch.cuda.Stream(priority=int(os.environ.get('PRIO', '0')))) if os.environ.get('OWN_STREAM') else contextlib.nullcontext()
with ctx:
    model, batches, elapsed, _ = bench.train_config('r34_hyper', 'bf16', 32, 'lovasz', 30, 8, dev)
print('step %.3f ms' % (1e3 * elapsed / 30))
sys.exit(0)
torch.cuda.synchronize()
ts = []
for i in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model._fit_loop(list(batches[i % 8]))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
import numpy as np
a = np.array(ts) * 1e3
print('host enqueue %.3f ms (median), enqueue+drain %.3f ms; nproc %d, loadavg %s' % (np.median(a[:, 0]), np.median(a[:, 1]), os.cpu_count(), open('/proc/loadavg').read().strip()))
