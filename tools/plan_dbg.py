import os, sys, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['SALT_FORCE_DP_PATH'] = '1'
import bench
from salt_amd.parallel import plan_buckets
dev = torch.device('cuda:0')
model = bench.train_config('r34_hyper', 'bf16', 32, 'lovasz', 2, 1, dev)[0]
eng = model.model.engine()
net = eng.net((32, 3, 128, 128), True)
plan = model.dp._plans[id(net)] if id(net) in model.dp._plans else plan_buckets(net.g.grad_ready, eng.n_live, model.dp.bucket_bytes)
print('n_ops', len(net.bwd), 'plan', [(lo, hi, r) for lo, hi, r in plan])
pos = 0
for lo, hi, r in plan:
    side = sum(1 for s in net.bwd.streams[pos:r] if s == 1)
    print('segment ops %d..%d: %d side entries, %.1f MB' % (pos, r, side, (hi - lo) * 4 / 1e6)); pos = r
