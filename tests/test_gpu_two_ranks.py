"""world = 2 on DEVICE tensors (SURVEY 8(e) was the only row whose GPU code path had never seen a second rank).

The gpurun box exposes ONE MI355X and RCCL refuses two ranks on one device, so both processes sit on cuda:0 and the process group is
gloo: `DataParallel._all_reduce` stages each gradient bucket through the host ON THE COMMUNICATION STREAM (the D2H copy waits for the
bucket's two events exactly like a collective kernel would) and copies the sum back.  Everything else is the shipped multi-GPU path:
`SegmentationModel._fit_loop` -> fused step -> `DataParallel.backward` (ONE executor call with bucket marks, comm stream, per-bucket
events on both compute queues), 1 / world folded into Adam's gradient scale, rank-0 `persist`, `any_rank` early stopping.

Replaces the reference's `nn.DataParallel` (common_blocks/models.py:81-85) and its checkpoint round trip (models.py:196-208)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
mode, out = sys.argv[1], sys.argv[2]
world = int(os.environ.get('WORLD_SIZE', '1'))
rank = int(os.environ.get('RANK', '0'))
torch.cuda.set_device(0)                                  # BOTH ranks on the one GPU of the box
if world > 1:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo')
from salt_amd import models
arch = {'model_params': {'architecture': 'UNetResNet', 'out_channels': 2, 'activation': 'sigmoid', 'loss': 'lovasz', 'compute_dtype': 'bf16'},
        'optimizer_params': {'lr': 1e-3}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
torch.manual_seed(3 + 17 * rank)                          # DIFFERENT initial weights per rank: broadcast_parameters must make them rank 0's
m = models.SegmentationModel(arch, {'epochs': 1}, {})
assert m.dp.world == world and m.dp.rank == rank
m.dp.bucket_bytes = 8 << 20
m._to_device(); m.model.train()
m.dp.broadcast_parameters(m.model)
eng = m.model.engine()
g = torch.Generator().manual_seed(5)
X = torch.randn(8, 3, 128, 128, generator=g)
M = (torch.rand(8, 1, 128, 128, generator=g) > 0.6).float()
T = torch.cat([1 - M, M], 1)
res = {'world': world}
if mode == 'same':
    # every rank trains on the SAME 4 images: the summed gradient is exactly 2 x, the average exactly the 1-rank gradient
    Xs, Ts = X[:4].cuda(), T[:4].cuda()
    res['losses'] = [float(m._fit_loop([Xs, Ts])['sum']) for _ in range(3)]
elif world > 1:
    # rank r trains on ITS half of the batch of 8 (shard_batch, the reference's scatter): per-rank BatchNorm, one gradient average
    from salt_amd.parallel import shard_batch
    sh = shard_batch(8, rank, world)
    Xs, Ts = X[sh].cuda(), T[sh].cuda()
    res['losses'] = [float(m._fit_loop([Xs, Ts])['sum']) for _ in range(2)]
    res['buckets'] = len(list(m.dp._plans.values())[0])
    # trainer decisions: early stopping on ONE rank ends fit() on every rank; only rank 0 writes the checkpoint
    res['any_rank'] = (m.dp.any_rank(rank == 1), m.dp.any_rank(False))
    m.persist(out + '.ckpt%%d' %% rank)
else:
    # ONE process emulating the two ranks: per step the gradient of each half from the same weights (per-half BatchNorm statistics),
    # summed, 1/2 on the optimizer.  Half B first, then the BatchNorm buffers are put back and half A runs: the running statistics
    # end up rank 0's (what nn.DataParallel keeps, models.py:81-85).
    (name, loss_fn, weight), kind = m.loss_function[0], m.loss_function[0][1].native_kind
    res['losses'] = []
    for step in range(2):
        bufs = [b.clone() for b in m.model.buffers()]
        grads = []
        for half in (slice(4, 8), slice(0, 4)):
            if half.start == 0:
                for b, s in zip(m.model.buffers(), bufs):
                    b.copy_(s)
            net = eng.forward(X[half].cuda(), True)
            net.target.copy_(T[half].cuda())
            net.loss_program(kind, weight).run()
            net.bwd.run(side=eng.side_stream)
            torch.cuda.synchronize()
            grads.append(eng.grads.clone())
        res['losses'].append(float(net.loss[0]))
        eng.grads.copy_(grads[1] + grads[0])              # rank 0's + rank 1's, as gloo sums them
        m.optimizer.grad_scale = 0.5
        m.optimizer.step()
torch.cuda.synchronize()
if rank == 0:
    res.update(flat=eng.flat.cpu(), grads=eng.grads.cpu(), scale=m.optimizer.grad_scale,
               bn={k: v.cpu() for k, v in m.model.state_dict().items() if 'running_' in k or 'num_batches' in k})
    torch.save(res, out)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


def _run(tmp_path, mode, world, tag):
    script = tmp_path / 'two_rank_worker.py'
    script.write_text(_WORKER % {'root': ROOT})
    out = tmp_path / (tag + '.pt')
    # fixed-order sums (conftest.deterministic_sums): bit-equality is then by construction
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', SALT_BN_FIN='0', SALT_SE_SHARDS='0')
    if world == 1:
        cmd = [sys.executable, str(script), mode, str(out)]
    else:
        port = str(29800 + (os.getpid() + 7 * len(tag)) % 150)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
               '--master-port', port, str(script), mode, str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return torch.load(out), out


def test_two_ranks_on_one_gpu_same_batch_equals_single_rank(tmp_path):
    """Both ranks hold the SAME batch: the bucketed sum is exactly 2 x the 1-rank gradient, Adam's 1/2 makes the update the 1-rank
    update - weights bit-identical after three steps (the body of test_two_rank_rccl_fit_loop_equals_single_rank, which needs 2 GPUs)."""
    one, _ = _run(tmp_path, 'same', 1, 'one')
    two, _ = _run(tmp_path, 'same', 2, 'two')
    assert two['world'] == 2 and two['scale'] == 0.5 and one['scale'] == 1.0
    assert one['losses'] == two['losses']
    assert torch.equal(two['grads'], one['grads'] * 2)
    assert torch.equal(one['flat'], two['flat'])


def test_two_ranks_on_one_gpu_half_batches_equal_the_gradient_average(tmp_path):
    """Each rank trains on its half of a batch of 8 (per-rank BatchNorm) for two steps; one process that computes the two half-batch
    gradients from the same weights, sums them and steps Adam with 1/2 must land on the same parameters; rank 0 keeps ITS running
    statistics; `persist` writes on rank 0 only; `any_rank` is true on both ranks when one rank raises it."""
    emu, _ = _run(tmp_path, 'halves', 1, 'emu')
    two, out = _run(tmp_path, 'halves', 2, 'dp2')
    assert two['world'] == 2 and two['scale'] == 0.5 and two['buckets'] >= 2
    assert two['any_rank'] == (True, False)
    assert os.path.exists(str(out) + '.ckpt0') and not os.path.exists(str(out) + '.ckpt1')
    ck = torch.load(str(out) + '.ckpt0')
    assert all(k.startswith('module.') for k in ck)                          # the reference's DataParallel key format (models.py:199-204)
    assert two['losses'] == emu['losses'], (two['losses'], emu['losses'])     # rank 0's loss = half A's
    d = (two['flat'] - emu['flat']).abs().max().item()
    assert torch.equal(two['grads'], emu['grads']), (two['grads'] - emu['grads']).abs().max().item()
    assert torch.equal(two['flat'], emu['flat']), d
    for k, v in emu['bn'].items():
        assert torch.equal(v, two['bn'][k]), k
