"""GPU end-to-end: SegmentationModel.fit with the reference's callback stack (validation monitor with the on-device threshold sweep,
best-metric checkpoint with 'module.' keys, ReduceLROnPlateau, early stopping) and checkpoint reload."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _data(n, seed):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(128), torch.arange(128), indexing='ij')
    X = torch.randn(n, 1, 128, 128, generator=g) * 0.3
    M = torch.zeros(n, 1, 128, 128)
    for i in range(n):
        if i % 3 == 0:
            continue
        cy, cx, r = [int(v) for v in torch.randint(30, 98, (3,), generator=g)]
        M[i, 0] = (((yy - cy) ** 2 + (xx - cx) ** 2) < (r // 2) ** 2).float()
    X = X + 0.9 * M
    return X, torch.cat([1 - M, M], 1)


def test_fit_with_reference_callback_stack(tmp_path):
    from salt_amd import models, inference as I
    from oracle import metrics as OM
    arch = {'model_params': {'architecture': 'VanillaUNet', 'out_channels': 2, 'activation': 'sigmoid', 'compute_dtype': 'f32'},
            'optimizer_params': {'lr': 2e-3}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
    ck = str(tmp_path / 'checkpoints' / 'network' / 'best.torch')
    cfg = {'model_checkpoint': {'filepath': ck, 'epoch_every': 1, 'metric_name': 'iout', 'minimize': False},
           'reduce_lr_on_plateau_scheduler': {'metric_name': 'iout', 'minimize': False, 'reduce_factor': 0.1, 'reduce_patience': 10, 'min_lr': 1e-7},
           'training_monitor': {'batch_every': 0, 'epoch_every': 1}, 'experiment_timing': {'batch_every': 0, 'epoch_every': 1},
           'validation_monitor': {'epoch_every': 1, 'data_dir': None, 'loader_mode': 'resize_and_pad', 'use_depth': False},
           'neptune_monitor': {'model_name': 'network', 'image_nr': 16, 'image_resize': 1.0, 'image_every': 10, 'use_depth': False},
           'early_stopping': {'patience': 20, 'metric_name': 'iout', 'minimize': False}}
    torch.manual_seed(0)
    m = models.SegmentationModel(arch, {'epochs': 3}, cfg)
    Xt, Mt = _data(32, 1)
    Xv, Mv = _data(16, 2)
    train = ([[Xt[i:i + 8], Mt[i:i + 8]] for i in range(0, 32, 8)], 3)
    valid = ([[Xv[i:i + 8], Mv[i:i + 8]] for i in range(0, 16, 8)], 1)
    m.fit(train, valid)
    assert sorted(m.validation_loss) == [0, 1, 2]
    for e in range(3):
        v = m.validation_loss[e]
        assert set(v) == {'sum', 'iou', 'iout'} and all(torch.isfinite(x).all() for x in v.values())
        assert 0.0 <= float(v['iou']) <= 1.0 and 0.0 <= float(v['iout']) <= 1.0
    # the validation score equals the numpy restatement on the model's own probabilities at the selected threshold
    m.model.eval()
    with torch.no_grad():
        prob = torch.sigmoid(torch.cat([m.model(Xv[i:i + 8].to(DEV)).float() for i in range(0, 16, 8)], 0)).cpu().numpy()
    m.model.train()
    gts = [OM.crop_image(Mv[b].numpy(), (101, 101))[1] for b in range(16)]
    preds = [(OM.crop_image(prob[b], (101, 101))[1].astype(np.float64) > m.best_threshold).astype(np.uint8) for b in range(16)]
    assert abs(OM.intersection_over_union(gts, preds) - float(m.validation_loss[2]['iou'])) < 1e-6
    assert abs(OM.intersection_over_union_thresholds(gts, preds) - float(m.validation_loss[2]['iout'])) < 1e-6
    # checkpoint: reference format ('module.' prefix) and loadable into a fresh model through load()
    assert os.path.exists(ck)
    sd = torch.load(ck, map_location='cpu')
    assert all(k.startswith('module.') for k in sd) and len(sd) == len(m.model.state_dict())
    m2 = models.SegmentationModel(arch, {'epochs': 1}, {}).load(ck)
    m2.model.eval()
    with torch.no_grad():
        a = m2.model(Xv[:4].to(DEV)).float().cpu()
    for k, v in m2.model.state_dict().items():
        assert torch.equal(v.cpu(), sd['module.' + k]), k
    assert torch.isfinite(a).all()
    # training reduced the loss
    tm = [c for c in m.callbacks.callbacks if type(c).__name__ == 'TrainingMonitor'][0]
    assert tm.history[-1]['sum'] < tm.history[0]['sum']


def test_early_stopping_and_lr_schedule_on_gpu(tmp_path):
    from salt_amd import models
    arch = {'model_params': {'architecture': 'VanillaUNet', 'out_channels': 2, 'activation': 'sigmoid', 'compute_dtype': 'bf16'},
            'optimizer_params': {'lr': 1e-3}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
    cfg = {'reduce_lr_on_plateau_scheduler': {'metric_name': 'sum', 'minimize': False, 'reduce_factor': 0.1, 'reduce_patience': 0, 'min_lr': 1e-7},
           'early_stopping': {'patience': 1, 'metric_name': 'sum', 'minimize': False}}     # "maximise the loss": it will not improve
    m = models.SegmentationModel(arch, {'epochs': 10}, cfg)
    Xt, Mt = _data(16, 3)
    gen = ([[Xt[i:i + 8], Mt[i:i + 8]] for i in range(0, 16, 8)], 1)
    m.fit(gen, gen)
    assert 2 <= len(m.validation_loss) < 10                                 # stopped early
    assert m.optimizer.param_groups[0]['lr'] < 1e-3                         # the fused Adam saw the scheduler's write


_DP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=%(port)r, RANK='0', WORLD_SIZE='1')
forced = sys.argv[1] == 'dp'
if forced:
    os.environ['SALT_FORCE_DP_PATH'] = '1'
    dist.init_process_group('nccl')
from salt_amd import models
arch = {'model_params': {'architecture': 'UNetResNet', 'out_channels': 2, 'activation': 'sigmoid', 'loss': 'lovasz', 'compute_dtype': 'bf16'},
        'optimizer_params': {'lr': 1e-3}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
torch.manual_seed(3)
m = models.SegmentationModel(arch, {'epochs': 1}, {})
m._to_device(); m.model.train()
g = torch.Generator().manual_seed(5)
X = torch.randn(4, 3, 128, 128, generator=g).cuda()
M = (torch.rand(4, 1, 128, 128, generator=g) > 0.6).float()
T = torch.cat([1 - M, M], 1).cuda()
losses = [float(m._fit_loop([X, T])['sum']) for _ in range(3)]
eng = m.model.engine()
torch.save({'losses': losses, 'flat': eng.flat.cpu(), 'grads': eng.grads.cpu(),
            'buckets': len(list(m.dp._plans.values())[0]) if forced else 0}, sys.argv[2])
'''


def test_bucketed_data_parallel_path_matches_plain_step(tmp_path):
    """The multi-GPU code path on one GPU: backward in bucket segments, RCCL all-reduce on the communication stream overlapped with
    the remaining segments (1-rank communicator), 1/world folded into Adam - three steps must reproduce the plain single-GPU
    steps bit for bit (a sum over one rank is the identity; the kernels and their order are the same)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'dp_worker.py'
    script.write_text(_DP_WORKER % {'root': root, 'port': str(29600 + os.getpid() % 300)})
    outs = {}
    for mode in ('plain', 'dp'):
        out = tmp_path / (mode + '.pt')
        env = dict(os.environ, SALT_BN_FIN='0', SALT_SE_SHARDS='0')          # fixed-order sums: bit-equality is by construction (conftest.deterministic_sums)
        r = subprocess.run([sys.executable, str(script), mode, str(out)], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        outs[mode] = torch.load(out)
    assert outs['dp']['buckets'] >= 2                                  # the R34 net is split into several collectives
    assert outs['plain']['losses'] == outs['dp']['losses']
    assert torch.equal(outs['plain']['grads'], outs['dp']['grads'])
    assert torch.equal(outs['plain']['flat'], outs['dp']['flat'])
