"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, ctypes mirrors match,
module trees reproduce the reference's state_dict layout, bucket planner, and the N>1 reducer under gloo."""
import ast
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'open-solution-salt-identification_amd')


def test_library_exports_every_declared_symbol():
    import salt_amd
    abi = salt_amd._abi
    hdr = open(abi.HEADER).read()
    declared = set(re.findall(r'\b(salt_[a-z0-9_]+)\s*\(', re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)))
    declared.discard('salt_op_fn')
    assert len(declared) >= 40
    for name in sorted(declared):
        assert hasattr(abi.lib, name), name            # ctypes resolves the symbol or raises AttributeError
    assert abi.lib.salt_abi_version() >= 1
    assert set(abi.DECLARED_SYMBOLS) <= declared | {'salt_last_error'}
    # ... and the other way round: the shipped library exports NO salt_* symbol the header does not declare
    import subprocess
    nm = None
    for tool in ('nm', '/opt/rocm/lib/llvm/bin/llvm-nm'):
        try:
            nm = subprocess.run([tool, '-D', '--defined-only', abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
            break
        except (OSError, subprocess.CalledProcessError):
            continue
    assert nm is not None, 'no nm / llvm-nm to list the exports'
    exported = {ln.split()[-1] for ln in nm.splitlines() if ln.split() and ln.split()[-1].startswith('salt_') and ln.split()[-2] == 'T'}
    exported = {e for e in exported if re.fullmatch(r'salt_[a-z0-9_]+', e)}
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))


def test_every_operator_rejects_null_args_without_touching_a_gpu():
    import ctypes
    import salt_amd
    abi = salt_amd._abi
    for name, (fn, S) in abi.OP_FUNCS.items():
        rc = fn(ctypes.byref(S()), None)
        assert rc != 0, name                              # zeroed struct -> SALT_E_BADARG, never a launch
        assert len(abi.lib.salt_last_error()) > 0


@pytest.mark.parametrize('tag,make', [
    ('unet_resnet34_hyper', lambda A: A.UNetResNet(34, 2, dropout_2d=0.0, pretrained=False, use_hypercolumn=True)),
    ('ternaus_resnet34_deconv', lambda A: A.TernausUNetResNet(34, 2, dropout_2d=0.0, pretrained=False, is_deconv=True)),
    ('ternaus_resnet34_upsample', lambda A: A.TernausUNetResNet(34, 2, dropout_2d=0.0, pretrained=False, is_deconv=False)),
    ('salt_unet', lambda A: A.SaltUNet(2, dropout_2d=0.0, is_deconv=True)),
    ('salt_linknet', lambda A: A.SaltLinkNet(2, dropout_2d=0.0, is_deconv=True)),
    ('unet_resnet152_hyper', lambda A: A.UNetResNet(152, 2, dropout_2d=0.0, pretrained=False, use_hypercolumn=True)),
    ('ternaus_resnet101_deconv', lambda A: A.TernausUNetResNet(101, 2, dropout_2d=0.0, pretrained=False, is_deconv=True)),
])
def test_state_dict_layout_equals_reference(tag, make):
    from salt_amd import architectures as A
    from oracle import specs as OS
    fx = golden('F8_' + tag)
    net = make(A)
    sd = net.state_dict()
    assert list(sd.keys()) == fx['keys'].tolist()
    arch = {'unet': 'UNetResNet', 'tern': 'TernausUNetResNet', 'salt_unet': 'SaltUNet', 'salt_link': 'SaltLinkNet'}[tag[:9] if tag.startswith('salt') else tag[:4]]
    depth = {'unet_resnet152_hyper': 152, 'ternaus_resnet101_deconv': 101}.get(tag)
    spec = OS.SPECS[arch](with_fc=True, **({'depth': depth} if depth else {}))
    for k, (shape, _) in spec.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    # parameters that never receive a gradient in the reference are exactly the ones kept out of the flat buffers
    has = dict(zip(fx['param_names'].tolist(), fx['param_has_grad'].tolist()))
    dead = set(net.dead_parameter_names())
    for k, _ in net.named_parameters():
        assert has[k] == (k not in dead), k


def test_registry_and_config_surface():
    from salt_amd import models
    assert models.ARCHITECTURES['UNetResNet']['model_config']['encoder_depth'] == 34
    assert models.ARCHITECTURES['UNetResNet']['model_config']['use_hypercolumn'] is True
    arch = {'model_params': {'architecture': 'UNetResNet', 'out_channels': 2, 'activation': 'sigmoid'},
            'optimizer_params': {'lr': 1e-4}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
    m = models.SegmentationModel(arch, {'epochs': 1}, {})
    assert m.output_names == ['mask'] and m.loss_function[0][0] == 'mask'
    g = m.optimizer.param_groups[0]
    assert g['lr'] == 1e-4 and g['weight_decay'] == 1e-4 and len(g['params']) == len([p for p in m.model.parameters() if p.requires_grad])
    with pytest.raises(models.SaltError):
        m.model(torch.zeros(1, 3, 64, 64))               # CPU tensor: loud failure, no fallback
    arch['model_params']['activation'] = 'softmax'
    with pytest.raises(NotImplementedError):
        models.SegmentationModel(arch, {'epochs': 1}, {})


def test_bucket_planner_covers_flat_buffer_in_reverse_order():
    from salt_amd.parallel import plan_buckets, shard_batch
    ready, off = [], 0
    sizes = [1000, 50_000, 3_000_000, 10, 9_000_000, 2_000_000, 64]
    for i, n in enumerate(sizes):
        ready.append((off, n, 100 - 10 * i))           # later parameters are ready earlier
        off += (n + 3) // 4 * 4
    b = plan_buckets(ready, off, bucket_bytes=8 << 20)
    assert b[0][1] == off and b[-1][0] == 0
    for (lo, hi, r), (lo2, hi2, r2) in zip(b, b[1:]):
        assert lo == hi2 and r2 >= r                     # contiguous, issued in readiness order
    assert sum(hi - lo for lo, hi, _ in b) == off
    assert [shard_batch(10, r, 4) for r in range(4)] == [slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 10)]
    # the collective that is issued last (the first parameters: final only when backward ends) is kept small
    t = plan_buckets(ready, off, bucket_bytes=32 << 20, tail_bytes=1 << 20)
    assert t[-1][0] == 0 and t[-1][1] * 4 <= 1 << 20 and t[-1][1] > 0
    assert sum(hi - lo for lo, hi, _ in t) == off and all(a[0] == b2[1] for a, b2 in zip(t, t[1:]))
    assert all(r2 >= r for (_, _, r), (_, _, r2) in zip(t, t[1:]))
    assert plan_buckets(ready, off, bucket_bytes=8 << 20, tail_bytes=0) == plan_buckets(ready, off, bucket_bytes=8 << 20, tail_bytes=1 << 40)


def test_first_bucket_closes_early_and_rank_pinning_reports_a_mask():
    """Round 6: (1) plan_buckets(first_pos=...) closes the FIRST bucket as soon as it holds 1 MB and the next gradient would
    only be final after that backward-program position - the wire starts by ~25 % of backward instead of when 32 MB have piled up;
    coverage / order invariants unchanged.  (2) pin_rank_threads splits the allowed cores between the ranks of a node and reports it."""
    from salt_amd.parallel import plan_buckets, pin_rank_threads, _cpulist
    ready, off = [], 0
    sizes = [1000, 50_000, 3_000_000, 10, 9_000_000, 2_000_000, 64, 500_000, 300_000]
    for i, n in enumerate(sizes):
        ready.append((off, n, 100 - 10 * i))           # later parameters are ready earlier (positions 20, 30, ... from the end)
        off += (n + 3) // 4 * 4
    plain = plan_buckets(ready, off, bucket_bytes=32 << 20)
    early = plan_buckets(ready, off, bucket_bytes=32 << 20, first_pos=25)
    assert len(early) == len(plain) + 1 and early[0][2] <= 25 < plain[0][2]        # issued by position 25 instead of 60
    assert (early[0][1] - early[0][0]) * 4 >= 1 << 20
    for b in (plain, early):
        assert b[0][1] == off and b[-1][0] == 0 and sum(hi - lo for lo, hi, _ in b) == off
        assert all(a[0] == c[1] and c[2] >= a[2] for a, c in zip(b, b[1:]))
    assert plan_buckets(ready, off, bucket_bytes=32 << 20, first_pos=10 ** 9) == plain      # nothing is final that late: size rule only
    assert _cpulist('0-3,8,10-11\n') == [0, 1, 2, 3, 8, 10, 11]
    before = os.sched_getaffinity(0)
    try:
        a = pin_rank_threads(1, 2)                       # no GPU here: the allowed cores are split between the two ranks of the node
        if a is not None and len(before) >= 2:
            assert a['ranks_on_node'] == 2 and a['cpus'] == len(before) // 2 and set(os.sched_getaffinity(0)) < set(before)
    finally:
        os.sched_setaffinity(0, before)


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import salt_amd
from salt_amd.parallel import DataParallel, plan_buckets, shard_batch
dp = DataParallel.init_process_group_from_env('gloo')
W = int(os.environ['WORLD_SIZE'])
assert dp.world == W
torch.manual_seed(0)
NS = 8 if W == 2 else 30                         # world 8: an UNEVEN last shard (30 samples: 4 x 7 ranks + 2)
full = torch.randn(NS, 1000)                     # per-sample gradients of a flat 1000-float parameter buffer
sh = shard_batch(NS, dp.rank, dp.world)
assert (sh.stop - sh.start) == (4 if dp.rank < W - 1 or W == 2 else 2)
mine = full[sh].sum(0)
ready = [(0, 300, 30), (300, 300, 20), (600, 400, 10)]
buckets = plan_buckets(ready, 1000, bucket_bytes=1200)
# every rank must derive the SAME bucket plan (same ranges, same order): a rank that issued its collectives in another order would
# dead-lock or, worse, sum mismatched ranges
plans = [None] * W
dist.all_gather_object(plans, [tuple(b) for b in buckets])
assert all(p == plans[0] for p in plans), plans
assert len(buckets) >= 2 and buckets[0][1] == 1000 and buckets[-1][0] == 0
flat = mine.clone()
dp.allreduce_flat(flat, buckets)
assert torch.allclose(flat, full.sum(0), atol=1e-5), (flat - full.sum(0)).abs().max()
w = torch.full((10,), float(dp.rank + 1))
class M(torch.nn.Module):
    def __init__(s):
        super().__init__(); s.p = torch.nn.Parameter(w.clone())
m = M(); dp.broadcast_parameters(m)
assert float(m.p[0]) == 1.0
# autograd-bridge branch of _fit_loop: SUM over ranks + grad_scale = 1/world on the optimizer == the average of the rank gradients
class Eng: pass
class Opt: grad_scale = 1.0
eng, opt = Eng(), Opt()
eng.grads, eng.n_live = mine.clone(), 1000
dp.allreduce_gradients(eng, opt)
assert opt.grad_scale == 1.0 / W
assert torch.allclose(eng.grads * opt.grad_scale, full.sum(0) / W, atol=1e-5)
# trainer decisions are rank consistent: early stopping on ONE rank ends every rank; validation scores are rank 0's
assert dp.any_rank(dp.rank == W - 1) is True and dp.any_rank(False) is False
assert dp.broadcast_scalars([0.25 + dp.rank, 7.0 * (dp.rank + 1)]) == [0.25, 7.0]
dist.barrier(); open(os.path.join(%(out)r, 'rank%%d.ok' %% dp.rank), 'w').write('ok')
'''


@pytest.mark.parametrize('world', [2, 8])
def test_gloo_ranks_gradient_average_and_bucket_plan(tmp_path, world):
    """The N > 1 path on CPU (gloo), world 2 and world 8 (the node the driver scales to): bucketed all-reduce of the flat gradient buffer
    (identical bucket plan on every rank, uneven last minibatch shard), parameter broadcast, 1 / world on the optimizer, rank-consistent
    trainer decisions."""
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % {'root': ROOT, 'out': str(tmp_path)})
    env = dict(os.environ, OMP_NUM_THREADS='1')
    for attempt in range(2):                     # (a rendezvous port left in TIME_WAIT by an earlier run: one retry on another port)
        port = 29500 + (os.getpid() % 500) + world + 37 * attempt
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % world, '--master-addr', '127.0.0.1',
               '--master-port', str(port), str(script)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        if r.returncode == 0:
            break
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert all((tmp_path / ('rank%d.ok' % k)).exists() for k in range(world))


def test_product_never_imports_the_oracle_or_reads_the_reference():
    """Layout rule: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                tree = ast.parse(src)
                for node in ast.walk(tree):
                    names = []
                    if isinstance(node, ast.Import):
                        names = [a.name for a in node.names]
                    elif isinstance(node, ast.ImportFrom):
                        names = [node.module or '']
                    assert not any(n.split('.')[0] == 'oracle' for n in names), f
                assert '/root/reference' not in src, f
    for f in ('bench.py', '__graft_entry__.py'):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            assert '/root/reference' not in open(p).read(), f
    for f in os.listdir(os.path.join(ROOT, 'tests')):
        if f.startswith('test_gpu') and f.endswith('.py'):
            assert 'ref_import' not in open(os.path.join(ROOT, 'tests', f)).read(), f


def test_inference_host_logic_matches_oracle():
    """Host halves of the inference epilogue: variant order, crop window, IoU/IOUT from counts, threshold selection."""
    from salt_amd import inference as I
    from oracle import metrics as OM
    for ud in (False, True):
        for lr in (False, True):
            assert I.tta_variants(ud, lr) == [(s['ud_flip'], s['lr_flip']) for s in OM.tta_specs(ud, lr)]
    for H, W, tgt in [(128, 128, (101, 101)), (256, 256, (202, 202)), (64, 96, (50, 71)), (128, 128, (128, 128))]:
        top, right, bottom, left = OM.crop_pad_sequence(H - tgt[0], W - tgt[1])
        assert I.crop_window(H, W, tgt) == (top, left)
    r = np.random.RandomState(0)
    B, h, w = 9, 20, 17
    gt = (r.rand(B, h, w) > 0.6).astype(np.uint8)
    gt[0] = 0; gt[1] = 0
    prob = r.rand(B, h, w)
    prob[0] = 0.0                                         # both empty
    ths = np.linspace(0.5, 0.3, 21)
    inter = np.array([[int(((prob[b] > t) & (gt[b] > 0)).sum()) for t in ths] for b in range(B)])
    pred = np.array([[int((prob[b] > t).sum()) for t in ths] for b in range(B)])
    iou, iout = I.scores_from_counts(inter, pred, gt.reshape(B, -1).sum(1))
    for k, t in enumerate(ths):
        preds = [(prob[b] > t).astype(np.uint8) for b in range(B)]
        assert abs(iou[k] - OM.intersection_over_union(list(gt), preds)) < 1e-12
        assert abs(iout[k] - OM.intersection_over_union_thresholds(list(gt), preds)) < 1e-12
    t_best, iou_b, iout_b = I.select_threshold([(inter, pred, gt.reshape(B, -1).sum(1))])
    assert 0.3 <= t_best <= 0.5 and 0.0 <= iou_b <= 1.0 and 0.0 <= iout_b <= 1.0
    with pytest.raises(I.SaltError):
        I.crop_threshold(torch.zeros(1, 2, 8, 8))         # CPU tensor: loud failure, no fallback


def test_input_pipeline_geometry_matches_reference_pad_sequence():
    """InferencePad split (augmentation.py:262-277 + utils.py:308-313): 101 -> 128 pads top 13 / bottom 14, left 14 / right 13."""
    from salt_amd.input_pipeline import DevicePreprocessor, pad_split
    from oracle import inputs as OI
    fx = golden('F10_post_metric')
    assert tuple(fx['crop_seq_27'].tolist()) == OI.crop_pad_sequence(27, 27) == (13, 13, 14, 14)       # (top, right, bottom, left)
    assert pad_split(101) == (13, 14) and pad_split(128) == (0, 0) and pad_split(202) == (27, 27)
    assert DevicePreprocessor(False).geometry(101, 101) == (0, 0, 13, 14, 128, 128)
    assert DevicePreprocessor(True).geometry(101, 101) == (102, 102, 13, 13, 128, 128)
    top, right, bottom, left = OI.crop_pad_sequence(155 % 64 and 64 - 155 % 64, 0)
    assert DevicePreprocessor(False).geometry(155, 128)[2] == top


class _FakeTransformer:
    """Just the attributes the callbacks touch (callbacks.py:42-48)."""

    def __init__(self, scores):
        self.model = torch.nn.Linear(2, 2)
        self.optimizer = torch.optim.SGD(self.model.parameters(), lr=0.1)
        self.loss_function = [('mask', None, 1.0)]
        self.output_names = ['mask']
        self.validation_loss = {}
        self.scores = list(scores)
        self.calls = 0
        self.saved = []

    def score_validation(self, datagen):
        v = self.scores[self.calls]
        self.calls += 1
        return {'sum': torch.tensor([1.0 - v]), 'iou': torch.tensor([v]), 'iout': torch.tensor([v])}

    def persist(self, path):
        self.saved.append((len(self.validation_loss) - 1, path))


def test_callbacks_follow_reference_semantics(tmp_path):
    """ModelCheckpoint / EarlyStopping / ReduceLROnPlateau driven by a metric sequence (callbacks.py:204-241,758-829); the
    validation pass runs once per epoch and is shared through transformer.validation_loss (callbacks.py:522-527)."""
    from salt_amd import callbacks as C
    scores = [0.50, 0.60, 0.55, 0.58, 0.59, 0.40, 0.30]
    tr = _FakeTransformer(scores)
    cfg = {'model_checkpoint': {'filepath': str(tmp_path / 'ck' / 'best.torch'), 'epoch_every': 1, 'metric_name': 'iout', 'minimize': False},
           'reduce_lr_on_plateau_scheduler': {'metric_name': 'iout', 'minimize': False, 'reduce_factor': 0.1, 'reduce_patience': 1, 'min_lr': 1e-7},
           'training_monitor': {'batch_every': 0, 'epoch_every': 1}, 'experiment_timing': {'batch_every': 0, 'epoch_every': 1},
           'validation_monitor': {'epoch_every': 1, 'data_dir': None, 'loader_mode': 'resize_and_pad', 'use_depth': False},
           'neptune_monitor': {'model_name': 'network'},
           'early_stopping': {'patience': 2, 'metric_name': 'iout', 'minimize': False}}
    cl = C.callbacks_network(cfg)
    assert [type(c).__name__ for c in cl.callbacks] == ['ExperimentTiming', 'TrainingMonitor', 'ValidationMonitor', 'ModelCheckpoint',
                                                        'ReduceLROnPlateauScheduler', 'EarlyStopping']
    cl.set_params(tr, validation_datagen=('gen', 0), meta_valid=None)
    cl.on_train_begin()
    lrs, stopped_at = [], None
    for epoch in range(len(scores)):
        cl.on_epoch_begin()
        cl.on_batch_begin()
        cl.on_batch_end(metrics={'sum': torch.tensor([0.5])})
        cl.on_epoch_end()
        lrs.append(tr.optimizer.param_groups[0]['lr'])
        if cl.training_break():
            stopped_at = epoch
            break
    assert tr.calls == stopped_at + 1                                    # one validation pass per epoch, shared by 4 callbacks
    assert [e for e, _ in tr.saved] == [0, 1]                            # epoch 0 always, then only strict improvements (0.60)
    # ReduceLROnPlateau(max, 0.1, patience 1): best 0.60 at epoch 1; bad epochs 2,3 -> reduced at epoch 3
    assert lrs[:3] == [0.1, 0.1, 0.1] and abs(lrs[3] - 0.01) < 1e-12 and abs(lrs[4] - 0.01) < 1e-12
    # EarlyStopping(patience 2): epochs since best (epoch 1) = 1,2,3 at epochs 2,3,4 -> break when > 2, i.e. after epoch 4
    assert stopped_at == 4
    tm = [c for c in cl.callbacks if isinstance(c, C.TrainingMonitor)][0]
    assert len(tm.history) == stopped_at + 1 and abs(tm.history[0]['sum'] - 0.5) < 1e-7


def test_fused_adam_is_a_torch_optimizer_for_schedulers():
    from salt_amd import models
    arch = {'model_params': {'architecture': 'VanillaUNet', 'out_channels': 2, 'activation': 'sigmoid'},
            'optimizer_params': {'lr': 1e-3}, 'regularizer_params': {'regularize': False, 'weight_decay_conv2d': 0.0}}
    m = models.SegmentationModel(arch, {'epochs': 1}, {})
    assert isinstance(m.optimizer, torch.optim.Optimizer) and m.optimizer.param_groups[0]['weight_decay'] == 0.0
    sch = torch.optim.lr_scheduler.ExponentialLR(m.optimizer, gamma=0.5)     # callbacks.py:170-201
    m.optimizer.steps = 1
    sch.step()
    assert abs(m.optimizer.param_groups[0]['lr'] - 5e-4) < 1e-12


def test_run_length_encoding_matches_reference_golden():
    """utils.py:99-133 executed by the reference (fixture F12) vs the vectorised host implementation."""
    from salt_amd import inference as I
    fx = golden('F12_rle')
    masks = list(fx['masks']) + [fx['small']]
    offs = fx['rle_offsets']
    for i, m in enumerate(masks):
        ref = fx['rle_flat'][offs[i]:offs[i + 1]].tolist()
        assert I.run_length_encoding(m) == ref, i
        assert I.run_length_encoding(torch.from_numpy(m)) == ref
        assert np.array_equal(I.run_length_decoding(ref, m.shape), m)
        assert np.array_equal(I.run_length_decoding(' '.join(str(v) for v in ref), m.shape), m)


def test_bench_self_launches_its_ranks_under_gloo():
    """`python bench.py --gpus 2` with no launcher in the environment re-launches itself through torch.distributed.run (VERDICT r2
    missing #1; the reference's nn.DataParallel, models.py:81-85, needed no launcher).  The hidden --selftest-launch mode brings the
    ranks up under gloo on CPU and prints the one JSON line from rank 0."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--selftest-launch'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks'] == 2 and out['allreduce_sum'] == out['expected_sum'] == 3.0


def test_bench_refuses_more_gpus_than_visible_with_a_message():
    """On a box with fewer GPUs than --gpus the self-launcher exits non-zero with a clear message instead of an assertion."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '64'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 3 and 'needs 64 visible GPUs' in r.stderr


def test_wgrad_ls_index_model_is_consistent():
    """conv_wgrad_ls_kernel's LDS layout on the CPU (tools/wgrad_ls_model.py): the loader's (piece, lane) -> (row, 16-byte slot) -> (pixel,
    channel slot) map, every transposed-read address of the MFMA waves (unit-step and stride-2 geometry, one or two images per k-step)
    and the bank sets of every 32-lane group - the checks that let the kernel pass its parity tests on its first GPU run."""
    import importlib.util
    path = os.path.join(ROOT, 'tools', 'wgrad_ls_model.py')
    spec = importlib.util.spec_from_file_location('wgrad_ls_model', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for NB in (1, 2):
        assert mod.check(NB, 4)['pieces'] == (17 if NB == 1 else 18)
        assert mod.check(NB, 8)['pre_pieces'] == 5
        assert mod.check_stride2(NB, 2) == {'pieces': 22, 'pre_first_q_piece': 13, 'pre_pieces': 5}


def test_no_wrong_result_switches_in_the_shipped_engine():
    """the product engine has no branch that computes wrong gradients - the SALT_EXP_* timing switches of rounds 3 - 4 are
    gone; what is left of the BatchNorm-fold experiment (SALT_FWD_BN_FOLD) builds a FORWARD-ONLY graph whose backward closure raises."""
    import re
    src = ''
    for fn in ('engine.py', 'architectures.py', 'runtime.py', 'models.py', 'optim.py', 'parallel.py'):
        with open(os.path.join(ROOT, 'open-solution-salt-identification_amd', fn)) as f:
            src += f.read()
    assert not re.search(r'SALT_EXP_|SALT_TIMING_ONLY|timing_experiment', src)
    assert 'forward-only' in src and 'SALT_FWD_BN_FOLD' in src

