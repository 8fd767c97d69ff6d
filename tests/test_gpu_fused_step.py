"""Parity of the SHIPPED training entry point: ``SegmentationModel._fit_loop`` -> ``_fused_step`` (forward program, native loss
writing dlogits in place, ``dp.backward`` in bucket segments, fused Adam with its gradient scale) against the goldens the
reference produced (common_blocks/models.py:105-136 executed on the reference's own modules) and against the oracle.

The autograd-bridge tests (test_gpu_models.py) drive ``net(x) -> loss.backward() -> opt.step()``; bench.py times THIS path."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import golden, T, assert_close
import closed_form as CF
from test_gpu_models import _fill_closed_form, _grad_report

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _segmentation_model(architecture, loss='lovasz', dtype='f32', lr=1e-4):
    from salt_amd.models import SegmentationModel
    arch = {'model_params': {'architecture': architecture, 'out_channels': 2, 'activation': 'sigmoid', 'loss': loss, 'compute_dtype': dtype},
            'optimizer_params': {'lr': lr}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
    m = SegmentationModel(arch, {'epochs': 1}, {})
    return m


@pytest.mark.parametrize('tag,architecture', [('unet_resnet34_hyper', 'UNetResNet'), ('ternaus_resnet34_deconv', 'TernausUNetResNet')])
def test_fit_loop_fused_step_matches_reference_golden(tag, architecture):
    """models.py:105-136 through the fused path, Lovasz hinge: loss, every per-tensor gradient norm, selected full gradients,
    post-Adam weight norms and BatchNorm running-statistics sums exactly as test_one_training_step_matches_reference asserts
    them for the autograd-bridge path."""
    fx = golden('F8_' + tag)
    m = _segmentation_model(architecture)
    _fill_closed_form(m.model)
    m._to_device()
    m.model.train()
    net = m.model
    metrics = m._fit_loop([T(fx['x']), T(fx['t'])])
    torch.cuda.synchronize()
    loss, ref = float(metrics['sum']), float(fx['train_loss'])
    assert abs(loss - ref) < 2e-3 * max(1.0, abs(ref)), (loss, ref)
    eng = net.engine()
    cnet = eng.net(tuple(fx['x'].shape), True)
    assert_close(cnet.logits.cpu(), fx['train_logits'], 2e-3, 'train logits')
    names = fx['param_names'].tolist()
    idx = {n: i for i, n in enumerate(names)}
    dead = set(net.dead_parameter_names())
    own = dict(net.named_parameters())
    checked, worst = 0, (0.0, '')
    for k, p in own.items():
        i = idx[k]
        has = bool(fx['param_has_grad'][i])
        assert has == (k not in dead), k
        if has and fx['grad_norm'][i] > 1e-4:
            off, n = eng.grad_range(p)
            gn = float(eng.grads[off:off + n].double().norm())        # Adam reads the gradients, it does not modify them
            worst = max(worst, (abs(gn - fx['grad_norm'][i]) / fx['grad_norm'][i], k))
            checked += 1
    assert checked > 100 and worst[0] < 1e-2, (checked, worst)
    for k in fx:
        if k.startswith('fullgrad:'):
            p = own[k[9:]]
            off, n = eng.grad_range(p)
            assert_close(eng.grads[off:off + n].view(p.shape).cpu(), fx[k], 2e-2, k)
    for k, p in own.items():
        i = idx[k]
        if fx['param_has_grad'][i] and fx['grad_norm'][i] > 1e-4:
            pn = float(p.detach().double().norm())
            assert abs(pn - fx['post_norm'][i]) <= 1e-4 * max(fx['post_norm'][i], 1e-3), (k, pn, fx['post_norm'][i])
    sd = net.state_dict()
    for k, s in zip(fx['bn_keys'].tolist(), fx['bn_sum'].tolist()):
        assert abs(float(sd[k].double().sum()) - s) <= 1e-3 * max(1.0, abs(s)), k


@pytest.mark.parametrize('loss', ['lovasz', 'bce_dice'])
def test_fit_loop_fused_step_matches_oracle_step(loss):
    """One fused step of the hypercolumn ResNet34 U-Net from default initialisation vs the oracle's step (forward, loss, backward,
    Adam + L2): loss value, every parameter gradient (relative L2 / cosine), and the updated weights."""
    from oracle import nets as ON, specs as OS, losses as OL
    torch.manual_seed(5)
    m = _segmentation_model('UNetResNet', loss)
    spec = OS.SPECS['UNetResNet'](with_fc=True)
    sd = OS.init_state(spec, seed=7)
    m.model.load_state_dict({k: sd[k] for k in m.model.state_dict() if k in sd}, strict=False)
    sd = {k: v.detach().clone() for k, v in m.model.state_dict().items() if k in spec}
    x = CF.input_for('r34', (4, 3, 64, 64))
    t = CF.mask_for('r34', (4, 64, 64))
    m._to_device()
    m.model.train()
    dead = set(m.model.dead_parameter_names())
    keys = [k for k in OS.trainable_keys(spec) if k not in dead]
    for k in keys:
        sd[k].requires_grad_(True)
    out_r = ON.unet_resnet(sd, x, True)
    loss_r = OL.LOSSES[loss](out_r, t)
    loss_r.backward()
    metrics = m._fit_loop([x, t])
    torch.cuda.synchronize()
    assert abs(float(metrics['sum']) - float(loss_r)) < 1e-4 * max(1.0, abs(float(loss_r)))
    # Lovasz: g_k = J_k - J_(k-1) carries ~1e-3 relative fp32 cancellation noise in torch itself (test_gpu_models.py)
    worst, cos, n = _grad_report(m.model, {k: sd[k].grad for k in keys})
    assert n > 120 and worst[0] < 5e-2 and cos > 0.9999, (worst, cos, n)
    # the oracle's Adam + L2 step on the oracle's gradients vs the fused Adam kernel on the HIP gradients
    params = [sd[k] for k in keys]
    before = [p.detach().clone() for p in params]
    with torch.no_grad():
        OL.adam_l2_step([p.data for p in params], [p.grad for p in params], [torch.zeros_like(p) for p in params],
                        [torch.zeros_like(p) for p in params], 1)
    own = dict(m.model.named_parameters())
    moved = 0
    for k, p0, p1 in zip(keys, before, params):
        mine = own[k].detach().cpu()
        step_ref, step_mine = (p1.detach() - p0), (mine - p0)
        # first Adam step: |delta| = lr * g / (|g| + eps) -> +-1e-4 wherever the gradient is not tiny; compare the updates themselves
        big = sd[k].grad.abs() > 1e-6
        if int(big.sum()) == 0:
            continue
        moved += 1
        agree = float(((step_ref - step_mine).abs()[big] < 2e-5).float().mean())
        assert agree > 0.98, (k, agree)
    assert moved > 100


class _TwoIdenticalRanks:
    """DataParallel stand-in for ONE GPU: world = 2 with both ranks holding the same minibatch, i.e. the SUM all-reduce of a bucket
    is exactly 2 x the local gradient (a power of two: exact in fp32).  Everything else - the bucket plan, segmented backward,
    the communication stream's event ordering, grad_scale = 1/2 folded into Adam - is the production code."""

    @staticmethod
    def make():
        from salt_amd import parallel

        class Two(parallel.DataParallel):
            def _all_reduce(self, t):
                t.mul_(2.0)
                return None
        return Two(rank=0, world=2, bucket_bytes=8 << 20)


@pytest.mark.parametrize('branch', ['fused', 'bridge'])
def test_grad_scale_half_on_doubled_gradients_equals_plain_step(branch, deterministic_sums):
    """world = 2 semantics on one GPU: gradients are summed over two identical ranks and Adam applies grad_scale = 0.5; three steps
    must reproduce the plain single-rank steps BIT FOR BIT (scaling by 2 and by 1/2 is exact).  'bridge' = a loss without
    ``native_kind`` (models.py autograd branch), whose all-reduce used to leave grad_scale at 1."""
    from salt_amd import losses
    results = {}
    for mode in ('plain', 'two'):
        torch.manual_seed(11)
        m = _segmentation_model('UNetResNet', 'lovasz', dtype='bf16', lr=1e-3)
        if branch == 'bridge':
            m.loss_function = [('mask', lambda o, t: losses.lovasz_loss(o, t), 1.0)]
        if mode == 'two':
            m.dp = _TwoIdenticalRanks.make()
        m._to_device()
        m.model.train()
        g = torch.Generator().manual_seed(5)
        X = torch.randn(4, 3, 128, 128, generator=g)
        M = (torch.rand(4, 1, 128, 128, generator=g) > 0.6).float()
        Tt = torch.cat([1 - M, M], 1)
        ls = [float(m._fit_loop([X, Tt])['sum']) for _ in range(3)]
        torch.cuda.synchronize()
        eng = m.model.engine()
        results[mode] = (ls, eng.flat.clone(), eng.grads.clone(), m.optimizer.grad_scale,
                         len(list(m.dp._plans.values())[0]) if (mode == 'two' and branch == 'fused') else 0)
    assert results['plain'][3] == 1.0 and results['two'][3] == 0.5
    if branch == 'fused':
        assert results['two'][4] >= 3                                   # several buckets: the segmented path really ran
    assert results['plain'][0] == results['two'][0]
    assert torch.equal(results['two'][2], results['plain'][2] * 2)     # the reduced (summed) gradients
    assert torch.equal(results['plain'][1], results['two'][1])         # the weights after three steps


def test_bf16_whole_network_vs_fp32_oracle_c2_shape():
    """BASELINE C2's dtype and shape ([32,3,128,128], bf16 activations / MFMA inputs, fp32 accumulation, masters, statistics and
    loss).  Yardsticks: (1) the fp32 oracle - eval logits, masks and both losses must agree to bf16 resolution; (2) the SAME oracle
    with bf16 STORAGE emulated at the tensor boundaries where the HIP path stores bf16 (oracle.blocks.bf16_storage): at random
    initialisation train-mode BatchNorm through ~55 layers amplifies storage rounding to ~7 % in the logits and to a gradient
    cosine of ~0.76 in ANY bf16-storage implementation, so the train-step bounds are "no worse than the independent emulation",
    not chosen constants."""
    from oracle import nets as ON, specs as OS, losses as OL, blocks as OB
    from salt_amd import losses as HL
    torch.manual_seed(5)
    m = _segmentation_model('UNetResNet', 'bce_dice', dtype='bf16')
    spec = OS.SPECS['UNetResNet'](with_fc=True)
    sd = OS.init_state(spec, seed=7)
    m.model.load_state_dict({k: sd[k] for k in m.model.state_dict() if k in sd}, strict=False)
    sd = {k: v.detach().clone() for k, v in m.model.state_dict().items() if k in spec}
    x = CF.input_for('c2', (32, 3, 128, 128))
    t = CF.mask_for('c2', (32, 128, 128))
    m._to_device()
    m.model.eval()
    with torch.no_grad():
        y = m.model(x.to(DEV)).float().cpu()
        yr = ON.unet_resnet(sd, x, False)

    def rel_l2(a, b):
        return float((a.double() - b.double()).norm() / b.double().norm())
    e_eval = rel_l2(y, yr)
    agree = float(((y[:, 1] > 0) == (yr[:, 1] > 0)).float().mean())
    m.model.train()
    dead = set(m.model.dead_parameter_names())
    keys = [k for k in OS.trainable_keys(spec) if k not in dead]

    def oracle_step(emulate):
        s2 = {k: v.detach().clone() for k, v in sd.items()}
        for k in keys:
            s2[k].requires_grad_(True)
        if emulate:
            with OB.bf16_storage():
                o = ON.unet_resnet(s2, x, True)
        else:
            o = ON.unet_resnet(s2, x, True)
        # gradients under the smooth BCE+Dice loss: at initialisation every hinge error is 1 +- 1e-2, so the Lovasz sort order (and
        # with it dL/dlogits) is decided by noise below bf16 resolution - that would measure the loss, not the network
        l = OL.mixed_dice_bce_loss(o, t)
        l.backward()
        return o.detach(), float(l), {k: s2[k].grad for k in keys}
    out_r, loss_r, g_r = oracle_step(False)
    out_e, loss_e, g_e = oracle_step(True)
    metrics = m._fit_loop([x, t])
    torch.cuda.synchronize()
    cnet = m.model.engine().net((32, 3, 128, 128), True)
    logits = cnet.logits.cpu()
    e_train, e_train_emu = rel_l2(logits, out_r), rel_l2(out_e, out_r)
    worst, cos, n = _grad_report(m.model, g_r)
    flat_r = torch.cat([g_r[k].double().reshape(-1) for k in keys])
    flat_e = torch.cat([g_e[k].double().reshape(-1) for k in keys])
    cos_emu = float((flat_r * flat_e).sum() / (flat_r.norm() * flat_e.norm()))
    lv, _ = HL.native_loss(cnet.logits, t.to(DEV), 'lovasz', want_grad=False)
    lv_r = float(OL.lovasz_loss(out_r, t))
    print('bf16 C2: eval logits relL2 %.3e (mask agreement %.5f) | train logits relL2 HIP %.3e / emulated bf16 storage %.3e | BCE+Dice %.5f '
          'vs %.5f | Lovasz %.5f vs %.5f | gradient cosine HIP %.4f / emulated %.4f' % (e_eval, agree, e_train, e_train_emu,
                                                                                      float(metrics['sum']), loss_r, float(lv), lv_r, cos, cos_emu))
    from helpers import record_parity
    mis = (y[:, 1] > 0) != (yr[:, 1] > 0)
    record_parity('C2_r34_hypercolumn_bf16_eval_masks', config='[32,3,128,128] bf16 HIP vs fp32 oracle.nets.unet_resnet, logit[1] > 0',
                  decisions=int(mis.numel()), differ=int(mis.sum()), agreement=agree, eval_logits_rel_l2=e_eval,
                  max_abs_ref_logit_at_differing=float(yr[:, 1][mis].abs().max()) if bool(mis.any()) else 0.0,
                  train_logits_rel_l2_hip=e_train, train_logits_rel_l2_emulated_bf16_storage=e_train_emu, gradient_cosine_hip=float(cos),
                  gradient_cosine_emulated=cos_emu)
    assert e_eval < 3e-2, e_eval                                          # measured 1e-2: ~55 bf16 roundings of 2^-9 each
    assert agree > 0.995, agree                                           # masks differ only where |logit| is within bf16 noise of 0
    assert abs(float(metrics['sum']) - loss_r) < 1e-2 * max(1.0, abs(loss_r)) and abs(float(lv) - lv_r) < 1e-2 * max(1.0, abs(lv_r))
    assert e_train <= 1.25 * e_train_emu + 5e-3, (e_train, e_train_emu)
    assert n > 120 and (1 - cos) <= 1.25 * (1 - cos_emu) + 1e-2, (cos, cos_emu)
    own = dict(m.model.named_parameters())
    eng = m.model.engine()
    for k in ('final.1.weight', 'final.0.conv.weight', 'dec1.conv2.conv.weight', 'dec3.conv1.conv.weight', 'center.0.conv.weight',
              'encoders.encoder.layer3.2.conv1.weight', 'encoders.encoder.layer1.0.conv1.weight', 'encoders.encoder.conv1.weight'):
        off, cnt = eng.grad_range(own[k])
        e_hip = rel_l2(eng.grads[off:off + cnt].view(own[k].shape).cpu(), g_r[k])
        e_emu = rel_l2(g_e[k], g_r[k])
        assert e_hip <= 1.3 * e_emu + 1e-2, (k, e_hip, e_emu)


def test_step_graph_replay_equals_eager_steps(deterministic_sums):
    """The training step captured into ONE hipGraph (pack, two-stream forward, loss, two-stream backward, Adam) and replayed must
    reproduce the eagerly launched steps bit for bit: same kernels, same order, same static buffers."""
    results = {}
    for mode in ('eager', 'graph'):
        torch.manual_seed(11)
        m = _segmentation_model('UNetResNet', 'lovasz', dtype='bf16', lr=1e-3)
        m.step_graph = mode == 'graph'
        m._to_device()
        m.model.train()
        g = torch.Generator().manual_seed(5)
        X = torch.randn(4, 3, 128, 128, generator=g)
        M = (torch.rand(4, 1, 128, 128, generator=g) > 0.6).float()
        Tt = torch.cat([1 - M, M], 1)
        ls = [float(m._fit_loop([X + 0.01 * i, Tt])['sum']) for i in range(5)]        # step 0 eager, step 1 captures, 2.. replay
        torch.cuda.synchronize()
        eng = m.model.engine()
        sd = {k: v.detach().clone() for k, v in m.model.state_dict().items()}
        results[mode] = (ls, eng.flat.clone(), eng.grads.clone(), sd, m.optimizer.steps,
                         sum(len(n.__dict__.get('_step_graphs', {})) for n in eng.nets.values()))
        # the eval-mode program sees the weights the captured Adam wrote (packed copies are refreshed)
        m.model.eval()
        with torch.no_grad():
            results[mode] += (m.model(X.to(DEV)).float().cpu(),)
    assert results['graph'][5] == 1 and results['eager'][5] == 0
    assert results['eager'][4] == results['graph'][4] == 5
    assert results['eager'][0] == results['graph'][0]
    assert torch.equal(results['eager'][2], results['graph'][2]) and torch.equal(results['eager'][1], results['graph'][1])
    for k, v in results['eager'][3].items():
        assert torch.equal(v, results['graph'][3][k]), k                         # BatchNorm running statistics, num_batches_tracked
    assert torch.equal(results['eager'][6], results['graph'][6])


_RCCL_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
world = int(os.environ.get('WORLD_SIZE', '1'))
if world > 1:
    torch.cuda.set_device(int(os.environ['LOCAL_RANK']))
    dist.init_process_group('nccl')
from salt_amd import models
arch = {'model_params': {'architecture': 'UNetResNet', 'out_channels': 2, 'activation': 'sigmoid', 'loss': 'lovasz', 'compute_dtype': 'bf16'},
        'optimizer_params': {'lr': 1e-3}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
torch.manual_seed(3)
m = models.SegmentationModel(arch, {'epochs': 1}, {})
m.dp.bucket_bytes = 8 << 20
m._to_device(); m.model.train()
m.dp.broadcast_parameters(m.model)
g = torch.Generator().manual_seed(5)
X = torch.randn(4, 3, 128, 128, generator=g).cuda()
M = (torch.rand(4, 1, 128, 128, generator=g) > 0.6).float()
T = torch.cat([1 - M, M], 1).cuda()
losses = [float(m._fit_loop([X, T])['sum']) for _ in range(3)]     # every rank trains on the SAME batch
torch.cuda.synchronize()
eng = m.model.engine()
if int(os.environ.get('RANK', '0')) == 0:
    torch.save({'losses': losses, 'flat': eng.flat.cpu(), 'grads': eng.grads.cpu(), 'world': m.dp.world, 'scale': m.optimizer.grad_scale}, sys.argv[1])
if world > 1:
    dist.barrier(); dist.destroy_process_group()
'''


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs >= 2 GPUs on the node (the gpurun box exposes one)')
def test_two_rank_rccl_fit_loop_equals_single_rank(tmp_path):
    """Real RCCL over xGMI: torchrun 2 ranks of _fit_loop on the SAME batch -> summed gradients are exactly 2 x, grad_scale = 1/2,
    bucketed all-reduce overlapped with backward -> weights bit-identical to the 1-rank run."""
    script = tmp_path / 'rccl_worker.py'
    script.write_text(_RCCL_WORKER % {'root': ROOT})
    outs = {}
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', SALT_BN_FIN='0', SALT_SE_SHARDS='0')      # fixed-order sums: bit-equality is by construction
    r = subprocess.run([sys.executable, str(script), str(tmp_path / 'one.pt')], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    outs['one'] = torch.load(tmp_path / 'one.pt')
    port = str(29700 + os.getpid() % 200)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', port, str(script), str(tmp_path / 'two.pt')], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    outs['two'] = torch.load(tmp_path / 'two.pt')
    assert outs['two']['world'] == 2 and outs['two']['scale'] == 0.5
    assert outs['one']['losses'] == outs['two']['losses']
    assert torch.equal(outs['two']['grads'], outs['one']['grads'] * 2)
    assert torch.equal(outs['one']['flat'], outs['two']['flat'])


def test_planar_hypercolumn_equals_interleaved(monkeypatch):
    """The hypercolumn (architectures/unet.py:101-107) as five dense 64-channel planes - the final convolution reads them by plane
    (salt_conv_args.x_plane, conv_ls_kernel), its data gradient writes them (y_plane, conv_ws_kernel), its weight gradient reads them
    (salt_conv_wgrad_args.q_plane), the up-samplings / their adjoints / dec1's scSE each stream one dense plane - against the
    channel-interleaved 320-channel buffer (SALT_NO_PLANAR=1).  Same arithmetic, other addresses: eval logits bit for bit; two train
    steps to the reproducibility of the fp64 statistics shards (last bits of the BatchNorm sums, conftest.deterministic_sums)."""
    results = {}
    for mode in ('planar', 'interleaved'):
        if mode == 'interleaved':
            monkeypatch.setenv('SALT_NO_PLANAR', '1')
        torch.manual_seed(11)
        m = _segmentation_model('UNetResNet', 'lovasz', dtype='bf16', lr=1e-3)
        m._to_device()
        g = torch.Generator().manual_seed(5)
        X = torch.randn(4, 3, 128, 128, generator=g)
        M = (torch.rand(4, 1, 128, 128, generator=g) > 0.6).float()
        Tt = torch.cat([1 - M, M], 1)
        m.model.eval()
        with torch.no_grad():
            y_eval = m.model(X.to(DEV)).float().cpu()
        eng = m.model.engine()
        n_eval = sum(1 for name, _, s in eng.net((4, 3, 128, 128), False).fwd.ops if name == 'conv' and s.x_plane)
        m.model.train()
        ls = [float(m._fit_loop([X, Tt])['sum']) for _ in range(2)]
        torch.cuda.synchronize()
        net = eng.net((4, 3, 128, 128), True)
        n = (sum(1 for name, _, s in net.fwd.ops if name == 'conv' and s.x_plane), sum(1 for name, _, s in net.bwd.ops if name == 'conv' and s.y_plane),
             sum(1 for name, _, s in net.bwd.ops if name == 'conv_wgrad' and s.q_plane), n_eval)
        assert n == ((1, 1, 1, 1) if mode == 'planar' else (0, 0, 0, 0)), n
        results[mode] = (ls, eng.flat.clone(), eng.grads.clone(), y_eval)
    a, b = results['planar'], results['interleaved']
    assert torch.equal(a[3], b[3])
    assert max(abs(x - y) for x, y in zip(a[0], b[0])) <= 1e-5 * max(1.0, abs(b[0][0])), (a[0], b[0])
    gn = float(b[2].norm())
    assert float((a[2] - b[2]).norm()) <= 2e-3 * gn, (float((a[2] - b[2]).norm()), gn)
    assert float((a[1] - b[1]).norm()) <= 1e-4 * float(b[1].norm())


@pytest.mark.parametrize('architecture,dtype', [('UNetResNet', 'bf16'), ('UNetResNet', 'f32'), ('VanillaUNet', 'f32')])
def test_auxiliary_stream_modes_equal_the_plain_step(architecture, dtype, monkeypatch, deterministic_sums):
    """Round 6, both opt-in (measured slower than the default on the C2 step) but shipped and therefore tested:
    * `SALT_ADAM_IN_BWD=1`: `FusedAdam.backward_program` inserts the update of every parameter range whose gradients are final into the
      backward program (stream tag 5: auxiliary stream, behind both queues) and `step()` only runs the remaining ranges.  Same arithmetic
      per element: parameters, both moments and the step counter after four `_fit_loop` steps are bit-identical to the step that runs
      Adam as ONE launch after backward; the reference is `optimizer.step()` after `batch_loss.backward()` (models.py:127-129).
    * `SALT_REDUCE_AUX=1`: the weight-gradient slab reductions on the auxiliary stream (stream tag 4) over two alternating slabs -
      the same kernels on the same values in another queue.
    Also both with the auxiliary stream withheld (`SALT_NO_AUX_STREAM`: the tags fall back to the weight-gradient queue)."""
    shape = (4, 3, 64, 64) if architecture == 'UNetResNet' else (4, 1, 64, 64)
    g = torch.Generator().manual_seed(11)
    X = torch.randn(*shape, generator=g).to(DEV)
    M = (torch.rand(shape[0], 1, shape[2], shape[3], generator=g) > 0.6).float()
    Tg = torch.cat([1 - M, M], 1).to(DEV)
    modes = {'plain': {}, 'adam': {'SALT_ADAM_IN_BWD': '1', 'SALT_ADAM_CHUNK_MB': '8'}, 'reduce': {'SALT_REDUCE_AUX': '1'},
             'both_no_aux': {'SALT_ADAM_IN_BWD': '1', 'SALT_ADAM_CHUNK_MB': '8', 'SALT_REDUCE_AUX': '1', 'SALT_NO_AUX_STREAM': '1'},
             'both': {'SALT_ADAM_IN_BWD': '1', 'SALT_ADAM_CHUNK_MB': '8', 'SALT_REDUCE_AUX': '1'}}
    res = {}
    for mode, env in modes.items():
        for k in ('SALT_ADAM_IN_BWD', 'SALT_ADAM_CHUNK_MB', 'SALT_REDUCE_AUX', 'SALT_NO_AUX_STREAM'):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        torch.manual_seed(3)
        m = _segmentation_model(architecture, dtype=dtype, lr=1e-3)
        m._to_device(); m.model.train()
        losses = [float(m._fit_loop([X, Tg])['sum']) for _ in range(4)]
        torch.cuda.synchronize()
        eng = m.model.engine()
        net = eng.net(shape, True)
        early = m.optimizer._bwd_progs.get(net)
        res[mode] = (losses, eng.flat.clone(), m.optimizer.exp_avg.clone(), m.optimizer.exp_avg_sq.clone(), int(m.optimizer.step_t.item()),
                     m.optimizer.steps, None if early is None or early[1] is None else (len(early[1][0]) - len(net.bwd), len(early[1][1])),
                     net.bwd.streams.count(4))
    assert res['plain'][6] is None and res['plain'][7] == 0 and res['reduce'][7] > 4
    if architecture == 'UNetResNet':
        assert res['adam'][6] is not None and res['adam'][6][0] >= 2, res['adam'][6]     # tick + at least one early range
    for mode in modes:
        assert res[mode][0] == res['plain'][0], (mode, res[mode][0], res['plain'][0])
        for i in (1, 2, 3):
            assert torch.equal(res[mode][i], res['plain'][i]), (mode, i)
        assert res[mode][4] == 4 and res[mode][5] == 4
