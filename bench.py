#!/usr/bin/env python
"""bench.py — training images/s (+ val IoU) of the ResNet34 hypercolumn U-Net on synthetic 101x101 salt tiles.

    python bench.py --gpus N --steps K --warmup W [--dtype bf16|f32] [--workload r34_hyper|ternaus34|vanilla] [--batch 32]
    N > 1 works both ways: started by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or as a plain
    `python bench.py --gpus N`, which re-launches itself through torch.distributed.run (one rank per GPU, 127.0.0.1 rendezvous).

A "step" is one `_fit_loop`-equivalent pass of the hot path (reference models.py:105-136) over one resident minibatch:
pack weights -> forward -> Lovasz hinge -> backward (bucketed RCCL all-reduce on a side stream when N > 1) -> fused Adam.
W untimed warm-up steps, then EXACTLY K timed steps bracketed by barrier + torch.cuda.synchronize(), MAX over ranks.
Rank 0 prints ONE JSON line.  Inputs are resident in HBM before the timed region (no PCIe inside it).

Extra objects in the line:
  roofline     the kernel that takes the most time in a step, measured live with HIP event pairs on the launch stream
               (salt_program_run_timed): achieved = algorithmic FLOPs of its launches / their summed duration.
  cpu_baseline the oracle (plain-PyTorch CPU restatement, kind "port") timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL reads its NCCL_DEBUG* variables once, when librccl initialises its logger: set them before torch is imported (round 2 set them
# next to init_process_group and the file never appeared).  %p = pid, one file per rank; rccl_summary() quotes rank 0's.
if __name__ == '__main__' and 'NCCL_DEBUG_FILE' not in os.environ:
    if os.environ.get('NCCL_DEBUG', '').upper() not in ('INFO', 'TRACE'):     # (the image may preset WARN: the summary needs the INFO lines)
        os.environ['NCCL_DEBUG'] = 'INFO'
    os.environ.setdefault('NCCL_DEBUG_SUBSYS', 'INIT,GRAPH,TUNING')
    os.environ['NCCL_DEBUG_FILE'] = '/tmp/salt_rccl_%p.log'

import numpy as np
import torch
import torch.distributed as dist

MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f32': 157.3}      # dense peaks, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


# ----------------------------------------------------------------------------- synthetic data
def synth_tiles(n, seed, size=101):
    """Seismic-like 101x101 gray tiles + salt masks: smoothed noise texture, ellipse/half-plane bodies, ~38 % empty."""
    r = np.random.RandomState(seed)
    img = r.randn(n, size + 8, size + 8).astype(np.float32)
    k = np.array([1, 4, 6, 4, 1], np.float32) / 16
    for _ in range(2):
        img = sum(k[i] * np.roll(img, i - 2, axis=1) for i in range(5))
        img = sum(k[i] * np.roll(img, i - 2, axis=2) for i in range(5))
    img = img[:, 4:-4, 4:-4]
    img = (img - img.mean()) / (img.std() + 1e-6)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    mask = np.zeros((n, size, size), np.float32)
    for i in range(n):
        u = r.rand()
        if u < 0.38:
            continue
        if u < 0.7:
            cy, cx = r.uniform(0, size, 2)
            ry, rx = r.uniform(12, 70, 2)
            mask[i] = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1).astype(np.float32)
        else:
            a = r.uniform(0, 2 * np.pi)
            off = r.uniform(20, 80)
            mask[i] = ((yy * np.sin(a) + xx * np.cos(a)) > off).astype(np.float32)
    img = img * 0.18 + 0.45 + 0.22 * mask * (0.6 + 0.4 * r.rand(n, 1, 1).astype(np.float32))     # intensity shift inside salt
    return np.clip(img, 0, 1), mask


def preprocess(img, mask, train, channels):
    """Reference loader geometry: train = resize 101->102 + edge-pad 13 -> 128 (neptune.yaml:22-26, augmentation.py:79-85);
    inference = edge-pad to 128 with the (13,14,14,13) split (augmentation.py:247-284).  Normalise with the ImageNet
    statistics; 3-channel mode applies AddDepthChannels (loaders.py:607-612, utils.py:494-500)."""
    x = torch.from_numpy(img)[:, None]
    m = torch.from_numpy(mask)[:, None]
    if train:
        # iaa.Scale's default in the reference's imgaug 0.2.5 is cubic (cv2.INTER_CUBIC), on the uint8 tile and the uint8 mask alike
        x = torch.clamp(torch.floor(torch.nn.functional.interpolate(x, size=(102, 102), mode='bicubic', align_corners=False) * 255 + 0.5), 0, 255) / 255
        m = torch.clamp(torch.floor(torch.nn.functional.interpolate(m, size=(102, 102), mode='bicubic', align_corners=False) + 0.5), 0, 1)
        pad = (13, 13, 13, 13)
    else:
        pad = (14, 13, 13, 14)                      # (left, right, top, bottom)
    x = torch.nn.functional.pad(x, pad, mode='replicate')
    m = torch.nn.functional.pad(m, pad, mode='replicate')
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    if channels == 1:
        x = (x - mean[0]) / std[0]
    else:
        x = torch.cat([(x - mean[c]) / std[c] for c in range(3)], 1)
        h = x.shape[2]
        x[:, 1] = torch.linspace(0, 1, h)[None, :, None]
        x[:, 2] = x[:, 0] * x[:, 1]
    t = torch.cat([1 - m, m], 1)
    return x.contiguous(), t.contiguous()


def iou_metric(pred, gt):
    """mean IoU with the reference's empty-mask conventions (metrics.py:21-34,53-59) on 101x101 crops."""
    vals = []
    for p, g in zip(pred, gt):
        if not g.any() and not p.any():
            vals.append(1.0)
        elif g.any() != p.any():
            vals.append(0.0)
        else:
            vals.append(float((p & g).sum()) / float((p | g).sum()))
    return float(np.mean(vals))


# ----------------------------------------------------------------------------- conv FLOP accounting (algorithmic)
def op_flops(name, s):
    """Algorithmic FLOPs (2*MAC) of one native operator launch, from its argument struct; 0 for bandwidth ops."""
    if name == 'conv':
        return 2.0 * s.x.B * s.OH * s.OW * s.y.C * s.x.C * s.ntaps
    if name == 'conv_wgrad':
        return 2.0 * s.p.B * s.p.H * s.p.W * s.p.C * s.q.C * s.ntaps
    if name == 'conv_first':
        return 2.0 * s.B * s.y.H * s.y.W * s.y.C * s.Cin * s.K * s.K
    if name == 'conv_first_wgrad':
        return 2.0 * s.B * s.dy.H * s.dy.W * s.dy.C * s.Cin * s.K * s.K
    return 0.0


CONV_SYMBOL = {1: 'conv_mfma_kernel', 2: 'conv_mfma_kernel', 3: 'conv_mfma_kernel', 4: 'conv_mfma_kernel', 5: 'conv_mfma_kernel',
               6: 'conv_glds_kernel', 7: 'conv_glds_kernel', 8: 'conv_glds_kernel', 9: 'conv_ws_kernel', 10: 'conv_ls_kernel',
               11: 'conv1x1_ls_kernel', 12: 'conv_thin_kernel', 13: 'conv_stem16_kernel'}


def op_symbol(name, s):
    """kernel symbol a convolution / weight-gradient launch runs on (salt_conv_kernel_id / salt_conv_wgrad_kernel_id); None for the rest"""
    import ctypes
    from salt_amd._abi import lib
    if name == 'conv':
        return CONV_SYMBOL.get(int(lib.salt_conv_kernel_id(ctypes.byref(s))), 'conv_mfma_kernel')
    if name == 'conv_wgrad':
        return {1: 'conv_wgrad_ls_kernel', 2: 'conv_wgrad_thin_kernel', 3: 'conv_wgrad_fast_kernel', 4: 'conv_wgrad_fast32_kernel',
                5: 'conv_wgrad_kernel', 6: 'conv_wgrad_full_kernel'}.get(int(lib.salt_conv_wgrad_kernel_id(ctypes.byref(s))), 'conv_wgrad_kernel')
    return None


def _vb(v, es):
    """bytes of one pass over the channels of a salt_view (0 for a NULL view)"""
    return float(v.B) * v.H * v.W * v.C * es if v.p else 0.0


def op_bytes(name, s, es):
    """Algorithmic HBM bytes of one operator launch from its argument struct: every operand tensor read once and every result written
    once (convolutions: input pixels once, packed weights once, output once - twice when accumulating).  0 for operators without a rule."""
    if name == 'conv':
        return es * (s.x.B * s.x.H * s.x.W * s.x.C + s.ntaps * s.x.C * s.y.C + s.x.B * s.OH * s.OW * s.y.C * (2 if s.accumulate else 1))
    if name == 'conv_wgrad':
        return _vb(s.p, es) + _vb(s.q, es) + 4.0 * s.nsplit * s.ntaps * s.p.C * s.q.C
    if name == 'wgrad_reduce':
        return 4.0 * (s.nsplit + 1) * s.ntaps * s.Ca * s.Cb
    if name == 'affine_act':
        return _vb(s.y, es) + _vb(s.res, es) + _vb(s.a, es)
    if name == 'bn_bwd':          # reduction pass (da, y[, a]) unless the producer carried the sums, then the apply pass (da, y[, a] -> dy[, dres])
        rd = _vb(s.da, es) + _vb(s.y, es) + _vb(s.a, es)
        return (1 if s.partials_ready else 2) * rd + _vb(s.dy, es) + _vb(s.dres, es) * (2 if s.accumulate_dres else 1)
    if name == 'bilinear':
        return _vb(s.x, es) * (2 if (s.backward and s.accumulate) else 1) + _vb(s.y, es)
    if name == 'hyper_stencil':   # forward: partial sum + every level's z read, y written; adjoint: y read, every level's dz written
        return sum(_vb(s.z[k], es) for k in range(s.nlev)) + _vb(s.y, es) + (_vb(s.y_in, es) if not s.backward else 0.0)
    if name in ('maxpool2', 'maxpool3s2', 'avgpool2'):
        return _vb(s.x, es) + _vb(s.y, es)
    if name in ('maxpool2_bwd', 'maxpool3s2_bwd'):
        return _vb(s.x, es) + _vb(s.dy, es) + _vb(s.dx, es) * (2 if s.accumulate else 1)
    if name == 'scse':            # statistics pass + apply pass over x, y written once
        return 2 * _vb(s.x, es) + _vb(s.y, es)
    if name == 'scse_bwd':        # (x, y, dy) read twice (sums, then apply), dx written
        return 2 * (_vb(s.x, es) + _vb(s.y, es) + _vb(s.dy, es)) + _vb(s.dx, es) * (2 if s.accumulate else 1)
    if name == 'head1x1':
        return _vb(s.x, es) + 4.0 * s.x.B * s.x.H * s.x.W * s.Cout
    if name == 'head1x1_bwd':
        return _vb(s.x, es) + 4.0 * s.x.B * s.x.H * s.x.W * s.Cout + _vb(s.dx, es) * (2 if s.accumulate else 1)
    if name == 'add':
        return _vb(s.a, es) + _vb(s.b, es) + _vb(s.y, es) * (2 if s.accumulate else 1)
    if name == 'relu_bwd':
        return _vb(s.da, es) + _vb(s.a, es) + _vb(s.dy, es)
    if name == 'lovasz_hinge':    # logits + targets read, gradient written (fp32 NCHW), 4 radix passes over (key, payload) pairs
        return 0.0
    return 0.0


def _latest_profile(name):
    """Newest committed evidence file of that kind (rounds are re-measured with tools/rNN_evidence.sh): the highest round tag wins."""
    import glob
    import re
    best = None
    for path in glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]*_' + name)):
        m = re.match(r'r(\d+)([a-z]?)_' + re.escape(name) + '$', os.path.basename(path))
        if m:
            key = (int(m.group(1)), m.group(2))
            if best is None or key > best[0]:
                best = (key, 'profiles/' + os.path.basename(path))
    return best[1] if best else 'profiles/r06_%s' % name


PMC_FILE = _latest_profile('pmc_traffic.json')
INSTEP_FILE = _latest_profile('instep.json')    # tools/prof_instep.py: kernel time per class INSIDE the step (rocprofv3 kernel trace)


def pmc_commit():
    """Commit the PMC file was measured at (recorded inside it by tools/pmc_traffic.py), so that `roofline.traffic` names its build."""
    try:
        return json.load(open(os.path.join(ROOT, PMC_FILE))).get('commit', 'unrecorded')
    except (OSError, ValueError):
        return None


def pmc_traffic(kernel):
    """HBM bytes per launch, averaged over the kernels named in the tuple `kernel`, from the committed rocprofv3 --pmc passes (PMC_FILE;
    tools/pmc_traffic.sh regenerates it - PMC counters cannot be read from inside the timed process).  None when the file is absent."""
    for path in (os.path.join(ROOT, PMC_FILE), os.path.join(ROOT, 'profiles', 'r04_pmc_traffic.json')):
        try:
            ks = json.load(open(path))['kernels']
            sel = [v for k, v in ks.items() if k in kernel]
            n = sum(v.get('launches_per_step', 1) for v in sel)
            tot = sum((v['read_MB_per_launch'] + v['write_MB_per_launch']) * v.get('launches_per_step', 1) for v in sel)
            if n:
                return round(tot / n * 1e6)
        except (OSError, KeyError, ValueError):
            continue
    return None


CPU_LEG_THREADS = 16        # a fixed, modest thread count: GPU boxes advertise 256 logical CPUs but oversubscribing them makes
CPU_LEG_BATCH = 8           # the torch CPU kernels orders of magnitude slower; SURVEY.md 8(d): batch 8, median of 10 steps after 2 warm-ups
CPU_LEG_STEPS, CPU_LEG_WARMUP = int(os.environ.get('SALT_CPU_LEG_STEPS', '10')), int(os.environ.get('SALT_CPU_LEG_WARMUP', '2'))
CPU_LEG_TIMEOUT_S = 150


def cpu_baseline_leg(arch_name, loss_name, channels, threads=0, batch=0):
    """Child process: time the oracle (oracle/, plain PyTorch CPU fp32) on a bounded sample of the same workload."""
    threads = threads or CPU_LEG_THREADS
    torch.set_num_threads(threads)
    from oracle import nets as ON, specs as OS, losses as OL
    spec = OS.SPECS[arch_name]() if arch_name != 'VanillaUNet' else OS.spec_vanilla_unet()
    sd = OS.init_state(spec, seed=0)
    keys = OS.trainable_keys(spec)
    for k in keys:
        sd[k].requires_grad_(True)
    params = [sd[k] for k in keys]
    m_ = [torch.zeros_like(p) for p in params]
    v_ = [torch.zeros_like(p) for p in params]
    cb = batch or CPU_LEG_BATCH
    img, msk = synth_tiles(cb, seed=1234)
    xc, tc = preprocess(img, msk, True, channels)
    times = []
    t_start = time.perf_counter()
    for it in range(CPU_LEG_WARMUP + CPU_LEG_STEPS):
        t1 = time.perf_counter()
        for p in params:
            p.grad = None
        o = ON.FORWARDS[arch_name](sd, xc, True)
        l = OL.LOSSES[loss_name](o, tc)
        l.backward()
        with torch.no_grad():
            OL.adam_l2_step([p.data for p in params], [p.grad for p in params], m_, v_, it + 1)
        times.append(time.perf_counter() - t1)
        if time.perf_counter() - t_start > 100 and len(times) >= CPU_LEG_WARMUP + 2:
            break
    timed = times[CPU_LEG_WARMUP:]
    med = float(np.median(timed))
    print(json.dumps({'value': round(cb / med, 2), 'unit': 'images/s', 'cores': threads, 'host_logical_cpus': os.cpu_count(), 'kind': 'port',
                      'sample': 'oracle (plain PyTorch CPU fp32, %d threads of the %d logical CPUs of this host) %s, %s loss, Adam; batch %d, '
                                'median of %d steps after %d warm-ups' % (torch.get_num_threads(), os.cpu_count(), arch_name, loss_name, cb,
                                                                         len(timed), CPU_LEG_WARMUP)}))


def cpu_baseline_subprocess(loss, arch_name, threads=0, batch=0):
    """Run the CPU leg in a child with a hard time limit so that the default bench run always finishes in minutes."""
    import subprocess
    threads = min(threads or CPU_LEG_THREADS, os.cpu_count() or 1)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-leg', arch_name, '--loss', loss, '--cpu-threads', str(threads), '--cpu-batch', str(batch or CPU_LEG_BATCH)]
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=CPU_LEG_TIMEOUT_S)
        line = [x for x in r.stdout.splitlines() if x.startswith('{')]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        note = 'cpu leg failed: ' + (r.stderr.strip().splitlines() or ['?'])[-1][:200]
    except subprocess.TimeoutExpired:
        note = 'cpu leg exceeded %d s' % CPU_LEG_TIMEOUT_S
    return {'value': None, 'unit': 'images/s', 'cores': threads, 'host_logical_cpus': os.cpu_count(), 'kind': 'port', 'sample': note}


DP_LEG_TIMEOUT_S = 300


def dp_leg(workload, dtype, B, loss):
    """Child mode (--dp-leg): the headline step through parallel.DataParallel.backward with a 1-rank RCCL communicator against the plain
    step, alternated inside this one fresh process (2 x 20 steps each), plus the exposed all-reduce time and the per-bucket timeline."""
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    os.environ['SALT_FORCE_DP_PATH'] = '1'
    init_rccl(0)
    import salt_amd  # noqa: F401
    m2, b2, _, _ = train_config(workload, dtype, B, loss, 4, 6, dev)

    def timed(dp_on, n=20):
        if dp_on:
            os.environ['SALT_FORCE_DP_PATH'] = '1'
        else:
            os.environ.pop('SALT_FORCE_DP_PATH', None)
        m2.dp.measure = False
        for i in range(3):
            m2._fit_loop(list(b2[i % len(b2)]))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            m2._fit_loop(list(b2[i % len(b2)]))
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n
    ms_plain, ms_dp = [], []
    for _ in range(2):
        ms_plain.append(timed(False)); ms_dp.append(timed(True))
    os.environ['SALT_FORCE_DP_PATH'] = '1'
    m2.dp.measure = True
    for i in range(4):
        m2._fit_loop(list(b2[i % len(b2)]))
    ex = m2.dp.exposed_allreduce_ms()
    m2.dp.measure = False
    m2.dp.timeline = True
    for i in range(6):
        m2._fit_loop(list(b2[i % len(b2)]))
    res = {'config': 'the headline step through the bucketed all-reduce path (parallel.DataParallel.backward) with a 1-rank RCCL communicator, '
                     'alternated with the plain step in one fresh process (2 x 20 steps each)',
           'ms_per_step': round(min(ms_dp), 3), 'ms_per_step_plain_same_process': round(min(ms_plain), 3),
           'ms_per_step_all': [round(v, 3) for v in ms_dp], 'ms_per_step_plain_all': [round(v, 3) for v in ms_plain],
           'overhead_frac': round(min(ms_dp) / min(ms_plain) - 1.0, 4), 'images_per_s': round(B / min(ms_dp) * 1e3, 1),
           'buckets': len(list(m2.dp._plans.values())[0]) if m2.dp._plans else 0, 'rccl_info': rccl_summary(6),
           'bucket_mbytes': [round((hi - lo) * 4 / 1e6, 2) for lo, hi, _ in (list(m2.dp._plans.values())[0] if m2.dp._plans else [])],
           'rccl_max_nchannels': RCCL_MAX_NCHANNELS,
           'exposed_allreduce_ms_per_step': round(ex, 4) if ex is not None else None, 'bucket_timeline': m2.dp.bucket_timeline()}
    print(json.dumps(res))
    dist.destroy_process_group()


def dp_leg_subprocess(workload, dtype, B, loss):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--dp-leg', '--workload', workload, '--dtype', dtype, '--batch', str(B), '--loss', loss]
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'SALT_FORCE_DP_PATH'):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=DP_LEG_TIMEOUT_S)
        line = [x for x in r.stdout.splitlines() if x.startswith('{')]
        if r.returncode == 0 and line:
            return json.loads(line[-1])
        return {'error': 'dp leg failed: ' + (r.stderr.strip().splitlines() or ['?'])[-1][:200]}
    except subprocess.TimeoutExpired:
        return {'error': 'dp leg exceeded %d s' % DP_LEG_TIMEOUT_S}


WORKLOADS = {'r34_hyper': ('UNetResNet', 3, 'architectures.unet.UNetResNet(34, hypercolumn)'),
             'ternaus34': ('TernausUNetResNet', 3, 'unet_models.UNetResNet(34, deconv)'),
             'vanilla': ('VanillaUNet', 1, 'vanilla 4-level U-Net (16 filters, 1 channel)')}


def conv_roofline(model, B, channels, dtype, loss, reps):
    """Event pair around every operator of one step (salt_program_run_timed): the operator with the largest share and its
    algorithmic FLOP rate.  Returns (roofline dict, op table, serial ms, side-stream ms, algorithmic FLOPs per step)."""
    eng = model.model.engine()
    net = eng.net((B, channels, 128, 128), True)
    groups = {}
    symbols = {}                                     # kernel symbol -> [ms, flops, launches] over the convolution / weight-gradient launches
    for _ in range(reps):
        eng.refresh(True)
        torch.cuda.synchronize()                     # the data-gradient weight packs run on the side stream
        for prog in (net.fwd, net.loss_program(loss, 1.0), net.bwd):
            for name, s, ms in prog.run_timed():
                g = groups.setdefault(name, [0.0, 0.0, 0, 0.0])
                g[0] += ms; g[1] += op_flops(name, s); g[2] += 1
                sym = op_symbol(name, s)
                if sym:
                    y = symbols.setdefault(sym, [0.0, 0.0, 0])
                    y[0] += ms; y[1] += op_flops(name, s); y[2] += 1
                try:
                    g[3] += op_bytes(name, s, 2 if dtype == 'bf16' else 4)
                except AttributeError:
                    pass
    total_ms = sum(g[0] for g in groups.values()) / reps
    dn, (dms, dfl, dcnt, dby) = max(groups.items(), key=lambda kv: kv[1][0])
    peak = MFMA_PEAK_TFLOPS[dtype]
    ach = dfl / (dms * 1e-3) / 1e12 if dms > 0 else 0.0
    kern = {'conv': 'conv_ws_kernel + conv_ls_kernel + conv1x1_ls_kernel + conv_mfma_kernel + conv_glds_kernel (fwd + dgrad launches)', 'conv_wgrad': 'conv_wgrad_ls_kernel + conv_wgrad_fast_kernel + conv_wgrad_kernel'}.get(dn, dn)
    roof = {'kernel': kern, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(ach / peak, 4),
            'algorithmic_bytes_per_launch': round(dby / dcnt) if dcnt else None, 'launches_per_step': dcnt // reps,
            'avg_launch_us': round(1e3 * dms / dcnt, 2), 'share_of_step': round(dms / reps / total_ms, 3),
            # FLOPs the launches of this class EXECUTE (2 MAC per tap of every launch, from the argument structs).  Since round 5 the
            # x4 / x8 / x16 hypercolumn levels are contracted at their own resolution (salt_hyper_stencil), so this is LESS than the
            # reference network's algorithmic count for the same step (`flops` in the line has both); `frac` is priced on what ran
            'executed_gflop_per_step': round(dfl / reps / 1e9, 2)}
    if symbols:
        # the largest SINGLE kernel symbol of the step: `kernel` above names a class of five symbols; this is the one
        # symbol with the most time, its own executed FLOP rate and fraction of the dense MFMA peak
        sn, (sms, sfl, scnt) = max(symbols.items(), key=lambda kv: kv[1][0])
        roof['kernel_symbol'] = {'symbol': sn, 'launches_per_step': scnt // reps, 'ms_per_step': round(sms / reps, 3),
                                 'achieved': round(sfl / (sms * 1e-3) / 1e12, 1), 'unit': 'TFLOP/s', 'peak': peak,
                                 'frac': round(sfl / (sms * 1e-3) / 1e12 / peak, 4), 'share_of_step': round(sms / reps / total_ms, 3)}
        roof['by_symbol'] = {k: {'launches_per_step': v[2] // reps, 'ms_per_step': round(v[0] / reps, 3),
                                 'achieved_tflops': round(v[1] / (v[0] * 1e-3) / 1e12, 1), 'frac': round(v[1] / (v[0] * 1e-3) / 1e12 / peak, 4)}
                             for k, v in sorted(symbols.items(), key=lambda kv: -kv[1][0])}
    ops = {k: round(v[0] / reps, 3) for k, v in sorted(groups.items(), key=lambda kv: -kv[1][0])[:8]}
    side = sum(v[0] for k, v in groups.items() if k in ('conv_wgrad', 'wgrad_reduce', 'wgrad_reduce_batched', 'conv_first_wgrad', 'stem_grad_unfold')) / reps
    # per kernel class: matrix kernels against the dense MFMA peak of the compute dtype, streaming kernels as
    # algorithmic GB/s against the HBM peak; every launch timed alone with its own HIP event pair
    by_class = {}
    for k, (ms, fl, cnt, by) in sorted(groups.items(), key=lambda kv: -kv[1][0]):
        if ms <= 0 or ms / reps < 0.02:
            continue
        e = {'launches_per_step': cnt // reps, 'ms_per_step': round(ms / reps, 3)}
        if fl > 0:
            e.update(bound='mfma', achieved=round(fl / (ms * 1e-3) / 1e12, 1), peak=peak, unit='TFLOP/s', frac=round(fl / (ms * 1e-3) / 1e12 / peak, 4))
        elif by > 0:
            e.update(bound='hbm', achieved=round(by / (ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
        else:
            e.update(bound='latency', achieved=None, peak=None, unit=None, frac=None)
        by_class[k] = e
    roof['by_class'] = by_class
    return roof, ops, total_ms, side, sum(g[1] for g in groups.values()) / reps, dn


RANK_MS = []
STEP_MS = []


def train_config(workload, dtype, B, loss, steps, warmup, dev, rank=0, world=1, reps=1):
    """Build the model of one training configuration, run `warmup` + `steps` timed fused steps on resident data; -> (model, batches, s)."""
    from salt_amd.models import SegmentationModel
    arch_name, channels, _ = WORKLOADS[workload]
    arch = {'model_params': {'architecture': arch_name, 'out_channels': 2, 'activation': 'sigmoid', 'loss': loss, 'compute_dtype': dtype},
            'optimizer_params': {'lr': 1e-4}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
    torch.manual_seed(1234)
    model = SegmentationModel(arch, {'epochs': 1}, {})
    model._to_device()
    model.model.train()
    model.dp.broadcast_parameters(model.model)
    pool_batches = 8
    img, msk = synth_tiles(B * pool_batches, seed=1234 + 17 * rank)     # resident pool, different tiles on every rank
    X, T = preprocess(img, msk, True, channels)
    X, T = X.to(dev), T.to(dev)
    batches = [(X[i * B:(i + 1) * B], T[i * B:(i + 1) * B]) for i in range(pool_batches)]
    if os.environ.get('SALT_MAIN_PRIORITY'):                 # A/B: the step's compute stream with a HIP stream priority
        torch.cuda.synchronize()
        torch.cuda.set_stream(torch.cuda.Stream(priority=int(os.environ['SALT_MAIN_PRIORITY'])))
    for i in range(warmup):
        model._fit_loop(list(batches[i % pool_batches]))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    model.dp.measure = model.dp._active()
    t0 = time.perf_counter()
    for i in range(steps):
        loss_v = model._fit_loop(list(batches[(warmup + i) % pool_batches]))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    global RANK_MS, STEP_MS
    RANK_MS = [1e3 * elapsed / steps]
    # SURVEY 8d asks for the MEDIAN step: a second leg behind the contract's timed region, one event per step on the compute stream
    # (rank 0's own steps; >= 50 of them), reported beside the mean of the K timed steps
    n_med = max(50, steps)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_med + 1)]
    evs[0].record()
    for i in range(n_med):
        model._fit_loop(list(batches[(warmup + steps + i) % pool_batches]))
        evs[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    STEP_MS = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n_med))
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(te) for _ in range(world)]
        dist.all_gather(every, te)
        RANK_MS = [round(1e3 * float(t[0]) / steps, 4) for t in every]      # every rank's own clock around the same K steps
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te[0])
    return model, batches, elapsed, float(loss_v['sum'])


def fit_e2e_leg(dev, workload, dtype, B, loss, n_train=3200, n_val=800, epochs=2):
    """The product the way a user runs it (reference models.py:78-103 driven by main.py): SegmentationModel.fit() over HOST uint8
    101x101 tiles + masks (3 200 train / 800 validation, seed 1234) -> one pinned H2D copy per batch -> DevicePreprocessor
    (resize 101 -> 102, edge pad, normalise, depth channels, one-hot target: loaders.py:603-624 on the device) -> fused step, with the
    reference's callback stack: TrainingMonitor, ValidationMonitor (GPU threshold sweep, callbacks.py:499-527), ReduceLROnPlateau,
    ModelCheckpoint, EarlyStopping.  Reports training images/s of the epoch loops (host generator + H2D + preprocessing + step + batch
    callbacks inside the timed region; the validation pass of each epoch is timed separately) and the validation metrics."""
    from salt_amd.models import SegmentationModel
    from salt_amd.input_pipeline import DevicePreprocessor
    from salt_amd.callbacks import Callback
    arch_name, channels, _ = WORKLOADS[workload]
    img, msk = synth_tiles(n_train + n_val, seed=1234)
    img8, msk8 = np.clip(img * 255 + 0.5, 0, 255).astype(np.uint8), msk.astype(np.uint8)
    host = {'train': (img8[:n_train], msk8[:n_train]), 'valid': (img8[n_train:], msk8[n_train:])}      # numpy uint8 [N,101,101] on the host
    pre = {'train': DevicePreprocessor(True, channels), 'valid': DevicePreprocessor(False, channels)}
    host_s = {'train': 0.0, 'valid': 0.0}

    class Batches:
        """re-iterable host generator: shuffled index -> gather on the host into a pinned staging buffer -> async H2D -> device preprocessing"""
        def __init__(self, split, shuffle):
            self.split, self.shuffle, self.epoch = split, shuffle, 0
            X8, M8 = host[split]
            self.n = X8.shape[0] // B
            self.stage = [(torch.empty((B,) + tuple(X8.shape[1:]), dtype=torch.uint8).pin_memory(),
                           torch.empty((B,) + tuple(M8.shape[1:]), dtype=torch.uint8).pin_memory(), torch.cuda.Event()) for _ in range(4)]
            self.stage_np = [(a.numpy(), b_.numpy()) for a, b_, _ in self.stage]       # numpy views of the pinned buffers (np.take gathers into them)

        def __iter__(self):
            X8, M8 = host[self.split]
            order = np.random.RandomState(1234 + self.epoch).permutation(X8.shape[0]) if self.shuffle else np.arange(X8.shape[0])
            self.epoch += 1
            for i in range(self.n):
                t0 = time.perf_counter()
                idx = order[i * B:(i + 1) * B]
                sx, sm, ev = self.stage[i % len(self.stage)]
                ev.synchronize()                               # the copy that last used this staging slot has left it
                nx, nm = self.stage_np[i % len(self.stage)]
                np.take(X8, idx, axis=0, out=nx); np.take(M8, idx, axis=0, out=nm)      # (torch.index_select(out=pinned) took 17 ms per call here)
                xb, mb = sx.to(dev, non_blocking=True), sm.to(dev, non_blocking=True)
                ev.record()
                out = list(pre[self.split](xb, mb))
                host_s[self.split] += time.perf_counter() - t0
                yield out

    class EpochTimer(Callback):
        """first in the callback list: its on_epoch_end runs BEFORE the validation pass of the monitors behind it"""
        def __init__(self):
            super().__init__()
            self.train_s, self.t0 = [], None

        def on_epoch_begin(self, *a, **k):
            torch.cuda.synchronize(); self.t0 = time.perf_counter()

        def on_epoch_end(self, *a, **k):
            torch.cuda.synchronize(); self.train_s.append(time.perf_counter() - self.t0)
            self.epoch_id += 1

    timer = EpochTimer()
    ckpt = '/tmp/salt_bench_ckpt_%d/best.torch' % os.getpid()
    cbs = {'callbacks': [timer], 'training_monitor': {'batch_every': 0, 'epoch_every': 1}, 'validation_monitor': {'epoch_every': 1},
           'model_checkpoint': {'filepath': ckpt, 'metric_name': 'iout', 'epoch_every': 1, 'minimize': False},
           'reduce_lr_on_plateau_scheduler': {'metric_name': 'iout', 'minimize': False, 'reduce_factor': 0.5, 'reduce_patience': 10, 'min_lr': 1e-6},
           'early_stopping': {'metric_name': 'iout', 'patience': 100, 'minimize': False}}
    arch = {'model_params': {'architecture': arch_name, 'out_channels': 2, 'activation': 'sigmoid', 'loss': loss, 'compute_dtype': dtype},
            'optimizer_params': {'lr': 1e-4}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
    torch.manual_seed(1234)
    model = SegmentationModel(arch, {'epochs': epochs}, cbs)
    tr, va = Batches('train', True), Batches('valid', False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.fit((tr, tr.n - 1), (va, va.n - 1))
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    last = max(model.validation_loss)
    val = model.validation_loss[last]
    steps = tr.n * epochs
    # the first epoch builds the programs (one-time: graph construction, kernel attributes): the steady-state figure is the LAST epoch's
    out = {'what': 'SegmentationModel.fit(): host uint8 tiles -> pinned H2D -> DevicePreprocessor -> fused step, callbacks: training / validation '
                   'monitor (GPU threshold sweep), ReduceLROnPlateau, ModelCheckpoint, EarlyStopping',
           'n_train': n_train, 'n_val': n_val, 'batch': B, 'epochs': epochs, 'steps': steps, 'dtype': dtype,
           'images_per_s': round(tr.n * B / timer.train_s[-1], 1), 'ms_per_step': round(1e3 * timer.train_s[-1] / tr.n, 3),
           'images_per_s_first_epoch_incl_program_build': round(tr.n * B / timer.train_s[0], 1),
           'host_ms_per_step': round(1e3 * host_s['train'] / steps, 3),
           'validation_s_per_epoch': round((wall - sum(timer.train_s)) / epochs, 3),
           'val_iou': round(float(val['iou']), 4), 'val_iout': round(float(val['iout']), 4), 'val_loss': round(float(val['sum']), 5),
           'best_threshold': round(float(model.best_threshold), 4), 'lr_after': model.optimizer.param_groups[0]['lr'],
           'checkpoint_written': os.path.exists(ckpt), 'wall_s': round(wall, 2)}
    try:
        os.remove(ckpt); os.rmdir(os.path.dirname(ckpt))
    except OSError:
        pass
    del model
    torch.cuda.empty_cache()
    return out


# ----------------------------------------------------------------------------- host side of an 8-rank node, on a 1-GPU box
def _core_share(k, n):
    """the k-th of n equal slices of the cores of GPU 0's NUMA node (what parallel.pin_rank_threads gives rank k when n ranks share a node)"""
    from salt_amd.parallel import gpu_numa_node, _cpulist
    allowed = sorted(os.sched_getaffinity(0))
    node = gpu_numa_node(0) if torch.cuda.is_available() else None
    cpus = allowed
    if node is not None:
        try:
            cpus = [c for c in _cpulist(open('/sys/devices/system/node/node%d/cpulist' % node).read()) if c in set(allowed)] or allowed
        except OSError:
            pass
    per = max(len(cpus) // n, 1)
    return node, cpus[k * per:(k + 1) * per] or cpus


def host_load_child(k, B):
    """CPU-only stand-in for the host work of ONE more rank of the node: the loader thread's gather of a shuffled uint8 batch into a staging
    buffer + ~3.4 ms of launch-thread work per 5 ms step (a busy loop of ctypes calls into the library, which is what the executor's host side
    is made of), on core share k of 8; runs until killed."""
    import ctypes
    import threading
    _, mine = _core_share(k, 8)
    os.sched_setaffinity(0, mine)
    import salt_amd  # noqa: F401
    from salt_amd._abi import lib
    img, msk = synth_tiles(1600, seed=99 + k)
    X8, M8 = np.clip(img * 255 + 0.5, 0, 255).astype(np.uint8), msk.astype(np.uint8)
    sx, sm = np.empty((B,) + X8.shape[1:], np.uint8), np.empty((B,) + M8.shape[1:], np.uint8)
    r = np.random.RandomState(k)

    def loader():
        while True:
            t0 = time.perf_counter()
            idx = r.randint(0, X8.shape[0], B)
            np.take(X8, idx, axis=0, out=sx); np.take(M8, idx, axis=0, out=sm)
            time.sleep(max(0.0, 0.005 - (time.perf_counter() - t0)))
    threading.Thread(target=loader, daemon=True).start()
    sys.stdout.write('ready\n'); sys.stdout.flush()
    while True:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.0034:
            lib.salt_abi_version()
        time.sleep(max(0.0, 0.005 - (time.perf_counter() - t0)))


def host_contention(workload, dtype, B, loss):
    """fit_e2e twice on core share 0 of 8 of the GPU's NUMA node: alone, then beside 7 host_load_child processes on shares 1..7 - the host
    side of an 8-rank node as far as a 1-GPU box can show it.  Prints one JSON object."""
    import subprocess
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    node, mine = _core_share(0, 8)
    os.sched_setaffinity(0, mine)
    import salt_amd  # noqa: F401
    alone = fit_e2e_leg(dev, workload, dtype, B, loss)
    kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--host-load', str(k), '--batch', str(B)], stdout=subprocess.PIPE, text=True)
            for k in range(1, 8)]
    try:
        for p in kids:
            p.stdout.readline()                       # 'ready': tiles synthesised, loops running
        time.sleep(1.0)
        loaded = fit_e2e_leg(dev, workload, dtype, B, loss)
    finally:
        for p in kids:
            p.kill()
    keep = ('images_per_s', 'ms_per_step', 'host_ms_per_step', 'val_iou')
    print(json.dumps({'what': 'SegmentationModel.fit() (fit_e2e) pinned to core share 0 of 8 of the GPU\'s NUMA node, alone and beside 7 CPU-only '
                              'stand-ins for the other ranks\' host work (loader gather + 3.4 ms launch-thread busy loop per 5 ms) on shares 1..7',
                      'numa_node': node, 'cores_per_rank': len(mine), 'logical_cpus': os.cpu_count(),
                      'alone': {k: alone[k] for k in keep}, 'with_7_host_loads': {k: loaded[k] for k in keep},
                      'ratio': round(loaded['images_per_s'] / alone['images_per_s'], 4)}))


def extra_configs(dev, steps=12, warmup=4):
    """The other BASELINE.json configurations, measured in the same run (each a few hundred milliseconds): C1 vanilla fp32,
    ResNet34 fp32, C3's per-GPU shape (batch 64), C4 inference with 4-flip TTA."""
    out = {}

    def one(tag, workload, dtype, B, note, steps=steps, warmup=warmup):
        model, batches, elapsed, _ = train_config(workload, dtype, B, 'lovasz', steps, warmup, dev)
        _, channels, _ = WORKLOADS[workload]
        net = model.model.engine().net((B, channels, 128, 128), True)
        net.x.copy_(batches[0][0]); net.target.copy_(batches[0][1])
        roof, _, _, _, fl, _ = conv_roofline(model, B, channels, dtype, 'lovasz', 1)
        roof.pop('by_class', None)
        med = STEP_MS[len(STEP_MS) // 2] if STEP_MS else None      # train_config's second leg: >= 50 further steps, one event per step
        out[tag] = {'config': note, 'images_per_s': round(B * steps / elapsed, 1), 'ms_per_step': round(1e3 * elapsed / steps, 3),
                    'ms_per_step_median_of_%d' % len(STEP_MS): round(med, 3) if med else None, 'dtype': dtype,
                    'steps': steps, 'warmup': warmup, 'step_tflops': round(fl / (elapsed / steps) / 1e12, 1),
                    'conv_tflops': roof['achieved'], 'conv_frac_of_mfma_peak': roof['frac'], 'mfma_peak_tflops': roof['peak']}
        del model, batches, net
        torch.cuda.empty_cache()
    # (timed regions of >= 0.25 s: the 12-step region of the 3 ms vanilla step once read 4.0 ms where every other run reads 3.1 - one
    # host hiccup of 10 ms is 30 % of 37 ms)
    one('C1_vanilla_f32_b32', 'vanilla', 'f32', 32, 'vanilla 4-level U-Net, 101x101 pad->128, batch 32, fp32, training step (Lovasz, Adam)', steps=80, warmup=8)
    one('R34_hyper_f32_b32', 'r34_hyper', 'f32', 32, 'ResNet34 hypercolumn U-Net, 128x128, batch 32, fp32 (exact-f32 MFMA), training step')
    one('C3_R34_hyper_bf16_b64', 'r34_hyper', 'bf16', 64, 'ResNet34 hypercolumn U-Net + Lovasz hinge, 128x128, batch 64 per GPU, bf16, training step', steps=30, warmup=6)
    # C4: ResNet152 hypercolumn U-Net, 256x256, batch 16, 4-flip TTA inference (flip -> one forward of 64 -> sigmoid -> inverse flip -> mean
    # -> centre crop -> threshold), bf16
    from salt_amd import architectures as A, inference as I
    torch.manual_seed(0)
    net = A.UNetResNet(152, 2, use_hypercolumn=True, dropout_2d=0.0, pretrained=False)
    net.set_compute_dtype('bf16')
    net.to(dev).eval()
    X = torch.randn(16, 3, 256, 256, device=dev)

    def step():
        return I.crop_threshold(I.predict_tta(net, X, True, True, depth_channels=False), (202, 202), 0.5, cls=1)
    step(); step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n4 = 4
    for _ in range(n4):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n4
    cnet = net.engine().net((64, 3, 256, 256), False)
    fl = ms = 0.0
    for name, s_, m_ in cnet.fwd.run_timed():
        if name == 'conv':
            fl += op_flops(name, s_); ms += m_
    out['C4_R152_hyper_bf16_256_b16_tta4'] = {
        'config': 'ResNet152 hypercolumn U-Net, 256x256, batch 16 per GPU, 4-flip TTA inference incl. crop + threshold, bf16',
        'images_per_s': round(16 / dt, 1), 'forward_images_per_s': round(64 / dt, 1), 'ms_per_batch': round(dt * 1e3, 2), 'steps': n4, 'warmup': 2,
        'whole_pass_tflops': round(2 * 384.7e9 * 64 / dt / 1e12, 1), 'conv_tflops': round(fl / (ms * 1e-3) / 1e12, 1) if ms else None,
        'conv_frac_of_mfma_peak': round(fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS['bf16'], 4) if ms else None, 'dtype': 'bf16'}
    del net, cnet, X
    torch.cuda.empty_cache()
    return out


RCCL_LOG = '/tmp/salt_rccl_%d.log' % os.getpid()
RCCL_MAX_NCHANNELS = None


def init_rccl(rank):
    """Bring up the RCCL process group.  RCCL prints a version banner on stdout when the first communicator is created; the contract
    is ONE JSON line on stdout, so stdout is pointed at stderr while the process group and its communicator come up.  NCCL_DEBUG=INFO
    goes to a per-process file; rccl_summary() quotes the lines that say which algorithm / protocol / channel count RCCL chose."""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
    from salt_amd.parallel import configure_rccl_env
    global RCCL_MAX_NCHANNELS
    RCCL_MAX_NCHANNELS = configure_rccl_env()          # NCCL_MAX_NCHANNELS (default 32: one CU per channel - parallel.py has the reasoning)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        dist.init_process_group('nccl')
        warm = torch.zeros(1 << 20, device='cuda')
        dist.all_reduce(warm)
        torch.cuda.synchronize()
    finally:
        sys.stdout.flush()
        try:                                 # the banner sits in the C stdio buffer (fully buffered on a pipe): flush it to stderr
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        os.dup2(saved, 1)
        os.close(saved)


def rccl_summary(limit=12):
    """The NCCL_DEBUG=INFO lines that name RCCL's choices (rings / trees / channels / protocol / transport), echoed to stderr once
    and returned for the JSON line, so a scaling run says WHICH collective ran over xGMI."""
    import re
    try:
        lines = open(RCCL_LOG, errors='replace').read().splitlines()
    except OSError:
        return None
    pat = re.compile(r'(Ring|Tree|Channel|channels|Algo|algo|Proto|proto|via P2P|via SHM|XGMI|xgmi|nranks|comm 0x|RCCL version|Using network)')
    keep = [re.sub(r'^\S+:\d+:\d+ \[\d+\] ', '', ln) for ln in lines if pat.search(ln)]
    out = []
    for ln in keep:                           # one line per distinct message shape
        key = re.sub(r'\d+', '#', ln)[:60]
        if key not in [k for k, _ in out]:
            out.append((key, ln[:200]))
    out = [ln for _, ln in out[:limit]]
    for ln in out:
        sys.stderr.write('[rccl] ' + ln + '\n')
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` with no launcher in the environment: start N ranks of this script through torch.distributed.run
    (the reference's nn.DataParallel needed no launcher, models.py:81-85; one process per GPU does).  Rank 0's JSON line is the
    only thing the ranks write to stdout; the children's return code is ours."""
    import socket
    import subprocess
    n = args.gpus
    if not args.selftest_launch:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            sys.stderr.write('bench.py: --gpus %d needs %d visible GPUs, this host shows %d (HIP_VISIBLE_DEVICES=%s); nothing was run\n'
                             % (n, n, have, os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')))
            return 3
    with socket.socket() as s_:                       # a free rendezvous port on the loopback interface
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    env.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(cmd, env=env)


def selftest_launch_rank():
    """Launcher self-test (tests/test_host_cpu.py): the ranks come up under gloo on CPU, agree on a MAX-reduced clock exactly as the
    timed region does, and rank 0 prints the one JSON line.  No GPU, no model - it proves `bench.py --gpus N` starts N ranks."""
    world, rank = int(os.environ['WORLD_SIZE']), int(os.environ['RANK'])
    dist.init_process_group('gloo')
    dist.barrier()
    t0 = time.perf_counter()
    x = torch.full((4,), float(rank + 1))
    dist.all_reduce(x)
    dist.barrier()
    te = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(te, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({'metric': 'launcher self-test', 'n_gpus': world, 'ranks': dist.get_world_size(), 'backend': 'gloo',
                          'allreduce_sum': float(x[0]), 'expected_sum': world * (world + 1) / 2.0, 'elapsed_s': float(te[0])}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--dtype', default=os.environ.get('SALT_BENCH_DTYPE', 'bf16'), choices=['bf16', 'f32'])
    ap.add_argument('--workload', default='r34_hyper', choices=['r34_hyper', 'ternaus34', 'vanilla'])
    ap.add_argument('--batch', type=int, default=32, help='images per GPU per step')
    ap.add_argument('--loss', default='lovasz', choices=['lovasz', 'bce_dice'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-iou', action='store_true', help='skip the fit_e2e leg (SegmentationModel.fit over host tiles with the callback stack + validation IoU)')
    ap.add_argument('--no-configs', action='store_true', help='skip the other BASELINE configurations (C1, fp32, C3 shape, C4)')
    ap.add_argument('--cpu-leg', default=None, help=argparse.SUPPRESS)       # child mode of the cpu_baseline leg
    ap.add_argument('--dp-leg', action='store_true', help=argparse.SUPPRESS)     # child mode of configs.dp_path_1rank
    ap.add_argument('--cpu-threads', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--cpu-batch', type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument('--selftest-launch', action='store_true', help=argparse.SUPPRESS)   # launcher self-test under gloo (no GPU)
    ap.add_argument('--host-contention', action='store_true', help='ONLY the host-contention proxy of an 8-rank node on a 1-GPU box: fit_e2e on core share 0 of 8 '
                    'while 7 CPU-only copies of a rank\'s host work (loader gather + a launch-thread busy loop) run on the other shares')
    ap.add_argument('--host-load', type=int, default=-1, help=argparse.SUPPRESS)      # child mode of --host-contention: core share k of 8
    args = ap.parse_args()
    if args.host_load >= 0:
        host_load_child(args.host_load, args.batch)
        return
    if args.host_contention:
        host_contention(args.workload, args.dtype, args.batch, args.loss)
        return
    if args.dp_leg:
        dp_leg(args.workload, args.dtype, args.batch, args.loss)
        return
    if args.cpu_leg:
        cpu_baseline_leg(args.cpu_leg, args.loss, 1 if args.cpu_leg == 'VanillaUNet' else 3, args.cpu_threads, args.cpu_batch)
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    if args.selftest_launch:
        selftest_launch_rank()
        return

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    torch.cuda.set_device(local)
    # one process per GPU: the launch thread and the loader thread of this rank stay on the cores of its GPU's NUMA node (parallel.py)
    from salt_amd.parallel import pin_rank_threads
    affinity = pin_rank_threads(local, int(os.environ.get('LOCAL_WORLD_SIZE', str(world))))
    if world > 1 or os.environ.get('SALT_FORCE_DP_PATH'):
        init_rccl(rank)
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    dev = torch.device('cuda', local)

    import salt_amd  # noqa: F401

    arch_name, channels, wl_desc = WORKLOADS[args.workload]
    B = args.batch
    model, batches, elapsed, final_loss = train_config(args.workload, args.dtype, B, args.loss, args.steps, args.warmup, dev, rank, world)
    value = world * B * args.steps / elapsed

    out = {'metric': 'training images/sec, U-Net ResNet34 101x101 (pad->128) bs32/GPU', 'value': round(value, 2), 'unit': 'images/s',
           'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
           'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
           'config': {'workload': '%s 128x128 (101 resized+edge-padded) batch %d/GPU, %s loss, Adam lr1e-4 L2 1e-4, train-mode BN'
                                  % (wl_desc, B, args.loss),
                      'global_batch': B * world, 'image': [128, 128], 'parallelism': 'dp%d' % world},
           'final_loss': round(final_loss, 5)}
    out['host'] = {'logical_cpus': os.cpu_count(), 'cpu_affinity': affinity,
                   'note': 'cpu_affinity: what parallel.pin_rank_threads gave this rank (cores of its GPU\'s NUMA node, split between the ranks that share it); None = left alone'}
    if STEP_MS:
        n_ = len(STEP_MS)
        out['ms_per_step_median'] = round(STEP_MS[n_ // 2], 3)
        out['step_ms_distribution'] = {'steps': n_, 'median': round(STEP_MS[n_ // 2], 3), 'p10': round(STEP_MS[n_ // 10], 3), 'p90': round(STEP_MS[(9 * n_) // 10], 3),
                                       'min': round(STEP_MS[0], 3), 'max': round(STEP_MS[-1], 3),
                                       'note': 'a second leg of >= 50 steps behind the timed region, one HIP event per step on the compute stream (rank 0); '
                                               '`ms_per_step` / `value` are the MEAN of the K timed steps the contract brackets with barrier + synchronize'}
        out['value_at_median'] = round(world * B / (STEP_MS[n_ // 2] * 1e-3), 2)
    if world > 1 or model.dp._active():
        # how to read a scaling run: every rank is one RCCL rank; `exposed_allreduce_ms` is what the compute stream still waited for
        # after its last backward kernel (the collectives of the earlier buckets ran underneath backward)
        ex = model.dp.exposed_allreduce_ms()
        out['rccl_ranks'] = dist.get_world_size() if dist.is_initialized() else 1
        out['per_rank_ms_per_step'] = {'min': min(RANK_MS), 'max': max(RANK_MS), 'all': RANK_MS}
        out['rccl_max_nchannels'] = RCCL_MAX_NCHANNELS      # NCCL_MAX_NCHANNELS in effect (None: RCCL's default); rccl_info has what RCCL reports
        if rank == 0:
            out['rccl_info'] = rccl_summary()
        out['allreduce'] = {'buckets': len(list(model.dp._plans.values())[0]) if model.dp._plans else 0,
                            'gradient_bytes': int(model.model.engine().n_live * 4),
                            'exposed_allreduce_ms_per_step_rank0': round(ex, 4) if ex is not None else None}
        # per bucket: when its gradients were final and when its all-reduce had finished, relative to the END of backward on the compute
        # stream (negative = underneath backward): 6 extra steps behind the timed region, every rank takes part in their collectives
        model.dp.measure = False
        model.dp.timeline = True
        for i in range(6):
            model._fit_loop(list(batches[i % len(batches)]))
        out['allreduce']['bucket_timeline_rank0'] = model.dp.bucket_timeline()
        model.dp.timeline = False
        if out['rccl_ranks'] != args.gpus:
            sys.stderr.write('bench.py: --gpus %d but the RCCL communicator has %d ranks\n' % (args.gpus, out['rccl_ranks']))
            if dist.is_initialized():
                dist.destroy_process_group()
            sys.exit(4)

    # ------------------------------------------------------------------ live roofline (rank 0): event pair around every operator
    if rank == 0:
        eng = model.model.engine()
        net = eng.net((B, channels, 128, 128), True)
        net.x.copy_(batches[0][0]); net.target.copy_(batches[0][1])
        roof, ops, total_ms, side_ms, all_fl, dn = conv_roofline(model, B, channels, args.dtype, args.loss, 3)
        headline = args.dtype == 'bf16' and args.workload == 'r34_hyper' and B == 32
        roof['traffic'] = pmc_traffic({'conv': ('conv_ws_kernel', 'conv_ls_kernel', 'conv1x1_ls_kernel', 'conv_thin_kernel', 'conv_mfma_kernel', 'conv_glds_kernel'), 'conv_wgrad': ('conv_wgrad_kernel',)}.get(dn, (dn,))) if headline else None
        roof['traffic_unit'] = ('bytes per launch (rocprofv3 PMC FETCH_SIZE*2 + WRITE_SIZE, separate --pmc passes; file %s measured at commit %s - '
                                'PMC counters cannot be read inside the timed process)' % (PMC_FILE, pmc_commit()))
        out['roofline_by_class'] = roof.pop('by_class')
        if headline:
            # whole-step HBM bytes: the rocprofv3 PMC counters (quoted from the committed file, commit inside) against the algorithmic
            # bytes of the step (sum over convolutions of 3|x| + 2|y| activations + weight / optimizer bytes = 155.4 MB per
            # image for this network at bf16, batch 32) - the ratio is what is left in re-reads, slabs and standalone BatchNorm passes
            try:
                pm = json.load(open(os.path.join(ROOT, PMC_FILE)))
                cnt = (pm['step_total_MB']['read'] + pm['step_total_MB']['write']) * 1e6
                alg = 155.4e6 * B
                out['hbm_bytes_per_step'] = {'counter': round(cnt), 'algorithmic': round(alg), 'ratio': round(cnt / alg, 3),
                                             'counter_GBps_at_this_step_time': round(cnt / (elapsed / args.steps) / 1e9, 1), 'hbm_peak_GBps': HBM_PEAK_GBS,
                                             'quoted_from': PMC_FILE, 'commit': pm.get('commit')}
            except (OSError, KeyError, ValueError):
                out['hbm_bytes_per_step'] = None
        # the same class INSIDE the step (both queues running), from the committed rocprofv3 kernel trace: quoted, not measured here
        if headline:
            try:
                ins = json.load(open(os.path.join(ROOT, INSTEP_FILE)))
                for cls in ('conv', 'conv_wgrad'):
                    bc = out['roofline_by_class'].get(cls)
                    us = ins['classes'].get(cls, {}).get('us_per_step')
                    if bc and us:
                        fl = bc['achieved'] * 1e12 * bc['ms_per_step'] * 1e-3              # algorithmic FLOPs per step of the class
                        bc['in_step'] = {'quoted_from': INSTEP_FILE, 'commit': ins.get('commit'), 'kernel_us_per_step': us,
                                         'achieved': round(fl / (us * 1e-6) / 1e12, 1), 'frac': round(fl / (us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS[args.dtype], 4)}
                if dn in ('conv', 'conv_wgrad') and 'in_step' in out['roofline_by_class'].get(dn, {}):
                    roof['in_step'] = out['roofline_by_class'][dn]['in_step']
            except (OSError, KeyError, ValueError):
                pass
        out['roofline'] = roof
        out['op_time_ms'] = ops
        out['op_time_serial_ms'] = round(total_ms, 3)          # every operator run back to back on ONE stream (no wgrad overlap)
        out['op_time_side_stream_ms'] = round(side_ms, 3)
        out['step_tflops'] = round(all_fl / (elapsed / args.steps) / 1e12, 2)
        # reference-algorithmic work of the same step: 2 MAC over every Conv / ConvT of the network as the reference defines it (SURVEY 3.4 / 8d:
        # 9.75 GMAC per 128 x 128 image forward for the ResNet34 hypercolumn U-Net), x3 for forward + data gradient + weight gradient
        ref_gf = 6 * 9.75 * B if args.workload == 'r34_hyper' else None
        out['flops'] = {'executed_matrix_gflop_per_step': round(all_fl / 1e9, 1), 'reference_algorithmic_gflop_per_step': ref_gf,
                        'step_tflops_executed': out['step_tflops'],
                        'step_tflops_reference_algorithmic': round(ref_gf * 1e9 / (elapsed / args.steps) / 1e12, 2) if ref_gf else None,
                        'note': 'executed < reference: the factored hypercolumn replaces 3/5 of the final 3x3 convolution (fwd + both gradients) by 1x1 '
                                'contractions at 1/16 .. 1/256 of the pixels + a vector-ALU stencil; roofline fractions are priced on EXECUTED matrix FLOPs'}

    # ------------------------------------------------------------------ end to end: SegmentationModel.fit() over host tiles (N = 1 only)
    fit_pending = not args.no_iou and rank == 0 and world == 1

    if rank == 0 and 'roofline_by_class' in out:           # the fused Adam + L2 kernel (28 bytes per parameter), timed after the evaluation
        aprog = model.optimizer._adam_pack_program() or model.optimizer.prog      # (bf16: Adam also writes the forward weight packs, 2 more bytes per packed parameter)
        for name, _, ms in aprog.run_timed():
            if name in ('adam', 'adam_pack') and ms > 0:
                by = (28.0 + (2.0 if name == 'adam_pack' else 0.0)) * model.model.engine().n_live
                out['roofline_by_class']['adam'] = {'launches_per_step': 1, 'ms_per_step': round(ms, 3), 'bound': 'hbm', 'achieved': round(by / (ms * 1e-3) / 1e9, 1),
                                                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # ------------------------------------------------------------------ the other BASELINE configurations, same run (N = 1 only)
    if fit_pending:
        del model, batches
        torch.cuda.empty_cache()
        model = batches = None
        fe = fit_e2e_leg(dev, args.workload, args.dtype, B, args.loss)
        out['fit_e2e'] = fe
        out['fit_e2e']['vs_fit_loop_headline'] = round(fe['images_per_s'] / (B * args.steps / elapsed), 4)
        out['val_iou'] = fe['val_iou']
        out['val_iou_note'] = ('validation IoU (reference metric, threshold from the sweep) on %d held-out synthetic tiles after fit(): %d epochs x %d '
                               'steps from random init - a signal that the shipped trainer learns, not an accuracy result; convergence PARITY is '
                               'tests/test_gpu_convergence.py' % (fe['n_val'], fe['epochs'], fe['steps'] // fe['epochs']))
    if rank == 0 and world == 1 and not args.no_configs:
        if model is not None:
            del model, batches
            torch.cuda.empty_cache()
        out['configs'] = extra_configs(dev)
        if not dist.is_initialized():
            # the bucketed data-parallel step against a 1-rank RCCL communicator, alternated with the plain step - in a FRESH child
            # process, which is what a rank of a real run looks like: inside this process (a dozen HIP streams created and destroyed by
            # the configurations above) the communication stream aliases a hardware queue of the compute streams and the same A/B
            # reads +6.5 % instead of +3 %
            out['configs']['dp_path_1rank'] = dp_leg_subprocess(args.workload, args.dtype, B, args.loss)

    # ------------------------------------------------------------------ CPU baseline: the oracle on the host cores (bounded sample)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline_subprocess(args.loss, arch_name)
        if not args.no_configs:
            # BASELINE C0: the vanilla U-Net on the CPU (batch 8), the reference's own CPU-runnable case
            out['configs']['C0_vanilla_cpu_b8'] = cpu_baseline_subprocess('lovasz', 'VanillaUNet')
    if rank == 0:
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
