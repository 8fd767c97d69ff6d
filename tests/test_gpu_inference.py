"""GPU parity of the inference epilogue (TTA, crop + threshold) and the validation metric vs the reference goldens (F9, F10) and
the oracle (oracle/metrics.py restates augmentation.py / postprocessing.py / metrics.py / callbacks.py:503-513)."""
import numpy as np
import pytest
import torch

from helpers import golden, T, assert_close

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_flip_and_inverse_vs_reference_golden():
    """augmentation.py:143-163: forward flips of an HWC image and inverse flips of CHW predictions (rotation-free variants)."""
    from salt_amd import inference as I
    fx = golden('F9_tta')
    img = T(fx['img']).permute(2, 0, 1)[None].contiguous().to(DEV)          # HWC -> NCHW
    pred = T(fx['pred']).to(DEV)
    for i, (ud, lr, rot) in enumerate(fx['specs'].tolist()):
        if rot:
            continue
        y = I.flip(img, ud, lr)[0].permute(1, 2, 0).cpu().numpy()
        assert np.array_equal(y, fx['fwd%d' % i]), i
        inv = I.flip(pred[i % 4][None], ud, lr)[0].cpu().numpy()              # a flip is its own inverse
        assert np.array_equal(inv, fx['inv%d' % i]), i


@pytest.mark.parametrize('ud,lr', [(True, True), (False, True), (True, False), (False, False)])
def test_tta_mean_vs_oracle(ud, lr):
    from salt_amd import inference as I
    from oracle import metrics as OM
    g = torch.Generator().manual_seed(11)
    B, C, H, W = 3, 2, 16, 24
    variants = I.tta_variants(ud, lr)
    specs = OM.tta_specs(ud, lr)
    assert [(s['ud_flip'], s['lr_flip']) for s in specs] == variants           # same order as the reference generator
    logits = torch.randn(len(variants) * B, C, H, W, generator=g) * 3
    prob = I.tta_mean(logits.to(DEV), variants, B).cpu().numpy()
    for b in range(B):
        preds = [OM.sigmoid(logits[v * B + b].numpy()) for v in range(len(variants))]
        ref = OM.tta_aggregate(preds, specs, 'mean')
        assert_close(prob[b], ref, 2e-6, 'tta mean')


def test_crop_threshold_vs_reference_golden_and_oracle():
    from salt_amd import inference as I
    from oracle import metrics as OM
    fx = golden('F10_post_metric')
    p128 = fx['p128']                                                        # (2,128,128) "probability" planes
    assert I.crop_window(128, 128, (101, 101)) == (13, 14)
    prob = T(OM.sigmoid(p128).astype(np.float32))[None].to(DEV)              # [1, 2, 128, 128]
    m = I.crop_threshold(prob, (101, 101), 0.5, cls=1).cpu().numpy()[0]
    ref = OM.binarize(OM.crop_image(OM.sigmoid(p128).astype(np.float32), (101, 101)), 0.5)
    assert np.array_equal(m, ref)
    assert np.array_equal(OM.crop_image(p128, (101, 101)), fx['crop101'])
    # other geometry + threshold
    g = torch.Generator().manual_seed(5)
    pr = torch.rand(4, 2, 64, 96, generator=g)
    for th in (0.3, 0.5, 0.72):
        m = I.crop_threshold(pr.to(DEV), (50, 71), th, cls=0).cpu().numpy()
        for b in range(4):
            ref = (OM.crop_image(pr[b].numpy(), (50, 71))[0] > np.float32(th)).astype(np.uint8)
            assert np.array_equal(m[b], ref)


def test_iou_sweep_vs_oracle_metric():
    """Counts -> IoU / IOUT for the whole callbacks.py:503-513 sweep equal the per-threshold numpy restatement, incl. empty masks."""
    from salt_amd import inference as I
    from oracle import metrics as OM
    g = torch.Generator().manual_seed(9)
    B, H, W, h, w = 12, 128, 128, 101, 101
    prob = torch.rand(B, 2, H, W, generator=g)
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
    gt = torch.zeros(B, h, w, dtype=torch.uint8)
    for b in range(B):
        if b % 4 == 0:
            continue                                                         # empty ground truth
        r = 10 + 6 * b
        gt[b] = (((yy - 50) ** 2 + (xx - 40 - b) ** 2) < r * r).to(torch.uint8)
        top, left = I.crop_window(H, W, (h, w))
        prob[b, 1, top:top + h, left:left + w] = 0.15 + 0.6 * gt[b].float() + 0.2 * torch.rand(h, w, generator=g)
    prob[0, 1] = 0.05                                                        # empty prediction on an empty mask -> IoU 1
    prob[4, 1] = 0.9                                                         # full prediction on an empty mask -> IoU 0
    ths = np.linspace(0.5, 0.3, 21)
    counts = I.iou_counts(prob.to(DEV), gt.to(DEV), ths, cls=1)
    iou, iout = I.scores_from_counts(*counts)
    for k, th in enumerate(ths):
        preds = [(OM.crop_image(prob[b].numpy(), (h, w))[1].astype(np.float64) > th).astype(np.uint8) for b in range(B)]
        gts = [gt[b].numpy() for b in range(B)]
        assert abs(iou[k] - OM.intersection_over_union(gts, preds)) < 1e-12, k
        assert abs(iout[k] - OM.intersection_over_union_thresholds(gts, preds)) < 1e-12, k
    # threshold selection walks the sweep while IOUT improves
    t_best, iou_b, iout_b = I.select_threshold([counts])
    best, tb = 0.0, 0.5
    for k, th in enumerate(ths):
        if iout[k] > best:
            best, tb = iout[k], float(th)
        else:
            break
    assert t_best == tb and abs(iout_b - best) < 1e-12


def test_predict_tta_end_to_end_vs_oracle():
    """4-flip TTA of an eval-mode network: device pipeline vs oracle forward of numpy-flipped inputs + numpy aggregation."""
    from salt_amd import architectures as A, inference as I
    from oracle import nets as ON, specs as OS, metrics as OM
    torch.manual_seed(3)
    net = A.VanillaUNet(2, in_channels=1, base_filters=8, levels=3)
    spec = OS.spec_vanilla_unet(2, 1, 8, 3)
    sd = OS.init_state(spec, seed=2)
    net.load_state_dict(sd)
    net.to(DEV).eval()
    X = torch.randn(2, 1, 32, 48)
    prob = I.predict_tta(net, X.to(DEV), True, True).cpu().numpy()
    specs = OM.tta_specs(True, True)
    for b in range(2):
        preds = []
        for s in specs:
            xv = OM.tta_transform(X[b].permute(1, 2, 0).numpy(), s)
            xv = torch.from_numpy(np.ascontiguousarray(xv)).permute(2, 0, 1)[None]
            with torch.no_grad():
                preds.append(OM.sigmoid(ON.vanilla_unet(sd, xv, False, levels=3)[0].numpy()))
        ref = OM.tta_aggregate(preds, specs, 'mean')
        assert_close(prob[b], ref, 1e-4, 'tta probabilities')
    with pytest.raises(Exception):
        net.train(); I.predict_tta(net, X.to(DEV))


@pytest.mark.parametrize('method', ['mean', 'max', 'min', 'gmean'])
def test_predict_tta_tiles_rotation_and_aggregators_vs_oracle(method):
    """The reference's TTA in its own order of operations (loaders.py:662-760, augmentation.py:143-163): flipud / fliplr / rot90 of
    the RAW square tile, inference preprocessing of every variant, forward, inverse transform of the probability maps, aggregation by
    mean / max / min / gmean - 16 variants with rotation - against the numpy restatement around the oracle network."""
    from salt_amd import architectures as A, inference as I
    from salt_amd.input_pipeline import DevicePreprocessor
    from oracle import nets as ON, specs as OS, metrics as OM, inputs as OI
    torch.manual_seed(3)
    net = A.VanillaUNet(2, in_channels=1, base_filters=8, levels=3)
    spec = OS.spec_vanilla_unet(2, 1, 8, 3)
    sd = OS.init_state(spec, seed=2)
    net.load_state_dict(sd)
    net.to(DEV).eval()
    r = np.random.RandomState(8)
    img = r.rand(2, 101, 101).astype(np.float32)
    pre = DevicePreprocessor(False, 1)
    prob = I.predict_tta_tiles(net, pre, T(img).to(DEV), flip_ud=True, flip_lr=True, rotation=True, method=method).cpu().numpy()
    specs = OM.tta_specs(True, True, True)
    assert len(specs) == 16
    for b in range(2):
        preds = []
        for sp in specs:
            v = np.ascontiguousarray(OM.tta_transform(img[b][:, :, None], sp)[:, :, 0])
            xv, _ = OI.preprocess(v[None], None, False, 1)
            with torch.no_grad():
                preds.append(OM.sigmoid(ON.vanilla_unet(sd, xv, False, levels=3)[0].numpy()))
        ref = OM.tta_aggregate(preds, specs, method)
        assert_close(prob[b], ref, 1e-4, 'tta %s probabilities' % method)


@pytest.mark.parametrize('method', ['max', 'min', 'gmean'])
def test_predict_tta_batch_aggregators_vs_oracle(method):
    """predict_tta (flips of the preprocessed batch) with the non-default aggregators."""
    from salt_amd import architectures as A, inference as I
    from oracle import nets as ON, specs as OS, metrics as OM
    torch.manual_seed(3)
    net = A.VanillaUNet(2, in_channels=1, base_filters=8, levels=3)
    spec = OS.spec_vanilla_unet(2, 1, 8, 3)
    sd = OS.init_state(spec, seed=2)
    net.load_state_dict(sd)
    net.to(DEV).eval()
    X = torch.randn(2, 1, 32, 48)
    prob = I.predict_tta(net, X.to(DEV), True, True, depth_channels=False, method=method).cpu().numpy()
    specs = OM.tta_specs(True, True)
    for b in range(2):
        preds = []
        for sp in specs:
            xv = OM.tta_transform(X[b].permute(1, 2, 0).numpy(), sp)
            xv = torch.from_numpy(np.ascontiguousarray(xv)).permute(2, 0, 1)[None]
            with torch.no_grad():
                preds.append(OM.sigmoid(ON.vanilla_unet(sd, xv, False, levels=3)[0].numpy()))
        assert_close(prob[b], OM.tta_aggregate(preds, specs, method), 1e-4, 'tta %s' % method)


@pytest.mark.parametrize('train', [True, False])
@pytest.mark.parametrize('channels', [1, 3])
@pytest.mark.parametrize('as_u8', [False, True])
def test_device_input_pipeline_vs_oracle(train, channels, as_u8, interpolation='cubic'):
    """f-1: resize + edge-pad + normalise + depth channels + one-hot target in one kernel vs the CPU restatement.  The resize kernel
    restated is cv2.INTER_CUBIC - imgaug 0.2.5's iaa.Scale default, which augmentation.py:79-85 gets by passing no interpolation
    (oracle/inputs.py: unpinned by the reference, imgaug / cv2 absent; pinned to the public definition by its own KATs)."""
    from salt_amd.input_pipeline import DevicePreprocessor
    from oracle import inputs as OI
    r = np.random.RandomState(4)
    B, h, w = 5, 101, 101
    img = r.rand(B, h, w).astype(np.float32)
    if as_u8:
        u8 = (img * 255).astype(np.uint8)
        img = u8.astype(np.float32) * np.float32(1.0 / 255.0)              # what the kernel sees
        dev_img = T(u8).to(DEV)
    else:
        dev_img = T(img).to(DEV)
    msk = np.zeros((B, h, w), np.uint8)                                     # blobs, not salt-and-pepper: what a salt mask looks like
    yy, xx = np.mgrid[0:h, 0:w]
    for b in range(1, B):
        msk[b] = (((yy - r.uniform(0, h)) / r.uniform(10, 60)) ** 2 + ((xx - r.uniform(0, w)) / r.uniform(10, 60)) ** 2 <= 1).astype(np.uint8)
    pre = DevicePreprocessor(train, channels, interpolation=interpolation)
    X, Tg = pre(dev_img, T(msk).to(DEV))
    Xr, Tr = OI.preprocess(img, msk, train, channels, interpolation=interpolation, uint8_grid=as_u8)
    assert tuple(X.shape) == tuple(Xr.shape) == (B, channels, 128, 128)
    if train and interpolation == 'cubic' and as_u8:
        # uint8 tiles: cv2's fixed-point evaluation (integer sums) on both sides - the resized uint8 values are EQUAL, what remains is
        # the float normalisation
        g_dev = torch.round((X.cpu()[:, 0] * 0.229 + 0.485) * 255)
        g_ref = torch.round((Xr[:, 0] * 0.229 + 0.485) * 255)
        assert torch.equal(g_dev, g_ref), int((g_dev != g_ref).sum())
        assert_close(X.cpu(), Xr, 2e-6, 'input batch (fixed-point cubic)')
    elif train and interpolation == 'cubic_float' and as_u8:
        # the float form rounds onto the uint8 grid on both sides; float summation order may move a value across a rounding boundary:
        # at most one grid step (1 / 255 / std), on a vanishing fraction of the pixels
        d = (X.cpu() - Xr)[:, 0].abs()
        assert float(d.max()) <= 1.0 / 255 / 0.229 + 1e-5, float(d.max())
        assert float((d > 1e-5).float().mean()) < 1e-3, float((d > 1e-5).float().mean())
    else:
        assert_close(X.cpu(), Xr, 2e-5 if (train and interpolation == 'cubic') else 2e-6, 'input batch')
    assert torch.equal(Tg.cpu(), Tr)                                         # one-hot target: exact
    X2, T2 = pre(dev_img)                                                    # inference batches carry no target
    assert T2 is None and torch.equal(X2, X)


def test_device_input_pipeline_cubic_float_mode_for_uint8_tiles():
    """interpolation='cubic_float': round 3's float form of the cubic for uint8 tiles, kept selectable; it is NOT what cv2 computes
    (the fixed-point default differs from it by one grey level on a few percent of the pixels of a noise tile)."""
    test_device_input_pipeline_vs_oracle(True, 3, True, interpolation='cubic_float')
    from salt_amd.input_pipeline import DevicePreprocessor
    u8 = torch.from_numpy(np.random.RandomState(7).randint(0, 256, (2, 101, 101)).astype(np.uint8)).to(DEV)
    a, _ = DevicePreprocessor(True, 1)(u8)
    b, _ = DevicePreprocessor(True, 1, interpolation='cubic_float')(u8)
    frac = float(((a - b).abs() > 1e-4).float().mean())
    assert 0.005 < frac < 0.15, frac


@pytest.mark.parametrize('as_u8', [True, False])
def test_device_input_pipeline_bilinear_mode(as_u8):
    """interpolation='bilinear': rounds 1-2's train-branch resize (bilinear tile, nearest mask), kept selectable."""
    test_device_input_pipeline_vs_oracle(True, 3, as_u8, interpolation='bilinear')


@pytest.mark.parametrize('R', [2, 4, 8, 16])
@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
def test_bilinear_align_corners_vs_torch(R, dtype):
    """salt_bilinear_args.align_corners = 1: nn.Upsample(mode='bilinear') as torch 0.3.1 - the reference's pinned version - evaluated
    it (base.py:70, unet.py:103-106): forward and the adjoint against torch's align_corners=True autograd."""
    from gpu_harness import BlockRun
    from torch import nn
    g = torch.Generator().manual_seed(R)
    x = torch.randn(2, 16, 5, 7, generator=g)
    if dtype == 'bf16':
        x = x.bfloat16().float()
    dummy = nn.Linear(1, 1)
    dummy.align_corners = True
    run = BlockRun(dummy, [x], lambda gr, a: gr.upsample(a, R), train=True, dtype=dtype)
    assert [int(s.align_corners) for name, _, s in run.g.fwd.ops if name == 'bilinear'] == [1]
    y = run.forward()
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.interpolate(xr, scale_factor=R, mode='bilinear', align_corners=True)
    tol = 2e-6 if dtype == 'f32' else 1e-2
    assert_close(y, yr, tol, 'up x%d align_corners' % R)
    gy = torch.randn(*yr.shape, generator=g)
    yr.backward(gy)
    gx, _ = run.backward(gy.to(DEV))
    assert_close(gx[0], xr.grad, 1e-5 if dtype == 'f32' else 2e-2, 'adjoint x%d align_corners' % R)


def test_network_align_corners_switch_vs_oracle():
    """HipNetwork.set_align_corners(True) - how a checkpoint trained under torch 0.3.1 saw its decoder / hypercolumn features: eval logits
    of the ResNet34 hypercolumn U-Net against the oracle with the same switch; the default (False) is what the goldens pin."""
    from salt_amd import architectures as A
    from oracle import nets as ON, specs as OS, blocks as OB
    spec = OS.SPECS['UNetResNet'](with_fc=True)
    sd = OS.init_state(spec, seed=3)
    net = A.UNetResNet(34, 2, use_hypercolumn=True, dropout_2d=0.0, pretrained=False)
    net.load_state_dict({k: sd[k] for k in net.state_dict() if k in sd}, strict=False)
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    outs = {}
    for ac in (False, True):
        net.set_align_corners(ac)
        net.to(DEV).eval()
        with torch.no_grad():
            outs[ac] = net(x.to(DEV)).float().cpu()
        OB.ALIGN_CORNERS = ac
        try:
            with torch.no_grad():
                ref = ON.unet_resnet({k: v for k, v in sd.items()}, x, False)
        finally:
            OB.ALIGN_CORNERS = False
        assert_close(outs[ac], ref, 1e-3, 'eval logits align_corners=%s' % ac)
    assert float((outs[True] - outs[False]).abs().max()) > 1e-3               # the switch is not a no-op


def test_device_input_pipeline_other_sizes():
    from salt_amd.input_pipeline import DevicePreprocessor
    from oracle import inputs as OI
    r = np.random.RandomState(6)
    img = r.rand(2, 202, 202).astype(np.float32)                            # C4: 256x256 inputs (pad/resize x2)
    X, _ = DevicePreprocessor(False, 3)(T(img).to(DEV))
    Xr, _ = OI.preprocess(img, None, False, 3)
    assert tuple(X.shape) == (2, 3, 256, 256)
    assert_close(X.cpu(), Xr, 2e-6, 'inference pad 202 -> 256')
    X, Tg = DevicePreprocessor(True, 3, resize=204, pad=26)(T(img).to(DEV), T((img > 0.5).astype(np.uint8)).to(DEV))
    Xr, Tr = OI.preprocess(img, (img > 0.5).astype(np.uint8), True, 3, resize=204, pad=26, uint8_grid=False)     # float tiles: no uint8 rounding
    assert tuple(X.shape) == (2, 3, 256, 256)
    assert_close(X.cpu(), Xr, 2e-5, 'train cubic resize 202 -> 204 + pad 26')
    assert torch.equal(Tg.cpu(), Tr)


def test_hipgraph_capture_replays_the_eval_program():
    """salt_graph_capture / salt_graph_launch: a captured eval forward gives the same logits as the op-by-op executor."""
    from salt_amd import architectures as A
    torch.manual_seed(1)
    net = A.VanillaUNet(2, in_channels=1, base_filters=8, levels=3).to(DEV).eval()
    x1 = torch.randn(2, 1, 32, 32, device=DEV)
    x2 = torch.randn(2, 1, 32, 32, device=DEV)
    with torch.no_grad():
        ref1 = net(x1).clone()
        ref2 = net(x2).clone()
    eng = net.engine()
    cn = eng.net((2, 1, 32, 32), False)
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    cn.fwd.capture(st)
    for x, ref in ((x1, ref1), (x2, ref2), (x1, ref1)):
        with torch.cuda.stream(st):
            cn.x.copy_(x)
            cn.fwd.replay(st)
        st.synchronize()
        assert torch.equal(cn.logits, ref)
    cn.fwd.release_graph()
    with pytest.raises(Exception):
        cn.fwd.replay(st)


@pytest.mark.parametrize('dtype', ['f32', 'bf16'])
@pytest.mark.parametrize('ac', [False, True])
def test_hyper_rows_equals_per_level_bilinear(dtype, ac):
    """salt_hyper_rows (all hypercolumn levels of a pixel row in one pass, architectures/unet.py:101-107) against one salt_bilinear launch
    per level into channel slices: bit-identical, both align_corners conventions."""
    from torch import nn
    from salt_amd.engine import Graph
    from salt_amd.runtime import Engine
    mod = nn.Linear(1, 1).to(DEV)
    mod.align_corners = ac
    eng = Engine(mod, torch.device(DEV), dtype)
    g = Graph(eng, False)
    B, H, W, C, Rs = 2, 32, 48, 16, [2, 4, 8, 16]
    xs = [g.new_act(B, H // R, W // R, C, 'x%d' % R) for R in Rs]
    gen = torch.Generator(device=DEV).manual_seed(5)
    for x in xs:
        x.buf.t.copy_(torch.randn(x.buf.t.shape, generator=gen, device=DEV))
    a, b = g.new_act(B, H, W, 5 * C, 'a'), g.new_act(B, H, W, 5 * C, 'b')
    for k, (x, R) in enumerate(zip(xs, Rs)):
        g.upsample(x, R, out=a.slice((k + 1) * C, C))
    g.hyper_rows(xs, Rs, b.slice(C, 4 * C))
    assert [n for n, _, _ in g.fwd.ops] == ['bilinear'] * 4 + ['hyper_rows']
    g.finalize()
    g.fwd.run()
    torch.cuda.synchronize()
    assert float(a.tensor()[..., C:].float().abs().max()) > 0.1
    assert torch.equal(a.tensor()[..., C:], b.tensor()[..., C:])
    assert float(b.tensor()[..., :C].float().abs().max()) == 0.0               # the first slice is not touched


def test_eval_network_hyper_rows_is_bit_identical(monkeypatch):
    """SALT_HYPER_ROWS=1 (opt-in: architectures.py defaults to 0 because the fused pass measured slower) against the per-level
    launches: the same logits, bit for bit.  Both are forms of the MATERIALISED hypercolumn (SALT_HYPER_FACTOR=0; the default since
    round 5 factors the x4 / x8 / x16 levels out - same values to rounding, not to the bit)."""
    from salt_amd import architectures as A
    from oracle import specs as OS
    spec = OS.SPECS['UNetResNet'](with_fc=True)
    sd = OS.init_state(spec, seed=3)
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(1))
    outs = {}
    monkeypatch.setenv('SALT_HYPER_FACTOR', '0')
    for mode in ('0', '1'):
        monkeypatch.setenv('SALT_HYPER_ROWS', mode)
        net = A.UNetResNet(34, 2, use_hypercolumn=True, dropout_2d=0.0, pretrained=False)
        net.load_state_dict({k: sd[k] for k in net.state_dict() if k in sd}, strict=False)
        net.compute_dtype = 'bf16'
        net.to(DEV).eval()
        with torch.no_grad():
            outs[mode] = net(x.to(DEV)).float().cpu()
        names = [n for n, _, _ in net.engine().net((2, 3, 64, 64), False).fwd.ops]
        assert ('hyper_rows' in names) == (mode == '1')
    assert torch.equal(outs['0'], outs['1'])


@pytest.mark.parametrize('method', ['mean', 'max', 'min', 'gmean'])
def test_tta_kernel_rotations_and_aggregators_vs_oracle(method):
    """salt_tta_mean with rot / method: sigmoid, inverse transform (rot90(-k), fliplr, flipud: augmentation.py:156-163) and aggregation
    (loaders.py:727-735) of the 16 flip x rotation variants in one launch, against the numpy restatement on random logits."""
    from salt_amd import inference as I
    from oracle import metrics as OM
    g = torch.Generator().manual_seed(21)
    B, C, n = 2, 2, 24
    specs = OM.tta_specs(True, True, True)
    variants = [(s['ud_flip'], s['lr_flip'], s['rotation'] // 90) for s in specs]
    assert len(variants) == 16
    logits = torch.randn(len(variants) * B, C, n, n, generator=g) * 3
    prob = I.tta_mean(logits.to(DEV), variants, B, method).cpu().numpy()
    for b in range(B):
        preds = [OM.sigmoid(logits[v * B + b].numpy()) for v in range(len(variants))]
        assert_close(prob[b], OM.tta_aggregate(preds, specs, method), 5e-6, 'tta %s' % method)
    with pytest.raises(Exception):
        I.tta_mean(torch.zeros(2 * B, C, 8, 12, device=DEV), [(False, False, 0), (False, False, 1)], B)      # rotation of a non-square map
