"""GPU parity of the COMPOSED BASELINE configurations at their full shapes: the C4 inference pipeline
(ResNet152 hypercolumn U-Net, 256x256, batch 16, 4-flip TTA, crop + threshold, bf16) and the C3 per-GPU training shape
(ResNet34 hypercolumn U-Net, batch 64).  The oracle runs on a bounded part of each workload (one image of the C4 batch; one
fp32 step of the C3 batch), size-independent properties cover the rest."""
import numpy as np
import pytest
import torch

from helpers import assert_close
import closed_form as CF

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _synthetic_256(n, seed):
    """n synthetic 202 x 202 salt tiles through the inference-geometry preprocessor -> (X [n,3,256,256], target [n,2,256,256]) on the device."""
    import bench
    from salt_amd.input_pipeline import DevicePreprocessor
    img, msk = bench.synth_tiles(n, seed=seed, size=202)
    u8 = torch.from_numpy((img * 255).astype(np.uint8)).to(DEV)
    return DevicePreprocessor(False, 3)(u8, torch.from_numpy(msk.astype(np.uint8)).to(DEV))


def test_c4_r152_tta_pipeline_bf16_full_shape():
    """R152 + predict_tta (flip_ud x flip_lr) + crop_threshold at [16,3,256,256] bf16 on a CONDITIONED network: round 3 ran this on a
    random-init R152 whose probabilities all sat within 0.12 of the threshold, so 17 % of the mask pixels flipped under ANY bf16
    implementation and the test could only bound the error by an emulation.  Here the network is first trained
    for a few steps on synthetic salt tiles (the shipped fused step), which pulls the logits apart like a checkpoint's, and then:
    (1) ALL 16 images against the fp32 oracle pipeline (numpy flips, 4 oracle forwards of the batch, sigmoid, inverse flips, mean,
        crop, > 0.5): mask agreement >= 0.999 per image, every differing pixel within the measured probability error of the threshold;
    (2) image 0 also against the oracle with bf16 STORAGE emulated (oracle.blocks.bf16_storage): the HIP path is no further from the
        fp32 pipeline than 1.5x / 2x what bf16 storage costs any implementation;
    (3) equivariance over the whole batch: the TTA mean over the flip group commutes with a flip of the input;
    (4) a probability is a mean of sigmoids: in [0, 1], finite."""
    from salt_amd import inference as I
    from oracle import nets as ON, specs as OS, metrics as OM, blocks as OB
    from test_gpu_fused_step import _segmentation_model
    from helpers import record_parity
    torch.manual_seed(11)
    m = _segmentation_model('UNetResNet152', 'lovasz', dtype='bf16', lr=2e-4)
    spec = OS.SPECS['UNetResNet'](with_fc=True, depth=152)
    sd0 = OS.init_state(spec, seed=9)
    m.model.load_state_dict({k: sd0[k] for k in m.model.state_dict() if k in sd0}, strict=False)
    m._to_device()
    m.model.train()
    STEPS = 24
    Xt, Tt = _synthetic_256(16 * 8, seed=77)                              # 8 distinct batches, cycled
    losses = []
    for it in range(STEPS):
        j = (it % 8) * 16
        losses.append(float(m._fit_loop([Xt[j:j + 16], Tt[j:j + 16]])['sum']))
    torch.cuda.synchronize()
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    net = m.model
    net.eval()
    sd = {k: v.detach().cpu().float().clone() for k, v in net.state_dict().items() if k in spec}
    Xd, _ = _synthetic_256(16, seed=78)                                    # held-out tiles, [16,3,256,256]
    X = Xd.cpu()
    prob = I.predict_tta(net, Xd, True, True, depth_channels=True)
    mask = I.crop_threshold(prob, (202, 202), 0.5, cls=1)
    torch.cuda.synchronize()
    p = prob.float().cpu()
    assert torch.isfinite(p).all() and float(p.min()) >= 0.0 and float(p.max()) <= 1.0                          # (4)
    assert tuple(mask.shape) == (16, 202, 202) and mask.dtype == torch.uint8
    # (3) flip the whole input batch left-right (channel by channel: a left-right flip leaves the depth ramp alone): the aggregated
    # probabilities flip with it (same four forwards per image in another order; the fp32 mean is order dependent in its last bit only)
    prob_f = I.predict_tta(net, torch.flip(Xd, dims=[3]).contiguous(), True, True, depth_channels=True).float().cpu()
    assert float((torch.flip(prob_f, dims=[3]) - p).abs().max()) <= 1e-5
    # (1) oracle pipeline for the whole batch.  A variant of the reference flips the RAW tile and then normalises / adds the depth
    # channels (loaders.py:401-423 -> 603-612): gray flipped, ramp kept, channel 2 = flipped gray x ramp
    specs = OM.tta_specs(True, True)

    def variant(xb, sp):
        ud, lr = bool(sp['ud_flip']), bool(sp['lr_flip'])
        dims = [d for d, f in ((2, ud), (3, lr)) if f]
        if not dims:
            return xb
        y = torch.flip(xb, dims=dims).contiguous()
        if ud:
            y[:, 1] = xb[:, 1]
            y[:, 2] = y[:, 0] * xb[:, 1]
        return y

    def oracle_tta(xb, emulate):
        preds = []
        for sp in specs:
            with torch.no_grad():
                if emulate:
                    with OB.bf16_storage():
                        o = ON.unet_resnet(sd, variant(xb, sp), False, depth=152)
                else:
                    o = ON.unet_resnet(sd, variant(xb, sp), False, depth=152)
            preds.append(OM.sigmoid(o.float().numpy()))
        out = []
        for b in range(xb.shape[0]):
            out.append(OM.tta_aggregate([q[b] for q in preds], specs, 'mean'))
        return np.stack(out)
    ref = oracle_tta(X, False)                                             # [16, 2, 256, 256]
    got = p.numpy()
    err = np.abs(got - ref)
    worst_agree, total_dis, decisions = 1.0, 0, 0
    for b in range(16):
        ref_crop = OM.crop_image(ref[b], (202, 202))[1]
        ref_mask = ref_crop > 0.5
        dis = mask[b].cpu().numpy().astype(bool) != ref_mask
        agree = 1.0 - float(dis.mean())
        worst_agree = min(worst_agree, agree)
        total_dis += int(dis.sum()); decisions += int(dis.size)
        assert agree >= 0.999, 'image %d: mask agreement %.5f' % (b, agree)
        # every disagreeing mask pixel has a reference probability within the measured bf16 error of the threshold
        assert not dis.any() or float(np.abs(ref_crop - 0.5)[dis].max()) <= float(err.max()) + 1e-6
    near = float((np.abs(ref[:, 1] - 0.5) < 0.05).mean())
    # (2) the emulation yardstick on image 0
    emu = oracle_tta(X[:1], True)[0]
    err0, err_emu = err[0], np.abs(emu - ref[0])
    print('C4 composed, 16 images vs fp32 oracle: probability error HIP bf16 mean %.3e max %.3e; worst per-image mask agreement %.5f '
          '(%d of %d decisions differ); %.2f %% of the reference probabilities within 0.05 of the threshold | image 0: HIP mean %.3e max %.3e, '
          'emulated bf16 storage mean %.3e max %.3e | training loss %.3f -> %.3f'
          % (err.mean(), err.max(), worst_agree, total_dis, decisions, 100 * near, err0.mean(), err0.max(), err_emu.mean(), err_emu.max(),
             losses[0], losses[-1]))
    # measured (round 4): HIP 8.9e-5 / 5.9e-4 (mean / max) on image 0 against 1.06e-4 / 5.7e-4 for the emulation; 3.0e-3 max over the batch
    assert err0.mean() <= 1.5 * err_emu.mean() + 1e-4 and err0.max() <= 2.0 * err_emu.max() + 2e-3, (err0.mean(), err_emu.mean(), err0.max(), err_emu.max())
    assert err.max() <= 2e-2 and err.mean() <= 5e-4, (err.max(), err.mean())
    record_parity('C4_r152_tta_bf16_crop_threshold_masks',
                  config='[16,3,256,256] bf16, 4-flip TTA mean, crop 202, > 0.5; R152 hypercolumn conditioned by %d fused training steps; all 16 images vs the fp32 oracle pipeline' % STEPS,
                  decisions=decisions, differ=total_dis, worst_image_agreement=worst_agree, prob_error_mean=float(err.mean()), prob_error_max=float(err.max()),
                  frac_ref_within_0p05_of_threshold=near, image0_prob_error_mean=float(err0.mean()), image0_emulated_bf16_storage_error_mean=float(err_emu.mean()))


def test_c3_r34_batch64_train_step_fp32_vs_oracle():
    """BASELINE C3's per-GPU shape ([64,3,128,128], R34 hypercolumn, Lovasz + Adam) through SegmentationModel._fit_loop in fp32
    against ONE oracle step: logits, loss, every gradient norm (the batch-64 tiling: twice the pixel tiles, 64 Lovasz sorts,
    other split-K factors than batch 32)."""
    from oracle import nets as ON, specs as OS, losses as OL
    from test_gpu_fused_step import _segmentation_model
    from test_gpu_models import _grad_report
    torch.manual_seed(5)
    m = _segmentation_model('UNetResNet', 'lovasz', dtype='f32')
    spec = OS.SPECS['UNetResNet'](with_fc=True)
    sd = OS.init_state(spec, seed=7)
    m.model.load_state_dict({k: sd[k] for k in m.model.state_dict() if k in sd}, strict=False)
    sd = {k: v.detach().clone() for k, v in m.model.state_dict().items() if k in spec}
    x = CF.input_for('c3', (64, 3, 128, 128))
    t = CF.mask_for('c3', (64, 128, 128))
    m._to_device()
    m.model.train()
    dead = set(m.model.dead_parameter_names())
    keys = [k for k in OS.trainable_keys(spec) if k not in dead]
    for k in keys:
        sd[k].requires_grad_(True)
    out_r = ON.unet_resnet(sd, x, True)
    loss_r = OL.lovasz_loss(out_r, t)
    loss_r.backward()
    metrics = m._fit_loop([x, t])
    torch.cuda.synchronize()
    cnet = m.model.engine().net((64, 3, 128, 128), True)
    assert_close(cnet.logits.cpu(), out_r.detach(), 1e-3, 'train-mode logits, batch 64')
    assert abs(float(metrics['sum']) - float(loss_r)) < 1e-4 * max(1.0, abs(float(loss_r)))
    worst, cos, n = _grad_report(m.model, {k: sd[k].grad for k in keys})
    print('C3 shape: worst per-tensor gradient rel-L2 %.3e (%s), global cosine %.6f over %d tensors' % (worst[0], worst[1], cos, n))
    # Lovasz gradients carry ~1e-3 of fp32 cancellation noise (g_k = J_k - J_(k-1)), amplified by train-mode BN: same bounds as the
    # batch-32 test of the fused step
    assert n > 120 and cos > 0.999 and worst[0] < 5e-2, (worst, cos, n)


def test_r152_hypercolumn_trains_at_c3_batch_64():
    """VERDICT r5 missing #5: ResNet152 hypercolumn TRAINING at BASELINE C3's per-GPU batch (64) stopped with SALT_E_LDS in the scSE FC
    backward (256-channel decoders, one workgroup holding all 64 images' vectors).  Round 6 tiles the batch (se.hip): two fused
    steps at [64,3,128,128] bf16 run, the loss is finite and moves, every parameter that gets a gradient in the reference gets one."""
    from test_gpu_fused_step import _segmentation_model
    import bench
    torch.manual_seed(13)
    m = _segmentation_model('UNetResNet152', 'lovasz', dtype='bf16', lr=2e-4)
    m._to_device()
    m.model.train()
    img, msk = bench.synth_tiles(64, seed=91)
    X, Tt = bench.preprocess(img, msk, True, 3)
    X, Tt = X.to(DEV), Tt.to(DEV)
    losses = [float(m._fit_loop([X, Tt])['sum']) for _ in range(2)]
    torch.cuda.synchronize()
    eng = m.model.engine()
    assert np.isfinite(losses).all() and losses[1] != losses[0], losses
    g = eng.grads
    assert bool(torch.isfinite(g).all())
    dead = set(m.model.dead_parameter_names())
    for k, p in m.model.named_parameters():
        if k in dead or not k.startswith('dec') or '_se.' not in k:
            continue
        off, n = eng.grad_range(p)
        assert float(g[off:off + n].abs().max()) > 0.0, k                # the scSE parameters of every decoder received a gradient
