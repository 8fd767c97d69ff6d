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


def test_c4_r152_tta_pipeline_bf16_full_shape():
    """R152 + predict_tta (flip_ud x flip_lr) + crop_threshold at [16,3,256,256] bf16.
    (1) image 0 against the fp32 oracle pipeline (numpy flips, 4 oracle forwards, sigmoid, inverse flips, mean, crop, > 0.5);
    (2) equivariance over the whole batch: the TTA mean over the flip group commutes with a flip of the input;
    (3) a probability is a mean of sigmoids: in [0, 1], finite."""
    from salt_amd import architectures as A, inference as I
    from oracle import nets as ON, specs as OS, metrics as OM
    torch.manual_seed(11)
    net = A.UNetResNet(152, 2, use_hypercolumn=True, dropout_2d=0.0, pretrained=False)
    spec = OS.SPECS['UNetResNet'](with_fc=True, depth=152)
    sd = OS.init_state(spec, seed=9)
    net.load_state_dict({k: sd[k] for k in net.state_dict() if k in sd}, strict=False)
    X = CF.input_for('c4', (16, 3, 256, 256))
    Xd = X.to(DEV)
    # a checkpoint's BatchNorm running statistics are calibrated; random init + (0, 1) statistics through 152 layers saturates every
    # sigmoid.  Calibrate them with ONE train-mode forward at momentum 1 (fp32, on the device), then freeze: the test is about eval.
    for mod in net.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.momentum = 1.0
    net.to(DEV).train()
    with torch.no_grad():
        net(Xd)
    net.eval()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items() if k in spec}
    net.set_compute_dtype('bf16')
    prob = I.predict_tta(net, Xd, True, True, depth_channels=False)
    mask = I.crop_threshold(prob, (202, 202), 0.5, cls=1)
    torch.cuda.synchronize()
    p = prob.float().cpu()
    assert torch.isfinite(p).all() and float(p.min()) >= 0.0 and float(p.max()) <= 1.0
    assert tuple(mask.shape) == (16, 202, 202) and mask.dtype == torch.uint8
    # (2) flip the whole input batch left-right: the aggregated probabilities flip with it (same four forwards per image, in
    # another order: bf16 rounding is identical per variant, the fp32 mean is order dependent in its last bit only)
    prob_f = I.predict_tta(net, torch.flip(Xd, dims=[3]).contiguous(), True, True, depth_channels=False).float().cpu()
    assert float((torch.flip(prob_f, dims=[3]) - p).abs().max()) <= 1e-5
    # (1) oracle for image 0: fp32, and fp32 with bf16 STORAGE emulated at the boundaries where the HIP path stores bf16
    # (oracle.blocks.bf16_storage) - the yardstick for what bf16 through 152 layers costs in ANY implementation
    from oracle import blocks as OB
    specs = OM.tta_specs(True, True)

    def oracle_tta(emulate):
        preds = []
        for sp in specs:
            xv = OM.tta_transform(X[0].permute(1, 2, 0).numpy(), sp)
            xv = torch.from_numpy(np.ascontiguousarray(xv)).permute(2, 0, 1)[None]
            with torch.no_grad():
                if emulate:
                    with OB.bf16_storage():
                        o = ON.unet_resnet(sd, xv, False, depth=152)
                else:
                    o = ON.unet_resnet(sd, xv, False, depth=152)
            preds.append(OM.sigmoid(o[0].float().numpy()))
        return OM.tta_aggregate(preds, specs, 'mean')
    ref, emu = oracle_tta(False), oracle_tta(True)
    got = p[0].numpy()
    err, err_emu = np.abs(got - ref), np.abs(emu - ref)
    ref_crop = OM.crop_image(ref, (202, 202))[1]
    ref_mask = ref_crop > 0.5
    agree = float((mask[0].cpu().numpy().astype(bool) == ref_mask).mean())
    agree_emu = float(((OM.crop_image(emu, (202, 202))[1] > 0.5) == ref_mask).mean())
    print('C4 composed, image 0 vs fp32 oracle: probability error HIP bf16 mean %.3e max %.3e | emulated bf16 storage mean %.3e max %.3e | '
          'mask agreement HIP %.5f / emulation %.5f' % (err.mean(), err.max(), err_emu.mean(), err_emu.max(), agree, agree_emu))
    assert err.mean() <= 1.5 * err_emu.mean() + 1e-3 and err.max() <= 2.0 * err_emu.max() + 1e-2, (err.mean(), err_emu.mean(), err.max(), err_emu.max())
    assert agree >= agree_emu - 0.01, (agree, agree_emu)
    # every disagreeing mask pixel has a reference probability within the measured bf16 error of the threshold
    dis = mask[0].cpu().numpy().astype(bool) != ref_mask
    from helpers import record_parity
    record_parity('C4_r152_tta_bf16_crop_threshold_masks', config='[16,3,256,256] bf16, 4-flip TTA mean, crop 202, > 0.5; image 0 vs fp32 oracle pipeline',
                  decisions=int(dis.size), differ=int(dis.sum()), agreement=agree, agreement_emulated_bf16_storage=agree_emu,
                  prob_error_mean=float(err.mean()), prob_error_max=float(err.max()),
                  max_ref_distance_to_threshold_at_differing=float(np.abs(ref_crop - 0.5)[dis].max()) if dis.any() else 0.0)
    assert not dis.any() or float(np.abs(ref_crop - 0.5)[dis].max()) <= float(err.max()) + 1e-6


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
