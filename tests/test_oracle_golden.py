"""CPU tests: the oracle restatement vs the golden vectors the reference itself produced, and vs the
known-answer values KAT-1..KAT-11 of SURVEY.md §8c.  This is what pins the oracle."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden, T, state_from_fixture, assert_close
import closed_form as CF
from oracle import blocks as OB, nets as ON, losses as OL, metrics as OM, specs as OS

TOL = 2e-5


def _run(fn, sd, x, extras=()):
    x = x.clone().requires_grad_(True)
    extras = [e.clone().requires_grad_(True) for e in extras]
    leaves = {k: v for k, v in sd.items() if v.dtype.is_floating_point and not k.endswith(('running_mean', 'running_var'))}
    for v in leaves.values():
        v.requires_grad_(True)
    y = fn(x, *extras)
    return x, extras, leaves, y


BLOCKS = {
    'F1_conv2dbnrelu_k33': lambda sd, tr: (lambda x: OB.conv2d_bn_relu(sd, '', x, tr)),
    'F1_conv2dbnrelu_k31': lambda sd, tr: (lambda x: OB.conv2d_bn_relu(sd, '', x, tr)),
    'F1_conv2dbnrelu_k13': lambda sd, tr: (lambda x: OB.conv2d_bn_relu(sd, '', x, tr)),
    'F2_convbnrelu': lambda sd, tr: (lambda x: OB.conv_bn_relu(sd, '', x, tr)),
    'F3_decoderv1': lambda sd, tr: (lambda x: OB.decoder_block_v1(sd, '', x, tr)),
    'F3_decoderv2_deconv': lambda sd, tr: (lambda x: OB.decoder_block_v2(sd, '', x, tr, True)),
    'F3_decoderv2_upsample': lambda sd, tr: (lambda x: OB.decoder_block_v2(sd, '', x, tr, False)),
    'F3_deconvconv2dbnrelu': lambda sd, tr: (lambda x: OB.deconv_conv2d_bn_relu(sd, '', x, tr)),
    'F4_decoderblock_skip': lambda sd, tr: (lambda x, e: OB.decoder_block(sd, '', x, e, tr)),
    'F4_decoderblock_noskip': lambda sd, tr: (lambda x: OB.decoder_block(sd, '', x, None, tr)),
    'F4_channel_se': lambda sd, tr: (lambda x: OB.channel_se(sd, '', x)),
    'F4_spatial_se': lambda sd, tr: (lambda x: OB.spatial_se(sd, '', x)),
}


@pytest.mark.parametrize('name', sorted(BLOCKS))
@pytest.mark.parametrize('mode', ['train', 'eval'])
def test_block_matches_reference(name, mode):
    fx = golden('%s_%s' % (name, mode))
    sd = state_from_fixture(fx)
    extras = [T(fx['e0'])] if 'e0' in fx else []
    x, ex, leaves, y = _run(BLOCKS[name](sd, mode == 'train'), sd, T(fx['x']), extras)
    assert_close(y, fx['y'], TOL, 'y')
    y.backward(T(fx['gy']))
    assert_close(x.grad, fx['gx'], 5e-5, 'gx')
    for i, e in enumerate(ex):
        assert_close(e.grad, fx['ge%d' % i], 5e-5, 'ge%d' % i)
    for k, v in leaves.items():
        if ('g:' + k) in fx and v.grad is not None:
            ref = fx['g:' + k]
            if np.abs(ref).max() < 1e-5:      # conv bias in front of train-mode BN: mathematically zero
                assert float(v.grad.abs().max()) < 1e-4, k
            else:
                assert_close(v.grad, ref, 2e-4, 'g:' + k)
    for k in sd:                               # running statistics after the pass
        if k.endswith(('running_mean', 'running_var')):
            assert_close(sd[k], fx['s:' + k], TOL, k)


def test_pool_upsample_ops():
    fx = golden('F5_pool_upsample')
    x = T(fx['x'])
    for tag, fn in (('max2', lambda t: F.max_pool2d(t, 2, 2)), ('max3s2', lambda t: F.max_pool2d(t, 3, 2, 1)),
                    ('avg2', lambda t: F.avg_pool2d(t, 2, 2))):
        t = x.clone().requires_grad_(True)
        y = fn(t)
        y.backward(T(fx[tag + '_gy']))
        assert_close(y, fx[tag + '_y'], 0, tag)
        assert_close(t.grad, fx[tag + '_gx'], 1e-6, tag + ' grad')
    for r in (2, 4, 8, 16):
        t = T(fx['xb']).clone().requires_grad_(True)
        y = OB.upsample_bilinear(t, r)
        y.backward(T(fx['up%d_gy' % r]))
        assert_close(y, fx['up%d_y' % r], 1e-6, 'up%d' % r)
        assert_close(t.grad, fx['up%d_gx' % r], 1e-5, 'up%d grad' % r)


@pytest.mark.parametrize('case', ['random', 'all0', 'all1', 'p1', 'ties', 'big'])
def test_lovasz_matches_reference(case):
    fx = golden('F6_lovasz')
    z = T(fx[case + '_z']).clone().requires_grad_(True)
    t = T(fx[case + '_t'])
    loss = OL.lovasz_loss(z, t)
    loss.backward()
    assert abs(float(loss) - float(fx[case + '_loss'])) <= 2e-6 * max(1, abs(float(fx[case + '_loss'])))
    if case != 'ties':
        assert_close(z.grad, fx[case + '_gz'], 1e-5, 'grad')
    else:   # per-element gradients are order dependent among ties; compare sums per (image, error value, label)
        e = (1 - fx[case + '_z'] * (2 * fx[case + '_t'] - 1)).reshape(3, -1)
        g1 = z.grad.numpy().reshape(3, -1)
        g2 = fx[case + '_gz'].reshape(3, -1)
        for b in range(3):
            for v in np.unique(e[b]):
                m = e[b] == v
                assert abs(g1[b][m].sum() - g2[b][m].sum()) < 1e-6
    lb = OL.lovasz_hinge(T(fx[case + '_z']), T(fx[case + '_t']).long(), per_image=False)
    assert abs(float(lb) - float(fx[case + '_loss_batch'])) <= 3e-6 * max(1, abs(float(lb)))
    # closed-form gradient agrees with autograd (no ties)
    if case in ('random', 'big', 'all0', 'all1'):
        lv, g = OL.lovasz_hinge_grad_closed_form(T(fx[case + '_z']), T(fx[case + '_t']), dtype=torch.float32)
        assert abs(lv - float(fx[case + '_loss'])) < 1e-5
        assert_close(g, fx[case + '_gz'], 1e-5, 'closed form grad (fp32 op sequence)')
        lv, g = OL.lovasz_hinge_grad_closed_form(T(fx[case + '_z']), T(fx[case + '_t']))
        assert abs(lv - float(fx[case + '_loss'])) < 1e-5
        assert_close(g.float(), fx[case + '_gz'], 1e-3, 'closed form grad (float64: g_k cancellation bound)')


def test_bce_dice_matches_reference():
    fx = golden('F7_bce_dice')
    z = T(fx['z']).clone().requires_grad_(True)
    loss = OL.mixed_dice_bce_loss(z, T(fx['t']))
    loss.backward()
    assert abs(float(loss) - float(fx['loss'])) < 2e-6
    assert_close(z.grad, fx['gz'], 1e-5, 'grad')
    assert abs(float(OL.dice_loss(torch.sigmoid(T(fx['z'])), T(fx['t']))) - float(fx['dice'])) < 2e-6
    assert abs(float(OL.multiclass_dice_loss(T(fx['z']), T(fx['t']))) - float(fx['mc_dice'])) < 2e-6


def test_kat_values():
    """KAT-1..KAT-11: RNG-free known answers captured from the reference."""
    np.testing.assert_allclose(OL.lovasz_grad(torch.tensor([1, 0, 1, 1, 0])).numpy(), [1 / 3, 1 / 6, 1 / 4, 1 / 4, 0], atol=1e-7)
    logits = torch.linspace(-2, 2, 64).view(2, 2, 4, 4).clone().requires_grad_(True)
    labels = (torch.arange(64) % 3 == 0).float().view(2, 2, 4, 4)
    l2 = OL.lovasz_hinge(logits, labels.long(), per_image=True)
    l2.backward()
    assert abs(float(l2) - 1.79754961) < 2e-6
    assert abs(float(logits.grad.sum()) + 0.30850735) < 2e-6
    assert abs(float(logits.grad.abs().sum()) - 0.96475732) < 2e-6
    np.testing.assert_allclose(logits.grad[0, 0, 0].numpy(), [-0.0454545, 0, 0, -0.0454546], atol=2e-6)
    assert abs(float(OL.lovasz_hinge(logits.detach(), labels.long(), per_image=False)) - 1.70986664) < 2e-6
    assert abs(float(OL.lovasz_hinge(logits.detach(), torch.zeros_like(labels).long())) - 1.98412704) < 2e-6
    assert abs(float(OL.mixed_dice_bce_loss(logits.detach(), labels)) - 0.88487631) < 2e-6
    assert abs(float(OL.dice_loss(torch.sigmoid(logits.detach()), labels)) - 0.59259260) < 2e-6
    assert abs(float(OL.multiclass_dice_loss(logits.detach(), labels)) - 0.59855360) < 2e-6
    x = torch.arange(16.).view(1, 1, 4, 4)
    w = torch.arange(1., 10.).view(1, 1, 3, 3)
    bn = {'weight': torch.ones(1), 'bias': torch.zeros(1), 'running_mean': torch.zeros(1), 'running_var': torch.ones(1)}
    sd = {'conv.weight': w, 'conv.bias': torch.zeros(1)}
    sd.update({'batch_norm.' + k: v.clone() for k, v in bn.items()})
    k8 = OB.conv2d_bn_relu(sd, '', x, False).flatten() * (1 + 1e-5) ** 0.5
    np.testing.assert_allclose(k8.numpy(), [51, 96, 123, 135, 147, 192, 219, 231, 303, 348, 375, 387, 483, 528, 555, 567], rtol=1e-6)
    sd = {'conv.0.weight': w, 'conv.0.bias': torch.zeros(1)}
    sd.update({'conv.1.' + k: v.clone() for k, v in bn.items()})
    k9 = OB.conv_bn_relu(sd, '', x, False).flatten() * (1 + 1e-5) ** 0.5
    np.testing.assert_allclose(k9.numpy(), [83, 139, 178, 121, 198, 303, 348, 225, 330, 483, 528, 333, 181, 253, 274, 163], rtol=1e-6)
    x2 = torch.arange(4.).view(1, 1, 2, 2)
    k10 = F.conv_transpose2d(x2, w, None, 2, 1, 1).flatten()
    np.testing.assert_allclose(k10.numpy(), [0, 4, 5, 6, 4, 16, 14, 18, 10, 24, 15, 18, 16, 39, 24, 27])
    k11 = F.conv_transpose2d(x2, torch.arange(1., 17.).view(1, 1, 4, 4), None, 2, 1).flatten()
    np.testing.assert_allclose(k11.numpy(), [0, 5, 6, 7, 4, 18, 24, 20, 12, 42, 48, 36, 20, 49, 54, 33])


NETS = {'unet_resnet34_hyper': ('UNetResNet', {}), 'ternaus_resnet34_deconv': ('TernausUNetResNet', {'is_deconv': True}),
        'ternaus_resnet34_upsample': ('TernausUNetResNet', {'is_deconv': False}),
        'salt_unet': ('SaltUNet', {}), 'salt_linknet': ('SaltLinkNet', {}),
        # Bottleneck encoders (C4 runs ResNet152): spec / forward take the depth
        'unet_resnet152_hyper': ('UNetResNet', {'depth': 152}), 'ternaus_resnet101_deconv': ('TernausUNetResNet', {'depth': 101, 'is_deconv': True})}


@pytest.mark.parametrize('tag', sorted(NETS))
def test_whole_model_matches_reference(tag):
    fx = golden('F8_' + tag)
    arch, kw = NETS[tag]
    spec = OS.SPECS[arch](with_fc=True, **({'depth': kw['depth']} if 'depth' in kw else {}))
    # key set (with aliases) equals the reference's state_dict
    assert set(OS.expand_aliases(arch, {k: None for k in spec})) == set(fx['keys'].tolist())
    sd = CF.state_for((k, s) for k, (s, _) in spec.items())
    x, t = T(fx['x']), T(fx['t'])
    with torch.no_grad():
        logits = ON.FORWARDS[arch](sd, x, False, **kw)
    assert_close(logits, fx['eval_logits'], 1e-4, 'eval logits')
    assert np.array_equal((logits[:, 1] > 0).numpy().astype(np.uint8), fx['eval_mask'])
    # one training step: loss, gradient norms, post-Adam parameter sums, BN running stats
    train_keys = OS.trainable_keys(spec)
    for k in train_keys:
        sd[k].requires_grad_(True)
    out = ON.FORWARDS[arch](sd, x, True, **kw)
    loss = OL.lovasz_loss(out, t)
    loss.backward()
    assert abs(float(loss) - float(fx['train_loss'])) < 1e-4 * max(1.0, abs(float(fx['train_loss'])))
    names = fx['param_names'].tolist()
    amap = OS.alias_map(arch)
    idx = {n: i for i, n in enumerate(names)}
    checked = 0
    for k in train_keys:
        i = idx[k]
        has = bool(fx['param_has_grad'][i])
        assert (sd[k].grad is not None) == has, k
        if has and fx['grad_norm'][i] > 1e-4:
            gn = float(sd[k].grad.double().norm())
            assert abs(gn - fx['grad_norm'][i]) <= 2e-3 * fx['grad_norm'][i], (k, gn, fx['grad_norm'][i])
            checked += 1
    assert checked > 20
    for k in fx:
        if k.startswith('fullgrad:'):
            assert_close(sd[k[9:]].grad, fx[k], 2e-3, k)
    params = [sd[k] for k in train_keys]
    grads = [p.grad for p in params]
    with torch.no_grad():
        ps = [p.detach() for p in params]
        OL.adam_l2_step(ps, grads, [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps], 1)
    for k, p in zip(train_keys, params):
        i = idx[k]
        if fx['param_has_grad'][i] and fx['grad_norm'][i] > 1e-4:
            assert abs(float(p.detach().double().norm()) - fx['post_norm'][i]) <= 1e-5 * max(fx['post_norm'][i], 1e-3), k
    for k, s in zip(fx['bn_keys'].tolist(), fx['bn_sum'].tolist()):
        assert abs(float(sd[k].double().sum()) - s) <= 1e-4 * max(1.0, abs(s)), k


def test_tta_post_metric():
    fx = golden('F9_tta')
    for i, (ud, lr, rot) in enumerate(fx['specs'].tolist()):
        s = {'ud_flip': bool(ud), 'lr_flip': bool(lr), 'rotation': int(rot)}
        np.testing.assert_array_equal(OM.tta_transform(fx['img'], s), fx['fwd%d' % i])
        np.testing.assert_array_equal(OM.tta_inverse(fx['pred'][i % 4], s), fx['inv%d' % i])
    assert [tuple(sorted(d.items())) for d in OM.tta_specs(True, True)] == \
        [tuple(sorted(d.items())) for d in [{'ud_flip': False, 'lr_flip': False, 'rotation': 0},
                                            {'ud_flip': True, 'lr_flip': True, 'rotation': 0},
                                            {'ud_flip': True, 'lr_flip': False, 'rotation': 0},
                                            {'ud_flip': False, 'lr_flip': True, 'rotation': 0}]]
    fx = golden('F10_post_metric')
    np.testing.assert_array_equal(OM.crop_image(fx['p128'], (101, 101)), fx['crop101'])
    assert OM.crop_image(fx['p128'], (101, 101)).shape == (2, 101, 101)
    np.testing.assert_array_equal(np.array(OM.crop_pad_sequence(27, 27)), fx['crop_seq_27'])
    np.testing.assert_array_equal(np.array(OM.crop_pad_sequence(155, 155)), fx['crop_seq_155'])
    np.testing.assert_allclose(OM.sigmoid(fx['p128'][:, :4, :4]), fx['sigmoid'], rtol=1e-7)
    np.testing.assert_array_equal(OM.binarize(OM.sigmoid(fx['p128'])), fx['binarize'])
    np.testing.assert_array_equal(OM.binarize(OM.sigmoid(fx['p128'])), (fx['p128'][1] > 0).astype(np.uint8))
    np.testing.assert_allclose(OM.add_depth_channels(fx['depth_in'].copy()), fx['depth_out'], rtol=1e-6)
    for v, it in zip(fx['iou_values'], fx['iout_values']):
        assert abs(sum(1.0 if v >= th else 0.0 for th in OM.THRESHOLDS) / 10 - it) < 1e-12
    e = np.zeros((5, 5), np.uint8)
    a = e.copy(); a[:2] = 1
    b = e.copy(); b[1:3] = 1
    assert OM.iou_single(e, e) == 1.0 and OM.iou_single(a, e) == 0.0 and OM.iou_single(e, a) == 0.0
    assert abs(OM.iou_single(a, b) - 5 / 15) < 1e-12


def test_cubic_resize_restatement_kats():
    """f-1 train-branch resize (augmentation.py:79-85 -> imgaug 0.2.5 iaa.Scale default 'cubic' -> cv2.INTER_CUBIC): imgaug / cv2 are
    absent, so the oracle restates the public definition; these known answers pin THAT restatement (hand arithmetic):
    Keys weights with a = -0.75 at t = 0 -> [0, 1, 0, 0], t = 0.5 -> [-3/32, 19/32, 19/32, -3/32], sum 1 everywhere; a constant image
    stays constant; the direct loop form equals the executable torch form on 101 -> 102."""
    from oracle import inputs as OI
    assert np.allclose(OI.cubic_weights(0.0), [0, 1, 0, 0])
    assert np.allclose(OI.cubic_weights(0.5), [-0.09375, 0.59375, 0.59375, -0.09375])
    assert np.allclose(OI.cubic_weights(0.25), [-0.10546875, 0.87890625, 0.26171875, -0.03515625])
    for t in np.linspace(0, 1, 7):
        assert abs(sum(OI.cubic_weights(t)) - 1) < 1e-12
    r = np.random.RandomState(0)
    img = r.rand(13, 11)
    a = OI.resize_cubic_numpy(img, 14, 12)
    b = torch.nn.functional.interpolate(torch.from_numpy(img)[None, None], size=(14, 12), mode='bicubic', align_corners=False)[0, 0].numpy()
    assert np.abs(a - b).max() < 1e-12
    assert np.allclose(OI.resize_cubic_numpy(np.full((5, 5), 0.3), 6, 6), 0.3)
    # a step edge: the cubic overshoots (negative lobes) - which the uint8 path saturates, and the mask path rounds back to {0, 1}
    step = np.zeros((1, 8, 8), np.float32); step[:, :, 4:] = 1.0
    X, Tm = OI.preprocess(step, step.astype(np.uint8), True, 1, resize=9, pad=0)
    assert set(np.unique(Tm.numpy()).tolist()) <= {0.0, 1.0}
    assert float(Tm[0, 1].sum()) > 0 and float(Tm[0, 0].sum()) > 0
    g = X[0, 0].numpy() * 0.229 + 0.485
    assert g.min() >= -1e-6 and g.max() <= 1 + 1e-6                         # saturated onto [0, 255] / 255


def test_cv2_fixed_point_cubic_kats():
    """cv2.INTER_CUBIC on uint8 (opencv_python 3.4.0.12, resize.cpp) evaluates the separable cubic in 11-bit fixed point; known answers
    by hand arithmetic that tell it from the float form:
      * t = 0.5: coefficients (-3/32, 19/32, 19/32, -3/32) x 2048 = (-192, 1216, 1216, -192), sum 2048; t = 0.25: (-216, 1800, 536, -72);
      * 101 -> 102, output column 0: fx = 0.5 * 101 / 102 - 0.5 = -0.0049 -> first tap -2, t = 0.9951 -> (0, 8, 2048, -7): the four
        coefficients are rounded one by one and sum to 2049, not 2048;
      * a 1 x 4 row (0, 255, 0, 255) resized to width 4 (identity geometry, t = 0): every output equals its input exactly;
      * the 4 x 4 image below -> 5 x 5: fixed point gives 226 / 59 where the float form rounds to 227 / 60."""
    from oracle import inputs as OI
    assert OI.cv_cubic_coeffs_fixed(0.5) == [-192, 1216, 1216, -192]
    assert OI.cv_cubic_coeffs_fixed(0.25) == [-216, 1800, 536, -72]
    first, coef = OI.cv_axis_tables(101, 102)
    assert first[0] == -2 and coef[0] == [0, 8, 2048, -7] and sum(coef[0]) == 2049
    assert first[101] == 99 and sum(coef[101]) in (2047, 2048, 2049)
    row = np.array([[0, 255, 0, 255]], np.uint8)
    assert OI.resize_cubic_u8_fixed(row, 1, 4).tolist() == row.tolist()
    img = np.array([[19, 133, 248, 83], [57, 45, 4, 70], [176, 190, 162, 87], [191, 211, 224, 207]], np.uint8)
    fixed = OI.resize_cubic_u8_fixed(img, 5, 5)
    assert fixed.tolist() == [[9, 91, 229, 226, 73], [32, 48, 54, 59, 73], [119, 116, 79, 52, 67], [192, 208, 213, 174, 118], [191, 205, 223, 226, 213]]
    flt = torch.nn.functional.interpolate(torch.from_numpy(img.astype(np.float32) / 255)[None, None], size=(5, 5), mode='bicubic', align_corners=False)
    flt = torch.clamp(torch.floor(flt * 255 + 0.5), 0, 255)[0, 0].numpy().astype(np.uint8)
    assert flt[0, 3] == 227 and flt[1, 3] == 60 and int((flt != fixed).sum()) == 2
    # hand check of fixed[0][3]: output row 0 of 4 -> 5: fy = 0.5 * 0.8 - 0.5 = -0.1 -> taps rows -2 .. 1 (clamped: 0, 0, 0, 1), t = 0.9;
    # output column 3: fx = 3.5 * 0.8 - 0.5 = 2.3 -> taps columns 1 .. 4 (4 clamped to 3), t = 0.3
    cy, cx = OI.cv_cubic_coeffs_fixed(np.float32(np.float32(-0.1) + 1)), OI.cv_cubic_coeffs_fixed(np.float32(np.float32(2.3) - 2))
    rows = [0, 0, 0, 1]; cols = [1, 2, 3, 3]
    v = sum(cy[i] * sum(cx[j] * int(img[rows[i], cols[j]]) for j in range(4)) for i in range(4))
    assert min(max((v + (1 << 21)) >> 22, 0), 255) == 226
    # the preprocess entry point takes that path for uint8 tiles, the float form for float tiles
    tile = np.random.RandomState(3).randint(0, 256, (1, 101, 101)).astype(np.uint8)
    X, _ = OI.preprocess(tile.astype(np.float32) / 255, None, True, 1)
    g = np.round((X[0, 0].numpy() * 0.229 + 0.485) * 255).astype(np.uint8)[13:115, 13:115]
    assert (g == OI.resize_cubic_u8_fixed(tile[0], 102, 102)).all()


def test_align_corners_restatement_kat():
    """torch 0.3.1 evaluated nn.Upsample(mode='bilinear') with src = dst (H - 1) / (R H - 1): [0, 1] x2 -> [0, 1/3, 2/3, 1]
    (the torch >= 0.4 default gives [0, 0.25, 0.75, 1])."""
    from oracle import blocks as OB
    x = torch.tensor([0.0, 1.0]).view(1, 1, 1, 2).repeat(1, 1, 2, 1)
    assert torch.allclose(OB.upsample_bilinear(x, 2)[0, 0, 0], torch.tensor([0, 0.25, 0.75, 1.0]))
    OB.ALIGN_CORNERS = True
    try:
        assert torch.allclose(OB.upsample_bilinear(x, 2)[0, 0, 0], torch.tensor([0, 1 / 3, 2 / 3, 1.0]))
    finally:
        OB.ALIGN_CORNERS = False
