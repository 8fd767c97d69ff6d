#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN MODULES.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
The reference never travels to the GPU box; only the .npz data written here does.  Fixture ids
follow SURVEY.md §8c (F1..F10).  Weights/inputs are closed-form (closed_form.py), keyed by name.

Everything is computed by the reference source under torch 2.10 CPU fp32 through the import stubs
of ref_import.py.  Functions that live in reference files which cannot be imported at all
(augmentation.py needs cv2/imgaug, postprocessing.py needs skimage, utils.py needs py<3.10) are
pulled out of their source file with ``ast`` and executed unmodified against numpy.
"""
import ast
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import closed_form as CF          # noqa: E402
import ref_import as R            # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(8)


ONLY = set(sys.argv[1:])          # e.g. `make_golden.py F8_unet_resnet152_hyper` regenerates just that fixture


def save(name, **arrays):
    if ONLY and name not in ONLY:
        return
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote %-28s %7.1f KB' % (name + '.npz', os.path.getsize(path) / 1024))


def canonical_fn(module):
    first = {}
    for k, v in module.state_dict().items():
        first.setdefault(v.data_ptr() if v.numel() else ('e', k), k)
    amap = {k: first[v.data_ptr() if v.numel() else ('e', k)] for k, v in module.state_dict().items()}
    return lambda k: amap[k]


def run_block(mod, x, train, extra_inputs=()):
    """fwd + bwd of a module with upstream grad = closed-form tensor; returns dict of arrays."""
    mod.train(train)
    x = x.clone().requires_grad_(True)
    extras = [e.clone().requires_grad_(True) for e in extra_inputs]
    y = mod(x, *extras)
    gy = CF.input_for('gy:%s' % (tuple(y.shape),), y.shape)
    y.backward(gy)
    out = OrderedDict(x=x, y=y, gy=gy, gx=x.grad)
    for i, e in enumerate(extras):
        out['e%d' % i] = e
        out['ge%d' % i] = e.grad
    for k, p in mod.named_parameters():
        out['g:' + k] = p.grad if p.grad is not None else torch.zeros_like(p)
    for k, v in mod.state_dict().items():
        out['s:' + k] = v                     # state AFTER the pass (running stats updated in train mode)
    return out


def block_fixture(name, make, x, extra_inputs=()):
    for train in (True, False):
        mod = make()
        CF.fill_module(mod)
        save('%s_%s' % (name, 'train' if train else 'eval'), **run_block(mod, x, train, extra_inputs))


def extract_functions(relpath, names, namespace):
    src = open(os.path.join(R.REFERENCE_ROOT, relpath)).read()
    tree = ast.parse(src)
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            code = compile(ast.Module(body=[node], type_ignores=[]), relpath, 'exec')
            exec(code, namespace)
    return namespace


def main():
    assert R.reference_available(), 'reference not mounted'
    base = R.load('architectures.base')
    um = R.load('unet_models')
    unet = R.load('architectures.unet')
    ll = R.load('lovasz_losses')
    models = R.load_models_module()

    # ---- F1: base.Conv2dBnRelu, odd sizes, (3,3) (3,1) (1,3) kernels
    x = CF.input_for('f1', (2, 3, 9, 11))
    block_fixture('F1_conv2dbnrelu_k33', lambda: base.Conv2dBnRelu(3, 8), x)
    block_fixture('F1_conv2dbnrelu_k31', lambda: base.Conv2dBnRelu(3, 8, kernel_size=(3, 1)), x)
    block_fixture('F1_conv2dbnrelu_k13', lambda: base.Conv2dBnRelu(3, 8, kernel_size=(1, 3)), x)
    # ---- F2: unet_models.ConvBnRelu
    block_fixture('F2_convbnrelu', lambda: um.ConvBnRelu(3, 8), x)
    # ---- F3: DecoderBlockV1 / V2 (deconv + upsample branch) / DeconvConv2dBnRelu
    x3 = CF.input_for('f3', (2, 6, 5, 7))
    block_fixture('F3_decoderv1', lambda: um.DecoderBlockV1(6, 8, 4), x3)
    block_fixture('F3_decoderv2_deconv', lambda: um.DecoderBlockV2(6, 8, 4, is_deconv=True), x3)
    block_fixture('F3_decoderv2_upsample', lambda: um.DecoderBlockV2(6, 8, 4, is_deconv=False), x3)
    block_fixture('F3_deconvconv2dbnrelu', lambda: base.DeconvConv2dBnRelu(6, 4), x3)
    # ---- F4: base.DecoderBlock with / without skip (+ cSE / sSE alone)
    x4 = CF.input_for('f4', (2, 8, 4, 6))
    e4 = CF.input_for('f4e', (2, 5, 8, 12))
    block_fixture('F4_decoderblock_skip', lambda: base.DecoderBlock(13, 16, 32), x4, (e4,))
    block_fixture('F4_decoderblock_noskip', lambda: base.DecoderBlock(8, 16, 32), x4)
    xs = CF.input_for('f4s', (2, 32, 5, 6))
    block_fixture('F4_channel_se', lambda: base.ChannelSELayer(32, reduction=16), xs)
    block_fixture('F4_spatial_se', lambda: base.SpatialSELayer(32), xs)

    # ---- F5: torch ops the reference calls: MaxPool2d(2,2) with ties, MaxPool 3x3 s2 p1, AvgPool2d(2,2), bilinear x2..x16
    xp = torch.round(CF.input_for('f5', (2, 4, 8, 10)) * 2) / 2          # quantised -> ties
    out = {}
    for tag, fn in (('max2', lambda t: F.max_pool2d(t, 2, 2)), ('max3s2', lambda t: F.max_pool2d(t, 3, 2, 1)),
                    ('avg2', lambda t: F.avg_pool2d(t, 2, 2))):
        t = xp.clone().requires_grad_(True)
        y = fn(t)
        gy = CF.input_for('gy5' + tag, y.shape)
        y.backward(gy)
        out.update({tag + '_y': y, tag + '_gy': gy, tag + '_gx': t.grad})
    out['x'] = xp
    xb = CF.input_for('f5b', (2, 3, 4, 6))
    out['xb'] = xb
    for r in (2, 4, 8, 16):
        t = xb.clone().requires_grad_(True)
        y = F.upsample(t, scale_factor=r, mode='bilinear')             # exactly the reference's call (unet.py:103-106)
        gy = CF.input_for('gy5b%d' % r, y.shape)
        y.backward(gy)
        out.update({'up%d_y' % r: y, 'up%d_gy' % r: gy, 'up%d_gx' % r: t.grad})
    save('F5_pool_upsample', **out)

    # ---- F6: lovasz_hinge via models.lovasz_loss (target.long()), fwd + grad
    out = {}
    cases = {}
    z = CF.input_for('f6', (3, 2, 12, 10), 2.0)
    t = CF.mask_for('f6', (3, 12, 10))
    cases['random'] = (z, t)
    cases['all0'] = (z, torch.zeros_like(t))
    cases['all1'] = (z, torch.ones_like(t))
    cases['p1'] = (CF.input_for('f6p1', (2, 1, 1, 1)), torch.tensor([1.0, 0.0]).view(2, 1, 1, 1))
    zt = torch.round(z * 2) / 2                                           # constructed ties
    cases['ties'] = (zt, t)
    cases['big'] = (CF.input_for('f6big', (2, 2, 32, 32), 3.0), CF.mask_for('f6big', (2, 32, 32)))
    for tag, (zz, tt) in cases.items():
        zv = zz.clone().requires_grad_(True)
        loss = models.lovasz_loss(zv, tt)
        loss.backward()
        out.update({tag + '_z': zz, tag + '_t': tt, tag + '_loss': loss, tag + '_gz': zv.grad})
        zv2 = zz.clone().requires_grad_(True)
        out[tag + '_loss_batch'] = ll.lovasz_hinge(zv2, tt.long(), per_image=False)
    out['kat1'] = ll.lovasz_grad(torch.tensor([1, 0, 1, 1, 0]))
    save('F6_lovasz', **out)

    # ---- F7: BCE + Dice
    out = {}
    zv = z.clone().requires_grad_(True)
    loss = models.mixed_dice_bce_loss(zv, t.clone())
    loss.backward()
    out.update(z=z, t=t, loss=loss, gz=zv.grad)
    out['dice'] = models.DiceLoss()(torch.sigmoid(z), t)
    out['mc_dice'] = models.multiclass_dice_loss(z, t.clone(), 0, 'sigmoid')
    save('F7_bce_dice', **out)

    # ---- F8: whole models at 64x64, B=2: eval logits + mask; one _fit_loop-equivalent train step
    X = CF.input_for('f8', (2, 3, 64, 64))
    T = CF.mask_for('f8', (2, 64, 64))
    nets = {'unet_resnet34_hyper': lambda: unet.UNetResNet(34, 2, dropout_2d=0.0, pretrained=False, use_hypercolumn=True),
            'ternaus_resnet34_deconv': lambda: um.UNetResNet(34, 2, dropout_2d=0.0, pretrained=False, is_deconv=True),
            'ternaus_resnet34_upsample': lambda: um.UNetResNet(34, 2, dropout_2d=0.0, pretrained=False, is_deconv=False),
            'salt_unet': lambda: um.SaltUNet(2, dropout_2d=0.0, is_deconv=True),
            'salt_linknet': lambda: um.SaltLinkNet(2, dropout_2d=0.0, is_deconv=True),
            # Bottleneck encoders (BASELINE C4 runs the ResNet152 hypercolumn net)
            'unet_resnet152_hyper': lambda: unet.UNetResNet(152, 2, dropout_2d=0.0, pretrained=False, use_hypercolumn=True),
            'ternaus_resnet101_deconv': lambda: um.UNetResNet(101, 2, dropout_2d=0.0, pretrained=False, is_deconv=True)}
    only = [a for a in sys.argv[1:] if a.startswith('F8_')]
    for tag, make in nets.items():
        if ONLY and 'F8_' + tag not in ONLY:
            continue
        net = make()
        canon = canonical_fn(net)
        CF.fill_module(net, canonical=canon)
        net.eval()
        with torch.no_grad():
            logits = net(X)
        out = OrderedDict(x=X, t=T, eval_logits=logits, eval_mask=(logits[:, 1] > 0).to(torch.uint8))
        out['keys'] = np.array(list(net.state_dict().keys()))
        # one training step exactly as SegmentationModel._fit_loop (models.py:105-136)
        net.train()
        params = [p for p in net.parameters() if p.requires_grad]
        opt = torch.optim.Adam([{'params': params, 'weight_decay': 1e-4}], lr=1e-4)
        opt.zero_grad()
        o = net(X)
        loss = models.lovasz_loss(o, T) * 1.0
        loss.backward()
        out['train_logits'] = o
        out['train_loss'] = loss
        names, gnorm, gsum, has_grad = [], [], [], []
        for k, p in net.named_parameters():
            names.append(k)
            has_grad.append(p.grad is not None)
            g = p.grad if p.grad is not None else torch.zeros_like(p)
            gnorm.append(float(g.double().norm()))
            gsum.append(float(g.double().sum()))
        opt.step()
        out['param_names'] = np.array(names)
        out['param_has_grad'] = np.array(has_grad)
        out['grad_norm'] = np.array(gnorm)
        out['grad_sum'] = np.array(gsum)
        out['post_norm'] = np.array([float(p.detach().double().norm()) for _, p in net.named_parameters()])
        out['post_sum'] = np.array([float(p.detach().double().sum()) for _, p in net.named_parameters()])
        sd = net.state_dict()
        bn_keys = [k for k in sd if k.endswith('running_mean') or k.endswith('running_var')]
        bn_keys = [k for k in bn_keys if canon(k) == k]
        out['bn_keys'] = np.array(bn_keys)
        out['bn_sum'] = np.array([float(sd[k].double().sum()) for k in bn_keys])
        # a couple of full gradients for elementwise checks
        named = dict(net.named_parameters())
        for k in [n for n in names if n.endswith('final.1.weight') or n.endswith('final.weight')
                  or n.endswith('dec1.conv2.conv.weight') or n.endswith('encoder.conv1.weight')][:4]:
            out['fullgrad:' + k] = named[k].grad
        save('F8_' + tag, **out)

    # ---- F11: the deep (Bottleneck) networks once more in float64 - the same reference modules, `.double()` - so that the tests can
    # EARN their gradient tolerances: |HIP - f64| is compared with |reference fp32 - f64| instead of with a chosen constant
    for tag in ('unet_resnet152_hyper', 'ternaus_resnet101_deconv'):
        if ONLY and 'F11_' + tag + '_f64' not in ONLY:
            continue
        net = nets[tag]()
        CF.fill_module(net, canonical=canonical_fn(net))
        net.double().train()
        o = net(X.double())
        # the loss keeps the reference's fp32 arithmetic (lovasz_losses.py:107 casts the labels to float; models.py:326-328), the
        # network forward/backward is float64
        o32 = o.detach().float().requires_grad_(True)
        loss = models.lovasz_loss(o32, T) * 1.0
        loss.backward()
        o.backward(o32.grad.double())
        out = OrderedDict(train_logits64=o.detach(), train_loss64=loss.detach())
        names, gnorm = [], []
        for k, p in net.named_parameters():
            names.append(k)
            gnorm.append(float(p.grad.norm()) if p.grad is not None else 0.0)
        out['param_names'] = np.array(names)
        out['grad_norm64'] = np.array(gnorm)
        named = dict(net.named_parameters())
        for k in [n for n in names if n.endswith('final.1.weight') or n.endswith('final.weight')
                  or n.endswith('dec1.conv2.conv.weight') or n.endswith('encoder.conv1.weight')][:4]:
            out['fullgrad64:' + k] = named[k].grad
        save('F11_' + tag + '_f64', **out)

    # ---- F13: architectures.unet.UNetResNet(34, hypercolumn, pool0=True): the stem max-pool 3x3 s2 p1 (encoders.py:23-27); the
    # reference's logits then come out at half the input resolution, so the training step uses the 2x sub-sampled target
    if not ONLY or 'F13_unet_resnet34_hyper_pool0' in ONLY:
        net = unet.UNetResNet(34, 2, dropout_2d=0.0, pretrained=False, use_hypercolumn=True, pool0=True)
        canon = canonical_fn(net)
        CF.fill_module(net, canonical=canon)
        net.eval()
        with torch.no_grad():
            logits = net(X)
        out = OrderedDict(x=X, t=T[:, :, ::2, ::2].contiguous(), eval_logits=logits, eval_mask=(logits[:, 1] > 0).to(torch.uint8))
        out['keys'] = np.array(list(net.state_dict().keys()))
        net.train()
        params = [p for p in net.parameters() if p.requires_grad]
        opt = torch.optim.Adam([{'params': params, 'weight_decay': 1e-4}], lr=1e-4)
        opt.zero_grad()
        o = net(X)
        loss = models.lovasz_loss(o, out['t']) * 1.0
        loss.backward()
        out['train_logits'] = o
        out['train_loss'] = loss
        names, gnorm, has_grad = [], [], []
        for k, p in net.named_parameters():
            names.append(k)
            has_grad.append(p.grad is not None)
            gnorm.append(float(p.grad.double().norm()) if p.grad is not None else 0.0)
        out['param_names'] = np.array(names)
        out['param_has_grad'] = np.array(has_grad)
        out['grad_norm'] = np.array(gnorm)
        save('F13_unet_resnet34_hyper_pool0', **out)

    # ---- F9 / F10: numpy helpers pulled from files that cannot be imported (executed unmodified)
    ns = {'np': np}
    extract_functions('common_blocks/augmentation.py',
                      {'test_time_augmentation_transform', 'test_time_augmentation_inverse_transform',
                       'per_channel_flipud', 'per_channel_fliplr', 'per_channel_rotation', 'rotate'}, ns)
    extract_functions('common_blocks/utils.py', {'get_crop_pad_sequence', 'sigmoid', 'AddDepthChannels'}, ns)
    extract_functions('common_blocks/postprocessing.py', {'crop_image', 'binarize'}, ns)
    extract_functions('common_blocks/metrics.py', {'compute_precision_at'}, ns)
    img = CF.input_for('f9', (8, 8, 3)).numpy()
    pred = CF.input_for('f9p', (4, 2, 8, 8)).numpy()
    out = {'img': img, 'pred': pred}
    specs = []
    for ud in (False, True):
        for lr in (False, True):
            for rot in (0, 90):
                specs.append({'ud_flip': ud, 'lr_flip': lr, 'rotation': rot, 'color_shift': False})
    for i, s in enumerate(specs):
        out['fwd%d' % i] = np.ascontiguousarray(ns['test_time_augmentation_transform'](img, s))
        out['inv%d' % i] = np.ascontiguousarray(ns['test_time_augmentation_inverse_transform'](pred[i % 4], s))
    out['specs'] = np.array([[s['ud_flip'], s['lr_flip'], s['rotation']] for s in specs])
    save('F9_tta', **out)

    out = {}
    p128 = CF.input_for('f10', (2, 128, 128)).numpy()
    out['p128'] = p128
    out['crop101'] = ns['crop_image'](p128, (101, 101))
    out['crop_seq_27'] = np.array(ns['get_crop_pad_sequence'](27, 27))
    out['crop_seq_155'] = np.array(ns['get_crop_pad_sequence'](155, 155))
    out['sigmoid'] = ns['sigmoid'](p128[:, :4, :4])
    out['binarize'] = ns['binarize'](ns['sigmoid'](p128), 0.5)
    t3 = torch.from_numpy(CF.input_for('f10d', (3, 6, 5)).numpy().copy())
    out['depth_in'] = t3.numpy().copy()
    out['depth_out'] = ns['AddDepthChannels']()(t3).numpy()
    ths = [0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 0.95]
    ious = np.array([0.0, 0.3, 0.5, 0.62, 0.8, 0.95, 1.0])
    out['iou_values'] = ious
    out['iout_values'] = np.array([np.mean([ns['compute_precision_at'](np.array([[v]]), th) for th in ths]) for v in ious])
    save('F10_post_metric', **out)

    # ---- F12: run-length encoding / decoding of submission masks (utils.py:99-133), executed unmodified
    extract_functions('common_blocks/utils.py', {'run_length_encoding', 'run_length_decoding'}, ns)
    r = np.random.RandomState(12)
    yy, xx = np.mgrid[0:101, 0:101]
    masks = [np.zeros((101, 101), np.uint8), np.ones((101, 101), np.uint8)]
    m = np.zeros((101, 101), np.uint8); m[0, 0] = 1; masks.append(m)
    m = np.zeros((101, 101), np.uint8); m[100, 100] = 1; m[100, 0] = 1; m[0, 100] = 1; masks.append(m)
    m = np.zeros((101, 101), np.uint8); m[:, 50] = 1; m[37, :] = 1; masks.append(m)          # runs that wrap a column end
    for _ in range(5):
        cy, cx, ry, rx = r.uniform(0, 101), r.uniform(0, 101), r.uniform(5, 60), r.uniform(5, 60)
        masks.append(((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2) <= 1).astype(np.uint8))
    masks.append((r.rand(101, 101) > 0.5).astype(np.uint8))
    m = (r.rand(7, 13) > 0.4).astype(np.uint8); masks_small = [m]
    out = {'masks': np.stack(masks), 'small': masks_small[0]}
    flat, offs = [], [0]
    for mk in masks + masks_small:
        rle = ns['run_length_encoding'](mk)
        dec = ns['run_length_decoding'](' '.join(str(v) for v in rle), mk.shape) if rle else np.zeros(mk.shape, np.uint8)
        assert np.array_equal(dec, mk)
        flat += list(rle); offs.append(len(flat))
    out['rle_flat'] = np.array(flat, np.int64)
    out['rle_offsets'] = np.array(offs, np.int64)
    save('F12_rle', **out)


if __name__ == '__main__':
    main()
