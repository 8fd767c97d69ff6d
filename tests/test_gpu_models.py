"""GPU parity of whole networks (through the nn.Module / SegmentationModel surface) against the goldens the
reference produced (64x64, B=2) and against the oracle on seeded inputs at BASELINE sizes."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from helpers import golden, T, assert_close
import closed_form as CF

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _fill_closed_form(net):
    first = {}
    sd = net.state_dict()
    for k, v in sd.items():
        first.setdefault(v.data_ptr() if v.numel() else ('e', k), k)
    new = OrderedDict()
    for k, v in sd.items():
        ck = first[v.data_ptr() if v.numel() else ('e', k)]
        new[k] = CF.tensor_for(ck, v.shape).to(v.dtype)
    net.load_state_dict(new)
    return net


def _grad_report(net, ref_grads):
    """Per-parameter relative L2 error and global cosine of the HIP gradients vs reference gradients.

    Whole-network gradients are only piecewise continuous: one ReLU / max-pool decision that flips under fp32
    summation-order noise moves a conv weight gradient by ~1/sqrt(pixels) ~ 1e-3..1e-2 (torch's own fp32 result is
    that far from its float64 result, measured in round 1), so the element-wise 1e-7 agreement of the block
    tests cannot hold here; L2 / cosine measures are the meaningful whole-net statistics."""
    eng = net.engine()
    worst, dots, n1, n2, n = (0.0, ''), 0.0, 0.0, 0.0, 0
    gmax = max(float(g.abs().max()) for g in ref_grads.values() if g is not None)
    for k, p in net.named_parameters():
        gr = ref_grads.get(k)
        if gr is None or float(gr.abs().max()) < 1e-7:
            continue
        off, cnt = eng.grad_range(p)
        mine = eng.grads[off:off + cnt].view(p.shape).cpu().double()
        gr = gr.double()
        if float(mine.abs().max()) == 0.0:
            # conv / deconv bias in front of train-mode BatchNorm: analytically zero (sum of the BN input gradient);
            # the HIP path writes exact zeros, the reference holds rounding noise
            assert float(gr.abs().max()) < 1e-3 * gmax, (k, float(gr.abs().max()), gmax)
            continue
        e = float((mine - gr).norm() / gr.norm())
        worst = max(worst, (e, k))
        dots += float((mine * gr).sum()); n1 += float((mine * mine).sum()); n2 += float((gr * gr).sum())
        n += 1
    return worst, dots / (n1 ** 0.5 * n2 ** 0.5), n


def _nets():
    from salt_amd import architectures as A
    return {'unet_resnet34_hyper': lambda: A.UNetResNet(34, 2, dropout_2d=0.0, pretrained=False, use_hypercolumn=True),
            'ternaus_resnet34_deconv': lambda: A.TernausUNetResNet(34, 2, dropout_2d=0.0, pretrained=False, is_deconv=True),
            'ternaus_resnet34_upsample': lambda: A.TernausUNetResNet(34, 2, dropout_2d=0.0, pretrained=False, is_deconv=False),
            'salt_unet': lambda: A.SaltUNet(2, dropout_2d=0.0, is_deconv=True),
            'salt_linknet': lambda: A.SaltLinkNet(2, dropout_2d=0.0, is_deconv=True),
            'unet_resnet152_hyper': lambda: A.UNetResNet(152, 2, dropout_2d=0.0, pretrained=False, use_hypercolumn=True),
            'ternaus_resnet101_deconv': lambda: A.TernausUNetResNet(101, 2, dropout_2d=0.0, pretrained=False, is_deconv=True)}


@pytest.mark.parametrize('tag', ['unet_resnet34_hyper', 'ternaus_resnet34_deconv', 'ternaus_resnet34_upsample', 'salt_unet', 'salt_linknet',
                                 'unet_resnet152_hyper', 'ternaus_resnet101_deconv'])
def test_eval_logits_and_masks_match_reference(tag):
    fx = golden('F8_' + tag)
    net = _fill_closed_form(_nets()[tag]()).to(DEV)
    net.eval()
    with torch.no_grad():
        logits = net(T(fx['x']).to(DEV)).cpu()
    assert_close(logits, fx['eval_logits'], 1e-3, 'eval logits')              # north_star: <= 1e-3 relative, fp32
    # per-pixel class decision sigmoid(logit[1]) > 0.5 <=> logit[1] > 0 must be bit-exact (postprocessing.py:41-43)
    assert np.array_equal((logits[:, 1] > 0).numpy().astype(np.uint8), fx['eval_mask'])


@pytest.mark.parametrize('tag', ['unet_resnet34_hyper', 'ternaus_resnet34_deconv', 'salt_unet', 'salt_linknet', 'unet_resnet152_hyper',
                                 'ternaus_resnet101_deconv'])
def test_one_training_step_matches_reference(tag):
    """zero_grad -> forward -> lovasz -> backward -> Adam(lr 1e-4, L2 1e-4) exactly as models.py:105-136."""
    from salt_amd.optim import FusedAdam, weight_regularization
    from salt_amd import losses
    fx = golden('F8_' + tag)
    net = _fill_closed_form(_nets()[tag]()).to(DEV)
    net.train()
    opt = FusedAdam(weight_regularization(net, True, 1e-4), lr=1e-4, model=net)
    out = net(T(fx['x']).to(DEV))
    assert_close(out.detach().cpu(), fx['train_logits'], 2e-3, 'train logits')
    loss = losses.lovasz_loss(out, T(fx['t']).to(DEV)) * 1.0
    loss.backward()
    ref = float(fx['train_loss'])
    assert abs(float(loss) - ref) < 2e-3 * max(1.0, abs(ref)), (float(loss), ref)
    names = fx['param_names'].tolist()
    idx = {n: i for i, n in enumerate(names)}
    eng = net.engine()
    dead = set(net.dead_parameter_names())
    checked = 0
    worst = (0.0, '')
    own = dict(net.named_parameters())
    for k, p in own.items():
        i = idx[k]
        has = bool(fx['param_has_grad'][i])
        assert has == (k not in dead), k
        if has and fx['grad_norm'][i] > 1e-4:
            off, n = eng.grad_range(p)
            gn = float(eng.grads[off:off + n].double().norm())
            rel = abs(gn - fx['grad_norm'][i]) / fx['grad_norm'][i]
            worst = max(worst, (rel, k))
            checked += 1
    # 100+ layer encoders with the closed-form (badly conditioned) golden weights: a handful of ReLU decisions flip between two fp32
    # summation orders and the tiniest gradients (a 1-element SE bias) move by 10 %; the well-conditioned check of the deep nets
    # is test_resnet34_unets_train_step_vs_oracle_default_init[UNetResNet152]
    deep = '152' in tag or '101' in tag
    assert checked > (100 if 'salt' not in tag else 30) and worst[0] < (0.15 if deep else 1e-2), (checked, worst)
    if deep:
        # the 15 % above is not a chosen number: the reference's OWN fp32 gradients sit that far from the same modules run in
        # float64 (fixture F11, make_golden.py).  Measured against float64, the HIP gradients must be as good as the reference's.
        f64 = golden('F11_' + tag + '_f64')
        i64 = {n: i for i, n in enumerate(f64['param_names'].tolist())}
        e_hip, e_ref, knames, g64s = [], [], [], []
        for k, p in own.items():
            i = idx[k]
            if fx['param_has_grad'][i] and fx['grad_norm'][i] > 1e-4:
                g64 = float(f64['grad_norm64'][i64[k]])
                off, n = eng.grad_range(p)
                e_hip.append(abs(float(eng.grads[off:off + n].double().norm()) - g64) / g64)
                e_ref.append(abs(float(fx['grad_norm'][i]) - g64) / g64)
                knames.append(k); g64s.append(g64)
        e_hip, e_ref = np.array(e_hip), np.array(e_ref)
        top = np.argsort(-e_hip)[:6]
        print('worst HIP tensors:', [(knames[j], '%.2e' % e_hip[j], '%.2e' % e_ref[j], '%.2e' % g64s[j]) for j in top])
        print('%s: grad-norm error vs float64: HIP max %.3e median %.3e | reference fp32 max %.3e median %.3e'
              % (tag, e_hip.max(), np.median(e_hip), e_ref.max(), np.median(e_ref)))
        # the 1-element squeeze-excitation biases are sums over every pixel with heavy cancellation (norm 1e-4 next to 1e+2 for the
        # convolutions): they are bounded absolutely, relative to the largest gradient norm; everything else relatively
        g64s, numels = np.array(g64s), np.array([own[k].numel() for k in knames])
        big = numels >= 16
        assert e_hip[big].max() <= 2.5 * e_ref[big].max() + 1e-3 and np.median(e_hip) <= 2.5 * np.median(e_ref) + 1e-4
        assert (e_hip[~big] * g64s[~big]).max() <= 1e-6 * g64s.max()
        for k in f64:
            if k.startswith('fullgrad64:'):
                p = own[k[11:]]
                off, n = eng.grad_range(p)
                g64 = T(f64[k]).double()
                l2_hip = float((eng.grads[off:off + n].view(p.shape).cpu().double() - g64).norm() / g64.norm())
                l2_ref = float((T(fx['fullgrad:' + k[11:]]).double() - g64).norm() / g64.norm())
                assert l2_hip <= 2.5 * l2_ref + 1e-3, (k, l2_hip, l2_ref)
    for k in fx:
        if k.startswith('fullgrad:'):
            p = own[k[9:]]
            off, n = eng.grad_range(p)
            mine = eng.grads[off:off + n].view(p.shape).cpu()
            if deep:                                 # see above: L2 instead of max-norm for the 100+ layer encoders
                l2 = float((mine.double() - T(fx[k]).double()).norm() / T(fx[k]).double().norm())
                assert l2 < 0.1, (k, l2)
            else:
                assert_close(mine, fx[k], 2e-2, k)
    opt.step()
    torch.cuda.synchronize()
    for k, p in own.items():
        i = idx[k]
        if fx['param_has_grad'][i] and fx['grad_norm'][i] > 1e-4:
            pn = float(p.detach().double().norm())
            assert abs(pn - fx['post_norm'][i]) <= (3e-4 if deep else 1e-4) * max(fx['post_norm'][i], 1e-3), (k, pn, fx['post_norm'][i])
    sd = net.state_dict()
    for k, s in zip(fx['bn_keys'].tolist(), fx['bn_sum'].tolist()):
        assert abs(float(sd[k].double().sum()) - s) <= (3e-3 if deep else 1e-3) * max(1.0, abs(s)), k


def test_pool0_stem_maxpool_matches_reference_golden():
    """ResNetEncoders(pool0=True) (architectures/encoders.py:23-27): stem + MaxPool2d(3, 2, 1); the reference's logits then come out at
    half the input resolution (unet.py:89-109).  Eval logits / bit-exact masks and one Lovasz training step vs the golden."""
    from salt_amd import architectures as A, losses
    fx = golden('F13_unet_resnet34_hyper_pool0')
    net = _fill_closed_form(A.UNetResNet(34, 2, dropout_2d=0.0, pretrained=False, use_hypercolumn=True, pool0=True))
    assert list(net.state_dict().keys()) == fx['keys'].tolist()
    net.to(DEV).eval()
    with torch.no_grad():
        logits = net(T(fx['x']).to(DEV)).cpu()
    assert tuple(logits.shape) == (2, 2, 32, 32)
    assert_close(logits, fx['eval_logits'], 1e-3, 'eval logits')
    assert np.array_equal((logits[:, 1] > 0).numpy().astype(np.uint8), fx['eval_mask'])
    net.train()
    out = net(T(fx['x']).to(DEV))
    assert_close(out.detach().cpu(), fx['train_logits'], 2e-3, 'train logits')
    loss = losses.lovasz_loss(out, T(fx['t']).to(DEV))
    loss.backward()
    ref = float(fx['train_loss'])
    assert abs(float(loss) - ref) < 2e-3 * max(1.0, abs(ref)), (float(loss), ref)
    idx = {n: i for i, n in enumerate(fx['param_names'].tolist())}
    eng = net.engine()
    checked, worst = 0, (0.0, '')
    for k, p in net.named_parameters():
        i = idx[k]
        if fx['param_has_grad'][i] and fx['grad_norm'][i] > 1e-4:
            off, n = eng.grad_range(p)
            gn = float(eng.grads[off:off + n].double().norm())
            worst = max(worst, (abs(gn - fx['grad_norm'][i]) / fx['grad_norm'][i], k))
            checked += 1
    assert checked > 100 and worst[0] < 1e-2, (checked, worst)


def test_r34_hypercolumn_fp32_eval_matches_oracle_c2_shape():
    """BASELINE C2's network and FULL shape on the <= 1e-3 path: architectures.unet.UNetResNet(34, hypercolumn),
    [32,3,128,128], fp32 engine, eval - logits <= 1e-3 relative to the fp32 oracle and `logit[1] > 0` masks (postprocessing.py:41-43)
    array_equal.  A differing pixel is admitted ONLY where the decision is not defined at fp32: the oracle re-run in float64 has
    |logit| below the fp32 oracle's own distance from float64 there (the reference's fp32 result could have fallen either way).
    Counts are printed and recorded (profiles/rNN_parity_counts.json)."""
    from salt_amd import architectures as A
    from oracle import nets as ON, specs as OS
    net = A.UNetResNet(34, 2, use_hypercolumn=True, dropout_2d=0.0, pretrained=False)
    spec = OS.SPECS['UNetResNet'](with_fc=True)
    sd = OS.init_state(spec, seed=7)
    net.load_state_dict({k: sd[k] for k in net.state_dict() if k in sd}, strict=False)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items() if k in spec}
    x = CF.input_for('c2', (32, 3, 128, 128))
    net.to(DEV).eval()
    assert net.engine().dtype == 'f32'
    with torch.no_grad():
        y = net(x.to(DEV)).float().cpu()
        yr = ON.unet_resnet(sd, x, False)
        y64 = ON.unet_resnet({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}, x.double(), False)
    assert y64.dtype == torch.float64
    e = assert_close(y, yr, 1e-3, 'eval logits vs fp32 oracle')             # north_star: <= 1e-3 relative, fp32
    noise = float((yr.double() - y64).abs().max())                          # how far the REFERENCE arithmetic at fp32 sits from exact
    e64 = float((y.double() - y64).abs().max())
    mis = (y[:, 1] > 0) != (yr[:, 1] > 0)
    mis64 = (y[:, 1] > 0) != (y64[:, 1] > 0)
    ref_mis64 = (yr[:, 1] > 0) != (y64[:, 1] > 0)
    n_mis = int(mis.sum())
    print('C2 fp32 mask: %d of %d decisions differ from the fp32 oracle (%d from the fp64 oracle; the fp32 oracle itself %d); max |logit error| '
          '%.3e vs fp64 (fp32 oracle: %.3e), rel %.3e' % (n_mis, mis.numel(), int(mis64.sum()), int(ref_mis64.sum()), e64, noise, e))
    from helpers import record_parity
    record_parity('C2_r34_hypercolumn_fp32_eval_masks', config='[32,3,128,128] fp32 HIP vs oracle.nets.unet_resnet (fp32 and float64), logit[1] > 0',
                  decisions=int(mis.numel()), differ=n_mis, differ_from_fp64_oracle=int(mis64.sum()), fp32_oracle_differs_from_fp64=int(ref_mis64.sum()),
                  max_abs_logit_error_vs_fp64=e64, fp32_oracle_max_abs_error_vs_fp64=noise, eval_logits_rel_err=e,
                  max_abs_fp64_logit_at_differing=float(y64[:, 1][mis].abs().max()) if n_mis else 0.0)
    if n_mis:
        assert float(y64[:, 1][mis].abs().max()) <= 2.0 * noise, (n_mis, float(y64[:, 1][mis].abs().max()), noise)
    else:
        assert np.array_equal((y[:, 1] > 0).numpy(), (yr[:, 1] > 0).numpy())


def test_vanilla_unet_matches_oracle_c1_shape():
    """BASELINE C1: vanilla 4-level U-Net, [32,1,128,128] fp32 — eval logits/masks and one train step vs the oracle."""
    from salt_amd import architectures as A
    from oracle import nets as ON, specs as OS, losses as OL
    # default (torch) initialisation: every BatchNorm channel is well conditioned, so fp32 gradients are comparable
    # to ~1e-5.  (The closed-form golden weights leave a few near-constant channels whose 1/sqrt(var+eps) = 316
    # amplifies fp32 summation-order noise to ~1e-2 in torch itself: DESIGN.md 5.)
    net = A.VanillaUNet(2, 1, 16, 4)
    spec = OS.spec_vanilla_unet()
    assert list(spec.keys()) == list(net.state_dict().keys())
    sd = OS.init_state(spec, seed=11)
    net.load_state_dict(sd)
    net.to(DEV)
    x = CF.input_for('c1', (32, 1, 128, 128))
    t = CF.mask_for('c1', (32, 128, 128))
    net.eval()
    with torch.no_grad():
        y = net(x.to(DEV)).cpu()
        yr = ON.vanilla_unet(sd, x, False)
    assert_close(y, yr, 1e-3, 'eval logits')
    # north_star: per-pixel masks bit-exact.  524 288 decisions; the only admissible disagreement is a pixel whose reference logit
    # is itself within fp32 rounding of zero (|logit| < 2e-6 of the largest), and the count is reported
    mis = (y[:, 1] > 0) != (yr[:, 1] > 0)
    n_mis = int(mis.sum())
    print('C1 mask: %d of %d decisions differ from the oracle' % (n_mis, mis.numel()))
    from helpers import record_parity
    record_parity('C1_vanilla_unet_fp32_eval_masks', config='[32,1,128,128] fp32, logit[1] > 0 vs oracle.nets.vanilla_unet', decisions=int(mis.numel()),
                  differ=n_mis, max_abs_ref_logit_at_differing=float(yr[:, 1][mis].abs().max()) if n_mis else 0.0,
                  max_abs_logit_error=float((y - yr).abs().max()))
    if n_mis:
        # round 6: no count allowance - EVERY differing pixel must sit inside the fp32 noise floor of the reference logit
        assert float(yr[:, 1][mis].abs().max()) < 2e-6 * float(yr.abs().max()), (n_mis, float(yr[:, 1][mis].abs().max()))
    net.train()
    xs, ts = x[:8], t[:8]
    from salt_amd import losses
    for k in OS.trainable_keys(spec):
        sd[k].requires_grad_(True)
    # BCE+Dice is smooth: it isolates the network backward.  The Lovasz gradient g_k = J_k - J_(k-1) carries ~1e-3
    # relative fp32 cancellation noise in the reference itself, so its bar is looser.
    for kind, oracle_loss, hip_loss, gtol in (('bce_dice', OL.mixed_dice_bce_loss, losses.mixed_dice_bce_loss, 5e-2),
                                              ('lovasz', OL.lovasz_loss, losses.lovasz_loss, 5e-2)):
        sd0 = {k: v.detach().clone() for k, v in sd.items()}
        for k in OS.trainable_keys(spec):
            sd0[k].requires_grad_(True)
        out_r = ON.vanilla_unet(sd0, xs, True)
        loss_r = oracle_loss(out_r, ts)
        loss_r.backward()
        net.load_state_dict({k: v.detach() for k, v in sd.items()})
        out = net(xs.to(DEV))
        loss = hip_loss(out, ts.to(DEV))
        loss.backward()
        assert abs(float(loss) - float(loss_r)) < 1e-3 * max(1.0, abs(float(loss_r))), kind
        worst, cos, n = _grad_report(net, {k: v.grad for k, v in sd0.items()})
        assert n > 40 and worst[0] < gtol and cos > 0.9999, (kind, worst, cos)
        # the last layers have no data-dependent branch downstream of their gradient: tight element-wise check
        eng = net.engine()
        for k in ('final.weight', 'dec1.1.conv.1.weight', 'dec1.1.conv.1.bias'):
            p = dict(net.named_parameters())[k]
            off, cnt = eng.grad_range(p)
            assert_close(eng.grads[off:off + cnt].view(p.shape).cpu(), sd0[k].grad, 1e-4 if kind == 'bce_dice' else 5e-3, kind + ' ' + k)


@pytest.mark.parametrize('arch', ['UNetResNet', 'TernausUNetResNet', 'SaltUNet', 'SaltLinkNet', 'UNetResNet152'])
def test_resnet34_unets_train_step_vs_oracle_default_init(arch):
    """ResNet34 U-Nets with default initialisation (well conditioned): logits, loss and EVERY parameter gradient vs the oracle."""
    from salt_amd import architectures as A, losses
    from oracle import nets as ON, specs as OS, losses as OL
    torch.manual_seed(5)
    skw = {}
    if arch == 'UNetResNet152':                      # Bottleneck encoder (BASELINE C4's network), B = 2 keeps the CPU oracle short
        arch, skw = 'UNetResNet', {'depth': 152}
        net, kw = A.UNetResNet(152, 2, use_hypercolumn=True), {'depth': 152}
    elif arch == 'UNetResNet':
        net, kw = A.UNetResNet(34, 2, use_hypercolumn=True), {}
    elif arch == 'TernausUNetResNet':
        net, kw = A.TernausUNetResNet(34, 2, dropout_2d=0.0, is_deconv=True), {'is_deconv': True}
    else:
        net, kw = getattr(A, arch)(2, dropout_2d=0.0, is_deconv=True), {'is_deconv': True}
    spec = OS.SPECS[arch](with_fc=True, **skw)
    sd = OS.init_state(spec, seed=7)
    net.load_state_dict({k: sd[k] for k in net.state_dict() if k in sd}, strict=False)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items() if k in spec}
    nb, hw = (2, 128) if skw else (4, 64)           # the deep net gets 128x128 so that its last stage still sees 32 samples per channel
    x = CF.input_for('r34', (nb, 3, hw, hw))
    t = CF.mask_for('r34', (nb, hw, hw))
    net.to(DEV).train()
    dead = set(net.dead_parameter_names())
    for k in OS.trainable_keys(spec):
        sd[k].requires_grad_(True)
    out_r = ON.FORWARDS[arch](sd, x, True, **kw)
    loss_r = OL.mixed_dice_bce_loss(out_r, t)
    loss_r.backward()
    out = net(x.to(DEV))
    loss = losses.mixed_dice_bce_loss(out, t.to(DEV))
    loss.backward()
    # depth 152 at 64x64, B = 2: the last stage normalises over 8 samples per channel - train-mode BN amplifies fp32 noise there
    assert_close(out.detach().cpu(), out_r.detach(), 5e-3 if skw else 1e-4, 'train-mode logits')
    assert abs(float(loss) - float(loss_r)) < (1e-3 if skw else 1e-5) * max(1.0, abs(float(loss_r)))
    for k in dead:
        assert sd[k].grad is None, k
    worst, cos, n = _grad_report(net, {k: v.grad for k, v in sd.items() if k not in dead})
    # depth 152 (train-mode BN through 150 layers): torch's own fp32 result sits at cosine 0.9975 / worst tensor 8e-2 / logits 1.1e-3
    # from its float64 result on this very input, so that is the resolution any fp32 implementation can be compared at
    assert n > (120 if 'Salt' not in arch else 30) and worst[0] < (0.3 if skw else 5e-2) and cos > (0.99 if skw else 0.9999), (worst, cos, n)
    if skw:
        # earn the loose depth-152 bounds: the same oracle in float64 is the yardstick - the HIP result must be as close to it as
        # torch's fp32 result is (same input, same weights)
        sd64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
        out64 = ON.FORWARDS[arch](sd64, x.double(), True, **kw)
        OL.mixed_dice_bce_loss(out64, t.double()).backward()
        g64 = {k: v.grad for k, v in sd64.items() if k not in dead and v.dtype.is_floating_point and v.grad is not None}
        w_hip, cos_hip, _ = _grad_report(net, g64)
        flat64 = torch.cat([g64[k].reshape(-1) for k in g64 if sd[k].grad is not None])
        flat32 = torch.cat([sd[k].grad.double().reshape(-1) for k in g64 if sd[k].grad is not None])
        cos_ref = float((flat64 * flat32).sum() / (flat64.norm() * flat32.norm()))
        w_ref = max(float((sd[k].grad.double() - g64[k]).norm() / g64[k].norm()) for k in g64
                    if sd[k].grad is not None and float(g64[k].abs().max()) >= 1e-7)
        e_log_hip = float((out.detach().cpu().double() - out64.detach()).abs().max() / out64.detach().abs().max())
        e_log_ref = float((out_r.detach().double() - out64.detach()).abs().max() / out64.detach().abs().max())
        print('depth 152 vs float64: HIP cosine %.5f worst %.3e logits %.3e | torch fp32 cosine %.5f worst %.3e logits %.3e'
              % (cos_hip, w_hip[0], e_log_hip, cos_ref, w_ref, e_log_ref))
        assert 1 - cos_hip <= 2.5 * (1 - cos_ref) + 1e-5 and w_hip[0] <= 2.5 * w_ref + 1e-3 and e_log_hip <= 2.5 * e_log_ref + 1e-5
    eng = net.engine()
    for k in ('final.1.weight', 'final.1.bias') if arch == 'UNetResNet' else ('final.weight', 'final.bias'):
        p = dict(net.named_parameters())[k]
        off, cnt = eng.grad_range(p)
        assert_close(eng.grads[off:off + cnt].view(p.shape).cpu(), sd[k].grad, 5e-3 if skw else 2e-5, k)


def test_segmentation_model_surface_fit_and_transform(tmp_path):
    """The drop-in boundary: SegmentationModel(architecture_config, training_config, callbacks_config) with
    fit / transform / persist / load as main.py uses them (main.py:365-366,389; utils.py:444-467)."""
    from salt_amd.models import SegmentationModel
    arch = {'model_params': {'architecture': 'VanillaUNet', 'out_channels': 2, 'activation': 'sigmoid'},
            'optimizer_params': {'lr': 1e-3}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
    torch.manual_seed(0)
    m = SegmentationModel(arch, {'epochs': 2}, {})
    X = CF.input_for('seg', (8, 1, 64, 64))
    M = CF.mask_for('seg', (8, 64, 64))
    batches = [[X[:4], M[:4]], [X[4:], M[4:]]]
    losses = []

    class Rec:
        def set_params(self, *a, **k): pass
        def on_train_begin(self): pass
        def on_train_end(self): pass
        def on_epoch_begin(self): pass
        def on_epoch_end(self): pass
        def on_batch_begin(self): pass
        def on_batch_end(self, metrics): losses.append(float(metrics['sum']))
        def training_break(self): return False
    m.callbacks.callbacks.append(Rec())
    m.fit((batches, len(batches)))
    assert len(losses) == 4 and all(np.isfinite(losses))
    out = m.transform(([X[:4], X[4:]], 2))
    preds = out['mask_prediction']
    assert len(preds) == 8 and preds[0].shape == (2, 64, 64) and preds[0].dtype == np.float32
    assert float(np.min(preds[0])) >= 0 and float(np.max(preds[0])) <= 1
    path = str(tmp_path / 'best.torch')
    m.persist(path)
    sd = torch.load(path)
    assert all(k.startswith('module.') for k in sd)
    m2 = SegmentationModel(arch, {'epochs': 1}, {})
    m2.load(path)
    out2 = m2.transform(([X[:4], X[4:]], 2))
    np.testing.assert_allclose(out2['mask_prediction'][3], preds[3], rtol=1e-5, atol=1e-6)
    assert m.optimizer.state_dict()['param_groups'][0]['lr'] == 1e-3
    assert m.output_names == ['mask'] and m.loss_function[0][2] == 1.0


def test_cpu_tensor_is_rejected_loudly():
    from salt_amd import architectures as A, SaltError
    net = A.VanillaUNet(2, 1, 16, 2)
    with pytest.raises(SaltError):
        net(torch.zeros(1, 1, 16, 16))


@pytest.mark.parametrize('depth,dtype,shape', [(34, 'f32', (2, 3, 64, 64)), (34, 'bf16', (32, 3, 128, 128)), (50, 'bf16', (2, 3, 64, 64)), (50, 'f32', (2, 3, 64, 64))])
def test_eval_residual_epilogue_is_bit_identical_to_the_separate_pass(depth, dtype, shape, monkeypatch):
    """Eval-mode BasicBlock / Bottleneck (torchvision layout via architectures/encoders.py:6-45): out = relu(bn(conv(a)) + identity).
    The folded form (salt_conv_args.res: BN affine, identity add and ReLU in the convolution's epilogue - conv_mfma_kernel for the
    1x1 / small launches, conv_ws_kernel / conv_ls_kernel at the [32,3,128,128] shape) must give the logits of the unfused form
    (convolution, then salt_affine_act over its stored output) bit for bit."""
    from salt_amd import architectures as A
    outs = {}
    for fold in (True, False):
        if fold:
            monkeypatch.delenv('SALT_NO_RES_FOLD', raising=False)
        else:
            monkeypatch.setenv('SALT_NO_RES_FOLD', '1')
        torch.manual_seed(5)
        net = A.UNetResNet(depth, 2, use_hypercolumn=True, dropout_2d=0.0, pretrained=False)
        with torch.no_grad():
            for m in net.modules():                      # non-trivial running statistics
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.normal_(0, 0.1); m.running_var.uniform_(0.5, 1.5)
        net.set_compute_dtype(dtype)
        net.to(DEV).eval()
        g = torch.Generator().manual_seed(3)
        x = torch.randn(*shape, generator=g)
        with torch.no_grad():
            outs[fold] = net(x.to(DEV)).float().cpu()
        prog = net.engine().net(tuple(shape), False).fwd
        n_aff = sum(1 for name, _, s_ in prog.ops if name == 'affine_act' and s_.res.p)
        n_res = sum(1 for name, _, s_ in prog.ops if name == 'conv' and s_.res.p)
        nblocks = {34: 16, 50: 16}[depth]
        assert (n_res, n_aff) == ((nblocks, 0) if fold else (0, nblocks)), (n_res, n_aff)
    assert torch.isfinite(outs[True]).all()
    assert torch.equal(outs[True], outs[False])


def test_segmentation_model_transform_values_match_reference_golden():
    """SegmentationModel.transform (models.py:138-147 -> _transform 150-177): the probabilities handed to the post-processing are
    sigmoid(eval logits) of the reference's own forward (golden F8, architectures.unet.UNetResNet(34, hypercolumn)) - a VALUE check of
    the boundary function, not only its shapes."""
    from salt_amd.models import SegmentationModel
    fx = golden('F8_unet_resnet34_hyper')
    arch = {'model_params': {'architecture': 'UNetResNet', 'out_channels': 2, 'activation': 'sigmoid'},
            'optimizer_params': {'lr': 1e-4}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
    m = SegmentationModel(arch, {'epochs': 1}, {})
    _fill_closed_form(m.model)
    x = T(fx['x'])
    out = m.transform(([x[:1], x[1:]], 2))['mask_prediction']
    ref = 1.0 / (1.0 + np.exp(-fx['eval_logits'].astype(np.float64)))
    assert len(out) == x.shape[0]
    for i, pr in enumerate(out):
        assert pr.shape == ref[i].shape and pr.dtype == np.float32
        assert float(np.abs(pr - ref[i]).max()) <= 1e-4, float(np.abs(pr - ref[i]).max())      # 1e-3 relative on logits -> <= 2.5e-4 on sigmoid
        assert np.array_equal(pr[1] > 0.5, fx['eval_mask'][i].astype(bool))
