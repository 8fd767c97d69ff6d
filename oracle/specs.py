"""Oracle: state-dict key/shape specifications and default initialisation (TEST INFRASTRUCTURE ONLY).

``spec_*`` functions return an ordered ``{canonical key: (shape, kind)}`` for an architecture as the
reference's module tree lays it out (names from the ``nn.Module`` attributes cited below);
``alias_map`` gives the additional spellings under which shared sub-modules also appear in the
reference's ``state_dict()`` (SURVEY.md §8b "nn.Module contract").

kinds: conv_w conv_b convT_w convT_b lin_w lin_b bn_w bn_b bn_rm bn_rv bn_nbt
       (prefix 'tv_' = torchvision-ResNet initialisation: conv N(0, sqrt(2/(k*k*cout))))

Reference anchors: architectures/unet.py:46-87, architectures/encoders.py:23-36, architectures/base.py,
unet_models.py:100-138,160-177,198-216; torchvision 0.2.0 models/resnet.py (un-vendored).
"""
import math
from collections import OrderedDict

import torch

from .blocks import RESNET_CFG


def _bn(spec, p, c):
    spec[p + 'weight'] = ((c,), 'bn_w')
    spec[p + 'bias'] = ((c,), 'bn_b')
    spec[p + 'running_mean'] = ((c,), 'bn_rm')
    spec[p + 'running_var'] = ((c,), 'bn_rv')
    spec[p + 'num_batches_tracked'] = ((), 'bn_nbt')


def _resnet(spec, p, depth, in_ch=3, with_fc=True):
    kind, counts = RESNET_CFG[depth]
    spec[p + 'conv1.weight'] = ((64, in_ch, 7, 7), 'tv_conv_w')
    _bn(spec, p + 'bn1.', 64)
    inpl = 64
    exp = 1 if kind == 'basic' else 4
    for li, (planes, n) in enumerate(zip([64, 128, 256, 512], counts), start=1):
        for b in range(n):
            q = '%slayer%d.%d.' % (p, li, b)
            stride = 2 if (b == 0 and li > 1) else 1
            if kind == 'basic':
                spec[q + 'conv1.weight'] = ((planes, inpl, 3, 3), 'tv_conv_w'); _bn(spec, q + 'bn1.', planes)
                spec[q + 'conv2.weight'] = ((planes, planes, 3, 3), 'tv_conv_w'); _bn(spec, q + 'bn2.', planes)
            else:
                spec[q + 'conv1.weight'] = ((planes, inpl, 1, 1), 'tv_conv_w'); _bn(spec, q + 'bn1.', planes)
                spec[q + 'conv2.weight'] = ((planes, planes, 3, 3), 'tv_conv_w'); _bn(spec, q + 'bn2.', planes)
                spec[q + 'conv3.weight'] = ((planes * 4, planes, 1, 1), 'tv_conv_w'); _bn(spec, q + 'bn3.', planes * 4)
            if stride != 1 or inpl != planes * exp:
                spec[q + 'downsample.0.weight'] = ((planes * exp, inpl, 1, 1), 'tv_conv_w')
                _bn(spec, q + 'downsample.1.', planes * exp)
            inpl = planes * exp
    if with_fc:
        spec[p + 'fc.weight'] = ((1000, 512 * exp), 'lin_w')
        spec[p + 'fc.bias'] = ((1000,), 'lin_b')
    return 512 * exp


def _conv2d_bn_relu(spec, p, cin, cout, k=(3, 3)):          # base.Conv2dBnRelu (registers batch_norm first)
    _bn(spec, p + 'batch_norm.', cout)
    spec[p + 'conv.weight'] = ((cout, cin, k[0], k[1]), 'conv_w')
    spec[p + 'conv.bias'] = ((cout,), 'conv_b')


def _conv_bn_relu(spec, p, cin, cout):                      # unet_models.ConvBnRelu
    spec[p + 'conv.0.weight'] = ((cout, cin, 3, 3), 'conv_w')
    spec[p + 'conv.0.bias'] = ((cout,), 'conv_b')
    _bn(spec, p + 'conv.1.', cout)


def _decoder_block(spec, p, cin, mid, cout):                # base.DecoderBlock
    _conv2d_bn_relu(spec, p + 'conv1.', cin, mid)
    _conv2d_bn_relu(spec, p + 'conv2.', mid, cout)
    r = cout // 16
    spec[p + 'channel_se.fc.0.weight'] = ((r, cout), 'lin_w'); spec[p + 'channel_se.fc.0.bias'] = ((r,), 'lin_b')
    spec[p + 'channel_se.fc.2.weight'] = ((cout, r), 'lin_w'); spec[p + 'channel_se.fc.2.bias'] = ((cout,), 'lin_b')
    spec[p + 'spatial_se.fc.weight'] = ((1, cout, 1, 1), 'conv_w'); spec[p + 'spatial_se.fc.bias'] = ((1,), 'conv_b')


def _decoder_block_v2(spec, p, cin, mid, cout):             # unet_models.DecoderBlockV2 (both branches)
    _conv_bn_relu(spec, p + 'deconv.0.', cin, mid)
    spec[p + 'deconv.1.weight'] = ((mid, cout, 4, 4), 'convT_w'); spec[p + 'deconv.1.bias'] = ((cout,), 'convT_b')
    _bn(spec, p + 'deconv.2.', cout)
    _conv_bn_relu(spec, p + 'upsample.0.', cin, cout)


def spec_unet_resnet(depth=34, num_classes=2, in_ch=3, use_hypercolumn=True, with_fc=False):
    """architectures.unet.UNetResNet."""
    s = OrderedDict()
    bottom = _resnet(s, 'encoders.encoder.', depth, in_ch, with_fc)
    _conv2d_bn_relu(s, 'center.0.', bottom, bottom)
    _conv2d_bn_relu(s, 'center.1.', bottom, bottom // 2)
    _decoder_block(s, 'dec5.', bottom + bottom // 2, bottom, bottom // 8)
    _decoder_block(s, 'dec4.', bottom // 2 + bottom // 8, bottom // 2, bottom // 8)
    _decoder_block(s, 'dec3.', bottom // 4 + bottom // 8, bottom // 4, bottom // 8)
    _decoder_block(s, 'dec2.', bottom // 8 + bottom // 8, bottom // 8, bottom // 8)
    _decoder_block(s, 'dec1.', bottom // 8, bottom // 16, bottom // 8)
    _conv2d_bn_relu(s, 'final.0.', (5 if use_hypercolumn else 1) * bottom // 8, bottom // 8)
    s['final.1.weight'] = ((num_classes, bottom // 8, 1, 1), 'conv_w')
    s['final.1.bias'] = ((num_classes,), 'conv_b')
    return s


def spec_ternaus_unet_resnet(depth=34, num_classes=2, in_ch=3, num_filters=32, with_fc=False):
    """unet_models.UNetResNet."""
    s = OrderedDict()
    bottom = _resnet(s, 'encoder.', depth, in_ch, with_fc)
    nf = num_filters
    _decoder_block_v2(s, 'dec4.', bottom, nf * 16, nf * 8)
    _decoder_block_v2(s, 'dec3.', bottom // 2 + nf * 8, nf * 16, nf * 8)
    _decoder_block_v2(s, 'dec2.', bottom // 4 + nf * 8, nf * 8, nf * 2)
    _decoder_block_v2(s, 'dec1.', bottom // 8 + nf * 2, nf * 4, nf * 4)
    s['final.weight'] = ((num_classes, nf * 4, 1, 1), 'conv_w')
    s['final.bias'] = ((num_classes,), 'conv_b')
    return s


def spec_salt_unet(num_classes=2, in_ch=3, with_fc=False):
    s = OrderedDict()
    _resnet(s, 'encoder.', 34, in_ch, with_fc)
    _decoder_block_v2(s, 'dec3.', 256, 512, 256)
    _conv_bn_relu(s, 'dec2.', 256 + 64, 256)
    _decoder_block_v2(s, 'dec1.', 256 + 64, (256 + 64) * 2, 256)
    s['final.weight'] = ((num_classes, 256, 1, 1), 'conv_w'); s['final.bias'] = ((num_classes,), 'conv_b')
    return s


def spec_salt_linknet(num_classes=2, in_ch=3, with_fc=False):
    s = OrderedDict()
    _resnet(s, 'encoder.', 34, in_ch, with_fc)
    _decoder_block_v2(s, 'dec2.', 128, 256, 256)
    _decoder_block_v2(s, 'dec1.', 256 + 64, 512, 256)
    s['final.weight'] = ((num_classes, 256, 1, 1), 'conv_w'); s['final.bias'] = ((num_classes,), 'conv_b')
    return s


def spec_vanilla_unet(num_classes=2, in_ch=1, base=16, levels=4):
    s = OrderedDict()
    c = in_ch
    for i in range(1, levels + 1):
        f = base * 2 ** (i - 1)
        _conv_bn_relu(s, 'enc%d.0.' % i, c, f)
        _conv_bn_relu(s, 'enc%d.1.' % i, f, f)
        c = f
    f = base * 2 ** levels
    _conv_bn_relu(s, 'center.0.', c, f)
    _conv_bn_relu(s, 'center.1.', f, f)
    c = f
    for i in range(levels, 0, -1):
        f = base * 2 ** (i - 1)
        _bn(s, 'up%d.batch_norm.' % i, f)                 # base.DeconvConv2dBnRelu registers batch_norm first
        s['up%d.deconv.weight' % i] = ((c, f, 3, 3), 'convT_w'); s['up%d.deconv.bias' % i] = ((f,), 'convT_b')
        _conv_bn_relu(s, 'dec%d.0.' % i, 2 * f, f)
        _conv_bn_relu(s, 'dec%d.1.' % i, f, f)
        c = f
    s['final.weight'] = ((num_classes, c, 1, 1), 'conv_w'); s['final.bias'] = ((num_classes,), 'conv_b')
    return s


SPECS = {'UNetResNet': spec_unet_resnet, 'TernausUNetResNet': spec_ternaus_unet_resnet,
         'SaltUNet': spec_salt_unet, 'SaltLinkNet': spec_salt_linknet, 'VanillaUNet': spec_vanilla_unet}


def alias_map(arch, depth=34):
    """{alias prefix: canonical prefix} for shared sub-modules (reference registers them twice)."""
    if arch == 'UNetResNet':        # encoders.py:23-36
        return OrderedDict([('encoders.conv1.0.', 'encoders.encoder.conv1.'), ('encoders.conv1.1.', 'encoders.encoder.bn1.'),
                            ('encoders.encoder2.', 'encoders.encoder.layer1.'), ('encoders.encoder3.', 'encoders.encoder.layer2.'),
                            ('encoders.encoder4.', 'encoders.encoder.layer3.'), ('encoders.encoder5.', 'encoders.encoder.layer4.')])
    if arch == 'TernausUNetResNet':  # unet_models.py:123-130
        return OrderedDict([('input_adjust.0.', 'encoder.conv1.'), ('input_adjust.1.', 'encoder.bn1.'),
                            ('conv1.', 'encoder.layer1.'), ('conv2.', 'encoder.layer2.'),
                            ('conv3.', 'encoder.layer3.'), ('conv4.', 'encoder.layer4.')])
    if arch == 'SaltUNet':           # unet_models.py:164-173
        return OrderedDict([('input_adjust.0.', 'encoder.conv1.'), ('input_adjust.1.', 'encoder.bn1.'),
                            ('conv1.', 'encoder.layer1.1.'), ('conv2.', 'encoder.layer1.2.'),
                            ('conv3.', 'encoder.layer2.0.'), ('conv4.', 'encoder.layer2.1.')])
    if arch == 'SaltLinkNet':        # unet_models.py:202-212
        return OrderedDict([('input_adjust.0.', 'encoder.conv1.'), ('input_adjust.1.', 'encoder.bn1.'),
                            ('conv1_1.', 'encoder.layer1.1.'), ('conv1_2.', 'encoder.layer1.2.'),
                            ('conv2_0.', 'encoder.layer2.0.'), ('conv2_1.', 'encoder.layer2.1.'),
                            ('conv2_2.', 'encoder.layer2.2.'), ('conv2_3.', 'encoder.layer2.3.')])
    return OrderedDict()


def expand_aliases(arch, sd):
    """Return a dict that also contains every alias spelling (same tensor objects)."""
    out = OrderedDict(sd)
    for a, c in alias_map(arch).items():
        for k, v in sd.items():
            if k.startswith(c):
                out[a + k[len(c):]] = v
    return out


def _fans(shape, transposed=False):
    rf = 1
    for d in shape[2:]:
        rf *= d
    # torch.nn.init._calculate_fan_in_and_fan_out: fan_in = size(1) * receptive field (also for ConvTranspose)
    return shape[1] * rf, shape[0] * rf


def init_state(spec, seed=0, dtype=torch.float32):
    """Default initialisation as the reference's constructors would produce it (torch nn defaults:
    kaiming_uniform(a=sqrt 5) weights, U(-1/sqrt(fan_in), +) biases; BN 1/0/0/1; torchvision ResNet
    convs N(0, sqrt(2/(k*k*cout)))).  Values come from a torch.Generator(seed), key by key in spec order."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    last_w = None
    for k, (shape, kind) in spec.items():
        if kind in ('conv_w', 'convT_w', 'lin_w'):
            fan_in = _fans(shape)[0] if len(shape) > 2 else shape[1]
            b = 1.0 / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g, dtype=dtype) * 2 - 1) * b
            last_w = fan_in
        elif kind in ('conv_b', 'convT_b', 'lin_b'):
            b = 1.0 / math.sqrt(last_w)
            t = (torch.rand(shape, generator=g, dtype=dtype) * 2 - 1) * b
        elif kind == 'tv_conv_w':
            n = shape[2] * shape[3] * shape[0]
            t = torch.randn(shape, generator=g, dtype=dtype) * math.sqrt(2.0 / n)
        elif kind in ('bn_w', 'bn_rv'):
            t = torch.ones(shape, dtype=dtype)
        elif kind in ('bn_b', 'bn_rm'):
            t = torch.zeros(shape, dtype=dtype)
        elif kind == 'bn_nbt':
            t = torch.zeros((), dtype=torch.long)
        else:
            raise KeyError(kind)
        sd[k] = t
    return sd


def trainable_keys(spec):
    return [k for k, (_, kind) in spec.items() if kind not in ('bn_rm', 'bn_rv', 'bn_nbt')]
