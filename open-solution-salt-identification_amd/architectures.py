"""Host-side mirror of the reference's network classes for the U-Net hot path.

Each class keeps the reference's constructor signature, attribute names and therefore ``state_dict()``
keys/shapes (checked against the reference's own key lists in tests/), but instead of a torch
``forward`` it has ``emit(graph, ...)``, which appends hand-written HIP operators to a static program
(engine.py).  The torch ``nn.Conv2d`` / ``nn.BatchNorm2d`` / ``nn.Linear`` objects inside are used ONLY
as parameter containers (names, shapes, default initialisation); their ``forward`` is never called.

Reference classes mirrored (paths relative to the reference's common_blocks/):
  unet_models.py:21-30 ConvBnRelu · :38-50 DecoderBlockV1 · :53-75 DecoderBlockV2 · :78-151 UNetResNet
  (-> TernausUNetResNet here, exported under its reference name in unet_models.py) · :154-189 SaltUNet ·
  :192-233 SaltLinkNet
  architectures/base.py:7-37 Conv2dBnRelu · :40-57 DeconvConv2dBnRelu · :65-86 DecoderBlock ·
  :89-104 ChannelSELayer · :107-117 SpatialSELayer
  architectures/encoders.py:6-45 ResNetEncoders;  architectures/unet.py:22-109 UNetResNet
  torchvision 0.2.0 models/resnet.py (un-vendored dependency): ResNet / BasicBlock / Bottleneck layout
"""
import math
import os

import torch
from torch import nn

from ._abi import SaltError
from .runtime import Engine


class EmitOnly(nn.Module):
    """Sub-modules are parameter containers + emitters; only whole networks are callable."""

    def forward(self, *a, **k):
        raise SaltError('%s is executed as part of a compiled network (HIP program); call the enclosing network'
                        % type(self).__name__)


# ----------------------------------------------------------------------------- TernausNet-style blocks
class ConvBnRelu(EmitOnly):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_channels, out_channels, 3, padding=1), nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))

    def emit(self, g, x, out=None):
        return g.conv(x, self.conv[0], self.conv[1], relu=True, out=out, name='ConvBnRelu')


class DecoderBlockV1(EmitOnly):
    def __init__(self, in_channels, middle_channels, out_channels):
        super().__init__()
        self.block = nn.Sequential(ConvBnRelu(in_channels, middle_channels),
                                   nn.ConvTranspose2d(middle_channels, out_channels, kernel_size=3, stride=2, padding=1, output_padding=1),
                                   nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))

    def emit(self, g, x, out=None):
        return g.conv_transpose(self.block[0].emit(g, x), self.block[1], self.block[2], relu=True, out=out, name='DecoderBlockV1')


class DecoderBlockV2(EmitOnly):
    def __init__(self, in_channels, middle_channels, out_channels, is_deconv=True):
        super().__init__()
        self.is_deconv = is_deconv
        self.deconv = nn.Sequential(ConvBnRelu(in_channels, middle_channels),
                                    nn.ConvTranspose2d(middle_channels, out_channels, kernel_size=4, stride=2, padding=1),
                                    nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True))
        self.upsample = nn.Sequential(ConvBnRelu(in_channels, out_channels), nn.Upsample(scale_factor=2, mode='bilinear'))

    def emit(self, g, x, out=None):
        if self.is_deconv:
            return g.conv_transpose(self.deconv[0].emit(g, x), self.deconv[1], self.deconv[2], relu=True, out=out, name='DecoderBlockV2')
        return g.upsample(self.upsample[0].emit(g, x), 2, out=out, name='DecoderBlockV2.up')

    def dead_prefix(self):
        return 'upsample.' if self.is_deconv else 'deconv.'


# ----------------------------------------------------------------------------- architectures/base.py blocks
class Conv2dBnRelu(EmitOnly):
    def __init__(self, in_channels, out_channels, kernel_size=(3, 3), use_relu=True, use_batch_norm=True, use_padding=True,
                 padding_method='replication'):
        super().__init__()
        if padding_method != 'replication' or not use_padding:
            raise NotImplementedError('only the replication-padded variant is on the U-Net hot path')
        self.use_relu, self.use_batch_norm, self.use_padding = use_relu, use_batch_norm, use_padding
        self.batch_norm = nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.padding = nn.ReplicationPad2d(padding=(0, kernel_size[1] - 1, kernel_size[0] - 1, 0))
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, padding=0)

    def emit(self, g, x, out=None):
        return g.conv(x, self.conv, self.batch_norm if self.use_batch_norm else None, relu=self.use_relu, out=out, replicate=True, name='Conv2dBnRelu')


class DeconvConv2dBnRelu(EmitOnly):
    def __init__(self, in_channels, out_channels, use_relu=True, use_batch_norm=True):
        super().__init__()
        self.use_relu, self.use_batch_norm = use_relu, use_batch_norm
        self.batch_norm = nn.BatchNorm2d(out_channels)
        self.relu = nn.ReLU(inplace=True)
        self.deconv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size=3, stride=2, padding=1, output_padding=1)

    def emit(self, g, x, out=None):
        return g.conv_transpose(x, self.deconv, self.batch_norm if self.use_batch_norm else None, relu=self.use_relu, out=out, name='DeconvConv2dBnRelu')


class ChannelSELayer(EmitOnly):
    def __init__(self, channel, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction), nn.ReLU(inplace=True),
                                nn.Linear(channel // reduction, channel), nn.Sigmoid())


class SpatialSELayer(EmitOnly):
    def __init__(self, channels):
        super().__init__()
        self.fc = nn.Conv2d(channels, 1, kernel_size=1)
        self.sigmoid = nn.Sigmoid()


class DecoderBlock(EmitOnly):
    def __init__(self, in_channels, middle_channels, out_channels):
        super().__init__()
        self.conv1 = Conv2dBnRelu(in_channels, middle_channels)
        self.conv2 = Conv2dBnRelu(middle_channels, out_channels)
        self.upsample = nn.Upsample(scale_factor=2, mode='bilinear')
        self.relu = nn.ReLU(inplace=True)
        self.channel_se = ChannelSELayer(out_channels, reduction=16)
        self.spatial_se = SpatialSELayer(out_channels)

    def emit(self, g, x, e=None, cat=None, out=None):
        """``cat``: optional pre-allocated [up(x) | e] buffer whose tail slice IS ``e`` (concat-free skip)."""
        if e is None:
            xin = g.upsample(x, 2, name='DecoderBlock.up')
        else:
            if cat is None:
                cat = g.new_act(x.B, 2 * x.H, 2 * x.W, x.C + e.C, 'DecoderBlock.cat')
                g.copy(e, cat.slice(x.C, e.C))
            else:
                assert e.buf is cat.buf and e.c0 == cat.c0 + x.C and cat.C == x.C + e.C
            g.upsample(x, 2, out=cat.slice(0, x.C))
            xin = cat
        y = self.conv2.emit(g, self.conv1.emit(g, xin))
        return g.scse(y, self.channel_se, self.spatial_se, out=out, name='DecoderBlock.scse')


# ----------------------------------------------------------------------------- torchvision-layout ResNet (parameter layout only)
class BasicBlock(EmitOnly):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def emit(self, g, x, out=None):
        idt = x
        if self.downsample is not None:          # the projection shortcut is independent of conv1 -> conv2: side stream, joined at the add
            with g.side():
                idt = g.conv(x, self.downsample[0], self.downsample[1], relu=False, name='down')
        n0 = len(g.tape)
        a = g.conv(x, self.conv1, self.bn1, relu=True, name='block.conv1')
        y = g.conv(a, self.conv2, self.bn2, relu=True, res=idt, out=out, name='block.conv2')
        _shortcut_gradient_first(g, n0, self.downsample)
        return y


def _shortcut_gradient_first(g, n0, downsample):
    """Backward order inside a residual block with a projection shortcut (torchvision layout through architectures/encoders.py:38-45):
    the tape runs in reverse, so by default the shortcut's closure - emitted first - runs LAST and its stride-2 1x1 data gradient
    (every other pixel of the block input) is the last writer of dL/dx.  Moving it in front of the main branch's first convolution
    makes that convolution's data gradient - a full-coverage launch - the last writer, which can then carry the BatchNorm-backward
    sums of x's producer (Graph._bn_train_bwd) and saves that layer's reduction pass.  ``n0`` = tape length BEFORE the main branch was
    emitted; the shortcut's closure sits at n0 - 1."""
    if downsample is None or not g.train or os.environ.get('SALT_NO_SHORTCUT_FIRST') or n0 < 1 or len(g.tape) < n0 + 2:
        return
    g.tape.insert(n0, g.tape.pop(n0 - 1))        # [.., down, conv1, ..] -> [.., conv1, down, ..]: backward runs .., down, conv1


class Bottleneck(EmitOnly):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def emit(self, g, x, out=None):
        idt = x
        if self.downsample is not None:
            with g.side():
                idt = g.conv(x, self.downsample[0], self.downsample[1], relu=False, name='down')
        n0 = len(g.tape)
        a = g.conv(x, self.conv1, self.bn1, relu=True, name='bneck.conv1')
        a = g.conv(a, self.conv2, self.bn2, relu=True, name='bneck.conv2')
        y = g.conv(a, self.conv3, self.bn3, relu=True, res=idt, out=out, name='bneck.conv3')
        _shortcut_gradient_first(g, n0, self.downsample)
        return y


class ResNet(EmitOnly):
    CFG = {18: (BasicBlock, [2, 2, 2, 2]), 34: (BasicBlock, [3, 4, 6, 3]), 50: (Bottleneck, [3, 4, 6, 3]),
           101: (Bottleneck, [3, 4, 23, 3]), 152: (Bottleneck, [3, 8, 36, 3])}

    def __init__(self, depth, in_channels=3):
        super().__init__()
        block, counts = self.CFG[depth]
        self.inplanes = 64
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(block, 64, counts[0], 1)
        self.layer2 = self._make(block, 128, counts[1], 2)
        self.layer3 = self._make(block, 256, counts[2], 2)
        self.layer4 = self._make(block, 512, counts[3], 2)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(512 * block.expansion, 1000)        # present in the reference state_dict, never used
        for m in self.modules():                                 # torchvision 0.2.0 initialisation
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make(self, block, planes, n, stride):
        down = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, down)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*layers)


def emit_blocks(g, blocks, x, out=None):
    blocks = list(blocks)
    for i, b in enumerate(blocks):
        x = b.emit(g, x, out=out if i == len(blocks) - 1 else None)
    return x


def resnet(depth, pretrained=False):
    if pretrained:
        raise SaltError('pretrained ImageNet weights need a download; load a state_dict instead')
    return ResNet(depth)


class ResNetEncoders(EmitOnly):
    def __init__(self, encoder_depth, pretrained=False, pool0=False):
        super().__init__()
        if encoder_depth not in ResNet.CFG:
            raise NotImplementedError('only 18, 34, 50, 101, 152 version of Resnet are implemented')
        # pool0: the stem is followed by torchvision's MaxPool2d(3, 2, 1) (encoders.py:23-27).  The reference's decoder has no
        # compensating up-sampling, so the logits then come out at HALF the input resolution (unet.py:89-109); the registry never
        # sets it (models.py:15-19) - reproduced as is
        self.pool0 = pool0
        self.encoder = resnet(encoder_depth, pretrained)
        if pool0:
            self.conv1 = nn.Sequential(self.encoder.conv1, self.encoder.bn1, self.encoder.relu, self.encoder.maxpool)
        else:
            self.conv1 = nn.Sequential(self.encoder.conv1, self.encoder.bn1, self.encoder.relu)
        self.encoder2 = self.encoder.layer1
        self.encoder3 = self.encoder.layer2
        self.encoder4 = self.encoder.layer3
        self.encoder5 = self.encoder.layer4


# ----------------------------------------------------------------------------- whole networks
class HipNetwork(nn.Module):
    """Base of the callable networks: owns the Engine, dispatches forward to the compiled HIP program."""

    compute_dtype = os.environ.get('SALT_DTYPE', 'f32')
    # nn.Upsample / F.upsample(mode='bilinear') (base.py:70, unet.py:103-106).  False: torch >= 0.4 semantics, what the oracle and the
    # goldens executed (torch 2.10).  True: how torch 0.3.1 - the version the reference pins, environment.yml:17 - evaluated the very
    # same calls; a checkpoint TRAINED in the reference's environment saw these features, so evaluate / fine-tune it with True
    align_corners = False

    def __init__(self):
        super().__init__()
        self._engine = None

    # -- engine management -------------------------------------------------------------------------
    def engine(self, device=None):
        if self._engine is None:
            if device is None:
                device = next(self.parameters()).device
            self._engine = Engine(self, torch.device(device), self.compute_dtype)
        return self._engine

    def set_compute_dtype(self, dtype):
        if dtype != self.compute_dtype:
            self.compute_dtype = dtype
            self._drop_engine()
        return self

    def set_align_corners(self, flag):
        flag = bool(flag)
        if flag != self.align_corners:
            self.align_corners = flag
            self._drop_engine()
        return self

    def _drop_engine(self):
        if self._engine is not None:
            self._engine.release_step_graphs()
            for p in self.parameters():       # detach parameters from the flat buffers before they go away
                p.data = p.data.clone()
                p.grad = None
            self._engine = None

    def _apply(self, fn, *a, **k):
        self._drop_engine()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, state_dict, strict=True, **kw):
        sd = {k: v for k, v in state_dict.items()}
        own = self.state_dict()
        for k in own:                          # torch 0.3.1 checkpoints have no num_batches_tracked
            if k.endswith('num_batches_tracked') and k not in sd:
                sd[k] = own[k]
        r = super().load_state_dict(sd, strict=strict, **kw)
        if self._engine is not None:
            self._engine.touch()
        return r

    def dead_parameter_names(self):
        return []

    # -- execution -----------------------------------------------------------------------------------
    def forward(self, x):
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise SaltError('expected a [B,C,H,W] tensor')
        if x.device.type != 'cuda':
            raise SaltError('%s runs only as hand-written HIP kernels on a GPU; got a %s tensor (no CPU/PyTorch fallback)'
                            % (type(self).__name__, x.device.type))
        eng = self.engine(x.device)
        x = x.contiguous().float()
        if self.training and torch.is_grad_enabled():
            from .autograd import HipNetFunction
            return HipNetFunction.apply(self, x, *eng.live_params)
        net = eng.forward(x, self.training)
        return net.logits.clone()


class UNetResNet(HipNetwork):
    """architectures.unet.UNetResNet — the hypercolumn scSE U-Net that main.py trains (models.py:15-19)."""

    def __init__(self, encoder_depth, num_classes, dropout_2d=0.0, pretrained=False, use_hypercolumn=False, pool0=False):
        super().__init__()
        # F.dropout2d(encoder5, p=self.dropout_2d) (unet.py:91): the reference environment pins torch==0.3.1 (environment.yml:17), whose
        # F.dropout2d defaults to training=False - the call is an identity for every p there, and the registry passes p = 0 anyway
        self.num_classes, self.dropout_2d, self.use_hypercolumn = num_classes, dropout_2d, use_hypercolumn
        self.encoders = ResNetEncoders(encoder_depth, pretrained=pretrained, pool0=pool0)
        b = 512 if encoder_depth in (18, 34) else 2048
        self.center = nn.Sequential(Conv2dBnRelu(b, b), Conv2dBnRelu(b, b // 2), nn.AvgPool2d(kernel_size=2, stride=2))
        self.dec5 = DecoderBlock(b + b // 2, b, b // 8)
        self.dec4 = DecoderBlock(b // 2 + b // 8, b // 2, b // 8)
        self.dec3 = DecoderBlock(b // 4 + b // 8, b // 4, b // 8)
        self.dec2 = DecoderBlock(b // 8 + b // 8, b // 8, b // 8)
        self.dec1 = DecoderBlock(b // 8, b // 16, b // 8)
        self.final = nn.Sequential(Conv2dBnRelu((5 if use_hypercolumn else 1) * b // 8, b // 8),
                                   nn.Conv2d(b // 8, num_classes, kernel_size=1, padding=0))
        self.bottom = b

    def dead_parameter_names(self):
        return ['encoders.encoder.fc.weight', 'encoders.encoder.fc.bias']

    def output_shape(self, shape):
        B, _, H, W = shape
        s = 2 if self.encoders.pool0 else 1
        return (B, self.num_classes, H // s, W // s)

    def emit(self, g, x_nchw, logits):
        enc = self.encoders.encoder
        B, _, H, W = x_nchw.shape
        b, d = self.bottom, self.bottom // 8
        exp = 1 if b == 512 else 4
        c1 = g.conv_first(x_nchw, enc.conv1, enc.bn1, relu=True, name='stem')
        if self.encoders.pool0:
            c1 = g.maxpool3s2(c1, name='stem.pool')
            H, W = H // 2, W // 2                    # everything downstream runs at half the resolution
        # concat-free skips: the encoder writes each feature map straight into the decoder's input buffer
        cat5 = g.new_act(B, H // 16, W // 16, b // 2 + b, 'cat5')                   # [up(center) | e5]
        cat4 = g.new_act(B, H // 8, W // 8, d + 256 * exp, 'cat4')                   # [up(dec5)   | e4]
        cat3 = g.new_act(B, H // 4, W // 4, d + 128 * exp, 'cat3')                   # [up(dec4)   | e3]
        cat2 = g.new_act(B, H // 2, W // 2, d + 64 * exp, 'cat2')                    # [up(dec3)   | e2]
        e2 = emit_blocks(g, enc.layer1, c1, out=cat2.slice(d, 64 * exp))
        e3 = emit_blocks(g, enc.layer2, e2, out=cat3.slice(d, 128 * exp))
        e4 = emit_blocks(g, enc.layer3, e3, out=cat4.slice(d, 256 * exp))
        e5 = emit_blocks(g, enc.layer4, e4, out=cat5.slice(b // 2, 512 * exp))
        c = self.center[1].emit(g, self.center[0].emit(g, e5))
        c = g.avgpool2(c, name='center.pool')
        # SALT_HYPER_ROWS (default 0; 1: eval, 2: train too): ONE salt_hyper_rows pass writes the four up-sampled levels of every pixel row
        # instead of four launches into channel slices - measured SLOWER on both streams, kept selectable
        rows_mode = int(os.environ.get('SALT_HYPER_ROWS', '0'))
        fused_rows = self.use_hypercolumn and (rows_mode == 2 or (rows_mode == 1 and not g.train))
        # FACTORED hypercolumn (round 5; DESIGN 4, saltnet.h salt_hyper_stencil): the levels up-sampled by R >= SALT_HYPER_FACTOR (default 4;
        # 0 = off) are never up-sampled, stored or convolved at full resolution.  A 1x1 contraction commutes with the bilinear
        # interpolation and the tap shift, so each such level enters the final convolution as z_k = [W_tap] dec_k - ONE 1x1 launch at the
        # level's own resolution - plus a separable stencil that adds sum_tap shift_tap(up(z_k[tap])) to the convolution over the
        # remaining full-resolution channels [dec1 | up2(dec2)]
        fmin = int(os.environ.get('SALT_HYPER_FACTOR', '4'))
        levels = [(16, 4), (8, 3), (4, 2), (2, 1)]          # (R, channel block of the hypercolumn)
        fact = [(R, k) for R, k in levels if fmin and R >= fmin] if (self.use_hypercolumn and not fused_rows) else []
        if fact and not g.hyper_factor_ok(B, H, W, d, [R for R, _ in fact]):
            fact = []
        nfull = 5 - len(fact)                               # channel blocks that stay at full resolution (contiguous from block 0)
        if fact and sorted(k for _, k in fact) != list(range(nfull, 5)):
            raise SaltError('factored hypercolumn levels must be the deepest ones')
        zs = {}
        if self.use_hypercolumn:
            # PLANAR hypercolumn where the kernels allow it (bf16, conv_ls / conv_ws eligible): five dense [B,H,W,64] planes instead of
            # 320-channel rows.  Every producer (dec1's scSE, the four up-samplings) and every gradient consumer then streams ONE dense
            # tensor instead of 128-byte pieces at a 640-byte pitch (0.6 TB/s at the C4 size); the final convolution, its data gradient
            # and its weight gradient address the planes themselves (salt_conv_args.x_plane / y_plane, salt_conv_wgrad_args.q_plane)
            planes = d if (not fused_rows and nfull > 1 and g.planar_ok(B, H, W, nfull * d, d, self.final[0].conv)) else 0
            hyper = g.new_act(B, H, W, nfull * d, 'hypercolumn', planes=planes)
        # the hypercolumn up-samplings only feed the final convolution: each one goes to the side stream as soon as its decoder
        # level exists and overlaps the remaining decoder levels; the final convolution joins
        def hyper_up(x, R, k):
            if (R, k) in fact:
                with g.side():
                    zs[k] = g.hyper_level(x, self.final[0].conv, k * d, name='hyper.z%d' % k)
            elif self.use_hypercolumn and not fused_rows:
                with g.side():
                    g.upsample(x, R, out=hyper.slice(k * d, d))
        d5 = self.dec5.emit(g, c, e5, cat=cat5)
        hyper_up(d5, 16, 4)
        d4 = self.dec4.emit(g, d5, e4, cat=cat4)
        hyper_up(d4, 8, 3)
        d3 = self.dec3.emit(g, d4, e3, cat=cat3)
        hyper_up(d3, 4, 2)
        d2 = self.dec2.emit(g, d3, e2, cat=cat2)
        hyper_up(d2, 2, 1)
        if fused_rows:
            with g.side():
                g.hyper_rows([d2, d3, d4, d5], [2, 4, 8, 16], hyper.slice(d, 4 * d))
        if self.use_hypercolumn:
            d1 = self.dec1.emit(g, d2, None, out=hyper.slice(0, d))
            g.join()
            if fact:
                ks = sorted(zs)
                # eval: the 1x1 logit head is the block's only consumer - the stencil's epilogue applies it (SALT_HYPER_HEAD=0: separate launch)
                fuse_head = g.head_bn_ok(d, self.final[1]) if g.train else g.hyper_head_ok(d, self.final[1])
                f = g.conv_hyper(hyper, [zs[k] for k in ks], [dict((k_, R_) for R_, k_ in fact)[k] for k in ks], self.final[0].conv,
                                 self.final[0].batch_norm, relu=self.final[0].use_relu, head=(self.final[1], logits) if fuse_head else None)
                if fuse_head:
                    return
            else:
                f = self.final[0].emit(g, hyper)
        else:
            d1 = self.dec1.emit(g, d2, None)
            f = self.final[0].emit(g, d1)
        g.head(f, self.final[1], logits)


class TernausUNetResNet(HipNetwork):
    """unet_models.UNetResNet — TernausNet-style decoder (DecoderBlockV2), exported as unet_models.UNetResNet."""

    def __init__(self, encoder_depth, num_classes, num_filters=32, dropout_2d=0.2, pretrained=False, is_deconv=False):
        super().__init__()
        if dropout_2d != 0.0:
            # the reference environment pins torch==0.3.1 (environment.yml:17) where F.dropout2d(x, p) defaults to
            # training=False, i.e. the call at unet_models.py:150 is an identity there.  We keep that behaviour.
            pass
        if encoder_depth not in (34, 101, 152):
            raise NotImplementedError('only 34, 101, 152 version of Resnet are implemented')
        self.num_classes, self.dropout_2d, self.is_deconv = num_classes, dropout_2d, is_deconv
        self.encoder = resnet(encoder_depth, pretrained)
        b = 512 if encoder_depth == 34 else 2048
        self.pool = nn.MaxPool2d(2, 2)
        self.relu = nn.ReLU(inplace=True)
        self.input_adjust = nn.Sequential(self.encoder.conv1, self.encoder.bn1, self.encoder.relu)
        self.conv1, self.conv2, self.conv3, self.conv4 = self.encoder.layer1, self.encoder.layer2, self.encoder.layer3, self.encoder.layer4
        nf = num_filters
        self.dec4 = DecoderBlockV2(b, nf * 8 * 2, nf * 8, is_deconv)
        self.dec3 = DecoderBlockV2(b // 2 + nf * 8, nf * 8 * 2, nf * 8, is_deconv)
        self.dec2 = DecoderBlockV2(b // 4 + nf * 8, nf * 4 * 2, nf * 2, is_deconv)
        self.dec1 = DecoderBlockV2(b // 8 + nf * 2, nf * 2 * 2, nf * 2 * 2, is_deconv)
        self.final = nn.Conv2d(nf * 2 * 2, num_classes, kernel_size=1)
        self.bottom, self.nf = b, nf

    def dead_parameter_names(self):
        dead = ['encoder.fc.weight', 'encoder.fc.bias']
        for n in ('dec4', 'dec3', 'dec2', 'dec1'):
            blk = getattr(self, n)
            dead += ['%s.%s' % (n, k) for k, _ in blk.named_parameters() if k.startswith(blk.dead_prefix())]
        return dead

    def emit(self, g, x_nchw, logits):
        enc = self.encoder
        B, _, H, W = x_nchw.shape
        b, nf = self.bottom, self.nf
        a = g.conv_first(x_nchw, enc.conv1, enc.bn1, relu=True, name='stem')
        cat3 = g.new_act(B, H // 8, W // 8, nf * 8 + b // 2, 'cat3')        # [dec4 | conv3]
        cat2 = g.new_act(B, H // 4, W // 4, nf * 8 + b // 4, 'cat2')        # [dec3 | conv2]
        cat1 = g.new_act(B, H // 2, W // 2, nf * 2 + b // 8, 'cat1')        # [dec2 | conv1]
        c1 = emit_blocks(g, enc.layer1, a, out=cat1.slice(nf * 2, b // 8))
        c2 = emit_blocks(g, enc.layer2, c1, out=cat2.slice(nf * 8, b // 4))
        c3 = emit_blocks(g, enc.layer3, c2, out=cat3.slice(nf * 8, b // 2))
        ce = emit_blocks(g, enc.layer4, c3)
        self.dec4.emit(g, ce, out=cat3.slice(0, nf * 8))
        self.dec3.emit(g, cat3, out=cat2.slice(0, nf * 8))
        self.dec2.emit(g, cat2, out=cat1.slice(0, nf * 2))
        d1 = self.dec1.emit(g, cat1)
        g.head(d1, self.final, logits)


class _SaltResNet34Stub(HipNetwork):
    """Shared plumbing of SaltUNet / SaltLinkNet: a whole torchvision-layout resnet34 is constructed (its unused blocks stay in
    ``state_dict()`` and ``parameters()`` exactly like the reference) but only a few BasicBlocks are on the path."""

    used_blocks = ()

    def _make(self, num_classes, dropout_2d, pretrained):
        self.num_classes, self.dropout_2d = num_classes, dropout_2d
        self.encoder = resnet(34, pretrained)
        self.relu = nn.ReLU(inplace=True)
        self.input_adjust = nn.Sequential(self.encoder.conv1, self.encoder.bn1, self.encoder.relu)

    def dead_parameter_names(self):
        live = ['encoder.conv1.', 'encoder.bn1.'] + ['encoder.%s.' % b for b in self.used_blocks]
        dead = [k for k, _ in self.encoder.named_parameters(prefix='encoder') if not any(k.startswith(l) for l in live)]
        for n in self.decoders:
            blk = getattr(self, n)
            dead += ['%s.%s' % (n, k) for k, _ in blk.named_parameters() if k.startswith(blk.dead_prefix())]
        return dead


class SaltUNet(_SaltResNet34Stub):
    """unet_models.SaltUNet (unet_models.py:154-189): stem, layer1[1], layer1[2], layer2[0], layer2[1]; DecoderBlockV2 / ConvBnRelu
    decoders over concatenated skips; 1x1 head.  layer1[0] and everything after layer2[1] is constructed but never called."""

    used_blocks = ('layer1.1', 'layer1.2', 'layer2.0', 'layer2.1')
    decoders = ('dec3', 'dec1')

    def __init__(self, num_classes, dropout_2d=0.2, pretrained=False, is_deconv=False):
        super().__init__()
        self._make(num_classes, dropout_2d, pretrained)
        self.conv1 = self.encoder.layer1[1]
        self.conv2 = self.encoder.layer1[2]
        self.conv3 = self.encoder.layer2[0]
        self.conv4 = self.encoder.layer2[1]
        self.dec3 = DecoderBlockV2(256, 512, 256, is_deconv)
        self.dec2 = ConvBnRelu(256 + 64, 256)
        self.dec1 = DecoderBlockV2(256 + 64, (256 + 64) * 2, 256, is_deconv)
        self.final = nn.Conv2d(256, num_classes, kernel_size=1)

    def emit(self, g, x_nchw, logits):
        enc = self.encoder
        B, _, H, W = x_nchw.shape
        a = g.conv_first(x_nchw, enc.conv1, enc.bn1, relu=True, name='stem')
        cat1 = g.new_act(B, H // 2, W // 2, 256 + 64, 'cat1')            # [dec2 | conv1]
        cat2 = g.new_act(B, H // 2, W // 2, 256 + 64, 'cat2')            # [dec3 | conv2]
        cat3 = g.new_act(B, H // 4, W // 4, 128 + 128, 'cat3')           # [center | conv3]
        c1 = self.conv1.emit(g, a, out=cat1.slice(256, 64))
        c2 = self.conv2.emit(g, c1, out=cat2.slice(256, 64))
        c3 = self.conv3.emit(g, c2, out=cat3.slice(128, 128))
        self.conv4.emit(g, c3, out=cat3.slice(0, 128))
        self.dec3.emit(g, cat3, out=cat2.slice(0, 256))
        self.dec2.emit(g, cat2, out=cat1.slice(0, 256))
        d1 = self.dec1.emit(g, cat1)
        g.head(d1, self.final, logits)


class SaltLinkNet(_SaltResNet34Stub):
    """unet_models.SaltLinkNet (unet_models.py:192-233): block outputs of layer1[1..2] and layer2[0..3] are summed per stage."""

    used_blocks = ('layer1.1', 'layer1.2', 'layer2.0', 'layer2.1', 'layer2.2', 'layer2.3')
    decoders = ('dec2', 'dec1')

    def __init__(self, num_classes, dropout_2d=0.2, pretrained=False, is_deconv=False):
        super().__init__()
        self._make(num_classes, dropout_2d, pretrained)
        self.conv1_1 = self.encoder.layer1[1]
        self.conv1_2 = self.encoder.layer1[2]
        self.conv2_0 = self.encoder.layer2[0]
        self.conv2_1 = self.encoder.layer2[1]
        self.conv2_2 = self.encoder.layer2[2]
        self.conv2_3 = self.encoder.layer2[3]
        self.dec2 = DecoderBlockV2(128, 256, 256, is_deconv=is_deconv)
        self.dec1 = DecoderBlockV2(256 + 64, 512, 256, is_deconv=is_deconv)
        self.final = nn.Conv2d(256, num_classes, kernel_size=1)

    def emit(self, g, x_nchw, logits):
        enc = self.encoder
        B, _, H, W = x_nchw.shape
        a = g.conv_first(x_nchw, enc.conv1, enc.bn1, relu=True, name='stem')
        c11 = self.conv1_1.emit(g, a)
        c12 = self.conv1_2.emit(g, c11)
        c20 = self.conv2_0.emit(g, c12)
        c21 = self.conv2_1.emit(g, c20)
        c22 = self.conv2_2.emit(g, c21)
        c23 = self.conv2_3.emit(g, c22)
        cat1 = g.new_act(B, H // 2, W // 2, 256 + 64, 'cat1')            # [dec2 | conv1_sum]
        g.add(c11, c12, out=cat1.slice(256, 64), name='conv1_sum')
        s2 = g.add(g.add(g.add(c20, c21), c22), c23, name='conv2_sum')   # left-to-right like the reference expression
        self.dec2.emit(g, s2, out=cat1.slice(0, 256))
        d1 = self.dec1.emit(g, cat1)
        g.head(d1, self.final, logits)


class VanillaUNet(HipNetwork):
    """BASELINE C0/C1 "vanilla 4-level U-Net": the reference has no in-tree definition; it is
    assembled from the reference's own blocks: per level 2 x ConvBnRelu + MaxPool2d(2,2); centre 2 x ConvBnRelu;
    per level DeconvConv2dBnRelu (ConvT k3 s2 p1 op1 + BN + ReLU), concat skip, 2 x ConvBnRelu; 1x1 head."""

    def __init__(self, num_classes=2, in_channels=1, base_filters=16, levels=4):
        super().__init__()
        self.num_classes, self.levels = num_classes, levels
        c = in_channels
        for i in range(1, levels + 1):
            f = base_filters * 2 ** (i - 1)
            setattr(self, 'enc%d' % i, nn.Sequential(ConvBnRelu(c, f), ConvBnRelu(f, f)))
            c = f
        f = base_filters * 2 ** levels
        self.center = nn.Sequential(ConvBnRelu(c, f), ConvBnRelu(f, f))
        c = f
        for i in range(levels, 0, -1):
            f = base_filters * 2 ** (i - 1)
            setattr(self, 'up%d' % i, DeconvConv2dBnRelu(c, f))
            setattr(self, 'dec%d' % i, nn.Sequential(ConvBnRelu(2 * f, f), ConvBnRelu(f, f)))
            c = f
        self.final = nn.Conv2d(c, num_classes, kernel_size=1)
        self.base = base_filters

    def emit(self, g, x_nchw, logits):
        B, _, H, W = x_nchw.shape
        L = self.levels
        cats = {}
        x = None
        for i in range(1, L + 1):
            f = self.base * 2 ** (i - 1)
            h, w = H >> (i - 1), W >> (i - 1)
            cats[i] = g.new_act(B, h, w, 2 * f, 'cat%d' % i)                 # [up_i | enc_i]
            enc = getattr(self, 'enc%d' % i)
            if i == 1:
                c0 = enc[0].conv
                a = g.conv_first(x_nchw, c0[0], c0[1], relu=True, name='enc1.0')
            else:
                a = enc[0].emit(g, x)
            skip = enc[1].emit(g, a, out=cats[i].slice(f, f))
            x = g.maxpool2(skip, name='pool%d' % i)
        x = self.center[1].emit(g, self.center[0].emit(g, x))
        for i in range(L, 0, -1):
            f = self.base * 2 ** (i - 1)
            getattr(self, 'up%d' % i).emit(g, x, out=cats[i].slice(0, f))
            dec = getattr(self, 'dec%d' % i)
            x = dec[1].emit(g, dec[0].emit(g, cats[i]))
        g.head(x, self.final, logits)
