"""Oracle: building blocks of the reference U-Nets as pure functions (TEST INFRASTRUCTURE ONLY).

Every function takes ``sd`` (flat ``{state_dict key: tensor}``), a key prefix ``p`` ending in
'.', (or ''), the NCHW fp32 input and ``train`` (BatchNorm mode).  Key names are exactly the ones
the reference's ``nn.Module`` tree produces, so one closed-form weight set drives all three
implementations (reference import, this oracle, HIP path).

Reference anchors (relative to /root/reference/common_blocks):
  unet_models.py:21-30   ConvBnRelu            -> conv_bn_relu
  unet_models.py:38-50   DecoderBlockV1        -> decoder_block_v1
  unet_models.py:53-75   DecoderBlockV2        -> decoder_block_v2
  architectures/base.py:7-37    Conv2dBnRelu   -> conv2d_bn_relu
  architectures/base.py:40-57   DeconvConv2dBnRelu -> deconv_conv2d_bn_relu
  architectures/base.py:65-86   DecoderBlock   -> decoder_block
  architectures/base.py:89-104  ChannelSELayer -> channel_se
  architectures/base.py:107-117 SpatialSELayer -> spatial_se
  torchvision 0.2.0 models/resnet.py (un-vendored; environment.yml:18): BasicBlock, Bottleneck
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ------------------------------------------------------------------ storage-precision emulation (bf16 whole-network tests)
# The HIP bf16 path keeps fp32 accumulators / statistics / parameters but STORES every activation, every activation gradient and
# the packed convolution weights in bf16.  ``with bf16_storage():`` makes the oracle round at the same tensor boundaries (forward
# value and, through a straight-through autograd function, the gradient that flows back through the same boundary), so a test can
# compare "HIP bf16 vs fp32" with "what bf16 storage costs an independent implementation".  Default: identity (plain fp32 oracle).
class _RoundBoth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().to(g.dtype)


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


_EMULATE_BF16 = False


def _st(x):
    """an activation that the HIP path stores (and whose gradient it stores) in the compute dtype"""
    return _RoundBoth.apply(x) if _EMULATE_BF16 else x


def _w(w):
    """a convolution weight as the MFMA kernels read it (packed compute-dtype copy of the fp32 master)"""
    return _RoundFwd.apply(w) if _EMULATE_BF16 else w


class bf16_storage:
    def __enter__(self):
        global _EMULATE_BF16
        self._old, _EMULATE_BF16 = _EMULATE_BF16, True

    def __exit__(self, *a):
        global _EMULATE_BF16
        _EMULATE_BF16 = self._old


def batch_norm(sd, p, x, train):
    """nn.BatchNorm2d with torch defaults (eps 1e-5, momentum 0.1, affine, running stats)."""
    if train and (p + 'num_batches_tracked') in sd:
        sd[p + 'num_batches_tracked'] += 1
    return F.batch_norm(x, sd[p + 'running_mean'], sd[p + 'running_var'],
                        sd[p + 'weight'], sd[p + 'bias'], train, BN_MOMENTUM, BN_EPS)


def conv_bn_relu(sd, p, x, train):
    """unet_models.ConvBnRelu: 3x3 conv, zero pad 1, bias -> BN -> ReLU.  Keys p+'conv.{0,1}.*'."""
    y = _st(F.conv2d(x, _w(sd[p + 'conv.0.weight']), sd[p + 'conv.0.bias'], padding=1))
    return _st(F.relu(batch_norm(sd, p + 'conv.1.', y, train)))


def conv2d_bn_relu(sd, p, x, train, use_relu=True, use_batch_norm=True, use_padding=True):
    """base.Conv2dBnRelu: replicate-pad (left 0, right kh-1, top kw-1, bottom 0) -> conv pad 0 -> BN -> ReLU.

    ``kernel_size=(kw, kh)`` in the reference is handed to nn.Conv2d unchanged, i.e. torch reads it
    as (rows, cols); the pad is (left=0, right=kernel_size[1]-1, top=kernel_size[0]-1, bottom=0).
    """
    w = sd[p + 'conv.weight']
    rows, cols = w.shape[2], w.shape[3]
    if use_padding:
        x = F.pad(x, (0, cols - 1, rows - 1, 0), mode='replicate')
    y = _st(F.conv2d(x, _w(w), sd[p + 'conv.bias']))
    if use_batch_norm:
        y = batch_norm(sd, p + 'batch_norm.', y, train)
    if use_relu:
        y = F.relu(y)
    return _st(y) if (use_batch_norm or use_relu) else y


def deconv_conv2d_bn_relu(sd, p, x, train, use_relu=True, use_batch_norm=True):
    """base.DeconvConv2dBnRelu: ConvTranspose2d k3 s2 p1 op1 -> BN -> ReLU."""
    y = _st(F.conv_transpose2d(x, _w(sd[p + 'deconv.weight']), sd[p + 'deconv.bias'],
                               stride=2, padding=1, output_padding=1))
    if use_batch_norm:
        y = batch_norm(sd, p + 'batch_norm.', y, train)
    if use_relu:
        y = F.relu(y)
    return _st(y) if (use_batch_norm or use_relu) else y


ALIGN_CORNERS = False     # tests flip it to restate torch 0.3.1's evaluation of the same nn.Upsample call (version drift)


def upsample_bilinear(x, scale):
    """nn.Upsample(mode='bilinear') as executed by torch 2.10 (align_corners=False)."""
    return _st(F.interpolate(x, scale_factor=scale, mode='bilinear', align_corners=ALIGN_CORNERS))


def decoder_block_v1(sd, p, x, train):
    """unet_models.DecoderBlockV1: ConvBnRelu -> ConvT k3 s2 p1 op1 -> BN -> ReLU."""
    y = conv_bn_relu(sd, p + 'block.0.', x, train)
    y = _st(F.conv_transpose2d(y, _w(sd[p + 'block.1.weight']), sd[p + 'block.1.bias'],
                               stride=2, padding=1, output_padding=1))
    return _st(F.relu(batch_norm(sd, p + 'block.2.', y, train)))


def decoder_block_v2(sd, p, x, train, is_deconv=True):
    """unet_models.DecoderBlockV2: deconv branch (ConvBnRelu, ConvT k4 s2 p1, BN, ReLU) or
    upsample branch (ConvBnRelu, bilinear x2).  Both branches own parameters; one runs."""
    if is_deconv:
        y = conv_bn_relu(sd, p + 'deconv.0.', x, train)
        y = _st(F.conv_transpose2d(y, _w(sd[p + 'deconv.1.weight']), sd[p + 'deconv.1.bias'], stride=2, padding=1))
        return _st(F.relu(batch_norm(sd, p + 'deconv.2.', y, train)))
    y = conv_bn_relu(sd, p + 'upsample.0.', x, train)
    return upsample_bilinear(y, 2)


def channel_se(sd, p, x):
    """base.ChannelSELayer: x * sigmoid(W2 relu(W1 gap(x) + b1) + b2)."""
    g = x.mean(dim=(2, 3))
    h = F.relu(F.linear(g, sd[p + 'fc.0.weight'], sd[p + 'fc.0.bias']))
    s = torch.sigmoid(F.linear(h, sd[p + 'fc.2.weight'], sd[p + 'fc.2.bias']))
    return x * s[:, :, None, None]


def spatial_se(sd, p, x):
    """base.SpatialSELayer: x * sigmoid(conv1x1(x))."""
    return x * torch.sigmoid(F.conv2d(x, sd[p + 'fc.weight'], sd[p + 'fc.bias']))


def decoder_block(sd, p, x, e, train):
    """base.DecoderBlock: bilinear x2 -> cat skip -> 2x Conv2dBnRelu -> cSE + sSE -> ReLU."""
    x = upsample_bilinear(x, 2)
    if e is not None:
        x = torch.cat([x, e], 1)
    x = conv2d_bn_relu(sd, p + 'conv1.', x, train)
    x = conv2d_bn_relu(sd, p + 'conv2.', x, train)
    return _st(F.relu(channel_se(sd, p + 'channel_se.', x) + spatial_se(sd, p + 'spatial_se.', x)))


# ------------------------------------------------------------------ torchvision-layout ResNet

RESNET_CFG = {18: ('basic', [2, 2, 2, 2]), 34: ('basic', [3, 4, 6, 3]), 50: ('bottle', [3, 4, 6, 3]),
              101: ('bottle', [3, 4, 23, 3]), 152: ('bottle', [3, 8, 36, 3])}


def basic_block(sd, p, x, train, stride):
    y = _st(F.conv2d(x, _w(sd[p + 'conv1.weight']), None, stride=stride, padding=1))
    y = _st(F.relu(batch_norm(sd, p + 'bn1.', y, train)))
    y = _st(F.conv2d(y, _w(sd[p + 'conv2.weight']), None, padding=1))
    y = batch_norm(sd, p + 'bn2.', y, train)
    if (p + 'downsample.0.weight') in sd:
        x = _st(F.conv2d(x, _w(sd[p + 'downsample.0.weight']), None, stride=stride))
        x = _st(batch_norm(sd, p + 'downsample.1.', x, train))
    return _st(F.relu(y + x))


def bottleneck(sd, p, x, train, stride):
    """torchvision 0.2.0 Bottleneck: 1x1 -> 3x3 (stride here) -> 1x1 (x4), BN after each."""
    y = _st(F.relu(batch_norm(sd, p + 'bn1.', _st(F.conv2d(x, _w(sd[p + 'conv1.weight']))), train)))
    y = _st(F.relu(batch_norm(sd, p + 'bn2.', _st(F.conv2d(y, _w(sd[p + 'conv2.weight']), None, stride=stride, padding=1)), train)))
    y = batch_norm(sd, p + 'bn3.', _st(F.conv2d(y, _w(sd[p + 'conv3.weight']))), train)
    if (p + 'downsample.0.weight') in sd:
        x = _st(F.conv2d(x, _w(sd[p + 'downsample.0.weight']), None, stride=stride))
        x = _st(batch_norm(sd, p + 'downsample.1.', x, train))
    return _st(F.relu(y + x))


def resnet_stem(sd, p, x, train, pool0=False):
    """conv7x7 s2 p3 (no bias) -> BN -> ReLU [-> MaxPool 3x3 s2 p1 iff pool0]."""
    y = _st(F.conv2d(_st(x), _w(sd[p + 'conv1.weight']), None, stride=2, padding=3))
    y = _st(F.relu(batch_norm(sd, p + 'bn1.', y, train)))
    if pool0:
        y = F.max_pool2d(y, 3, 2, 1)
    return y


def resnet_layer(sd, p, x, train, depth, idx, blocks=None):
    """layer{idx} of a torchvision ResNet; ``blocks`` optionally restricts to some block indices."""
    kind, counts = RESNET_CFG[depth]
    fn = basic_block if kind == 'basic' else bottleneck
    for b in range(counts[idx - 1]):
        if blocks is not None and b not in blocks:
            continue
        stride = 2 if (b == 0 and idx > 1) else 1
        x = fn(sd, '%slayer%d.%d.' % (p, idx, b), x, train, stride)
    return x
