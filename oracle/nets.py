"""Oracle: whole-network forwards as pure functions over a flat state dict (TEST INFRASTRUCTURE ONLY).

Reference anchors (relative to /root/reference/common_blocks):
  architectures/unet.py:22-109      UNetResNet (hypercolumn U-Net main.py trains) -> unet_resnet
  architectures/encoders.py:6-45    ResNetEncoders                                -> inside unet_resnet
  unet_models.py:78-151             UNetResNet (TernausNet style, DecoderBlockV2) -> ternaus_unet_resnet
  unet_models.py:154-189            SaltUNet                                      -> salt_unet
  unet_models.py:192-233            SaltLinkNet                                   -> salt_linknet
  (no in-tree definition)           "vanilla 4-level U-Net" of BASELINE C0/C1     -> vanilla_unet
      built from unet_models.ConvBnRelu, nn.MaxPool2d(2,2) (unet_models.py:119) and
      ConvTranspose2d(k3,s2,p1,op1)+BN+ReLU (base.py:40-57), 16*2^i filters.

Canonical keys: the reference registers shared torchvision sub-modules under several names
(e.g. ``encoders.encoder.layer1`` and ``encoders.encoder2``).  The oracle reads the
``…encoder.*`` spelling; specs.alias_map() lists the extra spellings.
"""
import torch
import torch.nn.functional as F

from . import blocks as B


def unet_resnet(sd, x, train, depth=34, use_hypercolumn=True, pool0=False, p=''):
    """architectures.unet.UNetResNet.forward (unet.py:89-109); dropout_2d = 0."""
    e = p + 'encoders.encoder.'
    c1 = B.resnet_stem(sd, e, x, train, pool0)
    e2 = B.resnet_layer(sd, e, c1, train, depth, 1)
    e3 = B.resnet_layer(sd, e, e2, train, depth, 2)
    e4 = B.resnet_layer(sd, e, e3, train, depth, 3)
    e5 = B.resnet_layer(sd, e, e4, train, depth, 4)
    c = B.conv2d_bn_relu(sd, p + 'center.0.', e5, train)
    c = B.conv2d_bn_relu(sd, p + 'center.1.', c, train)
    c = B._st(F.avg_pool2d(c, 2, 2))
    d5 = B.decoder_block(sd, p + 'dec5.', c, e5, train)
    d4 = B.decoder_block(sd, p + 'dec4.', d5, e4, train)
    d3 = B.decoder_block(sd, p + 'dec3.', d4, e3, train)
    d2 = B.decoder_block(sd, p + 'dec2.', d3, e2, train)
    d1 = B.decoder_block(sd, p + 'dec1.', d2, None, train)
    if use_hypercolumn:
        d1 = torch.cat([d1, B.upsample_bilinear(d2, 2), B.upsample_bilinear(d3, 4),
                        B.upsample_bilinear(d4, 8), B.upsample_bilinear(d5, 16)], 1)
    y = B.conv2d_bn_relu(sd, p + 'final.0.', d1, train)
    return F.conv2d(y, sd[p + 'final.1.weight'], sd[p + 'final.1.bias'])


def ternaus_unet_resnet(sd, x, train, depth=34, is_deconv=True, p=''):
    """unet_models.UNetResNet.forward (unet_models.py:140-151); dropout_2d = 0; no stem max-pool."""
    e = p + 'encoder.'
    a = B.resnet_stem(sd, e, x, train, False)
    c1 = B.resnet_layer(sd, e, a, train, depth, 1)
    c2 = B.resnet_layer(sd, e, c1, train, depth, 2)
    c3 = B.resnet_layer(sd, e, c2, train, depth, 3)
    ce = B.resnet_layer(sd, e, c3, train, depth, 4)
    d4 = B.decoder_block_v2(sd, p + 'dec4.', ce, train, is_deconv)
    d3 = B.decoder_block_v2(sd, p + 'dec3.', torch.cat([d4, c3], 1), train, is_deconv)
    d2 = B.decoder_block_v2(sd, p + 'dec2.', torch.cat([d3, c2], 1), train, is_deconv)
    d1 = B.decoder_block_v2(sd, p + 'dec1.', torch.cat([d2, c1], 1), train, is_deconv)
    return F.conv2d(d1, sd[p + 'final.weight'], sd[p + 'final.bias'])


def salt_unet(sd, x, train, is_deconv=True, p=''):
    """unet_models.SaltUNet.forward (unet_models.py:179-189): layer1 blocks 1,2 and layer2 blocks 0,1."""
    e = p + 'encoder.'
    a = B.resnet_stem(sd, e, x, train, False)
    c1 = B.resnet_layer(sd, e, a, train, 34, 1, blocks=[1])
    c2 = B.resnet_layer(sd, e, c1, train, 34, 1, blocks=[2])
    c3 = B.resnet_layer(sd, e, c2, train, 34, 2, blocks=[0])
    ce = B.resnet_layer(sd, e, c3, train, 34, 2, blocks=[1])
    d3 = B.decoder_block_v2(sd, p + 'dec3.', torch.cat([ce, c3], 1), train, is_deconv)
    d2 = B.conv_bn_relu(sd, p + 'dec2.', torch.cat([d3, c2], 1), train)
    d1 = B.decoder_block_v2(sd, p + 'dec1.', torch.cat([d2, c1], 1), train, is_deconv)
    return F.conv2d(d1, sd[p + 'final.weight'], sd[p + 'final.bias'])


def salt_linknet(sd, x, train, is_deconv=True, p=''):
    """unet_models.SaltLinkNet.forward (unet_models.py:218-233)."""
    e = p + 'encoder.'
    a = B.resnet_stem(sd, e, x, train, False)
    c11 = B.resnet_layer(sd, e, a, train, 34, 1, blocks=[1])
    c12 = B.resnet_layer(sd, e, c11, train, 34, 1, blocks=[2])
    c20 = B.resnet_layer(sd, e, c12, train, 34, 2, blocks=[0])
    c21 = B.resnet_layer(sd, e, c20, train, 34, 2, blocks=[1])
    c22 = B.resnet_layer(sd, e, c21, train, 34, 2, blocks=[2])
    c23 = B.resnet_layer(sd, e, c22, train, 34, 2, blocks=[3])
    d2 = B.decoder_block_v2(sd, p + 'dec2.', c20 + c21 + c22 + c23, train, is_deconv)
    d1 = B.decoder_block_v2(sd, p + 'dec1.', torch.cat([d2, c11 + c12], 1), train, is_deconv)
    return F.conv2d(d1, sd[p + 'final.weight'], sd[p + 'final.bias'])


def vanilla_unet(sd, x, train, levels=4, p=''):
    """Build-defined vanilla U-Net (BASELINE C0/C1): per level two ConvBnRelu + MaxPool2d(2,2);
    center two ConvBnRelu; per level ConvT(k3,s2,p1,op1)+BN+ReLU, cat skip, two ConvBnRelu; 1x1 head."""
    skips = []
    for i in range(1, levels + 1):
        x = B.conv_bn_relu(sd, '%senc%d.0.' % (p, i), x, train)
        x = B.conv_bn_relu(sd, '%senc%d.1.' % (p, i), x, train)
        skips.append(x)
        x = F.max_pool2d(x, 2, 2)
    x = B.conv_bn_relu(sd, p + 'center.0.', x, train)
    x = B.conv_bn_relu(sd, p + 'center.1.', x, train)
    for i in range(levels, 0, -1):
        x = B.deconv_conv2d_bn_relu(sd, '%sup%d.' % (p, i), x, train)
        x = torch.cat([x, skips[i - 1]], 1)
        x = B.conv_bn_relu(sd, '%sdec%d.0.' % (p, i), x, train)
        x = B.conv_bn_relu(sd, '%sdec%d.1.' % (p, i), x, train)
    return F.conv2d(x, sd[p + 'final.weight'], sd[p + 'final.bias'])


FORWARDS = {
    'UNetResNet': unet_resnet,                 # architectures.unet.UNetResNet (models.py:15-19)
    'TernausUNetResNet': ternaus_unet_resnet,  # unet_models.UNetResNet
    'SaltUNet': salt_unet,
    'SaltLinkNet': salt_linknet,
    'VanillaUNet': vanilla_unet,
}
