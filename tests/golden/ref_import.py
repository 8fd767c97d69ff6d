"""Import the reference's hot-path modules in THIS container (never on the GPU box, never shipped).

/root/reference is a Python 3.6 / torch 0.3.1 / torchvision 0.2.0 code base whose third-party
dependencies are mostly absent here.  This helper installs ``sys.modules`` stubs for
the missing packages so that the reference's own source files can be executed by torch 2.10 and
used to (a) generate the golden vectors committed under tests/golden/ and (b) validate the oracle.

Stubs (all build-authored; none of this is reference code):
  torchvision.models.resnet{18,34,50,101,152}  ->  TvResNet below: the public torchvision layout
      (stem conv7x7 s2 p3 bias-free, BN, ReLU, maxpool 3x3 s2 p1; BasicBlock / Bottleneck with
      stride on the 3x3; 1x1-stride downsample + BN; avgpool + fc(1000)).
  pretrainedmodels                              ->  empty module (SE/DenseNet encoders unused)
  common_blocks (package shell) + common_blocks.utils  ->  pytorch_where / sigmoid / softmax /
      get_list_of_image_predictions (the real utils.py fails to import: collections.Iterable, cv2 …)
  toolkit.pytorch_transformers.models.Model, common_blocks.callbacks  ->  minimal shells
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
from torch import nn

REFERENCE_ROOT = os.environ.get('SALT_REFERENCE_ROOT', '/root/reference')


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'common_blocks'))


# --------------------------------------------------------------------------- torchvision stub

class _Basic(nn.Module):
    expansion = 1

    def __init__(self, inpl, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inpl, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = down
        self.stride = stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + idt)


class _Bottle(nn.Module):
    expansion = 4

    def __init__(self, inpl, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(inpl, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = down
        self.stride = stride

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + idt)


class TvResNet(nn.Module):
    def __init__(self, block, counts):
        super().__init__()
        self.inpl = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(block, 64, counts[0], 1)
        self.layer2 = self._make(block, 128, counts[1], 2)
        self.layer3 = self._make(block, 256, counts[2], 2)
        self.layer4 = self._make(block, 512, counts[3], 2)
        self.avgpool = nn.AvgPool2d(7, stride=1)
        self.fc = nn.Linear(512 * block.expansion, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, (2. / n) ** 0.5)

    def _make(self, block, planes, n, stride):
        down = None
        if stride != 1 or self.inpl != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self.inpl, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inpl, planes, stride, down)]
        self.inpl = planes * block.expansion
        layers += [block(self.inpl, planes, 1, None) for _ in range(1, n)]
        return nn.Sequential(*layers)


def _tv_factory(block, counts):
    def make(pretrained=False, **kw):
        if pretrained:
            raise RuntimeError('no network: pretrained weights unavailable')
        return TvResNet(block, counts)
    return make


_installed = {}


def install_stubs():
    if _installed:
        return _installed
    tv = types.ModuleType('torchvision')
    tvm = types.ModuleType('torchvision.models')
    for name, (blk, cnt) in {'resnet18': (_Basic, [2, 2, 2, 2]), 'resnet34': (_Basic, [3, 4, 6, 3]),
                             'resnet50': (_Bottle, [3, 4, 6, 3]), 'resnet101': (_Bottle, [3, 4, 23, 3]),
                             'resnet152': (_Bottle, [3, 8, 36, 3])}.items():
        setattr(tvm, name, _tv_factory(blk, cnt))
    tv.models = tvm
    sys.modules.setdefault('torchvision', tv)
    sys.modules.setdefault('torchvision.models', tvm)
    sys.modules.setdefault('pretrainedmodels', types.ModuleType('pretrainedmodels'))

    cb = types.ModuleType('common_blocks')
    cb.__path__ = [os.path.join(REFERENCE_ROOT, 'common_blocks')]
    sys.modules['common_blocks'] = cb
    ut = types.ModuleType('common_blocks.utils')
    ut.pytorch_where = lambda cond, a, b: cond.float() * a + (1 - cond.float()) * b
    ut.sigmoid = lambda x: 1. / (1 + np.exp(-x))

    def _softmax(X, theta=1.0, axis=None):
        y = np.atleast_2d(X) * float(theta)
        y = np.exp(y - np.expand_dims(np.max(y, axis=axis), axis))
        return y / np.expand_dims(np.sum(y, axis=axis), axis)
    ut.softmax = _softmax
    ut.get_list_of_image_predictions = lambda bp: [im for b in bp for im in list(b)]
    sys.modules['common_blocks.utils'] = ut

    tk = types.ModuleType('toolkit'); tkp = types.ModuleType('toolkit.pytorch_transformers')
    tkm = types.ModuleType('toolkit.pytorch_transformers.models')

    class Model:                                   # shell of steppy-toolkit 0.1.5 Model (un-vendored)
        def __init__(self, architecture_config, training_config, callbacks_config):
            self.architecture_config = architecture_config
            self.training_config = training_config
            self.callbacks_config = callbacks_config
    tkm.Model = Model
    sys.modules['toolkit'] = tk; sys.modules['toolkit.pytorch_transformers'] = tkp
    sys.modules['toolkit.pytorch_transformers.models'] = tkm
    cbk = types.ModuleType('common_blocks.callbacks')
    sys.modules['common_blocks.callbacks'] = cbk
    _installed['ok'] = True
    return _installed


def load(name):
    """Import ``common_blocks.<name>`` from the reference (e.g. 'architectures.unet', 'lovasz_losses')."""
    install_stubs()
    return importlib.import_module('common_blocks.' + name)


def load_models_module():
    """``common_blocks.models`` needs several sibling architectures that fail to import; stub them."""
    install_stubs()
    for extra in ('large_kernel_matters', 'misc', 'models_with_depth', 'pspnet'):
        full = 'common_blocks.architectures.' + extra
        try:
            importlib.import_module(full)
        except Exception:
            m = types.ModuleType(full)
            for cls in ('LargeKernelMatters', 'StackingFCN', 'StackingFCNWithDepth', 'EmptinessClassifier',
                        'UNetResNetWithDepth', 'PSPNet'):
                setattr(m, cls, type(cls, (), {}))
            sys.modules[full] = m
            setattr(importlib.import_module('common_blocks.architectures'), extra, m)
    return importlib.import_module('common_blocks.models')
