"""Drop-in for the reference's ``common_blocks/models.py`` on the U-Net hot path.

Same names and call contract as the reference (models.py:15-208,289-340):
  ARCHITECTURES registry, SegmentationModel(architecture_config, training_config, callbacks_config) with
  fit / _fit_loop / transform / _transform / set_model / set_loss / load (+ persist, fit_transform from the
  un-vendored steppy-toolkit ``Model`` base, restated from its call sites), weight_regularization,
  lovasz_loss, mixed_dice_bce_loss.

What is different underneath: ``self.model`` is a HipNetwork (compiled static HIP programs, NHWC, flat
parameter buffers), the optimizer is one fused Adam kernel, ``_fit_loop`` runs forward -> loss -> backward
-> (bucketed RCCL all-reduce overlapped with backward) -> Adam without touching torch autograd, and
``nn.DataParallel`` (models.py:81-85) is replaced by one process per GPU (parallel.py).
"""
import os

import numpy as np
import torch

from . import architectures as A
from . import parallel
from ._abi import SaltError
from .losses import lovasz_loss, mixed_dice_bce_loss
from .optim import FusedAdam, weight_regularization

ARCHITECTURES = {
    'UNetResNet': {'model': A.UNetResNet,
                   'model_config': {'encoder_depth': 34, 'use_hypercolumn': True, 'dropout_2d': 0.0, 'pretrained': False, 'pool0': False},
                   'init_weights': False},
    # unet_models.py zoo (named by BASELINE.json's north_star; not wired into the reference's registry)
    'TernausUNetResNet': {'model': A.TernausUNetResNet,
                          'model_config': {'encoder_depth': 34, 'num_filters': 32, 'dropout_2d': 0.0, 'pretrained': False, 'is_deconv': True},
                          'init_weights': False},
    'UNetResNet152': {'model': A.UNetResNet,
                      'model_config': {'encoder_depth': 152, 'use_hypercolumn': True, 'dropout_2d': 0.0, 'pretrained': False, 'pool0': False},
                      'init_weights': False},
    'SaltUNet': {'model': A.SaltUNet, 'model_config': {'dropout_2d': 0.0, 'pretrained': False, 'is_deconv': True}, 'init_weights': False},
    'SaltLinkNet': {'model': A.SaltLinkNet, 'model_config': {'dropout_2d': 0.0, 'pretrained': False, 'is_deconv': True}, 'init_weights': False},
    'VanillaUNet': {'model': A.VanillaUNet, 'model_config': {'in_channels': 1, 'base_filters': 16, 'levels': 4}, 'init_weights': False},
}


def sigmoid(x):                      # utils.py:173-174
    return 1. / (1 + np.exp(-x))


def softmax(X, axis=0):
    y = np.exp(X - np.max(X, axis=axis, keepdims=True))
    return y / np.sum(y, axis=axis, keepdims=True)


def get_list_of_image_predictions(batch_predictions):      # utils.py:316-320
    return [img for batch in batch_predictions for img in list(batch)]


from .callbacks import Callback, CallbackList, callbacks_network, score_validation  # noqa: E402,F401  (callbacks.py surface)


class Model:
    """steppy-toolkit 0.1.5 ``toolkit.pytorch_transformers.models.Model`` restated from its call sites
    (models.py:6,67-70; callbacks.py:42-48; utils.py:444-467): holds the three configs, fit_transform, persist."""

    def __init__(self, architecture_config, training_config, callbacks_config):
        self.architecture_config = architecture_config
        self.training_config = training_config
        self.callbacks_config = callbacks_config
        self.model = None
        self.optimizer = None
        self.loss_function = None
        self.callbacks = None
        self.validation_loss = {}

    @property
    def output_names(self):
        return [name for (name, _, _) in self.loss_function]

    def fit_transform(self, *args, **kwargs):
        return self.fit(*args, **kwargs).transform(*args, **kwargs)

    def score_validation(self, validation_datagen):
        """{'sum', 'iou', 'iout'} of one pass over the validation generator (callbacks.py:503-568), threshold sweep on the GPU."""
        return score_validation(self, validation_datagen)

    def persist(self, filepath):
        """Reference checkpoints are written through nn.DataParallel, so every key carries a 'module.' prefix
        (models.py:199-204, callbacks.py:776-792); keep that format.  One process per GPU: only rank 0 writes (the replicas are
        identical up to their per-rank BatchNorm running statistics, and DataParallel keeps device 0's as well)."""
        dp = getattr(self, 'dp', None)
        if dp is not None and dp.rank != 0:
            return
        self.model.eval()
        sd = {'module.' + k: v.detach().cpu() for k, v in self.model.state_dict().items()}
        torch.save(sd, filepath)
        self.model.train()


class SegmentationModel(Model):
    def __init__(self, architecture_config, training_config, callbacks_config):
        super().__init__(architecture_config, training_config, callbacks_config)
        self.activation_func = self.architecture_config['model_params']['activation']
        self.set_model()
        self.set_loss()
        self.weight_regularization = weight_regularization
        self.optimizer = FusedAdam(self.weight_regularization(self.model, **architecture_config['regularizer_params']),
                                   model=self.model, **architecture_config['optimizer_params'])
        self.callbacks = callbacks_network(self.callbacks_config)
        self.dp = parallel.DataParallel.from_env()
        # opt-in: one-GPU training replays the whole step as a hipGraph (training_config['step_graph'] / SALT_STEP_GRAPH=1);
        # the eager two-stream step is faster
        self.step_graph = bool(int(os.environ.get('SALT_STEP_GRAPH', '1' if (training_config or {}).get('step_graph', False) else '0')))

    # ------------------------------------------------------------------ reference surface
    def set_model(self):
        mp = self.architecture_config['model_params']
        config = ARCHITECTURES[mp['architecture']]
        self.model = config['model'](num_classes=mp['out_channels'], **config['model_config'])
        if 'compute_dtype' in mp:
            self.model.set_compute_dtype(mp['compute_dtype'])
        if 'align_corners' in mp:            # True: bilinear up-sampling as torch 0.3.1 (the reference's pinned version) evaluated it
            self.model.set_align_corners(mp['align_corners'])
        self._initialize_model_weights = lambda: None

    def set_loss(self):
        if self.activation_func == 'softmax':
            raise NotImplementedError('No softmax loss defined')
        elif self.activation_func == 'sigmoid':
            name = self.architecture_config['model_params'].get('loss', 'lovasz')
            loss_function = {'lovasz': lovasz_loss, 'bce_dice': mixed_dice_bce_loss}[name]
        else:
            raise Exception('Only softmax and sigmoid activations are allowed')
        self.loss_function = [('mask', loss_function, 1.0)]

    def fit(self, datagen, validation_datagen=None, meta_valid=None):
        self._initialize_model_weights()
        self._to_device()
        self.model.train()
        self.dp.broadcast_parameters(self.model)
        self.callbacks.set_params(self, validation_datagen=validation_datagen, meta_valid=meta_valid)
        self.callbacks.on_train_begin()
        batch_gen, steps = datagen
        for epoch_id in range(self.training_config['epochs']):
            self.callbacks.on_epoch_begin()
            for batch_id, data in enumerate(batch_gen):
                self.callbacks.on_batch_begin()
                metrics = self._fit_loop(data)
                self.callbacks.on_batch_end(metrics=metrics)
                if batch_id == steps:
                    break
            self.callbacks.on_epoch_end()
            if self.dp.any_rank(self.callbacks.training_break()):       # every rank leaves the loop in the same epoch
                break
        self.callbacks.on_train_end()
        return self

    def _to_device(self):
        if not torch.cuda.is_available():
            raise SaltError('SegmentationModel needs a GPU: the hot path is hand-written HIP with no CPU fallback')
        dev = torch.device('cuda', torch.cuda.current_device())
        if next(self.model.parameters()).device != dev:
            self.model.to(dev)
        return dev

    def _fit_loop(self, data):
        dev = self._to_device()
        X = data[0].to(dev, non_blocking=True)
        targets = [t.to(dev, non_blocking=True) for t in data[1:]]
        self.optimizer.zero_grad()
        if len(self.loss_function) != 1:
            raise NotImplementedError('multi-output losses are off the reference default path')
        (name, loss_function, weight), target = self.loss_function[0], targets[0]
        kind = getattr(loss_function, 'native_kind', None)
        if kind is not None:
            batch_loss = self._fused_step(X, target, kind, weight, adam_in_backward=True)     # (optimizer.step() follows below)
        else:                                   # any torch-differentiable loss: through the autograd bridge
            outputs_batch = self.model(X)
            batch_loss = loss_function(outputs_batch, target) * weight
            batch_loss.backward()
            self.dp.allreduce_gradients(self.model.engine(), self.optimizer)   # SUM over ranks; Adam's grad_scale carries 1/world
        if getattr(self, '_graph_stepped', False):
            self._graph_stepped = False                  # Adam ran inside the captured step
        else:
            self.optimizer.step()
        return {'sum': batch_loss}

    def _fused_step(self, X, target, kind, weight, adam_in_backward=False):
        eng = self.model.engine(X.device)
        # 'first step of a shape ran eagerly' is remembered ON the engine (a rebuilt engine starts empty - no recycled id() can skip it)
        if self.step_graph and not self.dp._active() and (tuple(X.shape), kind) in eng.eager_done:
            # the whole step (pack, forward, loss, backward, Adam; both streams) replayed as ONE hipGraph launch (the first step of
            # a shape runs eagerly: lazy one-time work - kernel attributes, workspace allocation - must not fall into the capture)
            net = eng.net(tuple(X.shape), True)
            net.x.copy_(X)
            net.target.copy_(target[:, :net.logits.shape[1]])
            self.optimizer.grad_scale = 1.0
            eng.run_step_graph(net, kind, weight, self.optimizer)
            self._graph_stepped = True
            return net.loss[0].clone()
        # round 6: no torch copy / fill / clone kernel inside the step.  A resident fp32 batch / target is READ IN PLACE (the programs'
        # pointer fields are re-pointed for the duration of the enqueue calls, CompiledNet.bind), the loss scalar is written straight
        # into a fresh 0-dim tensor that is returned as is (read lazily by whoever wants the number: callbacks.TrainingMonitor).
        net = eng.net(tuple(X.shape), True)
        K = net.logits.shape[1]
        loss_prog = net.loss_program(kind, weight)
        zero_copy = os.environ.get('SALT_STEP_ZERO_COPY', '1') != '0'
        bx = zero_copy and net.bindable(X, net.x)
        bt = zero_copy and net.bindable(target, net.target)
        loss_t = torch.empty((1,), dtype=torch.float32, device=X.device) if zero_copy else None
        try:
            net.bind(x=X if bx else None, target=target if bt else None, loss=loss_t)
            eng.forward(X if bx else X.contiguous().float(), True, bound=bx)
            if not bt:
                net.target.copy_(target[:, :K])
            loss_prog.run()
            bwd = None
            if adam_in_backward and not self.dp._active() and getattr(self.optimizer, 'model', None) is self.model and hasattr(self.optimizer, 'begin_step'):
                # one rank: the optimizer updates each parameter range as soon as its gradients are final, beside the rest of backward
                self.optimizer.grad_scale = 1.0
                bwd = self.optimizer.begin_step(net)
            if bwd is not None:
                bwd.run(side=eng.side_stream)
            else:
                self.dp.backward(eng, net, self.optimizer)
        finally:
            net.bind()                           # back to the static buffers (tools / tests that run the programs on their own)
        eng.eager_done.add((tuple(X.shape), kind))
        return loss_t[0] if zero_copy else net.loss[0].clone()

    def transform(self, datagen, validation_datagen=None, *args, **kwargs):
        outputs = self._transform(datagen, validation_datagen)
        for name, prediction in outputs.items():
            if self.activation_func == 'softmax':
                outputs[name] = [softmax(single_prediction, axis=0) for single_prediction in prediction]
            elif self.activation_func == 'sigmoid':
                outputs[name] = [sigmoid(np.squeeze(mask)) for mask in prediction]
            else:
                raise Exception('Only softmax and sigmoid activations are allowed')
        return outputs

    def _transform(self, datagen, validation_datagen=None, **kwargs):
        dev = self._to_device()
        self.model.eval()
        batch_gen, steps = datagen
        outputs = {}
        with torch.no_grad():
            for batch_id, data in enumerate(batch_gen):
                X = data[0] if isinstance(data, (list, tuple)) else data
                outputs_batch = self.model(X.to(dev))
                outputs.setdefault(self.output_names[0], []).append(outputs_batch.cpu().numpy())
                if batch_id == steps:
                    break
        self.model.train()
        return {'{}_prediction'.format(name): get_list_of_image_predictions(outs) for name, outs in outputs.items()}

    def load(self, filepath):
        """Reference checkpoints ('module.'-prefixed keys, models.py:196-208).  Which bilinear semantics the loaded weights get is the
        network's ``align_corners`` (architecture_config['model_params']['align_corners'], default False = torch >= 0.4, the oracle's):
        a checkpoint trained in the reference's own environment (torch 0.3.1) expects True - nothing in the file says which, so the
        caller states it."""
        self.model.eval()
        sd = torch.load(filepath, map_location='cpu')
        sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
        self.model.load_state_dict(sd)
        if torch.cuda.is_available():
            self._to_device()
        return self
