"""Trainer conveniences of the reference (SURVEY.md §8 "next" row f-4), same class names / constructor arguments / behaviour as
``common_blocks/callbacks.py``, minus everything that talks to Neptune:

    Callback / CallbackList                callbacks.py:30-110
    TrainingMonitor                        callbacks.py:113-160   (running mean of the batch loss, logged per epoch)
    ExperimentTiming                       callbacks.py:271-330
    ValidationMonitor                      callbacks.py:462-568   validation loss + the IoU / IOUT threshold sweep; the sweep runs on
                                                                  the GPU (inference.iou_counts) instead of 21 CPU pipelines
    ReduceLROnPlateauScheduler             callbacks.py:204-241   torch ReduceLROnPlateau(mode, factor, patience, min_lr) on the fused
                                                                  Adam's param_groups (the kernel re-reads lr every step)
    ModelCheckpoint                        callbacks.py:758-792   best-metric ``state_dict`` with the reference's 'module.' key prefix
    EarlyStopping                          callbacks.py:795-829

The validation metric of an epoch is computed once and cached in ``transformer.validation_loss[epoch_id]`` exactly like the
reference (callbacks.py:522-527), so the scheduler / checkpoint / early-stopping callbacks do not re-run the validation set.
Metric values are plain 1-element tensors ({'sum', 'iou', 'iout'}).
"""
import logging
import os
import time

import numpy as np
import torch
from torch.optim.lr_scheduler import ReduceLROnPlateau

from . import inference

logger = logging.getLogger('salt_amd')


def _scalar(v):
    return float(v.detach().reshape(-1)[0]) if torch.is_tensor(v) else float(v)


class Callback:
    """The reference's callback PROTOCOL (common_blocks/callbacks.py:28-80): the attribute list `set_params` fills and the eight hooks are
    the interface user callbacks subclass and `SegmentationModel.fit` drives, so they are kept name for name and in the same order -
    an interface skeleton, restated on purpose.  Everything with behaviour (the monitors, checkpointing, schedulers below) is rewritten
    for the device-side metric sweep and the one-process-per-GPU layout."""

    def __init__(self):
        self.epoch_id = None
        self.batch_id = None
        self.model = None
        self.optimizer = None
        self.loss_function = None
        self.output_names = None
        self.validation_datagen = None
        self.transformer = None

    def set_params(self, transformer, validation_datagen=None, *args, **kwargs):
        self.transformer = transformer
        self.model = transformer.model
        self.optimizer = transformer.optimizer
        self.loss_function = transformer.loss_function
        self.output_names = transformer.output_names
        self.validation_datagen = validation_datagen

    def on_train_begin(self, *args, **kwargs):
        self.epoch_id = 0
        self.batch_id = 0

    def on_train_end(self, *args, **kwargs):
        pass

    def on_epoch_begin(self, *args, **kwargs):
        pass

    def on_epoch_end(self, *args, **kwargs):
        self.epoch_id += 1

    def training_break(self, *args, **kwargs):
        return False

    def on_batch_begin(self, *args, **kwargs):
        pass

    def on_batch_end(self, *args, **kwargs):
        self.batch_id += 1

    def get_validation_loss(self):
        """Shared, cached per epoch (callbacks.py:60-66): the first callback that asks triggers the validation pass."""
        if self.transformer is None:
            raise RuntimeError('callback is not attached to a transformer (set_params was not called)')
        if self.epoch_id not in self.transformer.validation_loss:
            val = self.transformer.score_validation(self.validation_datagen)
            dp = getattr(self.transformer, 'dp', None)
            if dp is not None and dp.world > 1:
                # one process per GPU: every rank's scheduler / checkpoint / early-stopping callbacks must see the SAME score
                # (rank 0's, the replica whose weights are persisted) or the replicas drift apart / dead-lock
                keys = sorted(val)
                nums = dp.broadcast_scalars([_scalar(val[k]) for k in keys] + [float(getattr(self.transformer, 'best_threshold', 0.5))])
                val = {k: torch.tensor([v], dtype=torch.float32) for k, v in zip(keys, nums)}
                self.transformer.best_threshold = nums[-1]
            self.transformer.validation_loss[self.epoch_id] = val
        return self.transformer.validation_loss[self.epoch_id]


class CallbackList:
    def __init__(self, callbacks=None):
        self.callbacks = list(callbacks or [])

    def __len__(self):
        return len(self.callbacks)

    def set_params(self, *args, **kwargs):
        for c in self.callbacks:
            c.set_params(*args, **kwargs)

    def on_train_begin(self, *args, **kwargs):
        for c in self.callbacks:
            c.on_train_begin(*args, **kwargs)

    def on_train_end(self, *args, **kwargs):
        for c in self.callbacks:
            c.on_train_end(*args, **kwargs)

    def on_epoch_begin(self, *args, **kwargs):
        for c in self.callbacks:
            c.on_epoch_begin(*args, **kwargs)

    def on_epoch_end(self, *args, **kwargs):
        for c in self.callbacks:
            c.on_epoch_end(*args, **kwargs)

    def training_break(self, *args, **kwargs):
        return any([c.training_break(*args, **kwargs) for c in self.callbacks])

    def on_batch_begin(self, *args, **kwargs):
        for c in self.callbacks:
            c.on_batch_begin(*args, **kwargs)

    def on_batch_end(self, *args, **kwargs):
        for c in self.callbacks:
            c.on_batch_end(*args, **kwargs)


def _every(n):
    return False if n == 0 else n


class TrainingMonitor(Callback):
    def __init__(self, epoch_every=None, batch_every=None):
        super().__init__()
        self.epoch_every, self.batch_every = _every(epoch_every), _every(batch_every)
        self.epoch_loss_sum, self.epoch_batches = {}, 0
        self.history = []

    def on_train_begin(self, *args, **kwargs):
        super().on_train_begin()
        self.epoch_loss_sum, self.epoch_batches = {}, 0

    def on_batch_end(self, metrics, *args, **kwargs):
        # the loss stays on the device; it is only read back when something is logged (no per-step synchronisation)
        # (round 6: kept as a list and summed ONCE per epoch - a `loss + sum` per step was one torch kernel on the step's queue)
        for name, loss in metrics.items():
            self.epoch_loss_sum.setdefault(name, []).append(loss.detach() if torch.is_tensor(loss) else loss)
        self.epoch_batches += 1
        if self.batch_every and (self.batch_id % self.batch_every) == 0:
            for name, loss in metrics.items():
                logger.info('epoch %d batch %d %s: %.5f', self.epoch_id, self.batch_id, name, _scalar(loss))
        self.batch_id += 1

    def on_epoch_end(self, *args, **kwargs):
        means = {k: _scalar(torch.stack([t.reshape(()) for t in v]).sum() if (v and torch.is_tensor(v[0])) else sum(v)) / max(self.epoch_batches, 1)
                 for k, v in self.epoch_loss_sum.items()}
        self.history.append(means)
        if self.epoch_every and (self.epoch_id % self.epoch_every) == 0:
            for name, v in means.items():
                logger.info('epoch %d %s:     %.5f', self.epoch_id, name, v)
        self.epoch_loss_sum, self.epoch_batches = {}, 0
        self.epoch_id += 1


class ExperimentTiming(Callback):
    def __init__(self, epoch_every=None, batch_every=None):
        super().__init__()
        self.epoch_every, self.batch_every = _every(epoch_every), _every(batch_every)
        self.epoch_start = None
        self.epoch_seconds = []

    def on_train_begin(self, *args, **kwargs):
        super().on_train_begin()
        logger.info('starting training...')

    def on_train_end(self, *args, **kwargs):
        logger.info('training finished')

    def on_epoch_begin(self, *args, **kwargs):
        if self.epoch_id > 0 and self.epoch_seconds and self.epoch_every and (self.epoch_id % self.epoch_every) == 0:
            logger.info('epoch %d time %.1f s', self.epoch_id - 1, self.epoch_seconds[-1])
        self.epoch_start = time.time()

    def on_epoch_end(self, *args, **kwargs):
        if self.epoch_start is not None:
            self.epoch_seconds.append(time.time() - self.epoch_start)
        self.epoch_id += 1


class ValidationMonitor(Callback):
    """callbacks.py:462-568.  ``data_dir`` / ``loader_mode`` / ``use_depth`` are accepted for signature compatibility; the
    ground-truth masks come from the validation batches themselves (target channel 1), not from files on disk."""

    def __init__(self, data_dir=None, loader_mode=None, epoch_every=None, batch_every=None, use_depth=False):
        super().__init__()
        self.epoch_every, self.batch_every = _every(epoch_every), _every(batch_every)
        self.data_dir, self.loader_mode, self.use_depth = data_dir, loader_mode, use_depth
        self.meta_valid = None

    def set_params(self, transformer, validation_datagen=None, meta_valid=None, *args, **kwargs):
        super().set_params(transformer, validation_datagen)
        self.meta_valid = meta_valid

    def on_epoch_end(self, *args, **kwargs):
        if self.epoch_every and (self.epoch_id % self.epoch_every) == 0 and self.validation_datagen is not None:
            val = self.get_validation_loss()
            for name, v in val.items():
                logger.info('epoch %d validation %s:     %.5f', self.epoch_id, name, _scalar(v))
        self.epoch_id += 1


class ReduceLROnPlateauScheduler(Callback):
    def __init__(self, metric_name, minimize, reduce_factor, reduce_patience, min_lr):
        super().__init__()
        self.metric_name, self.minimize = metric_name, minimize
        self.reduce_factor, self.reduce_patience, self.min_lr = reduce_factor, reduce_patience, min_lr
        self.lr_scheduler = None

    def set_params(self, transformer, validation_datagen=None, *args, **kwargs):
        super().set_params(transformer, validation_datagen)
        self.lr_scheduler = ReduceLROnPlateau(optimizer=self.optimizer, mode='min' if self.minimize else 'max',
                                              factor=self.reduce_factor, patience=self.reduce_patience, min_lr=self.min_lr)

    def on_epoch_end(self, *args, **kwargs):
        metric = _scalar(self.get_validation_loss()[self.metric_name])
        self.lr_scheduler.step(metric)
        logger.info('epoch %d current lr: %g', self.epoch_id + 1, self.optimizer.param_groups[0]['lr'])
        self.epoch_id += 1


class ModelCheckpoint(Callback):
    def __init__(self, filepath, metric_name='sum', epoch_every=1, minimize=True):
        super().__init__()
        self.filepath, self.metric_name, self.minimize = filepath, metric_name, minimize
        self.epoch_every = _every(epoch_every)
        self.best_score = None

    def on_train_begin(self, *args, **kwargs):
        super().on_train_begin()
        if os.path.dirname(self.filepath):
            os.makedirs(os.path.dirname(self.filepath), exist_ok=True)

    def on_epoch_end(self, *args, **kwargs):
        if self.epoch_every and (self.epoch_id % self.epoch_every) == 0:
            score = _scalar(self.get_validation_loss()[self.metric_name])
            if self.best_score is None:
                self.best_score = score
            if (self.minimize and score < self.best_score) or (not self.minimize and score > self.best_score) or self.epoch_id == 0:
                self.best_score = score
                self.transformer.persist(self.filepath)                # 'module.'-prefixed state_dict (models.py:199-204)
                logger.info('epoch %d model saved to %s', self.epoch_id, self.filepath)
        self.epoch_id += 1


class EarlyStopping(Callback):
    def __init__(self, metric_name='sum', patience=1000, minimize=True):
        super().__init__()
        self.metric_name, self.patience, self.minimize = metric_name, patience, minimize
        self.best_score = None
        self.epoch_since_best = 0
        self._training_break = False

    def training_break(self, *args, **kwargs):
        return self._training_break

    def on_epoch_end(self, *args, **kwargs):
        score = _scalar(self.get_validation_loss()[self.metric_name])
        if not self.best_score:                                         # sic (callbacks.py:815): a best score of 0.0 is re-seeded
            self.best_score = score
        if (self.minimize and score < self.best_score) or (not self.minimize and score > self.best_score):
            self.best_score = score
            self.epoch_since_best = 0
        else:
            self.epoch_since_best += 1
        if self.epoch_since_best > self.patience:
            self._training_break = True
        self.epoch_id += 1


def score_validation(transformer, validation_datagen, target_size=(101, 101)):
    """One pass over the validation generator (callbacks.py:529-568 + 503-527): mean loss, and IoU / IOUT at the threshold the
    reference's sweep selects.  Everything but a few integers per image stays on the GPU."""
    model = transformer.model
    was_training = model.training
    model.eval()
    batch_gen, steps = validation_datagen
    dev = transformer._to_device()
    (name, loss_function, weight) = transformer.loss_function[0]
    thresholds = np.linspace(0.5, 0.3, 21)
    losses, counts = [], []
    with torch.no_grad():
        for batch_id, data in enumerate(batch_gen):
            X, target = data[0].to(dev), data[1].to(dev)
            logits = model(X)
            losses.append((loss_function(logits, target) * weight).detach().reshape(1))
            prob = torch.sigmoid(logits.float())
            H, W = prob.shape[2:]
            hw = (min(target_size[0], H), min(target_size[1], W))
            top, left = inference.crop_window(H, W, hw)
            gt = (target[:, 1, top:top + hw[0], left:left + hw[1]] > 0.5).to(torch.uint8)
            counts.append(inference.iou_counts(prob, gt, thresholds, cls=1))
            if batch_id == steps:
                break
    if was_training:
        model.train()
    # the reference divides the summed batch losses by `steps` although steps + 1 batches are consumed (callbacks.py:556-557);
    # we report the plain mean over the batches actually seen
    mean_loss = torch.cat(losses).mean().reshape(1).cpu()
    t_best, iou, iout = inference.select_threshold(counts, thresholds)
    transformer.best_threshold = t_best
    return {'sum': mean_loss, 'iou': torch.tensor([iou], dtype=torch.float32), 'iout': torch.tensor([iout], dtype=torch.float32)}


def callbacks_network(callbacks_config):
    """models.py:300-312: build the callback list from the same config dict ('neptune_monitor' is ignored)."""
    cfg = callbacks_config or {}
    cbs = list(cfg.get('callbacks', []))
    if 'experiment_timing' in cfg:
        cbs.append(ExperimentTiming(**cfg['experiment_timing']))
    if 'training_monitor' in cfg:
        cbs.append(TrainingMonitor(**cfg['training_monitor']))
    if 'validation_monitor' in cfg:
        cbs.append(ValidationMonitor(**cfg['validation_monitor']))
    if 'model_checkpoint' in cfg:
        cbs.append(ModelCheckpoint(**cfg['model_checkpoint']))
    if 'reduce_lr_on_plateau_scheduler' in cfg:
        cbs.append(ReduceLROnPlateauScheduler(**cfg['reduce_lr_on_plateau_scheduler']))
    if 'early_stopping' in cfg:
        cbs.append(EarlyStopping(**cfg['early_stopping']))
    return CallbackList(cbs)
