"""Fused Adam with L2-in-gradient weight decay over the engine's flat parameter buffer.

Mirrors ``optim.Adam(weight_regularization(model, regularize, weight_decay_conv2d), lr=...)`` of the reference
(models.py:74-75,289-297): one parameter group over every parameter that receives a gradient (BN affine and
biases included), classic Adam (not AdamW), torch defaults beta=(0.9,0.999), eps=1e-8.  The surface the
reference's callbacks touch is kept: ``param_groups`` (lr is read AND written by the schedulers,
callbacks.py:262-275), ``state_dict()``, ``zero_grad()``, ``step()``."""
import torch

from ._abi import SaltError
from .engine import Program


def weight_regularization(model, regularize, weight_decay_conv2d):
    """models.py:289-297 — returns the param-group list in the reference's format."""
    params = [p for p in model.parameters() if p.requires_grad]
    if regularize:
        return [{'params': params, 'weight_decay': weight_decay_conv2d}]
    return [params]


class FusedAdam(torch.optim.Optimizer):
    """A real ``torch.optim.Optimizer`` (so torch's LR schedulers - ReduceLROnPlateau, ExponentialLR: callbacks.py:170-241 - accept
    it) whose ``step`` is one fused kernel over the flat buffers; ``param_groups[0]['lr']`` is re-read every step."""

    def __init__(self, param_groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, model=None):
        if isinstance(param_groups, (list, tuple)) and param_groups and isinstance(param_groups[0], dict):
            g = dict(param_groups[0])
        else:
            g = {'params': list(param_groups[0] if param_groups and isinstance(param_groups[0], (list, tuple)) else param_groups)}
        super().__init__([g], dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise SaltError('FusedAdam: one parameter group (models.py:289-297)')
        self.param_groups[0].setdefault('initial_lr', self.param_groups[0]['lr'])
        self.model = model
        self._eng = None
        self._host_hyper = None
        self.steps = 0
        self.grad_scale = 1.0

    # -- lazily bound to the engine's flat buffers
    def _bind(self):
        eng = self.model.engine()
        if self._eng is eng:
            return
        old = (self.exp_avg, self.exp_avg_sq) if self._eng is not None else None
        self._eng = eng
        dev = eng.device
        if old is not None and old[0].numel() == eng.flat.numel():
            # the model's engine was rebuilt (.to() / .float() / set_compute_dtype drop it): the flat layout is the same, so the Adam
            # moments and the bias-correction counter carry over instead of being silently reset
            self.exp_avg, self.exp_avg_sq = old[0].to(dev).clone(), old[1].to(dev).clone()
        else:
            if old is not None:
                self.steps = 0                       # different parameter set: moments AND bias correction restart together
            self.exp_avg = torch.zeros_like(eng.flat)
            self.exp_avg_sq = torch.zeros_like(eng.flat)
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        self.step_t = torch.full((1,), self.steps, dtype=torch.int64, device=dev)
        self._host_hyper = None
        self.prog = Program('adam')
        self.prog.add('adam_tick', hyper=self.hyper.data_ptr(), step=self.step_t.data_ptr())
        self.prog.add('adam', param=eng.flat.data_ptr(), grad=eng.grads.data_ptr(), exp_avg=self.exp_avg.data_ptr(),
                      exp_avg_sq=self.exp_avg_sq.data_ptr(), n=eng.n_live, hyper=self.hyper.data_ptr())
        self.prog.finalize()
        self._pack_prog, self._pack_gen = None, None

    def _adam_pack_program(self):
        """Adam + the bf16 forward weight packs in ONE launch (salt_adam_pack): the thread that updates 8 input channels x all taps of an
        output channel also stores their packed pieces, so the pack launch that opened every step on the critical queue (405 MB of
        traffic re-deriving what Adam had just written) shrinks to the few layers without a vector pack.  None when there is nothing to
        fuse (fp32 engines, no compiled network yet, SALT_ADAM_PACK=0).  Rebuilt whenever the engine's pack tables are."""
        import os
        import numpy as np
        eng = self._eng
        if eng.dtype != 'bf16' or os.environ.get('SALT_ADAM_PACK', '1') == '0':
            return None
        if getattr(eng, '_pack_batched_n', -1) != len(eng._pack_ops):
            eng._build_pack_batch()
            eng._packed_version = eng._packed_bwd_version = -1
        if self._pack_gen == eng._pack_generation:
            return self._pack_prog
        self._pack_gen, self._pack_prog = eng._pack_generation, None
        jobs = eng._adam_jobs
        if not jobs:
            return None
        import ctypes
        from ._abi import lib
        base = eng.flat.data_ptr()
        blocks = [lib.salt_pack_job_blocks(ctypes.byref(j)) for j in jobs]
        pref = np.concatenate([[0], np.cumsum(blocks)]).astype(np.int32)
        covered = sorted(((j.w - base) // 4, j.D0 * j.D1 * j.KH * j.KW) for j in jobs)
        rest, pos = [], 0
        for first, cnt in covered:
            if first % 4 or cnt % 4 or first < pos:
                return None                              # (cannot happen for whole 32-channel chunks; plain Adam + full packs then)
            if first > pos:
                rest.append((pos, first - pos))
            pos = first + cnt
        if pos < eng.n_live:
            rest.append((pos, eng.n_live - pos))
        rblocks = [(c + 1023) // 1024 for _, c in rest]
        rpref = np.concatenate([[0], np.cumsum(rblocks)]).astype(np.int32) if rest else np.zeros(1, np.int32)
        dev = eng.device
        self._pack_tables = [torch.frombuffer(bytearray(b''.join(bytes(j) for j in jobs)), dtype=torch.uint8).to(dev), torch.from_numpy(pref).to(dev),
                             torch.tensor(rest if rest else [[0, 0]], dtype=torch.int64).to(dev), torch.from_numpy(rpref).to(dev)]
        t = self._pack_tables
        prog = Program('adam_pack')
        prog.add('adam_tick', hyper=self.hyper.data_ptr(), step=self.step_t.data_ptr())
        prog.add('adam_pack', param=eng.flat.data_ptr(), grad=eng.grads.data_ptr(), exp_avg=self.exp_avg.data_ptr(), exp_avg_sq=self.exp_avg_sq.data_ptr(),
                 n=eng.n_live, hyper=self.hyper.data_ptr(), jobs=t[0].data_ptr(), job_block0=t[1].data_ptr(), njobs=len(jobs), pack_blocks=int(pref[-1]),
                 rest=t[2].data_ptr(), rest_block0=t[3].data_ptr(), nrest=len(rest), rest_blocks=int(rpref[-1]))
        prog.finalize()
        self._pack_prog = prog
        return prog

    def _sync_hyper(self):
        g = self.param_groups[0]
        cur = (float(g['lr']), float(g['betas'][0]), float(g['betas'][1]), float(g['eps']), float(g['weight_decay']), float(self.grad_scale))
        if cur != self._host_hyper:
            self.hyper[:5].copy_(torch.tensor(cur[:5], dtype=torch.float32))
            self.hyper[7:8].fill_(cur[5])
            self._host_hyper = cur

    def zero_grad(self, set_to_none=False):
        pass        # the backward program overwrites every gradient each step

    def step(self, closure=None):
        if self.model is None:
            raise SaltError('FusedAdam needs the HipNetwork it optimises (model=...)')
        self._bind()
        self._sync_hyper()
        fused = self._adam_pack_program()
        (fused or self.prog).run()
        self.steps += 1
        self._eng.touch(weights=True, stats=False)
        if fused is not None:
            self._eng._adam_packed_version = self._eng.wver      # Engine.refresh: only the remaining forward packs are stale

    def state_dict(self):
        g = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        g['params'] = list(range(len(self.param_groups[0]['params'])))
        st = {}
        if self._eng is None and self.model is not None and getattr(self.model, '_engine', None) is not None:
            self._bind()                             # zero moments before the first step, like torch's Adam after initialisation
        if self._eng is not None:
            st = {'step': self.steps, 'exp_avg': self.exp_avg.detach().cpu(), 'exp_avg_sq': self.exp_avg_sq.detach().cpu()}
        return {'state': st, 'param_groups': [g]}

    def load_state_dict(self, sd):
        for k, v in sd['param_groups'][0].items():
            if k != 'params':
                self.param_groups[0][k] = v
        st = sd.get('state', {})
        if st:
            self._bind()
            self.steps = int(st['step'])
            self.step_t.fill_(self.steps)
            self.exp_avg.copy_(st['exp_avg'])
            self.exp_avg_sq.copy_(st['exp_avg_sq'])
