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
        self.prog.run()
        self.steps += 1
        self._eng.touch(weights=True, stats=False)

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
