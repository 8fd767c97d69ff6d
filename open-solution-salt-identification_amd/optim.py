"""Fused Adam with L2-in-gradient weight decay over the engine's flat parameter buffer.

Mirrors ``optim.Adam(weight_regularization(model, regularize, weight_decay_conv2d), lr=...)`` of the reference
(models.py:74-75,289-297): one parameter group over every parameter that receives a gradient (BN affine and
biases included), classic Adam (not AdamW), torch defaults beta=(0.9,0.999), eps=1e-8.  The surface the
reference's callbacks touch is kept: ``param_groups`` (lr is read AND written by the schedulers,
callbacks.py:262-275), ``state_dict()``, ``zero_grad()``, ``step()``."""
import weakref

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
        self._bwd_progs = weakref.WeakKeyDictionary()        # per compiled net: (key, (backward + early updates, tail) | None)
        self._chunk_tables = []
        self._pending_tail = None

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
        self._rest_ranges = list(rest)
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

    # -- the update of a parameter range as soon as its gradients are final (round 6)
    def _range_ops(self, prog, ranges, stream):
        """Append to ``prog`` the Adam update (+ forward packs, bf16 engines) of the flat-buffer ``ranges`` [(lo, hi)]; False when a
        range cuts a pack job or is not float4-aligned (the caller then keeps the one-launch step)."""
        import numpy as np
        eng = self._eng
        fused = self._adam_pack_program()
        for lo, hi in ranges:
            if lo % 4 or (hi % 4 and hi != eng.n_live):
                return False
        if fused is None:
            for lo, hi in ranges:
                prog.add('adam', stream=stream, param=eng.flat.data_ptr() + 4 * lo, grad=eng.grads.data_ptr() + 4 * lo, exp_avg=self.exp_avg.data_ptr() + 4 * lo,
                         exp_avg_sq=self.exp_avg_sq.data_ptr() + 4 * lo, n=hi - lo, hyper=self.hyper.data_ptr())
            return True
        import ctypes
        from ._abi import lib
        base = eng.flat.data_ptr()
        jobs, rest = [], []
        for j in eng._adam_jobs:
            first, cnt = (j.w - base) // 4, j.D0 * j.D1 * j.KH * j.KW
            inside = [lo <= first and first + cnt <= hi for lo, hi in ranges]
            if any(inside):
                jobs.append(j)
            elif any(first < hi and lo < first + cnt for lo, hi in ranges):
                return False
        for first, cnt in self._rest_ranges:
            for lo, hi in ranges:
                a, b = max(first, lo), min(first + cnt, hi)
                if a < b:
                    if a % 4 or ((b - a) % 4 and b != eng.n_live):
                        return False
                    rest.append((a, b - a))
        blocks = [lib.salt_pack_job_blocks(ctypes.byref(j)) for j in jobs]
        pref = np.concatenate([[0], np.cumsum(blocks)]).astype(np.int32)
        rblocks = [(c + 1023) // 1024 for _, c in rest]
        rpref = np.concatenate([[0], np.cumsum(rblocks)]).astype(np.int32) if rest else np.zeros(1, np.int32)
        dev = eng.device
        t = [torch.frombuffer(bytearray(b''.join(bytes(j) for j in jobs) or b'\0'), dtype=torch.uint8).to(dev), torch.from_numpy(pref).to(dev),
             torch.tensor(rest if rest else [[0, 0]], dtype=torch.int64).to(dev), torch.from_numpy(rpref).to(dev)]
        self._chunk_tables.append(t)
        prog.add('adam_pack', stream=stream, param=eng.flat.data_ptr(), grad=eng.grads.data_ptr(), exp_avg=self.exp_avg.data_ptr(),
                 exp_avg_sq=self.exp_avg_sq.data_ptr(), n=eng.n_live, hyper=self.hyper.data_ptr(), jobs=t[0].data_ptr(), job_block0=t[1].data_ptr(),
                 njobs=len(jobs), pack_blocks=int(pref[-1]), rest=t[2].data_ptr(), rest_block0=t[3].data_ptr(), nrest=len(rest), rest_blocks=int(rpref[-1]))
        return True

    def backward_program(self, net):
        """-> (program, tail) or None.  ``program`` = the backward program of ``net`` with this optimizer's update of every parameter
        range whose gradients are final before the program ends inserted at that position as auxiliary-stream entries (stream tag 5,
        runtime.hip: behind everything both queues were given so far, beside what follows); ``tail`` = the update of the remaining
        ranges, which :meth:`step` runs after the join.  The launches of the one-kernel step cut into ranges: every element takes the
        same arithmetic (`adam4`), so parameters, moments and packs are bit-identical to it.  Why: Adam is a 0.17 ms streaming pass
        (R34 hypercolumn) that used to run alone at the end of the step, while the backward's two queues leave most of the HBM
        bandwidth unused; the decoder's and layer4's parameters - 3 / 4 of the network - are final 40 - 70 % into backward."""
        import os
        from .parallel import plan_buckets
        self._bind()
        self._adam_pack_program()
        key = (self._eng._pack_generation, len(net.bwd), id(self._eng))
        hit = self._bwd_progs.get(net)
        if hit is not None and hit[0] == key:
            return hit[1]
        self._bwd_progs[net] = (key, None)
        if not getattr(net.g, 'grad_ready', None):
            return None
        n_ops = len(net.bwd)
        chunk = int(float(os.environ.get('SALT_ADAM_CHUNK_MB', '24')) * (1 << 20))
        plan = plan_buckets(net.g.grad_ready, self._eng.n_live, bucket_bytes=chunk, tail_bytes=0)
        margin = int(os.environ.get('SALT_ADAM_TAIL_OPS', '4'))          # a range final this close to the end gains nothing from its own launch
        early = [(lo, hi, min(max(pos, 0), n_ops)) for lo, hi, pos in plan if pos < n_ops - margin and hi > lo]
        late = [(lo, hi) for lo, hi, pos in plan if not pos < n_ops - margin and hi > lo]
        if not early:
            return None
        if net.bwd._entries is None:
            net.bwd.finalize()
        self._chunk_tables = getattr(self, '_chunk_tables', [])
        ins = Program('adam_ranges')
        at = []
        ins.add('adam_tick', stream=5, hyper=self.hyper.data_ptr(), step=self.step_t.data_ptr())
        at.append(early[0][2])
        for lo, hi, pos in early:
            n0 = len(ins.ops)
            if not self._range_ops(ins, [(lo, hi)], 5):
                return None
            at.extend([pos] * (len(ins.ops) - n0))
        tail = Program('adam_tail')
        if late and not self._range_ops(tail, late, 0):
            return None
        tail.finalize()
        prog = Program('bwd+adam')
        k = 0
        for i in range(n_ops + 1):
            while k < len(at) and at[k] <= i:
                prog.ops.append(ins.ops[k]); prog.streams.append(ins.streams[k]); k += 1
            if i < n_ops:
                prog.ops.append(net.bwd.ops[i]); prog.streams.append(net.bwd.streams[i])
        prog._pre_run, prog._post_run = net.bwd._pre_run, net.bwd._post_run
        prog.finalize()
        self._bwd_progs[net] = (key, (prog, tail))
        return prog, tail

    def begin_step(self, net):
        """Called by the fused step BEFORE backward: -> the program to run instead of ``net.bwd`` (None: run ``net.bwd``, step() does
        everything).  After it ran, :meth:`step` only runs the tail."""
        import os
        if self.model is None or os.environ.get('SALT_ADAM_IN_BWD', '0') == '0':
            return None
        self._bind()
        self._sync_hyper()
        progs = self.backward_program(net)
        if progs is None:
            return None
        self._pending_tail = progs[1]
        return progs[0]

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
        fused = self._adam_pack_program()
        if self._pending_tail is not None:                 # begin_step: the ranges that were final early took their update inside backward
            tail, self._pending_tail = self._pending_tail, None
            if len(tail):
                tail.run()
        else:
            self._sync_hyper()
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
