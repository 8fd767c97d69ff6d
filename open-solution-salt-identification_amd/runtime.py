"""Engine: owns the flat parameter/gradient/optimizer buffers of one model, the packed compute copies of
its convolution weights, the BatchNorm work vectors, and the compiled (static) programs per batch shape.

Design notes (MI355X-first):
  * fp32 master parameters live in ONE flat buffer in the reference's own tensor layouts, so
    ``state_dict()`` is the reference's; gradients live in a second flat buffer of the same layout, so
    the optimizer is one fused kernel over 30 M floats and data-parallel all-reduce works on contiguous
    byte ranges (buckets) without gather/scatter copies.
  * convolution kernels read PACKED copies ([chunk][tap][n][64 B], bf16 or f32; plus the transposed copy
    the data-gradient needs) that a tiny pack program refreshes after every optimizer step.
  * a training step = pack -> forward -> loss -> backward (in bucket segments, all-reduce on a side
    stream) -> fused Adam; each is one call into the native executor.
"""
import ctypes
import os

import torch

from . import _abi
from ._abi import SaltError, lib
from .engine import Graph, Program, DT_CODE, TORCH_DT


def _require_gpu(device):
    if device.type != 'cuda':
        raise SaltError('the HIP path needs a GPU tensor/device (got %s); there is no CPU fallback' % device)


def _ptr_fields(struct, ptr, out):
    """(struct, field) of every pointer field of an argument struct (nested views / arrays included) that holds ``ptr``"""
    for name, ct in struct._fields_:
        v = getattr(struct, name)
        if isinstance(v, ctypes.Structure):
            _ptr_fields(v, ptr, out)
        elif isinstance(v, ctypes.Array):
            for e in v:
                if isinstance(e, ctypes.Structure):
                    _ptr_fields(e, ptr, out)
        elif ct is ctypes.c_void_p and v == ptr:
            out.append((struct, name))
    return out


class CompiledNet:
    """One static instance: input/target/logits buffers + forward/backward programs for fixed (B,C,H,W)."""

    def __init__(self, engine, shape, train, num_classes):
        B, C, H, W = shape
        self.engine, self.shape, self.train = engine, shape, train
        g = Graph(engine, train)
        self.g = g
        self.x = g.alloc((B, C, H, W), torch.float32)
        oshape = engine.module.output_shape(shape) if hasattr(engine.module, 'output_shape') else (B, num_classes, H, W)
        self.logits = g.alloc(tuple(oshape), torch.float32)
        engine.module.emit(g, self.x, self.logits)
        self.dlogits = getattr(g, 'dlogits', None)
        if train:
            self.target = g.alloc(tuple(oshape), torch.float32)
            self.loss = g.alloc((1,), torch.float32)
            self.loss_per_image = g.alloc((B,), torch.float32)
            g.build_backward()
        g.finalize()
        self.fwd, self.bwd = g.fwd, g.bwd
        self._loss_progs = {}
        self._slots = {}               # 'x' | 'target' | 'loss' -> [(struct, field)] that hold the static buffer's address
        self._bound = {}

    # ------------------------------------------------------------------ zero-copy step inputs (round 6)
    def _slot_list(self, which):
        """Argument-struct pointer fields of this instance's programs that read the static input / target buffer or write the loss
        scalar.  The executor reads an argument struct when it ENQUEUES the launch, so re-pointing such a field between two runs of a
        program is a host-side store and nothing else: the fused step (models.SegmentationModel._fused_step) points them at the
        caller's resident batch / target for the duration of its enqueue calls instead of copying 6 MB + 4 MB into the static
        buffers first, and puts them back (hipGraph captures bake pointers in - they keep the copies)."""
        buf = {'x': self.x, 'target': getattr(self, 'target', None), 'loss': getattr(self, 'loss', None)}[which]
        progs = [self.fwd, self.bwd] + list(self._loss_progs.values())
        key = (which, tuple(id(p) for p in progs))
        if self._slots.get(which, (None,))[0] != key:
            out = []
            if buf is not None:
                for prog in progs:
                    for _, _, st in prog.ops:
                        _ptr_fields(st, buf.data_ptr(), out)
                        if self._bound.get(which, buf.data_ptr()) != buf.data_ptr():
                            _ptr_fields(st, self._bound[which], out)
            self._slots[which] = (key, out)
        return self._slots[which][1]

    def bind(self, **tensors):
        """bind(x=t, target=t2, loss=t3): point the programs at these device tensors (None / omitted: the static buffer).  The caller
        keeps the tensors alive until the step's launches are enqueued on the stream it later frees / overwrites them on."""
        for which in ('x', 'target', 'loss'):
            t = tensors.get(which)
            buf = {'x': self.x, 'target': getattr(self, 'target', None), 'loss': getattr(self, 'loss', None)}[which]
            if buf is None:
                continue
            ptr = buf.data_ptr() if t is None else t.data_ptr()
            slots = self._slot_list(which)
            if t is not None and (t.dtype != buf.dtype or tuple(t.shape) != tuple(buf.shape) or not t.is_contiguous() or t.device != buf.device or ptr % 16):
                raise SaltError('bind(%s): needs a contiguous %s tensor of shape %s on %s' % (which, buf.dtype, tuple(buf.shape), buf.device))
            for st, field in slots:
                setattr(st, field, ptr)
            self._bound[which] = ptr

    @staticmethod
    def bindable(t, buf):
        return (t.dtype == buf.dtype and tuple(t.shape) == tuple(buf.shape) and t.is_contiguous() and t.device == buf.device
                and t.data_ptr() % 16 == 0)

    def loss_program(self, kind, loss_scale):
        key = (kind, float(loss_scale))
        if key in self._loss_progs:
            return self._loss_progs[key]
        g = self.g
        B, K, H, W = self.logits.shape
        p = Program('loss:' + kind)
        if kind == 'lovasz':
            P = K * H * W
            wk = g.alloc((2, B, P), torch.int32, zero=False)
            wv = g.alloc((2, B, P), torch.int32, zero=False)
            sw = int(lib.salt_lovasz_split_words(P))             # > 0: several workgroups per image (segments of the sort)
            ws = g.alloc((B * sw,), torch.int32, zero=False) if sw else None
            p.add('lovasz_hinge', logits=self.logits.data_ptr(), target=self.target.data_ptr(), B=B, P=P, ws_keys=wk.data_ptr(),
                  ws_vals=wv.data_ptr(), loss_per_image=self.loss_per_image.data_ptr(), loss=self.loss.data_ptr(),
                  dlogits=self.dlogits.data_ptr(), loss_scale=loss_scale, ws_split=ws.data_ptr() if ws is not None else None)
        elif kind == 'bce_dice':
            S = _abi.STRUCTS['salt_bce_dice_args']()
            _abi.fill(S, B=B, C=K, HW=H * W)
            nparts = lib.salt_bce_dice_parts(ctypes.byref(S))
            parts = g.alloc((nparts * 4,), torch.float32)
            sums = g.alloc((3 * K + 1,), torch.float32)
            p.add('bce_dice', logits=self.logits.data_ptr(), target=self.target.data_ptr(), B=B, C=K, HW=H * W, dice_weight=0.2, bce_weight=0.9,
                  partials=parts.data_ptr(), nparts=nparts, sums=sums.data_ptr(), loss=self.loss.data_ptr(), dlogits=self.dlogits.data_ptr(),
                  loss_scale=loss_scale)
        else:
            raise SaltError('unknown native loss %r' % kind)
        p.finalize()
        self._loss_progs[key] = p
        return p


class Engine:
    def __init__(self, module, device, dtype='f32'):
        _require_gpu(device)
        if dtype not in DT_CODE:
            raise SaltError('dtype must be f32 or bf16')
        self.module, self.device, self.dtype = module, device, dtype
        self._flatten()
        self._packed = {}
        self._pack_ops = Program('pack')
        self._packed_bwd_version = -1
        self._pack_is_bwd, self._pack_keys = {}, []      # conv-weight pack jobs in _pack_ops order; True = only backward reads it
        self._bn = {}
        self._fold_ops = Program('bn_fold')
        self.nets = {}
        self.wver = 0                   # bumped whenever master weights change
        self.sver = 0                   # bumped whenever BN running statistics change
        self._packed_version = -1
        self._folded_version = None
        self.world = 1
        pr = os.environ.get('SALT_SIDE_PRIORITY')                # (A/B: HIP stream priority of the weight-gradient queue; lower number = served first)
        self.side_stream = torch.cuda.Stream(device=device, priority=int(pr)) if pr else torch.cuda.Stream(device=device)     # weight-gradient kernels overlap the data-gradient chain
        self.eager_done = set()          # (shape, loss kind) whose first training step already ran eagerly (models.SegmentationModel._fused_step)

    # ------------------------------------------------------------------ flat parameter storage
    def _flatten(self):
        m = self.module
        dead = set(m.dead_parameter_names()) if hasattr(m, 'dead_parameter_names') else set()
        seen = {}
        live = []
        for name, p in m.named_parameters():
            if id(p) in seen:
                continue
            seen[id(p)] = name
            if p.requires_grad and name not in dead:
                live.append((name, p))
        total = sum(((p.numel() + 3) // 4) * 4 for _, p in live)
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.grads = torch.zeros(total, dtype=torch.float32, device=self.device)
        self._off = {}
        self.live_names = []
        off = 0
        for name, p in live:
            n = p.numel()
            view = self.flat[off:off + n].view(p.shape)
            view.copy_(p.data.to(self.device, torch.float32))
            p.data = view
            p.grad = self.grads[off:off + n].view(p.shape)
            self._off[id(p)] = (off, n)
            self.live_names.append(name)
            off += ((n + 3) // 4) * 4
        self.n_live = total
        self.live_params = [p for _, p in live]
        for name, p in m.named_parameters():              # dead parameters just move to the device
            if id(p) not in self._off and p.device != self.device:
                p.data = p.data.to(self.device)
        for name, b in m.named_buffers():
            if b.device != self.device:
                b.data = b.data.to(self.device)

    def grad_ptr(self, param):
        off, _ = self._off[id(param)]
        return self.grads.data_ptr() + 4 * off

    def grad_range(self, param):
        return self._off[id(param)]

    # ------------------------------------------------------------------ packed conv weights
    def packed(self, conv, taps_khkw, transposed, bwd=False, d1=None):
        """Packed copy of a convolution weight.  ``bwd``: only the backward program reads it (data-gradient packs): such copies are
        refreshed on the side stream while the forward pass runs.  ``d1`` = (first, count): only that slice of the weight's SECOND axis
        (the input channels of an nn.Conv2d) - the full-resolution part of the factored hypercolumn convolution (Graph.conv_hyper)."""
        key = (id(conv), tuple(taps_khkw), bool(transposed)) + ((tuple(d1),) if d1 else ())
        if key in self._packed:
            if not bwd and self._pack_is_bwd.get(key, False):
                self._pack_is_bwd[key] = False
                self._pack_batched_n = -1
            return self._packed[key]
        self._pack_is_bwd[key] = bool(bwd)
        self._pack_keys.append(key)
        D0, D1, KH, KW = conv.weight.shape
        c1 = d1[1] if d1 else D1
        n, c = (c1, D0) if transposed else (D0, c1)
        elems = lib.salt_packed_weight_elems(DT_CODE[self.dtype], len(taps_khkw), n, c)
        t = torch.zeros(elems, dtype=TORCH_DT[self.dtype], device=self.device)
        self._pack_ops.add('pack_conv_weight', dtype=DT_CODE[self.dtype], w=conv.weight.data_ptr() + (4 * d1[0] * KH * KW if d1 else 0), D0=D0, D1=D1,
                           KH=KH, KW=KW, ntaps=len(taps_khkw), tap_kh=[a for a, _ in taps_khkw], tap_kw=[b for _, b in taps_khkw],
                           transpose=int(transposed), wp=t.data_ptr(), d1_cnt=(d1[1] if d1 else 0))
        self._pack_ops._entries = None
        self._packed[key] = t
        self._packed_version = -1
        self._pack_batched_n = -1
        return t

    def packed_tapgemm(self, conv, c0, cn, transposed, bwd=False):
        """The KH x KW tap matrices W[:, c0 : c0 + cn, kh, kw] of a convolution weight as ONE packed 1x1 weight: forward
        (transposed=False) cn -> KH KW Cout output channels, channel t Cout + o = tap t of output channel o; transposed: the 1x1 data
        gradient KH KW Cout -> cn.  One pack job per tap into the shared tensor (salt_pack_conv_weight_args.n_off / chunk_off).  The
        low-resolution half of the factored hypercolumn convolution (Graph.hyper_level, saltnet.h salt_hyper_stencil)."""
        key = (id(conv), 'tapgemm', c0, cn, bool(transposed))
        if key in self._packed:
            return self._packed[key]
        D0, D1, KH, KW = conv.weight.shape
        nt = KH * KW
        kce = 16 if self.dtype == 'f32' else 32
        if transposed and D0 % kce:
            raise SaltError('tap GEMM: %d output channels are not a multiple of the %d-channel pack chunk' % (D0, kce))
        n, c = (cn, nt * D0) if transposed else (nt * D0, cn)
        elems = lib.salt_packed_weight_elems(DT_CODE[self.dtype], 1, n, c)
        t = torch.zeros(elems, dtype=TORCH_DT[self.dtype], device=self.device)
        for tap in range(nt):
            pkey = key + (tap,)
            self._pack_is_bwd[pkey] = bool(bwd)
            self._pack_keys.append(pkey)
            self._pack_ops.add('pack_conv_weight', dtype=DT_CODE[self.dtype], w=conv.weight.data_ptr() + 4 * c0 * KH * KW, D0=D0, D1=D1, KH=KH, KW=KW,
                               ntaps=1, tap_kh=[tap // KW], tap_kw=[tap % KW], transpose=int(transposed), wp=t.data_ptr(), d1_cnt=cn,
                               n_off=0 if transposed else tap * D0, n_total=n, chunk_off=(tap * D0 // kce) if transposed else 0)
        self._pack_ops._entries = None
        self._packed[key] = t
        self._packed_version = -1
        self._pack_batched_n = -1
        return t

    def packed_phases(self, conv, phase_taps, transposed, bwd=False):
        """Packed weights of a phase-fused stride-2 launch: one block of len(phase_taps[0]) taps per output-parity phase, contiguous;
        ``phase_taps[p][t]`` is (kh, kw) or None (a tap that phase does not have: packed as zeros).  -> (tensor, elements per phase)"""
        key = (id(conv), tuple(tuple(t) for t in phase_taps), bool(transposed), 'phases')
        D0, D1, KH, KW = conv.weight.shape
        n, c = (D1, D0) if transposed else (D0, D1)
        nt = len(phase_taps[0])
        elems = lib.salt_packed_weight_elems(DT_CODE[self.dtype], nt, n, c)
        if key in self._packed:
            return self._packed[key], elems
        t = torch.zeros(elems * len(phase_taps), dtype=TORCH_DT[self.dtype], device=self.device)
        for ph, taps in enumerate(phase_taps):
            pkey = key + (ph,)
            self._pack_is_bwd[pkey] = bool(bwd)
            self._pack_keys.append(pkey)
            self._pack_ops.add('pack_conv_weight', dtype=DT_CODE[self.dtype], w=conv.weight.data_ptr(), D0=D0, D1=D1, KH=KH, KW=KW, ntaps=nt,
                               tap_kh=[(tp[0] if tp is not None else -1) for tp in taps], tap_kw=[(tp[1] if tp is not None else 0) for tp in taps],
                               transpose=int(transposed), wp=t.data_ptr() + ph * elems * t.element_size())
        self._pack_ops._entries = None
        self._packed[key] = t
        self._packed_version = -1
        self._pack_batched_n = -1
        return t, elems

    def packed_stem(self, conv):
        key = (id(conv), 'stem')
        if key in self._packed:
            return self._packed[key]
        Cout, Cin, K, _ = conv.weight.shape
        TT = (K + 1) // 2
        elems = lib.salt_packed_weight_elems(DT_CODE[self.dtype], TT * TT, Cout, 16)
        t = torch.zeros(elems, dtype=TORCH_DT[self.dtype], device=self.device)
        self._pack_ops.add('pack_stem_weight', dtype=DT_CODE[self.dtype], w=conv.weight.data_ptr(), Cout=Cout, Cin=Cin, K=K, wp=t.data_ptr())
        self._pack_ops._entries = None
        self._packed[key] = t
        self._packed_version = -1
        return t

    def bn_work(self, bn):
        if id(bn) in self._bn:
            return self._bn[id(bn)]
        C = bn.num_features
        w = {k: torch.zeros(C, dtype=torch.float32, device=self.device) for k in ('mean', 'invstd', 'scale', 'shift')}
        self._fold_ops.add('bn_fold', C=C, gamma=bn.weight.data_ptr(), beta=bn.bias.data_ptr(), running_mean=bn.running_mean.data_ptr(),
                           running_var=bn.running_var.data_ptr(), eps=bn.eps, scale=w['scale'].data_ptr(), shift=w['shift'].data_ptr())
        self._fold_ops._entries = None
        self._bn[id(bn)] = w
        self._folded_version = None
        return w

    def touch(self, weights=True, stats=True):
        """Master weights / BN statistics changed: packed copies / folded BN are stale."""
        if weights:
            self.wver += 1
        if stats:
            self.sver += 1

    def _build_pack_batch(self):
        """Two launches pack every convolution weight (per-job argument structs live in device tables): the copies the forward
        pass reads on the main stream, the data-gradient copies on the side stream (they are not needed before backward)."""
        import numpy as np
        jobs = [s for (name, fn, s) in self._pack_ops.ops if name == 'pack_conv_weight']
        assert len(jobs) == len(self._pack_keys)
        others = Program('pack_misc')
        for (name, fn, s), st in zip(self._pack_ops.ops, self._pack_ops.streams):
            if name != 'pack_conv_weight':
                others.ops.append((name, fn, s)); others.streams.append(st)
        self._pack_tables = []

        def batch(sel, name):
            prog = Program(name)
            if sel:
                raw = b''.join(bytes(s) for s in sel)
                blocks = [lib.salt_pack_job_blocks(ctypes.byref(s)) for s in sel]
                pref = np.concatenate([[0], np.cumsum(blocks)]).astype(np.int32)
                table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
                pref_t = torch.from_numpy(pref).to(self.device)
                self._pack_tables += [table, pref_t]
                prog.add('pack_batched', jobs=table.data_ptr(), job_block0=pref_t.data_ptr(), njobs=len(sel), total_blocks=int(pref[-1]),
                         dtype=DT_CODE[self.dtype])
            return prog
        fwd_jobs = [s for s, k in zip(jobs, self._pack_keys) if not self._pack_is_bwd[k]]
        fwd = batch(fwd_jobs, 'pack')
        fwd.extend(others)
        fwd.finalize()
        # the forward packs FusedAdam writes itself while it updates the weights (salt_adam_pack): vector-path jobs, one per weight, whose
        # master lives in the flat buffer; `_pack_batched_rest` = what a refresh right behind such an optimizer step still has to run
        lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.n_live
        seen, self._adam_jobs = set(), []
        for s in fwd_jobs:
            if lib.salt_pack_job_is_vec(ctypes.byref(s)) and lo <= s.w < hi and s.w not in seen:
                seen.add(s.w)
                self._adam_jobs.append(s)
        ids = {id(s) for s in self._adam_jobs}
        rest = batch([s for s in fwd_jobs if id(s) not in ids], 'pack_rest')
        rest.extend(others)
        rest.finalize()
        self._pack_batched_rest = rest
        self._adam_packed_version = None                 # (new / moved pack tensors: the next refresh runs every job)
        bwd = batch([s for s, k in zip(jobs, self._pack_keys) if self._pack_is_bwd[k]], 'pack_bwd')
        bwd.finalize()
        self._pack_batched, self._pack_batched_bwd = fwd, bwd
        self._pack_batched_n = len(self._pack_ops)
        # a captured step graph bakes the device pointers of THESE tables in: every rebuild starts a new generation, and step_graph()
        # recaptures an executable of an older one instead of replaying it against freed tables
        self._pack_generation = getattr(self, '_pack_generation', 0) + 1

    def refresh(self, train, defer_bwd=False):
        """Bring the packed weight copies (and, in eval mode, the folded BN constants) up to date.  ``defer_bwd``: leave the
        data-gradient packs to a later ``refresh(True)`` (Engine.forward issues them after the forward program, so that they run under
        the loss kernel - 32 workgroups on an otherwise idle GPU - instead of competing with the first forward layers)."""
        if getattr(self, '_pack_batched_n', -1) != len(self._pack_ops):
            self._build_pack_batch()
            self._packed_version = self._packed_bwd_version = -1
        if self._packed_version != self.wver:
            # the optimizer step that produced this weight version wrote the vector-path forward packs itself (FusedAdam -> salt_adam_pack)
            (self._pack_batched_rest if getattr(self, '_adam_packed_version', None) == self.wver else self._pack_batched).run()
            self._packed_version = self.wver
        if train and not defer_bwd and self._packed_bwd_version != self.wver and len(self._pack_batched_bwd):
            # ordered after the optimizer step on the main stream; the backward program's first data-gradient joins (engine.py)
            if os.environ.get('SALT_PACK_BWD_MAIN'):                 # A/B: everything on the main stream
                self._pack_batched_bwd.run()
            else:
                self.side_stream.wait_stream(torch.cuda.current_stream())
                self._pack_batched_bwd.run(stream=self.side_stream)
            self._packed_bwd_version = self.wver
        if not train and self._folded_version != (self.wver, self.sver):
            self._fold_ops.run()
            self._folded_version = (self.wver, self.sver)

    # ------------------------------------------------------------------ compiled instances
    def net(self, shape, train):
        key = (tuple(shape), bool(train))
        if key not in self.nets:
            self.nets[key] = CompiledNet(self, tuple(shape), train, self.module.num_classes)
        return self.nets[key]

    # ------------------------------------------------------------------ one training step as ONE hipGraph
    def step_graph(self, net, kind, loss_scale, optimizer):
        """Capture pack -> forward (two streams) -> data-gradient packs -> loss -> backward (two streams) -> Adam of one compiled
        instance into a hipGraph (cached); replaying it is one launch instead of ~490.  Everything the step touches is static
        device memory; the learning rate / bias corrections live in the optimizer's device `hyper` vector (adam_tick)."""
        # the cache lives ON the compiled instance (which this engine owns) and holds the optimizer object itself: a rebuilt engine /
        # net / optimizer can never alias a stale executable through a recycled id()
        cache = net.__dict__.setdefault('_step_graphs', {})
        key = (kind, float(loss_scale))
        if getattr(self, '_pack_batched_n', -1) != len(self._pack_ops):
            self._build_pack_batch()
        hit = cache.get(key)
        if hit is not None and hit[0] is optimizer and hit[2] == self._pack_generation:
            return hit[1]
        if hit is not None:
            lib.salt_graph_destroy(hit[1])
        optimizer._bind()
        optimizer._sync_hyper()
        loss_prog = net.loss_program(kind, loss_scale)
        st = self._graph_stream = getattr(self, '_graph_stream', None) or torch.cuda.Stream(device=self.device)
        torch.cuda.synchronize()
        exec_ = ctypes.c_void_p()
        with torch.cuda.stream(st):
            _abi.check(lib.salt_graph_begin(ctypes.c_void_p(st.cuda_stream)), 'graph_begin')
            try:
                self._pack_batched.run(stream=st)
                net.fwd.run(stream=st, side=self.side_stream)
                if len(self._pack_batched_bwd):
                    # fork: the data-gradient packs run on the side stream under the loss kernel; backward's first operator joins
                    _abi.check(lib.salt_program_run_streams_ex(ctypes.cast(self._pack_fork_entries(), ctypes.c_void_p), 0, 1, ctypes.c_void_p(st.cuda_stream),
                                                               ctypes.c_void_p(self.side_stream.cuda_stream), 0), 'pack_fork')
                loss_prog.run(stream=st)
                net.bwd.run(stream=st, side=self.side_stream)
                optimizer.prog.run(stream=st)
            finally:
                rc = lib.salt_graph_end(ctypes.c_void_p(st.cuda_stream), ctypes.byref(exec_))
            _abi.check(rc, 'graph_end')
        torch.cuda.synchronize()
        cache[key] = (optimizer, exec_, self._pack_generation)
        return exec_

    def release_step_graphs(self):
        """Destroy the captured step executables of every compiled instance (called when the engine is dropped)."""
        for net in self.nets.values():
            for _, ex, *_gen in net.__dict__.pop('_step_graphs', {}).values():
                lib.salt_graph_destroy(ex)

    def _pack_fork_entries(self):
        """The data-gradient pack launch as a one-entry program tagged for the side stream (the executor forks with an event)."""
        if getattr(self, '_pack_fork', None) is None or self._pack_fork[0] is not self._pack_batched_bwd:
            Entry = _abi.STRUCTS['salt_program_entry']
            arr = (Entry * 1)()
            name, fn, s = self._pack_batched_bwd.ops[0]
            arr[0].fn = ctypes.cast(fn, ctypes.c_void_p).value
            arr[0].args = ctypes.addressof(s)
            arr[0].stream = 1
            self._pack_fork = (self._pack_batched_bwd, arr)
        return self._pack_fork[1]

    def run_step_graph(self, net, kind, loss_scale, optimizer):
        exec_ = self.step_graph(net, kind, loss_scale, optimizer)
        optimizer._sync_hyper()
        cur = torch.cuda.current_stream()
        st = self._graph_stream
        st.wait_stream(cur)                                  # the input / target copies of this step
        _abi.check(lib.salt_graph_launch(exec_, ctypes.c_void_p(st.cuda_stream)), 'graph_launch')
        cur.wait_stream(st)
        optimizer.steps += 1
        self._packed_version = self._packed_bwd_version = self.wver      # the graph packed the weights it started from ...
        self.touch(weights=True, stats=True)                             # ... and Adam / BatchNorm moved them

    def forward(self, x, train, bound=False):
        """``bound``: the caller pointed the programs at ``x`` itself (CompiledNet.bind) - no copy into the static input buffer."""
        _require_gpu(x.device)
        net = self.net(x.shape, train)
        if not bound:
            net.bind(x=None)
            net.x.copy_(x)
        late = train and bool(os.environ.get('SALT_PACK_BWD_LATE'))   # round 2: the loss is ~65 us now, too short to hide the packs
        self.refresh(train, defer_bwd=late)
        net.fwd.run(side=None if os.environ.get('SALT_NO_FWD_SIDE') else self.side_stream)
        if late:
            self.refresh(True)                               # data-gradient packs behind the forward pass (A/B: SALT_PACK_BWD_LATE)
        if train:
            self.touch(weights=False, stats=True)            # BN running statistics moved
        return net
