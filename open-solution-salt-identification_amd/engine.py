"""Static-program engine: builds the forward and backward operator lists of a network ONCE per
(batch shape, mode) and runs each as a single call into libsaltnet_hip.so.

The reference executes its U-Nets through PyTorch autograd, one operator at a time from Python
(common_blocks/models.py:105-136).  Here the module tree (architectures.py) "emits" its operators into
a :class:`Graph`; every emitter also pushes a closure that later emits the matching backward operators,
so the backward program is derived mechanically in reverse order with static gradient-accumulation
flags.  Activations are NHWC buffers that live for the lifetime of the graph (288 GB of HBM makes
re-materialisation pointless); skip connections and the hypercolumn are channel SLICES of wider
buffers that their producers write in place, so ``torch.cat`` never copies anything.

PyTorch is used for device memory (tensors), streams and nothing else on this path.
"""
import ctypes
import os

import numpy as np
import torch

from ._abi import OP_FUNCS, STRUCTS, SaltError, check, fill, lib

DT_CODE = {'f32': 0, 'bf16': 1}
TORCH_DT = {'f32': torch.float32, 'bf16': torch.bfloat16}
VEC = {'f32': 4, 'bf16': 8}


def fwd_bn_fold():
    """SALT_FWD_BN_FOLD=1 (measurement / verification switch, DESIGN history round 3): the BatchNorm apply + ReLU of a conv -> BN -> ReLU output whose
    only reader is ONE 3x3 convolution runs in that convolution's loader (salt_conv_args.in_*; forward values bit-identical, tested).
    The weight-gradient kernels cannot re-derive the activation, so such a graph is FORWARD-ONLY: its backward closure raises instead
    of computing a wrong gradient."""
    return os.environ.get('SALT_FWD_BN_FOLD', '0') not in ('', '0')


def _round_up(v, m):
    return (v + m - 1) // m * m


class Scratch:
    """Named shared workspace request; resolved to one allocation (max size) when the graph is finalized."""

    def __init__(self, name, nbytes):
        self.name, self.nbytes = name, int(nbytes)


_AUX_STREAMS = {}


def _aux_stream(side):
    """The auxiliary stream that goes with a side stream (salt_set_aux_stream): one per side stream, created on first use."""
    a = _AUX_STREAMS.get(side.cuda_stream)
    if a is None:
        a = _AUX_STREAMS[side.cuda_stream] = torch.cuda.Stream(device=side.device)
    return a


class Program:
    """A flat list of (operator, argument struct) pairs executed by salt_program_run."""

    def __init__(self, name=''):
        self.name = name
        self.ops = []          # (opname, fn, struct)
        self.default_stream = 0
        self.join_next = False
        self.streams = []      # 0 main / 1 side, per op
        self.patches = []      # (struct, path, Scratch)
        self._entries = None
        self.marks = {}
        self._has_aux = None           # any stream-tag-4 entry (slab reductions on the auxiliary stream)?
        self._pre_run = None           # callable(stream, begin) run before the entries are issued (Graph.finalize: shard hygiene)
        self._post_run = None          # callable(begin, end) after they were issued

    def add(self, opname, stream=None, **fields):
        fn, S = OP_FUNCS['salt_' + opname]
        s = S()
        if stream is None:                       # default tag: the enclosing Graph.side() block, or a pending join
            stream = self.default_stream
            if stream == 0 and self.join_next:
                stream, self.join_next = 2, False
        self.streams.append(stream)
        plain = {}
        for k, v in fields.items():
            if isinstance(v, Scratch):
                self.patches.append((s, (k,), v))
            elif isinstance(v, tuple) and len(v) == 2 and isinstance(v[1], Scratch):   # (view, scratch) -> view.p patched
                plain[k] = v[0]
                self.patches.append((s, (k, 'p'), v[1]))
            else:
                plain[k] = v
        fill(s, **plain)
        self.ops.append((opname, fn, s))
        return s

    def set_fields(self, s, **fields):
        """Set more fields of an argument struct that ``add`` returned (same conventions: Scratch values are patched at finalize)."""
        plain = {}
        for k, v in fields.items():
            if isinstance(v, Scratch):
                self.patches.append((s, (k,), v))
            else:
                plain[k] = v
        fill(s, **plain)

    def mark(self, label):
        self.marks[label] = len(self.ops)

    def extend(self, other):
        self.ops.extend(other.ops)
        self.streams.extend(other.streams)
        self.patches.extend(other.patches)
        self._entries = None

    def finalize(self):
        n = len(self.ops)
        Entry = STRUCTS['salt_program_entry']
        arr = (Entry * max(n, 1))()
        for i, (_, fn, s) in enumerate(self.ops):
            arr[i].fn = ctypes.cast(fn, ctypes.c_void_p).value
            arr[i].args = ctypes.addressof(s)
            arr[i].stream = self.streams[i]
        self._entries = arr
        return self

    def __len__(self):
        return len(self.ops)

    def run(self, stream=None, begin=0, end=None, side=None, join=True, marks=None):
        """Enqueue ops [begin, end).  With ``side`` (a torch.cuda.Stream) ops tagged stream=1 run on it concurrently; ``join=False``
        leaves the side stream un-joined at the end of the range (the caller orders its consumers after both streams).
        ``marks`` = (positions int array, n, main-event array, side-event array) prepared by parallel.DataParallel: the executor records
        the two events of a mark before it issues the entry at that position (salt_program_run_streams_marks)."""
        if self._entries is None:
            self.finalize()
        end = len(self.ops) if end is None else end
        if self._pre_run is not None:
            self._pre_run(stream, begin)
        st = _stream_ptr(stream)
        if side is None and 3 in self.streams[begin:end]:
            raise SaltError('program %s joins the side stream (data-gradient weight packs): run it with side=engine.side_stream' % self.name)
        if side is not None and marks is not None:
            rc = lib.salt_program_run_streams_marks(ctypes.cast(self._entries, ctypes.c_void_p), begin, end, st, ctypes.c_void_p(side.cuda_stream),
                                                    1 if join else 0, marks[0], marks[1], marks[2], marks[3])
        elif side is not None:
            if self._has_aux is None:
                self._has_aux = 4 in self.streams or 5 in self.streams
            if self._has_aux:
                lib.salt_set_aux_stream(None if os.environ.get('SALT_NO_AUX_STREAM') else ctypes.c_void_p(_aux_stream(side).cuda_stream))
            rc = lib.salt_program_run_streams_ex(ctypes.cast(self._entries, ctypes.c_void_p), begin, end, st, ctypes.c_void_p(side.cuda_stream),
                                                 1 if join else 0)
        else:
            rc = lib.salt_program_run_range(ctypes.cast(self._entries, ctypes.c_void_p), begin, end, st)
        if rc != 0:
            msg = lib.salt_last_error().decode(errors='replace')
            m = None
            try:
                m = int(msg.split('program entry')[1].split()[0])
            except Exception:
                pass
            where = (' [%s]' % self.ops[m][0]) if m is not None and m < len(self.ops) else ''
            raise SaltError('program %s failed (%d)%s: %s' % (self.name, rc, where, msg))
        if self._post_run is not None:
            self._post_run(begin, end)

    def capture(self, stream):
        """Capture the whole program (single stream, tags ignored) into a hipGraph on ``stream`` (not the default stream)."""
        if self._entries is None:
            self.finalize()
        self.release_graph()
        h = ctypes.c_void_p()
        check(lib.salt_graph_capture(ctypes.cast(self._entries, ctypes.c_void_p), len(self.ops), ctypes.c_void_p(stream.cuda_stream),
                                     ctypes.byref(h)), 'graph_capture')
        self._graph = h
        return self

    def replay(self, stream=None):
        if getattr(self, '_graph', None) is None:
            raise SaltError('program %s has no captured graph' % self.name)
        check(lib.salt_graph_launch(self._graph, _stream_ptr(stream)), 'graph_launch')

    def release_graph(self):
        if getattr(self, '_graph', None) is not None:
            lib.salt_graph_destroy(self._graph)
            self._graph = None

    def run_timed(self, stream=None):
        """Run with a HIP event pair around every entry (on the launch stream); returns [(opname, struct, ms)]."""
        if self._entries is None:
            self.finalize()
        n = len(self.ops)
        ms = (ctypes.c_float * max(n, 1))()
        if self._pre_run is not None:
            self._pre_run(stream, 0)
        check(lib.salt_program_run_timed(ctypes.cast(self._entries, ctypes.c_void_p), 0, n, _stream_ptr(stream), ms), 'run_timed')
        if self._post_run is not None:
            self._post_run(0, n)
        return [(self.ops[i][0], self.ops[i][2], float(ms[i])) for i in range(n)]

    def run_debug(self, stream=None):
        """Run op by op with a device sync after each (pinpoints a faulting kernel)."""
        if self._pre_run is not None:
            self._pre_run(stream, 0)
        st = _stream_ptr(stream)
        for i, (name, fn, s) in enumerate(self.ops):
            check(fn(ctypes.byref(s), st), '%s[%d] %s' % (self.name, i, name))
            torch.cuda.synchronize()
        if self._post_run is not None:
            self._post_run(0, len(self.ops))


def _stream_ptr(stream):
    if stream is None:
        stream = torch.cuda.current_stream()
    return ctypes.c_void_p(stream.cuda_stream)


class Buffer:
    """NHWC storage [B,H,W,Ct] (+ lazily allocated gradient storage of the same geometry).  ``planes`` > 0: PLANAR storage
    [C / planes][B,H,W,planes] for both - every channel slice of `planes` channels is a dense tensor of its own, and a launch that takes
    the whole buffer names the plane stride (salt_conv_args.x_plane / y_plane, salt_conv_wgrad_args.q_plane)."""

    def __init__(self, graph, B, H, W, C, name='', planes=0):
        self.g, self.B, self.H, self.W, self.C = graph, B, H, W, C
        self.Ct = _round_up(C, graph.ve)
        self.name = name
        self.planes = planes
        if planes and (C % planes or planes % graph.ve):
            raise SaltError('planar buffer %s: %d channels in planes of %d' % (name, C, planes))
        self.t = graph.alloc(self._shape(), graph.tdtype)
        self.grad_t = None
        self.grad_init = np.zeros(self.Ct, dtype=bool)
        self.grad_writers = []         # one entry per gradient writer in backward-program order: [c0, C, conv-args struct | None]
        self.reads = 0                 # views handed out (SALT_FWD_BN_FOLD: an activation nobody else reads need not be materialised)

    def _shape(self):
        return (self.C // self.planes, self.B, self.H, self.W, self.planes) if self.planes else (self.B, self.H, self.W, self.Ct)

    def plane_elems(self):
        return self.B * self.H * self.W * self.planes

    def grad(self):
        if self.grad_t is None:
            self.grad_t = self.g.alloc(self._shape(), self.g.tdtype)
        return self.grad_t


class Act:
    """A channel slice [c0, c0+C) of a Buffer."""

    def __init__(self, buf, c0=0, C=None):
        self.buf, self.c0 = buf, c0
        self.C = buf.C if C is None else C

    B = property(lambda s: s.buf.B)
    H = property(lambda s: s.buf.H)
    W = property(lambda s: s.buf.W)

    def slice(self, c0, C):
        return Act(self.buf, self.c0 + c0, C)

    def _view(self, t):
        v = STRUCTS['salt_view']()
        pc = self.buf.planes
        if pc:
            # one dense plane, or the whole buffer (cs < C: only a launch that also names the plane stride may take it)
            whole = self.c0 == 0 and self.C == self.buf.C
            if self.c0 % pc or not (self.C == pc or whole):
                raise SaltError('planar buffer %s: slice [%d, %d) is not one plane of %d channels' % (self.buf.name, self.c0, self.c0 + self.C, pc))
            v.p = t.data_ptr() + (self.c0 // pc) * self.buf.plane_elems() * t.element_size()
            v.B, v.H, v.W, v.C, v.cs = self.buf.B, self.buf.H, self.buf.W, self.C, pc
            return v
        v.p = t.data_ptr() + self.c0 * t.element_size()
        v.B, v.H, v.W, v.C, v.cs = self.buf.B, self.buf.H, self.buf.W, self.C, self.buf.Ct
        return v

    def plane_stride(self):
        """plane stride (elements) a launch over the WHOLE planar buffer passes as x_plane / y_plane / q_plane; 0 for ordinary buffers"""
        return self.buf.plane_elems() if (self.buf.planes and self.C != self.buf.planes) else 0

    def view(self):
        self.buf.reads += 1
        return self._view(self.buf.t)

    def gview(self):
        return self._view(self.buf.grad())

    def grad_state(self):
        """-> accumulate flag for the next writer of this slice's gradient; marks it initialised."""
        st = self.buf.grad_init[self.c0:self.c0 + self.C]
        self.buf.grad_writers.append([self.c0, self.C, None])
        if st.all():
            return 1
        if st.any():
            raise SaltError('gradient of %s partially initialised' % self.buf.name)
        self.buf.grad_init[self.c0:self.c0 + self.C] = True
        return 0

    def grad_ready(self):
        return bool(self.buf.grad_init[self.c0:self.c0 + self.C].all())

    def _nhwc(self, t):
        if self.buf.planes:
            t = t.permute(1, 2, 3, 0, 4).reshape(self.buf.B, self.buf.H, self.buf.W, self.buf.C)
        return t[..., self.c0:self.c0 + self.C]

    def tensor(self):
        """NHWC torch view [B,H,W,C] of the activation (tests / debugging; a copy for planar buffers)."""
        return self._nhwc(self.buf.t)

    def grad_tensor(self):
        return self._nhwc(self.buf.grad())


def null_view():
    return STRUCTS['salt_view']()


def shaped_view(ptr, B, H, W, C, cs=None):
    v = STRUCTS['salt_view']()
    v.p, v.B, v.H, v.W, v.C, v.cs = ptr, B, H, W, C, (C if cs is None else cs)
    return v


def conv_taps(KH, KW, pad_y, pad_x):
    """[(kh, kw, dy, dx)] for a cross-correlation with the given top/left padding."""
    return [(kh, kw, kh - pad_y, kw - pad_x) for kh in range(KH) for kw in range(KW)]


class Graph:
    """Forward + backward programs of one network instance for fixed (B, H, W), dtype and BN mode."""

    def __init__(self, engine, train):
        self.engine = engine
        self.device = engine.device
        self.dtype = engine.dtype
        self.dt = DT_CODE[self.dtype]
        self.tdtype = TORCH_DT[self.dtype]
        self.ve = VEC[self.dtype]
        self.train = train
        self.fwd = Program('forward')
        self.bwd = Program('backward')
        self.tape = []
        self.keep = []
        self.scratch = {}
        self.bytes = 0
        self._touched = []
        self.grad_ready = []
        self._scratch_sfx = ''
        # fp64 statistics shards of the train-mode BatchNorm layers (SALT_BN_FIN=2): slices of one arena per program, cleared by ONE
        # salt_zero at the head of the program
        self._pending_reduces, self._deferred_params, self._pending_bytes, self._reduce_batches = [], [], 0, []
        self._n_slab = 0
        self._fin_bytes = {'fwd': 0, 'bwd': 0}
        self._fin_patches = []                   # (struct, field, 'fwd' | 'bwd', byte offset)
        self._fin_zero = {}
        if train and self._fin_mode() == 2:
            self._fin_zero['fwd'] = self.fwd.add('zero', p=1, bytes=0)

    # ------------------------------------------------------------------ memory
    def alloc(self, shape, dtype, zero=True):
        t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
        self.keep.append(t)
        self.bytes += t.numel() * t.element_size()
        return t

    def new_act(self, B, H, W, C, name='', planes=0):
        return Act(Buffer(self, B, H, W, C, name, planes))

    # ------------------------------------------------------------------ forward branches on the side stream
    def side(self):
        """``with g.side():`` - forward operators emitted inside run on the side stream, concurrently with the main-stream operators
        emitted after the block.  The caller must call ``g.join()`` before emitting the first consumer of their results.  Shared
        scratch workspaces get their own copies ('@side').  Backward operators of the same layers are unaffected."""
        g = self

        class _Side:
            def __enter__(self_):
                g.fwd.default_stream, g._scratch_sfx = 1, '@side'

            def __exit__(self_, *a):
                g.fwd.default_stream, g._scratch_sfx = 0, ''
        return _Side()

    def join(self):
        """The next main-stream forward operator waits for everything the side stream has been given so far."""
        self.fwd.join_next = True

    def f32(self, n):
        return self.alloc((max(int(n), 1),), torch.float32)

    def finalize(self):
        for buf, reads in getattr(self, '_virtual_acts', []):
            if buf.reads != reads:
                raise SaltError('activation %s is applied on the fly by its only consumer (Graph._take_act_op) but another operator asked for its storage' % buf.name)
        sizes = {}
        for prog in (self.fwd, self.bwd):
            for _, _, sc in prog.patches:
                sizes[sc.name] = max(sizes.get(sc.name, 0), sc.nbytes)
        for name, nb in sizes.items():
            self.scratch[name] = self.alloc((_round_up(max(nb, 16), 16),), torch.uint8, zero=False)
        for prog in (self.fwd, self.bwd):
            for s, path, sc in prog.patches:
                obj = s
                for a in path[:-1]:
                    obj = getattr(obj, a)
                setattr(obj, path[-1], self.scratch[sc.name].data_ptr())
            prog.finalize()
        # device tables of the batched slab reductions (the jobs' slab pointers were patched just above)
        import numpy as np
        for op, jobs, blocks in self._reduce_batches:
            table = torch.frombuffer(bytearray(b''.join(bytes(j) for j in jobs)), dtype=torch.uint8).to(self.device)
            pref = torch.from_numpy(np.concatenate([[0], np.cumsum(blocks)]).astype(np.int32)).to(self.device)
            self.keep += [table, pref]
            fill(op, jobs=table.data_ptr(), job_block0=pref.data_ptr())
        # ONE arena for the fp64 statistics shards of both programs, cleared by ONE salt_zero at the head of the FORWARD program (a
        # training step runs forward -> loss -> backward exactly once each; the backward program's own clear - a 6 us launch at the
        # head of the critical queue - is dropped: its salt_zero entry stays in the program with 0 bytes, which launches nothing).
        # SALT_SPLIT_ZERO=1: each program clears its own half (for callers that replay a backward program on its own).
        split = bool(os.environ.get('SALT_SPLIT_ZERO')) or 'fwd' not in self._fin_zero
        nbf, nbb = self._fin_bytes['fwd'], self._fin_bytes['bwd']
        if self._fin_zero:
            arena = self.alloc((max((nbf + nbb) // 8, 1),), torch.float64)
            base = {'fwd': arena.data_ptr(), 'bwd': arena.data_ptr() + nbf}
            for which, z in self._fin_zero.items():
                if split:
                    fill(z, p=base[which], bytes=self._fin_bytes[which])
                else:
                    fill(z, p=base[which], bytes=(nbf + nbb) if which == 'fwd' else 0)
            for st, field, w, off in self._fin_patches:
                setattr(st, field, base[w] + off)
            if not split and nbb and 'bwd' in self._fin_zero:
                # The backward shards are accumulate-only and only the FORWARD program clears them: a backward program that is run
                # again without a fresh forward pass (loss.backward(retain_graph=True) through autograd.py, a second loss's backward,
                # bwd.run_timed in the profiling tools) would add its BatchNorm-backward sums on top of the previous run's.
                # The programs keep a 'backward shards dirty' flag: forward clears it, backward sets it, and a backward run that finds
                # it set clears its own half first (one salt_zero launch, only ever on that off-path case).
                rz = Program('rezero_bwd_shards')
                rz.add('zero', p=base['bwd'], bytes=nbb)
                rz.finalize()
                state = self._shard_state = {'dirty': False, 'rezeroed': 0}

                def fwd_done(begin, end, state=state):
                    if begin == 0:
                        state['dirty'] = False

                def bwd_begin(stream, begin, state=state, rz=rz):
                    if begin == 0:
                        if state['dirty']:
                            rz.run(stream=stream)
                            state['rezeroed'] += 1
                        state['dirty'] = True
                self.fwd._post_run, self.bwd._pre_run = fwd_done, bwd_begin
        return self

    def _fin_slot(self, which, ndoubles, *targets):
        """Reserve ``ndoubles`` fp64 of the program's statistics arena; ``targets`` = (struct, field) pairs that receive its address."""
        off = self._fin_bytes[which]
        self._fin_bytes[which] = off + _round_up(ndoubles * 8, 64)
        for st, field in targets:
            self._fin_patches.append((st, field, which, off))
        return off

    def _gp(self, param, deferred=False):
        """Gradient pointer of a parameter; remembers that the current backward closure finalises it (``deferred``: the NEXT batched
        slab reduction does - Graph._flush_reduces records the position)."""
        (self._deferred_params if deferred else self._touched).append(param)
        return self.engine.grad_ptr(param)

    # ------------------------------------------------------------------ batched weight-gradient slab reductions (round 6)
    def _reduce_batching(self):
        """SALT_WGRAD_BATCH_MB=<n> (OPT-IN; default 0 = one salt_wgrad_reduce launch per layer): the slab reductions of consecutive layers
        are collected - every layer then needs its OWN slab region (0.9 GB for the ResNet34 U-Net instead of one shared 25 MB workspace) -
        and issued as ONE salt_wgrad_reduce_batched launch whenever the collected gradients reach n megabytes or 16 layers.  Built for
        VERDICT r5 #3 and measured SLOWER on the step (same box: 5.06 ms per layer / 5.13 with per-layer slabs but a launch per layer /
        5.16 at 4 MB / 5.19 at 40 MB, profiles/r06_wgrad_batch2_ab.txt): the shared 25 MB workspace is written and read back out of the
        256 MB Infinity Cache, per-layer regions go to HBM - the slab round trip the counters show is cache traffic, not HBM time."""
        return float(os.environ.get('SALT_WGRAD_BATCH_MB', '0')) if self.train else 0.0

    def _queue_reduce(self, fields, weight, nbytes):
        """one layer's reduction: argument struct now, launch with the next batch"""
        S = STRUCTS['salt_wgrad_reduce_args']()
        self._n_slab = getattr(self, '_n_slab', 0) + 1
        sc = Scratch('wgrad#%d' % self._n_slab, nbytes)
        plain = dict(fields)
        plain['grad'] = self._gp(weight, deferred=True) + plain.pop('grad_off', 0)
        fill(S, **plain)
        self.bwd.patches.append((S, ('partials',), sc))
        self.keep.append(S)
        self._pending_reduces.append(S)
        self._pending_bytes += plain['ntaps'] * plain['Ca'] * plain['Cb'] * 4                      # bytes of gradient this job finishes
        if self._pending_bytes >= self._reduce_batching() * 1e6 or len(self._pending_reduces) >= 16:
            self._flush_reduces()
        return sc

    def _flush_reduces(self):
        if not getattr(self, '_pending_reduces', None):
            return
        jobs = self._pending_reduces
        blocks = [lib.salt_wgrad_reduce_job_blocks(ctypes.byref(j)) for j in jobs]
        if min(blocks) < 0:
            raise SaltError('wgrad_reduce_batched: bad job')
        op = self.bwd.add('wgrad_reduce_batched', stream=1, jobs=1, job_block0=1, njobs=len(jobs), total_blocks=int(sum(blocks)))
        self._reduce_batches.append((op, jobs, blocks))
        for p in self._deferred_params:
            off, n = self.engine.grad_range(p)
            self.grad_ready.append((off, n, len(self.bwd.ops)))
        self._pending_reduces, self._deferred_params, self._pending_bytes = [], [], 0

    def build_backward(self):
        """Emit the backward program (reverse tape order).  Also records, per parameter, the program position
        after which its gradient is final (parallel.plan_buckets turns that into all-reduce buckets)."""
        self.grad_ready = []
        self._resolve_lazies()
        if self.train and self._fin_mode() == 2 and 'bwd' not in self._fin_zero:
            self._fin_zero['bwd'] = self.bwd.add('zero', p=1, bytes=0)
        for fn in reversed(self.tape):
            self._touched = []
            fn()
            for p in self._touched:
                off, n = self.engine.grad_range(p)
                self.grad_ready.append((off, n, len(self.bwd.ops)))
        self._flush_reduces()
        self.tape = []
        for b in getattr(self, '_bias_bufs', []):        # every folded channel-SE term must have met its bn_bwd, or x.grad is incomplete
            if getattr(b, 'grad_bias', None) is not None:
                raise SaltError('gradient bias of %s was never consumed (no BatchNorm backward read that slice)' % b.name)
        # The data-gradient weight packs of the step were enqueued on the side stream during forward (Engine.refresh): the FIRST
        # main-stream operator of the backward program joins the side stream - before any weight-gradient kernel is enqueued there,
        # so the join waits for the packs only.
        if getattr(self, '_uses_bwd_packs', False):
            for i, st in enumerate(self.bwd.streams):
                if st == 0:
                    self.bwd.streams[i] = 3
                    break
                if st == 1:
                    raise SaltError('backward program starts with a side-stream operator')

    def planar_ok(self, B, H, W, C, pc, conv):
        """Can a [B,H,W,C] buffer that feeds ONE replicate-padded 3x3 convolution (the hypercolumn -> final Conv2dBnRelu,
        architectures/unet.py:101-109) be stored as C / pc dense planes?  Yes iff the library runs the forward launch on conv_ls_kernel
        (x_plane) and, in train mode, the data gradient on conv_ws_kernel (y_plane) and accepts q_plane for the weight gradient."""
        if self.dtype != 'bf16' or os.environ.get('SALT_NO_PLANAR') or C % pc or pc % self.ve:
            return False
        if self.train and self._fin_mode() != 2:
            return False                               # the per-tile partials protocols run on conv_mfma_kernel only
        Cout, _, KH, KW = conv.weight.shape
        if (KH, KW) != (3, 3):
            return False
        d = 1 << 20                                     # stand-in pointer (the probes only plan, nothing is launched)
        td = [(kh - (KH - 1), kw) for kh in range(KH) for kw in range(KW)]
        xv, yv, plane = shaped_view(d, B, H, W, C, pc), shaped_view(d, B, H, W, Cout), B * H * W * pc
        S = fill(STRUCTS['salt_conv_args'](), dtype=self.dt, x=xv, w=d, ntaps=9, tap_dy=[t[0] for t in td], tap_dx=[t[1] for t in td], in_step=1,
                 pad_mode=1, y=yv, OH=H, OW=W, out_step=1, x_plane=plane)
        if lib.salt_conv_kernel_id(ctypes.byref(S)) != 10:
            return False
        if not self.train:
            return True
        tg = [(-t[0] - (KH - 1), -t[1]) for t in td]
        S = fill(STRUCTS['salt_conv_args'](), dtype=self.dt, x=yv, w=d, ntaps=9, tap_dy=[t[0] for t in tg], tap_dx=[t[1] for t in tg], in_step=1,
                 pad_mode=0, y=xv, OH=H + KH - 1, OW=W + KW - 1, out_step=1, fold_top=KH - 1, fold_right=KW - 1, y_plane=plane)
        if lib.salt_conv_kernel_id(ctypes.byref(S)) != 9:
            return False
        Wg = fill(STRUCTS['salt_conv_wgrad_args'](), dtype=self.dt, p=yv, q=xv, ntaps=9, tap_dy=[t[0] for t in td], tap_dx=[t[1] for t in td],
                  q_step=1, pad_mode=1, q_plane=plane)
        return lib.salt_conv_wgrad_nsplit(ctypes.byref(Wg)) >= 1

    def _resolve_lazies(self):
        """SALT_FWD_BN_FOLD: drop the affine_act of every conv -> BN -> ReLU output whose only forward reader is ONE convolution that
        applies the transform in its loader; every other candidate goes back to reading the materialised activation."""
        drop = []
        for buf in getattr(self, '_lazies', []):
            lz = buf.lazy
            if len(lz['users']) == 1 and buf.reads == lz['reads0']:
                lz['dropped'] = True
                drop.append(next(i for i, (_, _, st) in enumerate(self.fwd.ops) if st is lz['sa']))
            else:
                for st in lz['users']:
                    fill(st, x=Act(buf).view(), in_fin=None, in_relu=0)
                    self._fin_patches = [q for q in self._fin_patches if q[0] is not st or q[1] != 'in_fin_acc']
        for i in sorted(drop, reverse=True):
            if self.fwd.streams[i] != 0:
                raise SaltError('a dropped affine_act carried a stream join')
            del self.fwd.ops[i]
            del self.fwd.streams[i]
        self._lazies = []
        self.n_folded = len(drop)

    # ------------------------------------------------------------------ helpers
    def _es(self):
        return 4 if self.dtype == 'f32' else 2

    def _conv_launch(self, prog, x_view, wp, taps_dydx, in_step, pad_mode, y_view, OH, OW, out_step=1, out_oy=0, out_ox=0,
                     bias=None, scale=None, shift=None, relu=0, accumulate=0, stats=None, stats_cnt=None, part0=0, cfg=0, **fold):
        kw = dict(dtype=self.dt, x=x_view, w=wp, ntaps=len(taps_dydx), tap_dy=[t[0] for t in taps_dydx], tap_dx=[t[1] for t in taps_dydx],
                  in_step=in_step, pad_mode=pad_mode, y=y_view, OH=OH, OW=OW, out_step=out_step, out_oy=out_oy, out_ox=out_ox,
                  bias=bias, scale=scale, shift=shift, relu=relu, accumulate=accumulate, stats=stats, stats_cnt=stats_cnt,
                  stats_part0=part0, cfg=cfg)
        stream = fold.pop('stream', None)
        kw.update(fold)
        kw.update(self._plane_args(x_view, y_view))
        if prog is self.bwd and not (cfg >> 8) and os.environ.get('SALT_CONV_WPX_BWD'):
            # A/B: workgroups per XCD of the whole-CU kernels (conv_ws / conv_ls) in BACKWARD, where they share the chip with the
            # weight-gradient stream (a whole-CU workgroup waits for a CU the other stream has left completely)
            kw['cfg'] = cfg | (int(os.environ['SALT_CONV_WPX_BWD']) << 8)
        return prog.add('conv', stream=stream, **kw)

    @staticmethod
    def _plane_args(x_view, y_view):
        """A view over a whole PLANAR buffer has cs < C (Act._view): the launch must name the plane stride (salt_conv_args.x_plane /
        y_plane; the library fails if its kernel for these arguments cannot address planes)."""
        kw = {}
        if not isinstance(y_view, tuple) and 0 < y_view.cs < y_view.C:
            kw['y_plane'] = y_view.B * y_view.H * y_view.W * y_view.cs
        if not isinstance(x_view, tuple) and 0 < x_view.cs < x_view.C:
            kw['x_plane'] = x_view.B * x_view.H * x_view.W * x_view.cs
        return kw

    def _conv_parts(self, x_view, taps_dydx, in_step, y_view, OH, OW, cfg=0):
        S = STRUCTS['salt_conv_args']()
        fill(S, dtype=self.dt, x=x_view, w=1, ntaps=len(taps_dydx), tap_dy=[t[0] for t in taps_dydx], tap_dx=[t[1] for t in taps_dydx],
             in_step=in_step, pad_mode=0, y=y_view, OH=OH, OW=OW, out_step=1, cfg=cfg, **self._plane_args(x_view, y_view))
        n = lib.salt_conv_stats_parts(ctypes.byref(S))
        if n < 0:
            raise SaltError('conv plan failed: ' + lib.salt_last_error().decode())
        return n

    # ------------------------------------------------------------------ BatchNorm plumbing
    @staticmethod
    def _fin_mode():
        """SALT_BN_FIN: how the train-mode BatchNorm sums travel.  2 (default): fp64 shard atomics in the producing launch, finalized
        by the consumer (salt_affine_act / the apply pass of salt_bn_bwd); 1: the same shards, finalized in the producing launch by the
        workgroup that draws the last ticket; 0: per-tile partials + separate finalize launches."""
        return int(os.environ.get('SALT_BN_FIN', '2'))

    @classmethod
    def _fin_on(cls):
        return cls._fin_mode() != 0

    def _fin_buffers(self, ndoubles):
        acc = self.alloc((ndoubles,), torch.float64)
        ticket = self.alloc((16,), torch.int32)
        return acc.data_ptr(), ticket.data_ptr()

    def _bn_train_fwd(self, y, bn, relu, res, out, nparts, stats, cnt, producer=None):
        """``producer``: the argument struct of the ONE convolution launch that writes y - it then finalizes the statistics itself
        (salt_conv_args.fin) and no bn_finalize operator is emitted."""
        w = self.engine.bn_work(bn)
        nbt = bn.num_batches_tracked.data_ptr() if bn.num_batches_tracked is not None else None
        fin = dict(C=bn.num_features, gamma=bn.weight.data_ptr(),
                   beta=bn.bias.data_ptr(), running_mean=bn.running_mean.data_ptr(), running_var=bn.running_var.data_ptr(),
                   num_batches_tracked=nbt, momentum=bn.momentum, eps=bn.eps, mean=w['mean'].data_ptr(), invstd=w['invstd'].data_ptr(),
                   scale=w['scale'].data_ptr(), shift=w['shift'].data_ptr())
        F = None
        if producer is not None:
            F = fill(STRUCTS['salt_bn_finalize_args'](), **fin)
            self.keep.append(F)
            if self._fin_mode() == 1:
                acc, ticket = self._fin_buffers(8 * (2 * bn.num_features + 1))
                self.fwd.set_fields(producer, fin=ctypes.addressof(F), fin_acc=acc, fin_ticket=ticket)
                F = None
        else:
            self.fwd.add('bn_finalize', stats=stats, stats_cnt=cnt, nparts=nparts, **fin)
        if res is not None and getattr(res, 'on_side', False):
            self.join()                          # the residual branch ran on the side stream
        out.buf.bn_train_out = (out.c0, out.C)      # dL/d(out) is consumed by this layer's bn_bwd only: it may take a per-image bias (scse)
        sa = self.fwd.add('affine_act', dtype=self.dt, y=y.view(), scale=w['scale'].data_ptr(), shift=w['shift'].data_ptr(),
                          res=res.view() if res is not None else null_view(), relu=int(relu), a=out.view())
        out.buf.act_op = (sa, out.buf.reads, out.c0, out.C)     # Graph.scse may take this operator over (the activation is then never stored)
        if F is not None:                        # the producer only adds to the shards; this operator finalizes them
            self.fwd.set_fields(sa, fin=ctypes.addressof(F))
            off = self._fin_slot('fwd', 8 * (2 * bn.num_features + 1), (producer, 'fin_acc'), (sa, 'fin_acc'))
            if (fwd_bn_fold() and res is None and relu and self.dtype == 'bf16' and out.c0 == 0 and out.C == out.buf.C
                    and self.fwd.streams[-1] == 0):
                # measurement switch: if the ONLY forward reader of `out` turns out to be one 3x3 convolution, that launch
                # applies this BatchNorm + ReLU in its loader (salt_conv_args.in_*) and this affine_act is dropped (build_backward)
                out.buf.lazy = dict(y=y, F=F, sa=sa, off=off, reads0=out.buf.reads, users=[], dropped=False)
                self._lazies = getattr(self, '_lazies', []) + [out.buf]
        return w

    def _bn_train_bwd(self, y, bn, relu, res, out, w):
        """emit backward of out = relu?(bn(y) (+res)); returns nothing, leaves grad in y.grad (and res.grad)."""
        hconv = getattr(out.buf, 'head_fused', None)
        if hconv is not None:
            # `out` was never stored: its only reader was the logit head (Graph.head took the affine_act over) - head gradients and this
            # BatchNorm's backward from (y, dlogits) alone
            if res is not None:
                raise SaltError('fused head behind a residual BatchNorm')
            C, Cout = bn.num_features, hconv.weight.shape[0]
            S = fill(STRUCTS['salt_head_bn_bwd_args'](), y=y.view())
            nparts = lib.salt_head_bn_bwd_parts(ctypes.byref(S))
            assert y.grad_state() == 0, 'conv output gradient has a single producer'
            sb = self.bwd.add('head_bn_bwd', dtype=self.dt, y=y.view(), relu=int(relu), mean=w['mean'].data_ptr(), invstd=w['invstd'].data_ptr(),
                              gamma=bn.weight.data_ptr(), beta=bn.bias.data_ptr(), w=hconv.weight.data_ptr(), Cout=Cout, dy_nchw=self.dlogits.data_ptr(),
                              partials=Scratch('head', nparts * Cout * (C + 1) * 4), nparts=nparts, gw=self._gp(hconv.weight),
                              gb=self._gp(hconv.bias) if hconv.bias is not None else None, dgamma=self._gp(bn.weight), dbeta=self._gp(bn.bias),
                              coef=self.f32(3 * C).data_ptr(), dy=y.gview())
            self._fin_slot('bwd', 8 * 2 * C, (sb, 'fin_acc'))
            return
        S = STRUCTS['salt_bn_bwd_args']()
        fill(S, y=y.view())
        nparts = lib.salt_bn_bwd_parts(ctypes.byref(S))
        C = bn.num_features
        gw, gb = self._gp(bn.weight), self._gp(bn.bias)
        coef = self.f32(3 * C)
        dres = null_view()
        acc_res = 0
        if res is not None:
            acc_res = res.grad_state()
            dres = res.gview()
        acc_y = y.grad_state()
        assert acc_y == 0, 'conv output gradient has a single producer'
        partials, ready = Scratch('bn_bwd', nparts * 2 * C * 4), 0
        wr = out.buf.grad_writers
        # (round 6: only the LAST writer has to cover exactly this slice - earlier writers of the same buffer, e.g. the decoder's data
        #  gradient over the whole concat buffer an encoder output lives in, are complete before it runs.  SALT_BNB_STRICT=1: round 5's rule)
        same = (lambda w_: (w_[0], w_[1]) == (out.c0, out.C))
        ok_wr = bool(wr) and same(wr[-1]) and (all(same(w_) for w_ in wr) or not os.environ.get('SALT_BNB_STRICT'))
        sec = wr[-1][2] if (ok_wr and isinstance(wr[-1][2], tuple) and wr[-1][2][0] == 'sec') else None
        if sec is not None and not relu and res is None and self._fin_mode() == 2:
            # round 6: dL/d(out) is the dres the main branch's bn_bwd wrote (this is a projection shortcut's BatchNorm): that apply pass
            # took this layer's sums as well (salt_bn_bwd_args.sec_*) - apply-only here
            partials, ready, producer = None, 3, sec[1]
            self.bwd.set_fields(producer, sec_y=y.view(), sec_mean=w['mean'].data_ptr(), sec_invstd=w['invstd'].data_ptr())
        elif sec is not None:
            pass                                   # (a shape the secondary sums do not cover: this layer keeps its reduction pass)
        elif (ok_wr and wr[-1][2] is not None and C % self.ve == 0
                and (res is None or not os.environ.get('SALT_NO_BNB_RES')) and not os.environ.get('SALT_NO_BNB_FUSE')):
            # the LAST writer of dL/d(out) is a plain data-gradient launch (it completes the gradient: earlier writers of the same
            # slice were accumulated): its epilogue also reduces this layer's BatchNorm-backward sums over its pixel tiles
            # (salt_conv_args.bnb_*), and bn_bwd skips its own pass over da and y.  With a residual the mask comes from `out`.
            s = wr[-1][2]
            nparts = lib.salt_conv_stats_parts(ctypes.byref(s))
            if nparts < 1:
                raise SaltError('conv plan failed: ' + lib.salt_last_error().decode())
            self._n_bnb = getattr(self, '_n_bnb', 0) + 1
            if self._fin_mode() == 2:
                partials, ready = None, 3          # the launch adds the sums to fp64 shards; the apply pass of bn_bwd finalizes them
            elif self._fin_mode() == 1:
                partials, ready = None, 2          # the launch also finalizes (salt_conv_args.bnb_fin): bn_bwd is the apply pass only
            else:
                partials, ready = Scratch('bnb%d' % self._n_bnb, nparts * 2 * C * 4), 1
            self.bwd.set_fields(s, bnb_y=y.view(), bnb_mean=w['mean'].data_ptr(), bnb_invstd=w['invstd'].data_ptr(), bnb_gamma=bn.weight.data_ptr(),
                                bnb_beta=bn.bias.data_ptr(), bnb_partials=partials, bnb_relu=int(relu),
                                bnb_a=out.view() if (relu and res is not None) else null_view())
            producer = s
        # without a residual a = relu(y*scale + shift): the kernel recomputes the mask from y and never reads `a`
        s2 = self.bwd.add('bn_bwd', dtype=self.dt, da=out.gview(), a=out.view() if (relu and res is not None) else null_view(), y=y.view(), relu=int(relu),
                          mean=w['mean'].data_ptr(), invstd=w['invstd'].data_ptr(), gamma=bn.weight.data_ptr(), beta=bn.bias.data_ptr(),
                          partials=partials, nparts=nparts, dgamma=gw, dbeta=gb, accumulate_param_grads=0,
                          coef=coef.data_ptr(), dy=y.gview(), dres=dres, accumulate_dres=acc_res, partials_ready=ready)
        bias = getattr(out.buf, 'grad_bias', None)
        if bias is not None:
            # (c0, C, dgap): the scSE backward left the channel-SE term out of dL/d(out); THIS bn_bwd - the slice's only consumer - adds it
            if (bias[0], bias[1]) != (out.c0, out.C):
                raise SaltError('gradient bias of %s covers channels [%d, %d), this BatchNorm backward reads [%d, %d)'
                                % (out.buf.name, bias[0], bias[0] + bias[1], out.c0, out.c0 + out.C))
            sums = bias[3] if len(bias) > 3 else None       # the scSE backward also took this layer's BatchNorm-backward sums (salt_scse_bwd_args.bnb_acc)
            if ready != 0:
                raise SaltError('a per-image gradient bias needs the reduction pass of bn_bwd (the producer of dL/da is not a convolution)')
            if sums is not None:
                ready, producer = 3, sums
                self.bwd.set_fields(s2, partials=None, partials_ready=3)
                self.bwd.patches = [q for q in self.bwd.patches if q[0] is not s2 or q[1] != ('partials',)]
                self.bwd.set_fields(sums, bn_mean=w['mean'].data_ptr(), bn_invstd=w['invstd'].data_ptr())
            self.bwd.set_fields(s2, da_bias=bias[2].data_ptr())
            out.buf.grad_bias = None
        if res is not None and acc_res == 0 and ready == 3 and self._fin_mode() == 2 and C % self.ve == 0 and 256 % (C // self.ve) == 0 \
                and getattr(res.buf, 'bn_train_out', None) == (res.c0, res.C) and bias is None and not res.plane_stride() \
                and os.environ.get('SALT_BNB_SEC', '1') != '0':
            res.buf.grad_writers[-1][2] = ('sec', s2)          # the residual's own BatchNorm (a projection shortcut) may ask this pass for its sums
        if sec is not None and ready == 3 and producer is sec[1]:
            self._fin_slot('bwd', 8 * 2 * C, (producer, 'sec_acc'), (s2, 'fin_acc'))
        elif ready == 3:
            peers = [(bias[4], 'bnb_acc')] if (bias is not None and len(bias) > 4 and bias[4] is not None) else []   # scse_fc_grads reads the same 6 C + 1 wide rows
            self._fin_slot('bwd', 8 * 2 * C, (producer, 'bnb_acc'), (s2, 'fin_acc'), *peers)
        elif ready == 2:
            acc, ticket = self._fin_buffers(8 * 2 * C)
            self.bwd.set_fields(producer, bnb_fin=ctypes.addressof(s2), bnb_acc=acc, bnb_ticket=ticket)
        elif ready == 0 and self._fin_mode() == 2:
            self._fin_slot('bwd', 8 * 2 * C, (s2, 'fin_acc'))       # reduction pass -> shards -> apply pass
        elif ready == 0 and self._fin_mode() == 1:
            acc, ticket = self._fin_buffers(8 * 2 * C)       # the reduction pass of bn_bwd finalizes in-launch
            self.bwd.set_fields(s2, fin_acc=acc, fin_ticket=ticket)

    # ------------------------------------------------------------------ dense convolution (+BN +ReLU +residual)
    def conv(self, x, conv, bn=None, relu=False, res=None, out=None, replicate=False, name=''):
        """nn.Conv2d (k in {1,3} or (k,1)/(1,k); stride 1|2; zero pad or the reference's replicate top/right pad)
        optionally followed by BatchNorm2d, residual add and ReLU."""
        eng = self.engine
        Cout, Cin, KH, KW = conv.weight.shape
        assert Cin == x.C, (name, Cin, x.C)
        stride = conv.stride[0]
        if replicate:
            assert stride == 1
            taps = [(kh, kw, kh - (KH - 1), kw) for kh in range(KH) for kw in range(KW)]
            pad_mode = 1
            OH, OW = x.H, x.W
        else:
            py, px = conv.padding
            taps = conv_taps(KH, KW, py, px)
            pad_mode = 0
            OH = (x.H + 2 * py - KH) // stride + 1
            OW = (x.W + 2 * px - KW) // stride + 1
        tk = [(t[0], t[1]) for t in taps]
        td = [(t[2], t[3]) for t in taps]
        pk = eng.packed(conv, tk, transposed=False)
        bias = conv.bias.data_ptr() if conv.bias is not None else None
        if out is None:
            out = self.new_act(x.B, OH, OW, Cout, name)
        assert (out.B, out.H, out.W, out.C) == (x.B, OH, OW, Cout), name
        y = None
        w = None
        lz = getattr(x.buf, 'lazy', None) if (self.train and x.c0 == 0 and x.C == x.buf.C and len(taps) == 9 and stride == 1) else None
        if bn is not None and self.train:
            y = self.new_act(x.B, OH, OW, Cout, name + '.y')
            if lz is not None and self._fin_on():
                # candidate for the loader fold: read the producer's raw output and transform it on the way into LDS
                nparts = self._conv_parts(lz['y'].view(), td, stride, y.view(), OH, OW)
                S = fill(STRUCTS['salt_conv_args'](), dtype=self.dt, x=lz['y'].view(), w=1, ntaps=9, tap_dy=[t[0] for t in td], tap_dx=[t[1] for t in td],
                         in_step=1, pad_mode=pad_mode, y=y.view(), OH=OH, OW=OW, out_step=1, in_fin=ctypes.addressof(lz['F']), in_fin_acc=8, in_relu=1)
                if lib.salt_conv_kernel_id(ctypes.byref(S)) in (1, 2, 3, 4):
                    prod = self._conv_launch(self.fwd, lz['y'].view(), pk.data_ptr(), td, stride, pad_mode, y.view(), OH, OW, bias=bias,
                                             in_fin=ctypes.addressof(lz['F']), in_relu=1)
                    self._fin_patches.append((prod, 'in_fin_acc', 'fwd', lz['off']))
                    lz['users'].append(prod)
                else:
                    lz = None
                    prod = self._conv_launch(self.fwd, x.view(), pk.data_ptr(), td, stride, pad_mode, y.view(), OH, OW, bias=bias)
                w = self._bn_train_fwd(y, bn, relu, res, out, nparts, None, None, producer=prod)
            elif self._fin_on():
                lz = None
                nparts = self._conv_parts(x.view(), td, stride, y.view(), OH, OW)
                prod = self._conv_launch(self.fwd, x.view(), pk.data_ptr(), td, stride, pad_mode, y.view(), OH, OW, bias=bias)
                w = self._bn_train_fwd(y, bn, relu, res, out, nparts, None, None, producer=prod)
            else:
                lz = None
                nparts = self._conv_parts(x.view(), td, stride, y.view(), OH, OW)
                stats, cnt = Scratch('stats' + self._scratch_sfx, 4 * lib.salt_bn_stats_floats(nparts, Cout)), Scratch('stats_cnt' + self._scratch_sfx, nparts * 4)
                self._conv_launch(self.fwd, x.view(), pk.data_ptr(), td, stride, pad_mode, y.view(), OH, OW, bias=bias, stats=stats, stats_cnt=cnt)
                w = self._bn_train_fwd(y, bn, relu, res, out, nparts, stats, cnt)
        elif bn is not None:
            w = eng.bn_work(bn)
            if res is None:
                self._conv_launch(self.fwd, x.view(), pk.data_ptr(), td, stride, pad_mode, out.view(), OH, OW, bias=bias,
                                  scale=w['scale'].data_ptr(), shift=w['shift'].data_ptr(), relu=int(relu))
            elif Cout % self.ve == 0 and not os.environ.get('SALT_NO_RES_FOLD'):
                # eval-mode residual block: folded BN + identity add + ReLU all in the convolution's epilogue (salt_conv_args.res) - the
                # same values as the separate affine_act pass below, bit for bit, without its launch and its trip over `out`
                if getattr(res, 'on_side', False):
                    self.join()
                self._conv_launch(self.fwd, x.view(), pk.data_ptr(), td, stride, pad_mode, out.view(), OH, OW, bias=bias,
                                  scale=w['scale'].data_ptr(), shift=w['shift'].data_ptr(), relu=int(relu), res=res.view())
            else:
                self._conv_launch(self.fwd, x.view(), pk.data_ptr(), td, stride, pad_mode, out.view(), OH, OW, bias=bias,
                                  scale=w['scale'].data_ptr(), shift=w['shift'].data_ptr(), relu=0)
                if getattr(res, 'on_side', False):
                    self.join()
                self.fwd.add('affine_act', dtype=self.dt, y=out.view(), scale=None, shift=None, res=res.view(), relu=int(relu), a=out.view())
        else:
            assert res is None
            self._conv_launch(self.fwd, x.view(), pk.data_ptr(), td, stride, pad_mode, out.view(), OH, OW, bias=bias, relu=int(relu))

        out.on_side = self.fwd.default_stream == 1
        if self.train:
            def backward():
                if bn is not None:
                    self._bn_train_bwd(y, bn, relu, res, out, w)
                    dy = y
                else:
                    if conv.bias is not None:
                        raise NotImplementedError('bias gradient of a dense conv without BatchNorm (not on the reference path)')
                    dy = out
                    if relu:
                        self.bwd.add('relu_bwd', dtype=self.dt, da=out.gview(), a=out.view(), dy=out.gview(), accumulate=0)
                # weight gradient: P = dY (a = cout), Q = X (b = cin)
                if lz is not None and lz['dropped']:
                    # the activation was never materialised and the weight-gradient kernels (LDS-DMA loaders) cannot transform their
                    # operand: a graph built under SALT_FWD_BN_FOLD has no backward pass (fail loudly, never a wrong gradient)
                    raise SaltError('SALT_FWD_BN_FOLD graphs are forward-only (the folded activation of %s does not exist for the weight gradient)' % name)
                xq = x
                self._wgrad(dy.gview(), xq.view(), td, tk, stride, pad_mode, conv.weight, KH, KW)
                # data gradient
                if x.buf.name != '__input__':
                    self._dgrad(conv, x, dy, taps, stride, replicate, KH, KW)
            self.tape.append(backward)
        return out

    # ------------------------------------------------------------------ factored hypercolumn (saltnet.h: salt_hyper_stencil)
    def hyper_factor_ok(self, B, H, W, C, Rs):
        """Can the levels up-sampled by ``Rs`` leave the hypercolumn (architectures/unet.py:101-109) and enter the final Conv2dBnRelu
        through the low-resolution tap GEMM + stencil instead?  (the stencil's shape rules, saltnet.h; consumer-side BatchNorm shards)"""
        if not Rs or C % 8 or (self.train and self._fin_mode() != 2) or (self.train and W > 256):
            return False
        if C % (16 if self.dtype == 'f32' else 32):
            return False                                # the transposed tap-GEMM pack places whole chunks
        return all(R >= 4 and R & (R - 1) == 0 and R <= 32 and H % R == 0 and W % R == 0 for R in Rs)

    def hyper_head_ok(self, C, hconv):
        """Can salt_hyper_stencil's epilogue apply the logit head ``hconv`` (eval; saltnet.h: 1..2 classes, 1 / 2 / 4 / 8 channel blocks)?"""
        if self.train or os.environ.get('SALT_HYPER_HEAD', '1') == '0':
            return False
        co, ci, kh, kw = hconv.weight.shape
        return (kh, kw) == (1, 1) and ci == C and 1 <= co <= 2 and C in ((64, 128, 256) if self.dtype == 'f32' else (64, 128, 256, 512))

    def head_bn_ok(self, C, hconv):
        """TRAIN mode: can the final block's BatchNorm apply + ReLU and the logit head ``hconv`` run as one pass over the raw convolution
        output (salt_head_bn / salt_head_bn_bwd; saltnet.h: <= 4 classes, C a power-of-two number of 16-byte pieces <= 64, consumer-side
        statistics shards)?  SALT_HEAD_BN=0 keeps the separate launches (A/B)."""
        if not self.train or self._fin_mode() != 2 or os.environ.get('SALT_HEAD_BN', '1') == '0':
            return False
        co, ci, kh, kw = hconv.weight.shape
        cpv = C // self.ve
        return (kh, kw) == (1, 1) and ci == C and 1 <= co <= 4 and C % self.ve == 0 and 1 <= cpv <= 64 and cpv & (cpv - 1) == 0

    def _head_bn(self, y, bn, relu, hconv, logits, producer):
        """forward salt_head_bn behind ``producer`` (which adds y's statistics to the fp64 shards) + the tape entry of salt_head_bn_bwd,
        which leaves dL/dy in y.grad: returned through ``self._head_bn_bwd`` - the caller's own backward closure calls it first."""
        eng = self.engine
        C = bn.num_features
        w = eng.bn_work(bn)
        nbt = bn.num_batches_tracked.data_ptr() if bn.num_batches_tracked is not None else None
        F = fill(STRUCTS['salt_bn_finalize_args'](), C=C, gamma=bn.weight.data_ptr(), beta=bn.bias.data_ptr(), running_mean=bn.running_mean.data_ptr(),
                 running_var=bn.running_var.data_ptr(), num_batches_tracked=nbt, momentum=bn.momentum, eps=bn.eps, mean=w['mean'].data_ptr(),
                 invstd=w['invstd'].data_ptr(), scale=w['scale'].data_ptr(), shift=w['shift'].data_ptr())
        self.keep.append(F)
        Cout = hconv.weight.shape[0]
        hb = self.fwd.add('head_bn', dtype=self.dt, y=y.view(), fin=ctypes.addressof(F), relu=int(relu), w=hconv.weight.data_ptr(),
                          bias=hconv.bias.data_ptr() if hconv.bias is not None else None, Cout=Cout, y_nchw=logits.data_ptr())
        self._fin_slot('fwd', 8 * (2 * C + 1), (producer, 'fin_acc'), (hb, 'fin_acc'))
        self.dlogits = self.alloc(tuple(logits.shape), torch.float32)

        def backward():
            S = fill(STRUCTS['salt_head_bn_bwd_args'](), y=y.view())
            nparts = lib.salt_head_bn_bwd_parts(ctypes.byref(S))
            assert y.grad_state() == 0, 'conv output gradient has a single producer'
            coef = self.f32(3 * C)
            sb = self.bwd.add('head_bn_bwd', dtype=self.dt, y=y.view(), relu=int(relu), mean=w['mean'].data_ptr(), invstd=w['invstd'].data_ptr(),
                              gamma=bn.weight.data_ptr(), beta=bn.bias.data_ptr(), w=hconv.weight.data_ptr(), Cout=Cout, dy_nchw=self.dlogits.data_ptr(),
                              partials=Scratch('head', nparts * Cout * (C + 1) * 4), nparts=nparts, gw=self._gp(hconv.weight),
                              gb=self._gp(hconv.bias) if hconv.bias is not None else None, dgamma=self._gp(bn.weight), dbeta=self._gp(bn.bias),
                              coef=coef.data_ptr(), dy=y.gview())
            self._fin_slot('bwd', 8 * 2 * C, (sb, 'fin_acc'))
        self._head_bn_bwd = backward
        return w

    def hyper_level(self, x, conv, c0, name='hyper.z'):
        """z = [W[:, c0 : c0 + x.C, kh, kw]]_taps x  at x's (LOW) resolution: one 1x1 convolution x.C -> 9 Cout whose output channel
        t Cout + o is tap t of output channel o of the 3x3 convolution ``conv`` restricted to the input channels of this level.
        Backward (from dz, which conv_hyper's stencil adjoint writes): the 1x1 weight gradient scattered into conv.weight.grad and
        the 1x1 data gradient into dL/dx."""
        eng = self.engine
        Cout, _, KH, KW = conv.weight.shape
        taps = [(kh, kw) for kh in range(KH) for kw in range(KW)]
        pk = eng.packed_tapgemm(conv, c0, x.C, transposed=False)
        z = self.new_act(x.B, x.H, x.W, len(taps) * Cout, name)
        # cfg 11 | 1 << 16: ask for conv1x1_ls_kernel with 32-channel items wherever it applies - the heuristic sends these few-pixel
        # launches to conv_mfma_kernel's 128 x 32 tiles (a serial chain of K / 32 staged chunks per workgroup: 30 us for 0.15 GFLOP)
        # (with thousands of 64-channel items - the 256 x 256 inference shapes - 64-channel items halve the re-reads of x instead)
        items64 = (x.B * x.H * x.W // 256) * (len(taps) * Cout // 64)
        ask = 11 | ((1 << 16) if items64 < 2048 else 0)
        self._conv_launch(self.fwd, x.view(), pk.data_ptr(), [(0, 0)], 1, 0, z.view(), x.H, x.W, cfg=ask)
        z.on_side = self.fwd.default_stream == 1
        if self.train:
            D1 = conv.weight.shape[1]

            def backward():
                if not z.grad_ready():
                    raise SaltError('hyper_level: dL/dz was never written (conv_hyper must consume %s)' % name)
                self._wgrad(z.gview(), x.view(), [(0, 0)], None, 1, 0, conv.weight, KH, KW, b_slice=(c0, D1), tapgemm=(Cout, taps))
                pkt = eng.packed_tapgemm(conv, c0, x.C, transposed=True, bwd=True)
                acc = x.grad_state()
                self._conv_launch(self.bwd, z.gview(), pkt.data_ptr(), [(0, 0)], 1, 0, x.gview(), x.H, x.W, accumulate=acc, stream=self._bwd_pack_tag(), cfg=ask)
            self.tape.append(backward)
        return z

    def conv_hyper(self, x, zs, Rs, conv, bn, relu=True, out=None, name='final', head=None):
        """Conv2dBnRelu over the hypercolumn (architectures/unet.py:84-87,101-109; base.py:21-37) with the levels of ``zs`` factored out:
        y = conv3x3_replicate(x; W[:, :x.C]) + bias, then salt_hyper_stencil adds sum_k stencil_k(z_k) (and takes the BatchNorm
        statistics of the sum in training / applies the folded BatchNorm + ReLU in eval); x holds the full-resolution channels only.
        ``head`` = (nn.Conv2d(Cout, classes, 1), fp32 NCHW logits), eval only: the block's only consumer in the reference's ``final``
        Sequential (architectures/unet.py:84-87) is applied by the stencil's epilogue (salt_hyper_stencil_args.head_*; same values as
        salt_head1x1 on the stored activation, which is then never written) and None is returned."""
        eng = self.engine
        Cout, Cin, KH, KW = conv.weight.shape
        assert (KH, KW) == (3, 3) and bn is not None and x.C <= Cin and all(z.C == 9 * Cout for z in zs)
        taps = [(kh, kw, kh - (KH - 1), kw) for kh in range(KH) for kw in range(KW)]
        tk = [(t[0], t[1]) for t in taps]
        td = [(t[2], t[3]) for t in taps]
        d1 = (0, x.C)
        pk = eng.packed(conv, tk, transposed=False, d1=d1)
        bias = conv.bias.data_ptr() if conv.bias is not None else None
        if head is not None and not (self.head_bn_ok(Cout, head[0]) if self.train else self.hyper_head_ok(Cout, head[0])):
            raise SaltError('conv_hyper: this head cannot be fused (ask hyper_head_ok / head_bn_ok first)')
        if out is None and head is None:
            out = self.new_act(x.B, x.H, x.W, Cout, name)
        ac = int(bool(getattr(getattr(self.engine, 'module', None), 'align_corners', False)))
        y = self.new_act(x.B, x.H, x.W, Cout, name + '.y')
        self._conv_launch(self.fwd, x.view(), pk.data_ptr(), td, 1, 1, y.view(), x.H, x.W, bias=bias)
        if any(getattr(z, 'on_side', False) for z in zs):
            self.join()
        st = dict(dtype=self.dt, nlev=len(zs), z=[z.view() for z in zs], R=list(Rs), y_in=y.view(), backward=0, align_corners=ac)
        if self.train and head is not None:
            # round 6: BatchNorm apply + ReLU + logit head in ONE pass over the raw sum (salt_head_bn): the block's activation and its
            # gradient are never stored.  Backward: salt_head_bn_bwd writes dL/dy directly (head gradients + BatchNorm-backward sums in
            # its first pass), then the stencil adjoint / weight gradient / data gradient as below.
            prod = self.fwd.add('hyper_stencil', y=y.view(), **st)
            w = self._head_bn(y, bn, relu, head[0], head[1], prod)
            zs_, Rs_ = list(zs), list(Rs)

            hb_bwd = self._head_bn_bwd

            def backward():
                hb_bwd()
                for z in zs_:
                    assert z.grad_state() == 0
                self.bwd.add('hyper_stencil', dtype=self.dt, nlev=len(zs_), z=[z.gview() for z in zs_], R=Rs_, y_in=null_view(), y=y.gview(),
                             backward=1, align_corners=ac)
                self._wgrad(y.gview(), x.view(), td, tk, 1, 1, conv.weight, KH, KW, b_slice=(0, Cin))
                self._dgrad(conv, x, y, taps, 1, True, KH, KW, d1=d1)
            self.tape.append(backward)
            return None
        if self.train:
            prod = self.fwd.add('hyper_stencil', y=y.view(), **st)
            w = self._bn_train_fwd(y, bn, relu, None, out, 0, None, None, producer=prod)
        elif head is not None:
            w = eng.bn_work(bn)
            hconv, logits = head
            yv = y.view()
            yv.p = None                                  # shape only: the activated block output is never stored
            self.fwd.add('hyper_stencil', y=yv, scale=w['scale'].data_ptr(), shift=w['shift'].data_ptr(), relu=int(relu),
                         head_w=hconv.weight.data_ptr(), head_b=hconv.bias.data_ptr() if hconv.bias is not None else None,
                         head_y_nchw=logits.data_ptr(), head_cout=hconv.weight.shape[0],
                         head_ws=self.alloc((x.B * (Cout // 64) * hconv.weight.shape[0] * x.H * x.W,), torch.float32).data_ptr() if Cout > 64 else None, **st)
            return None
        else:
            w = eng.bn_work(bn)
            self.fwd.add('hyper_stencil', y=out.view(), scale=w['scale'].data_ptr(), shift=w['shift'].data_ptr(), relu=int(relu), **st)
        out.on_side = False
        if self.train:
            def backward():
                self._bn_train_bwd(y, bn, relu, None, out, w)
                for z in zs:
                    assert z.grad_state() == 0
                self.bwd.add('hyper_stencil', dtype=self.dt, nlev=len(zs), z=[z.gview() for z in zs], R=list(Rs), y_in=null_view(), y=y.gview(),
                             backward=1, align_corners=ac)
                self._wgrad(y.gview(), x.view(), td, tk, 1, 1, conv.weight, KH, KW, b_slice=(0, Cin))
                self._dgrad(conv, x, y, taps, 1, True, KH, KW, d1=d1)
            self.tape.append(backward)
        return out

    def _wgrad(self, p_view, q_view, taps_dydx, taps_khkw, q_step, pad_mode, weight, KH, KW, b_slice=None, tapgemm=None):
        """dW = sum_p P[p,:]^T Q[p*q_step + tap, :] -> weight.grad (reference layout [Ca][Cb][KH][KW]).
        ``b_slice`` = (first, row stride): Q covers only channels [first, first + Cb) of the weight's second axis (salt_wgrad_reduce_args.ldb).
        ``tapgemm`` = (rows, [(kh, kw)]): a 1x1 launch whose P channels are t * rows + a (salt_wgrad_reduce_args.a_mod) - Graph.hyper_level."""
        batching = self._reduce_batching() > 0
        aux = not batching and os.environ.get('SALT_REDUCE_AUX', '0') != '0'
        gw = self.engine.grad_ptr(weight) if batching else self._gp(weight)
        goff = 0
        Ca, Cb = p_view.C, q_view.C
        extra = {}
        if b_slice is not None:
            goff = 4 * b_slice[0] * KH * KW
            gw += goff
            extra['ldb'] = b_slice[1]
        if tapgemm is not None:
            assert len(taps_dydx) == 1 and Ca == tapgemm[0] * len(tapgemm[1])
            extra['a_mod'] = tapgemm[0]
            taps_khkw = list(tapgemm[1])
        first = True
        for i in range(0, len(taps_dydx), 9 if len(taps_dydx) != 16 else 4):
            chunk = list(range(i, min(i + (9 if len(taps_dydx) != 16 else 4), len(taps_dydx))))
            S = STRUCTS['salt_conv_wgrad_args']()
            qp = q_view.B * q_view.H * q_view.W * q_view.cs if 0 < q_view.cs < q_view.C else 0       # planar Q (Act._view of a whole planar buffer)
            fill(S, dtype=self.dt, p=p_view, q=q_view, ntaps=len(chunk), tap_dy=[taps_dydx[j][0] for j in chunk],
                 tap_dx=[taps_dydx[j][1] for j in chunk], q_step=q_step, pad_mode=pad_mode, q_plane=qp)
            ns = lib.salt_conv_wgrad_nsplit(ctypes.byref(S))
            if ns < 0:
                raise SaltError('wgrad plan failed: ' + lib.salt_last_error().decode())
            nbytes = ns * len(chunk) * Ca * Cb * 4
            # two alternating slab workspaces: the reduction of pair k runs on the auxiliary stream beside conv_wgrad k + 1 (stream tag 4,
            # runtime.hip); both stay in the Infinity Cache (2 x <= 25 MB)
            slab = 'wgrad'
            if aux:
                slab = 'wgrad_' + 'ab'[self._n_slab & 1]
                self._n_slab += 1
            wg_op = self.bwd.add('conv_wgrad', stream=1, dtype=self.dt, p=p_view, q=q_view, ntaps=len(chunk), tap_dy=[taps_dydx[j][0] for j in chunk],
                                 tap_dx=[taps_dydx[j][1] for j in chunk], q_step=q_step, pad_mode=pad_mode,
                                 partials=None if batching else Scratch(slab, nbytes), nsplit=ns, q_plane=qp)
            rt = range(len(taps_khkw)) if tapgemm is not None else chunk
            rfields = dict(nsplit=ns, ntaps=len(chunk), Ca=Ca, Cb=Cb, KH=KH, KW=KW, tap_kh=[taps_khkw[j][0] for j in rt],
                           tap_kw=[taps_khkw[j][1] for j in rt], accumulate=0, **extra)
            if batching:
                sc = self._queue_reduce(dict(rfields, grad_off=goff), weight, nbytes)
                self.bwd.set_fields(wg_op, partials=sc)
            else:
                self.bwd.add('wgrad_reduce', stream=4 if aux else 1, partials=Scratch(slab, nbytes), grad=gw, **rfields)
            first = False

    def _bwd_pack_tag(self):
        """Stream tag of a data-gradient convolution: the first one of the backward program joins the side stream, where the
        data-gradient weight packs of this step were enqueued during forward (Engine.refresh)."""
        self._uses_bwd_packs = True
        return None

    def _dgrad(self, conv, x, dy, taps, stride, replicate, KH, KW, d1=None):
        eng = self.engine
        tk = [(t[0], t[1]) for t in taps]
        pk = eng.packed(conv, tk, transposed=True, bwd=True, d1=d1)       # n = cin, c = cout
        if replicate:
            # gradient w.r.t. the replicate-padded input on the extended domain, then fold the pad back
            top, right = KH - 1, KW - 1
            Hp, Wp = x.H + top, x.W + right
            td = [(-(t[2]) - top, -(t[3])) for t in taps]
            acc = x.grad_state()
            if os.environ.get('SALT_FOLD_FULL'):            # A/B: gradient on the extended grid in scratch, then a full fold pass
                ext = shaped_view(0, x.B, Hp, Wp, x.C, _round_up(x.C, self.ve))
                nbytes = x.B * Hp * Wp * ext.cs * self._es()
                self._conv_launch(self.bwd, dy.gview(), pk.data_ptr(), td, 1, 0, (ext, Scratch('dgrad_ext', nbytes)), Hp, Wp, stream=self._bwd_pack_tag())
                self.bwd.add('pad_fold', dtype=self.dt, xp=(ext, Scratch('dgrad_ext', nbytes)), top=top, bottom=0, left=0, right=right,
                             x=x.gview(), accumulate=acc)
                return
            if x.C % self.ve == 0 and not os.environ.get('SALT_FOLD_STRIP'):
                # fused fold: the launch tiles the extended grid so that every pad-ring pixel shares a tile with the edge pixel it folds
                # onto; the epilogue sums them from the staged tile - no strip, no second pass, and (a plain store of the complete
                # gradient) the launch can carry the BatchNorm-backward sums of x's producer
                s = self._conv_launch(self.bwd, dy.gview(), pk.data_ptr(), td, 1, 0, x.gview(), Hp, Wp, accumulate=acc,
                                      fold_top=top, fold_right=right, stream=self._bwd_pack_tag())
                if not x.plane_stride():               # (a planar dL/dx has one dense consumer per plane: nothing to ride along)
                    x.buf.grad_writers[-1][2] = s
                return
            # strip fold: interior pixels go straight into x.grad, only the pad ring takes the detour through scratch
            scs = _round_up(x.C, self.ve)
            ring = lib.salt_fold_strip_pixels(x.H, x.W, top, 0, 0, right)
            strip = Scratch('dgrad_ring', x.B * ring * scs * self._es())
            self._conv_launch(self.bwd, dy.gview(), pk.data_ptr(), td, 1, 0, x.gview(), Hp, Wp, accumulate=acc,
                              strip=strip, strip_cs=scs, fold_top=top, fold_right=right, stream=self._bwd_pack_tag())
            self.bwd.add('pad_fold_strip', dtype=self.dt, strip=strip, strip_cs=scs, top=top, bottom=0, left=0, right=right, x=x.gview())
            return
        acc = x.grad_state()
        if stride == 1:
            td = [(-t[2], -t[3]) for t in taps]
            s = self._conv_launch(self.bwd, dy.gview(), pk.data_ptr(), td, 1, 0, x.gview(), x.H, x.W, accumulate=acc, stream=self._bwd_pack_tag())
            if not x.plane_stride():
                x.buf.grad_writers[-1][2] = s      # a plain full-grid launch: can carry the BatchNorm-backward sums of x's producer
            return
        # stride 2: the data gradient is a transposed convolution - four output-parity phases
        phases = []
        for ay in range(2):
            for ax in range(2):
                sel = [i for i, t in enumerate(taps) if (t[2] - ay) % 2 == 0 and (t[3] - ax) % 2 == 0]
                phases.append((ay, ax, sel))
        fused = self._phase_fused([[(((ay - taps[i][2]) // 2, (ax - taps[i][3]) // 2), tk[i]) for i in sel] for ay, ax, sel in phases], x.H, x.W)
        if fused is not None:
            # ONE launch for all four phases (each its own packed weight block; taps a phase lacks are zero weights)
            td_u, phase_taps = fused
            pk_t, elems = eng.packed_phases(conv, phase_taps, transposed=True, bwd=True)
            s = self._conv_launch(self.bwd, dy.gview(), pk_t.data_ptr(), td_u, 1, 0, x.gview(), x.H // 2, x.W // 2, out_step=2, accumulate=acc,
                                  nphase=4, w_phase_elems=elems, stream=self._bwd_pack_tag())
            if not x.plane_stride() and not os.environ.get('SALT_NO_BNB_PHASE'):
                x.buf.grad_writers[-1][2] = s      # all four parities of an even grid: every pixel once - can carry the BatchNorm-backward sums (round 6)
            return
        if any(not sel for _, _, sel in phases) and not acc:
            self.fill(x, 0.0, grad=True)
            acc = 1
        for ay, ax, sel in phases:
            oh, ow = (x.H - ay + 1) // 2, (x.W - ax + 1) // 2
            if not sel or oh <= 0 or ow <= 0:
                continue
            pk_s = eng.packed(conv, [tk[i] for i in sel], transposed=True, bwd=True)
            td = [((ay - taps[i][2]) // 2, (ax - taps[i][3]) // 2) for i in sel]
            self._conv_launch(self.bwd, dy.gview(), pk_s.data_ptr(), td, 1, 0, x.gview(), oh, ow, out_step=2, out_oy=ay, out_ox=ax, accumulate=acc,
                              stream=self._bwd_pack_tag())

    def _phase_fused(self, per_phase, H, W):
        """per_phase: for each of the 4 output-parity phases a list of ((dy, dx), (kh, kw)).  -> (unified tap offsets, per-phase
        (kh, kw) | None lists) when the phases can share one launch (even grid, <= 9 distinct offsets, at least two non-empty
        phases), else None."""
        if os.environ.get('SALT_NO_PHASE_FUSE') or H % 2 or W % 2:
            return None
        offs = sorted({o for ph in per_phase for o, _ in ph})
        if not offs or len(offs) > 9 or sum(1 for ph in per_phase if ph) < 2:
            return None
        # every phase runs ALL unified offsets (missing ones against zero weights): worth one launch instead of four only while the
        # padded work stays below ~2x (k3: 16 taps for 9 real ones; k4: 36 for 16 - not fused)
        if 4 * len(offs) > 1.8 * sum(len(ph) for ph in per_phase):
            return None
        if len(offs) not in (4, 9):               # the kernels unroll 4- and 9-tap loops; pad other counts with a repeated zero tap
            offs = offs + [offs[0]] * ((4 if len(offs) < 4 else 9) - len(offs))
        phase_taps = []
        for ph in per_phase:
            d = dict(ph)
            used = set()
            row = []
            for o in offs:
                if o in d and o not in used:
                    row.append(d[o]); used.add(o)
                else:
                    row.append(None)
            phase_taps.append(row)
        return offs, phase_taps

    def fill(self, act, value, grad=False):
        assert value == 0.0
        t = act.buf.grad() if grad else act.buf.t
        if act.c0 == 0 and act.C == act.buf.C:
            (self.bwd if grad else self.fwd).add('zero', p=t.data_ptr(), bytes=t.numel() * t.element_size())
        else:   # channel slice: y = a - a via affine (scale 0) keeps it one pass
            z = self.f32(2 * act.C)
            v = act.gview() if grad else act.view()
            (self.bwd if grad else self.fwd).add('affine_act', dtype=self.dt, y=v, scale=z.data_ptr(), shift=z.data_ptr() + 4 * act.C,
                                                 res=null_view(), relu=0, a=v)

    # ------------------------------------------------------------------ transposed convolution (k3 op1 / k4, stride 2, pad 1)
    def conv_transpose(self, x, deconv, bn=None, relu=False, out=None, name=''):
        eng = self.engine
        Cin, Cout, KH, KW = deconv.weight.shape
        assert Cin == x.C and deconv.stride[0] == 2 and deconv.padding[0] == 1 and KH == KW and KH in (3, 4)
        p = 1
        OH, OW = 2 * x.H, 2 * x.W
        bias = deconv.bias.data_ptr() if deconv.bias is not None else None
        if out is None:
            out = self.new_act(x.B, OH, OW, Cout, name)
        train_bn = bn is not None and self.train
        tgt = self.new_act(x.B, OH, OW, Cout, name + '.y') if train_bn else out
        w = eng.bn_work(bn) if bn is not None else None
        phases = []
        for fy in range(2):
            for fx in range(2):
                sel = [(u, v, (fy + p - u) // 2, (fx + p - v) // 2) for u in range(KH) for v in range(KW)
                       if (fy + p - u) % 2 == 0 and (fx + p - v) % 2 == 0]
                phases.append((fy, fx, sel))
        fused = self._phase_fused([[((s_[2], s_[3]), (s_[0], s_[1])) for s_ in sel] for _, _, sel in phases], OH, OW)
        total_parts, stats, cnt, fin_prod = 0, None, None, None
        if fused is not None:
            # ONE launch for the four output-parity phases of the transposed convolution
            td_u, phase_taps = fused
            pk_t, elems = eng.packed_phases(deconv, phase_taps, transposed=True)
            if train_bn:
                S = STRUCTS['salt_conv_args']()
                fill(S, dtype=self.dt, x=x.view(), w=1, ntaps=len(td_u), tap_dy=[t[0] for t in td_u], tap_dx=[t[1] for t in td_u], in_step=1, pad_mode=0,
                     y=tgt.view(), OH=x.H, OW=x.W, out_step=2, nphase=4, w_phase_elems=elems)
                total_parts = lib.salt_conv_stats_parts(ctypes.byref(S))
                if total_parts < 0:
                    raise SaltError('conv plan failed: ' + lib.salt_last_error().decode())
                if self._fin_on():
                    fin_prod = self._conv_launch(self.fwd, x.view(), pk_t.data_ptr(), td_u, 1, 0, tgt.view(), x.H, x.W, out_step=2, bias=bias,
                                                 nphase=4, w_phase_elems=elems)
                else:
                    stats = Scratch('stats', 4 * lib.salt_bn_stats_floats(total_parts, Cout))
                    cnt = Scratch('stats_cnt', total_parts * 4)
                    self._conv_launch(self.fwd, x.view(), pk_t.data_ptr(), td_u, 1, 0, tgt.view(), x.H, x.W, out_step=2, bias=bias, stats=stats, stats_cnt=cnt,
                                      nphase=4, w_phase_elems=elems)
            elif bn is not None:
                self._conv_launch(self.fwd, x.view(), pk_t.data_ptr(), td_u, 1, 0, tgt.view(), x.H, x.W, out_step=2, bias=bias,
                                  scale=w['scale'].data_ptr(), shift=w['shift'].data_ptr(), relu=int(relu), nphase=4, w_phase_elems=elems)
            else:
                self._conv_launch(self.fwd, x.view(), pk_t.data_ptr(), td_u, 1, 0, tgt.view(), x.H, x.W, out_step=2, bias=bias, relu=int(relu),
                                  nphase=4, w_phase_elems=elems)
            phases = []
        part0 = 0
        plans = []
        for fy, fx, sel in phases:
            td = [(s[2], s[3]) for s in sel]
            n = self._conv_parts(x.view(), td, 1, shaped_view(1, x.B, x.H, x.W, Cout), x.H, x.W) if train_bn else 0
            plans.append(n)
            total_parts += n
        if fused is None:
            stats = Scratch('stats', 4 * lib.salt_bn_stats_floats(total_parts, Cout)) if train_bn else None
            cnt = Scratch('stats_cnt', total_parts * 4) if train_bn else None
        for (fy, fx, sel), n in zip(phases, plans):
            pk = eng.packed(deconv, [(s[0], s[1]) for s in sel], transposed=True)       # n = cout (D1), c = cin (D0)
            td = [(s[2], s[3]) for s in sel]
            if train_bn:
                self._conv_launch(self.fwd, x.view(), pk.data_ptr(), td, 1, 0, tgt.view(), x.H, x.W, out_step=2, out_oy=fy, out_ox=fx,
                                  bias=bias, stats=stats, stats_cnt=cnt, part0=part0)
            elif bn is not None:
                self._conv_launch(self.fwd, x.view(), pk.data_ptr(), td, 1, 0, tgt.view(), x.H, x.W, out_step=2, out_oy=fy, out_ox=fx,
                                  bias=bias, scale=w['scale'].data_ptr(), shift=w['shift'].data_ptr(), relu=int(relu))
            else:
                self._conv_launch(self.fwd, x.view(), pk.data_ptr(), td, 1, 0, tgt.view(), x.H, x.W, out_step=2, out_oy=fy, out_ox=fx,
                                  bias=bias, relu=int(relu))
            part0 += n
        if train_bn:
            self._bn_train_fwd(tgt, bn, relu, None, out, total_parts, stats, cnt, producer=fin_prod)
        if self.train:
            def backward():
                if bn is None:
                    raise NotImplementedError('ConvTranspose2d without BatchNorm in training')
                self._bn_train_bwd(tgt, bn, relu, None, out, w)
                taps_all = [(u, v, u - p, v - p) for u in range(KH) for v in range(KW)]
                # weight gradient: P = X (a = cin), Q = dY (b = cout), dY sampled at 2i - p + u
                self._wgrad(x.view(), tgt.gview(), [(t[2], t[3]) for t in taps_all], [(t[0], t[1]) for t in taps_all], 2, 0,
                            deconv.weight, KH, KW)
                # data gradient: stride-2 convolution of dY with W[cin][cout] (n = cin, c = cout)
                pk = eng.packed(deconv, [(t[0], t[1]) for t in taps_all], transposed=False, bwd=True)
                acc = x.grad_state()
                self._conv_launch(self.bwd, tgt.gview(), pk.data_ptr(), [(t[2], t[3]) for t in taps_all], 2, 0, x.gview(), x.H, x.W, accumulate=acc,
                                  stream=self._bwd_pack_tag())
            self.tape.append(backward)
        return out

    # ------------------------------------------------------------------ first layer (fp32 NCHW input image)
    def conv_first(self, x_nchw, conv, bn, relu=True, name=''):
        eng = self.engine
        B, Cin, H, W = x_nchw.shape
        Cout, _, K, _ = conv.weight.shape
        stride, pad = conv.stride[0], conv.padding[0]
        if stride == 2 and K % 2 == 1 and pad == K // 2 and H % 2 == 0 and W % 2 == 0 and conv.bias is None and bn is not None:
            return self._stem_s2d(x_nchw, conv, bn, relu, name)
        OH = (H + 2 * pad - K) // stride + 1
        OW = (W + 2 * pad - K) // stride + 1
        out = self.new_act(B, OH, OW, Cout, name)
        bias = conv.bias.data_ptr() if conv.bias is not None else None
        common = dict(dtype=self.dt, x=x_nchw.data_ptr(), B=B, Cin=Cin, H=H, W=W, w=conv.weight.data_ptr(), K=K, stride=stride, pad=pad, bias=bias)
        if self.train:
            y = self.new_act(B, OH, OW, Cout, name + '.y')
            S = STRUCTS['salt_conv_first_args']()
            fill(S, B=B, Cin=Cin, H=H, W=W, K=K, stride=stride, pad=pad)
            nparts = lib.salt_conv_first_stats_parts(ctypes.byref(S))
            stats, cnt = Scratch('stats' + self._scratch_sfx, 4 * lib.salt_bn_stats_floats(nparts, Cout)), Scratch('stats_cnt' + self._scratch_sfx, nparts * 4)
            self.fwd.add('conv_first', y=y.view(), relu=0, stats=stats, stats_cnt=cnt, **common)
            w = self._bn_train_fwd(y, bn, relu, None, out, nparts, stats, cnt)

            def backward():
                self._bn_train_bwd(y, bn, relu, None, out, w)
                S2 = STRUCTS['salt_conv_first_wgrad_args']()
                fill(S2, B=B, Cin=Cin, H=H, W=W, K=K, stride=stride, pad=pad)
                np_ = lib.salt_conv_first_wgrad_parts(ctypes.byref(S2))
                self.bwd.add('conv_first_wgrad', stream=1, dtype=self.dt, x=x_nchw.data_ptr(), B=B, Cin=Cin, H=H, W=W, K=K, stride=stride, pad=pad,
                             dy=y.gview(), partials=Scratch('wgrad', np_ * Cout * Cin * K * K * 4), nparts=np_,
                             grad=self._gp(conv.weight), accumulate=0)
            self.tape.append(backward)
        else:
            w = eng.bn_work(bn)
            self.fwd.add('conv_first', y=out.view(), scale=w['scale'].data_ptr(), shift=w['shift'].data_ptr(), relu=int(relu), **common)
        return out

    def _stem_s2d(self, x_nchw, conv, bn, relu, name):
        """ResNet stem (KxK stride 2 over <= 4 channels) on the matrix cores: 2x2 space-to-depth turns it into a
        ((K+1)/2)^2-tap stride-1 convolution over 16 channels, which the dense MFMA kernels run ~15x faster than the
        vector-ALU direct kernel; weights / weight-gradients are folded between the two forms by tiny kernels."""
        eng = self.engine
        B, Cin, H, W = x_nchw.shape
        Cout, _, K, _ = conv.weight.shape
        TT, half = (K + 1) // 2, (K + 1) // 4
        z = self.new_act(B, H // 2, W // 2, 16, name + '.s2d')
        self.fwd.add('s2d', dtype=self.dt, x=x_nchw.data_ptr(), B=B, Cin=Cin, H=H, W=W, z=z.view())
        wp = eng.packed_stem(conv)
        taps = [(dh - half, dw - half) for dh in range(TT) for dw in range(TT)]
        OH, OW = H // 2, W // 2
        out = self.new_act(B, OH, OW, Cout, name)
        if self.train:
            y = self.new_act(B, OH, OW, Cout, name + '.y')
            nparts = self._conv_parts(z.view(), taps, 1, y.view(), OH, OW)
            if self._fin_on():
                prod = self._conv_launch(self.fwd, z.view(), wp.data_ptr(), taps, 1, 0, y.view(), OH, OW)
                w = self._bn_train_fwd(y, bn, relu, None, out, nparts, None, None, producer=prod)
            else:
                stats, cnt = Scratch('stats' + self._scratch_sfx, 4 * lib.salt_bn_stats_floats(nparts, Cout)), Scratch('stats_cnt' + self._scratch_sfx, nparts * 4)
                self._conv_launch(self.fwd, z.view(), wp.data_ptr(), taps, 1, 0, y.view(), OH, OW, stats=stats, stats_cnt=cnt)
                w = self._bn_train_fwd(y, bn, relu, None, out, nparts, stats, cnt)

            def backward():
                self._bn_train_bwd(y, bn, relu, None, out, w)
                g16 = self.f32(Cout * 16 * TT * TT)
                # dW'[t][n][c'] : P = dY (a = cout), Q = z (b = 16 s2d channels); reduce into [Cout][16][T][T], then unfold
                gw_tmp = g16.data_ptr()
                per = 8 if len(taps) > 9 else 9
                for i in range(0, len(taps), per):
                    chunk = list(range(i, min(i + per, len(taps))))
                    S = STRUCTS['salt_conv_wgrad_args']()
                    fill(S, dtype=self.dt, p=y.gview(), q=z.view(), ntaps=len(chunk), tap_dy=[taps[j][0] for j in chunk],
                         tap_dx=[taps[j][1] for j in chunk], q_step=1, pad_mode=0)
                    ns = lib.salt_conv_wgrad_nsplit(ctypes.byref(S))
                    nbytes = ns * len(chunk) * Cout * 16 * 4
                    # the stem is the last layer of backward: its weight gradient runs on the MAIN stream (own workspace), which is
                    # idle by then, instead of queueing behind the side stream's remaining weight gradients
                    self.bwd.add('conv_wgrad', stream=0, dtype=self.dt, p=y.gview(), q=z.view(), ntaps=len(chunk), tap_dy=[taps[j][0] for j in chunk],
                                 tap_dx=[taps[j][1] for j in chunk], q_step=1, pad_mode=0, partials=Scratch('wgrad@main', nbytes), nsplit=ns)
                    self.bwd.add('wgrad_reduce', stream=0, partials=Scratch('wgrad@main', nbytes), nsplit=ns, ntaps=len(chunk), Ca=Cout, Cb=16, KH=TT, KW=TT,
                                 tap_kh=[j // TT for j in chunk], tap_kw=[j % TT for j in chunk], grad=gw_tmp, accumulate=0)
                self.bwd.add('stem_grad_unfold', stream=0, g16=gw_tmp, Cout=Cout, Cin=Cin, K=K, grad=self._gp(conv.weight), accumulate=0)
            self.tape.append(backward)
        else:
            w = eng.bn_work(bn)
            self._conv_launch(self.fwd, z.view(), wp.data_ptr(), taps, 1, 0, out.view(), OH, OW,
                              scale=w['scale'].data_ptr(), shift=w['shift'].data_ptr(), relu=int(relu))
        return out

    # ------------------------------------------------------------------ logit head: 1x1 conv to <= 4 channels, fp32 NCHW out
    def head(self, x, conv, logits_nchw):
        Cout = conv.weight.shape[0]
        # round 6: x = relu(bn(y)) stored by the last forward operator and read by nobody else (the `final` Sequential of every U-Net here:
        # unet_models.py:131-138, architectures/unet.py:84-87) - BatchNorm apply + ReLU + head in one pass over y (salt_head_bn), the
        # activation and its gradient are never stored; the producer layer's _bn_train_bwd emits salt_head_bn_bwd instead of salt_bn_bwd
        taken = self._take_act_op(x) if self.head_bn_ok(x.C, conv) else None
        if taken is not None:
            sa, Fbn = taken
            hb = self.fwd.add('head_bn', dtype=self.dt, y=sa.y, fin=ctypes.addressof(Fbn), relu=int(sa.relu), w=conv.weight.data_ptr(),
                              bias=conv.bias.data_ptr() if conv.bias is not None else None, Cout=Cout, y_nchw=logits_nchw.data_ptr())
            for i, q in enumerate(self._fin_patches):    # the statistics shards the dropped affine_act would have finalized
                if q[0] is sa and q[1] == 'fin_acc':
                    self._fin_patches[i] = (hb, 'fin_acc', q[2], q[3])
            self.dlogits = self.alloc(tuple(logits_nchw.shape), torch.float32)
            x.buf.head_fused = conv
            self.tape.append(lambda: None)               # the head's backward is part of the producer layer's closure (Graph._bn_train_bwd)
            return
        self.fwd.add('head1x1', dtype=self.dt, x=x.view(), w=conv.weight.data_ptr(), bias=conv.bias.data_ptr() if conv.bias is not None else None,
                     Cout=Cout, y_nchw=logits_nchw.data_ptr(), y=null_view())
        if self.train:
            self.dlogits = self.alloc(tuple(logits_nchw.shape), torch.float32)

            def backward():
                S = STRUCTS['salt_head1x1_bwd_args']()
                fill(S, x=x.view())
                nparts = lib.salt_head1x1_bwd_parts(ctypes.byref(S))
                acc = x.grad_state()
                self.bwd.add('head1x1_bwd', dtype=self.dt, x=x.view(), w=conv.weight.data_ptr(), Cout=Cout, dy_nchw=self.dlogits.data_ptr(),
                             dx=x.gview(), accumulate=acc, partials=Scratch('head', nparts * Cout * (x.C + 1) * 4), nparts=nparts,
                             gw=self._gp(conv.weight), gb=self._gp(conv.bias) if conv.bias is not None else None)
            self.tape.append(backward)

    # ------------------------------------------------------------------ layout boundary (fp32 NCHW <-> NHWC activations)
    def from_nchw(self, x_nchw, name='input'):
        B, C, H, W = x_nchw.shape
        a = self.new_act(B, H, W, C, name)
        self.fwd.add('layout', dtype=self.dt, nchw=x_nchw.data_ptr(), nhwc=a.view(), to_nhwc=1)
        if self.train:
            dx = self.alloc((B, C, H, W), torch.float32)
            self.input_grads = getattr(self, 'input_grads', []) + [dx]

            def backward():
                if a.grad_ready():
                    self.bwd.add('layout', dtype=self.dt, nchw=dx.data_ptr(), nhwc=a.gview(), to_nhwc=0)
            self.tape.append(backward)
        return a

    def to_nchw(self, a, out_nchw):
        self.fwd.add('layout', dtype=self.dt, nchw=out_nchw.data_ptr(), nhwc=a.view(), to_nhwc=0)
        if self.train:
            self.dlogits = self.alloc(tuple(out_nchw.shape), torch.float32)

            def backward():
                assert a.grad_state() == 0
                self.bwd.add('layout', dtype=self.dt, nchw=self.dlogits.data_ptr(), nhwc=a.gview(), to_nhwc=1)
            self.tape.append(backward)

    # ------------------------------------------------------------------ pooling / resize
    def maxpool2(self, x, out=None, name=''):
        if out is None:
            out = self.new_act(x.B, x.H // 2, x.W // 2, x.C, name)
        self.fwd.add('maxpool2', dtype=self.dt, x=x.view(), y=out.view())
        if self.train:
            def backward():
                acc = x.grad_state()
                self.bwd.add('maxpool2_bwd', dtype=self.dt, x=x.view(), dy=out.gview(), dx=x.gview(), accumulate=acc)
            self.tape.append(backward)
        return out

    def maxpool3s2(self, x, out=None, name=''):
        """nn.MaxPool2d(3, 2, 1) - the ResNet stem pool of ResNetEncoders(pool0=True) (architectures/encoders.py:23-27)."""
        if out is None:
            out = self.new_act(x.B, (x.H + 1) // 2, (x.W + 1) // 2, x.C, name)
        self.fwd.add('maxpool3s2', dtype=self.dt, x=x.view(), y=out.view())
        if self.train:
            def backward():
                acc = x.grad_state()
                self.bwd.add('maxpool3s2_bwd', dtype=self.dt, x=x.view(), dy=out.gview(), dx=x.gview(), accumulate=acc)
            self.tape.append(backward)
        return out

    def avgpool2(self, x, out=None, name=''):
        if out is None:
            out = self.new_act(x.B, x.H // 2, x.W // 2, x.C, name)
        self.fwd.add('avgpool2', dtype=self.dt, x=x.view(), y=out.view(), backward=0, accumulate=0)
        if self.train:
            def backward():
                acc = x.grad_state()
                self.bwd.add('avgpool2', dtype=self.dt, x=x.gview(), y=out.gview(), backward=1, accumulate=acc)
            self.tape.append(backward)
        return out

    def upsample(self, x, R, out=None, name=''):
        if out is None:
            out = self.new_act(x.B, x.H * R, x.W * R, x.C, name)
        ac = int(bool(getattr(getattr(self.engine, 'module', None), 'align_corners', False)))      # HipNetwork.set_align_corners
        self.fwd.add('bilinear', dtype=self.dt, x=x.view(), y=out.view(), R=R, backward=0, accumulate=0, align_corners=ac)
        if self.train:
            def backward():
                acc = x.grad_state()
                tmp = Scratch('bilinear', x.B * out.H * x.W * _round_up(x.C, self.ve) * self._es()) if R >= 4 else None
                self.bwd.add('bilinear', dtype=self.dt, x=x.gview(), y=out.gview(), R=R, backward=1, accumulate=acc, tmp=tmp, align_corners=ac)
            self.tape.append(backward)
        return out

    def hyper_rows(self, xs, Rs, out):
        """All up-sampled hypercolumn levels in ONE pass: out (a slice of len(xs) * C channels) <- [up(x_k, R_k)] (salt_hyper_rows); the
        adjoints stay one salt_bilinear launch per level."""
        ac = int(bool(getattr(getattr(self.engine, 'module', None), 'align_corners', False)))
        assert out.C == len(xs) * xs[0].C and all(x.C == xs[0].C for x in xs)
        self.fwd.add('hyper_rows', dtype=self.dt, nlev=len(xs), x=[x.view() for x in xs], R=list(Rs), y=out.view(), align_corners=ac)
        if self.train:
            def backward():
                for k in reversed(range(len(xs))):
                    x, R, o = xs[k], Rs[k], out.slice(k * xs[0].C, xs[0].C)
                    acc = x.grad_state()
                    tmp = Scratch('bilinear', x.B * o.H * x.W * _round_up(x.C, self.ve) * self._es()) if R >= 4 else None
                    self.bwd.add('bilinear', dtype=self.dt, x=x.gview(), y=o.gview(), R=R, backward=1, accumulate=acc, tmp=tmp, align_corners=ac)
            self.tape.append(backward)
        return out

    def add(self, a, b, out=None, name=''):
        if out is None:
            out = self.new_act(a.B, a.H, a.W, a.C, name)
        self.fwd.add('add', dtype=self.dt, a=a.view(), b=b.view(), y=out.view(), accumulate=0)
        if self.train:
            def backward():
                for t in (a, b):
                    acc = t.grad_state()
                    self.bwd.add('add', dtype=self.dt, a=out.gview(), b=null_view(), y=t.gview(), accumulate=acc)
            self.tape.append(backward)
        return out

    def copy(self, a, out):
        self.fwd.add('add', dtype=self.dt, a=a.view(), b=null_view(), y=out.view(), accumulate=0)
        if self.train:
            def backward():
                acc = a.grad_state()
                self.bwd.add('add', dtype=self.dt, a=out.gview(), b=null_view(), y=a.gview(), accumulate=acc)
            self.tape.append(backward)
        return out

    def _take_act_op(self, x):
        """x = relu?(bn(y)) written by the LAST forward operator (a consumer-side-finalize salt_affine_act without residual, main stream)
        and read by nobody yet: remove that operator from the program and hand back (its argument struct, the layer's
        salt_bn_finalize_args) - the caller's kernels apply the transform to y themselves.  Graph.finalize fails loudly if anybody asks
        for a storage view of the activation afterwards."""
        rec = getattr(x.buf, 'act_op', None)
        if rec is None or not self.fwd.ops or self.fwd.ops[-1][0] != 'affine_act':
            return None
        sa, reads0, c0, C = rec
        if (self.fwd.ops[-1][2] is not sa or (c0, C) != (x.c0, x.C) or x.c0 != 0 or x.C != x.buf.C or x.buf.reads != reads0 or not sa.fin or sa.res.p
                or self.fwd.streams[-1] != 0 or x.buf.planes):
            return None
        self.fwd.ops.pop(); self.fwd.streams.pop()
        self.fwd._entries = None
        x.buf.act_op = None
        self._virtual_acts = getattr(self, '_virtual_acts', []) + [(x.buf, x.buf.reads)]
        return sa, STRUCTS['salt_bn_finalize_args'].from_address(sa.fin)

    # ------------------------------------------------------------------ scSE
    def scse(self, x, cse, sse, out=None, name=''):
        """relu(x*cSE(x) + x*sSE(x)); cse.fc = Sequential(Linear, ReLU, Linear, Sigmoid), sse.fc = Conv2d(C,1,1)."""
        eng = self.engine
        if out is None:
            out = self.new_act(x.B, x.H, x.W, x.C, name)
        l1, l2, cs = cse.fc[0], cse.fc[2], sse.fc
        R, C, B = l1.weight.shape[0], x.C, x.B
        shards = self.train and self._fin_mode() == 2 and os.environ.get('SALT_SE_SHARDS', '1') != '0'   # per-image sums through the zeroed fp64 arena (see _fin_slot)
        # round 6: x = relu(bn(conv)) whose ONLY reader is this operator (base.DecoderBlock: conv2 -> scSE, architectures/base.py:60-85) -
        # the affine_act that would store it is taken back out of the program and the scSE kernels apply BatchNorm + ReLU to the raw
        # convolution output on the way in (salt_scse_args.in_fin); backward reads the raw output through the same transform
        taken = self._take_act_op(x) if (shards and os.environ.get('SALT_SE_IN_BN', '1') != '0') else None
        if taken is not None:
            sa, Fbn = taken
            xv_ = sa.y                                   # the raw convolution output; never x.view(): the activation has no storage reader
            in_kw = dict(in_fin=ctypes.addressof(Fbn), in_relu=int(sa.relu))
            in_bwd = dict(in_scale=Fbn.scale, in_shift=Fbn.shift, in_relu=int(sa.relu))
        else:
            xv_, in_kw, in_bwd = None, {}, {}
        xview = (lambda: xv_) if taken is not None else x.view
        S = STRUCTS['salt_scse_args']()
        fill(S, x=xview())
        nparts = lib.salt_scse_parts(ctypes.byref(S))
        gap, hid, gc, gs = self.f32(B * C), self.f32(B * R), self.f32(B * C), self.f32(B * x.H * x.W)
        sf = self.fwd.add('scse', dtype=self.dt, x=xview(), w1=l1.weight.data_ptr(), b1=l1.bias.data_ptr(), w2=l2.weight.data_ptr(),
                          b2=l2.bias.data_ptr(), R=R, ws=cs.weight.data_ptr(), bs=cs.bias.data_ptr(), gap_partials=Scratch('se', B * nparts * (2 * C + 1) * 4),
                          nparts=nparts, gap=gap.data_ptr(), hidden=hid.data_ptr(), gate_c=gc.data_ptr(), gate_s=gs.data_ptr(), y=out.view(), **in_kw)
        if taken is not None:
            for i, q in enumerate(self._fin_patches):    # the statistics shards the dropped affine_act would have finalized
                if q[0] is sa and q[1] == 'fin_acc':
                    self._fin_patches[i] = (sf, 'in_fin_acc', q[2], q[3])
        if shards:
            self._fin_slot('fwd', B * C, (sf, 'gap_acc'))
        if self.train:
            def backward():
                acc = x.grad_state()
                dgap = self.f32(B * C)
                gp = self._gp
                # round 6: the FC parameter gradients (one workgroup over the whole batch: 18 - 24 us) are nobody's input before the optimizer -
                # they run as their own operator on the weight-gradient queue; the critical queue keeps the per-image dgap kernel
                defer = shards and os.environ.get('SALT_SE_FC_SIDE', '1') != '0'
                kw = dict(dtype=self.dt, x=xview(), y=out.view(), dy=out.gview(), w1=l1.weight.data_ptr(), w2=l2.weight.data_ptr(),
                          R=R, ws=cs.weight.data_ptr(), gap=gap.data_ptr(), hidden=hid.data_ptr(), gate_c=gc.data_ptr(), gate_s=gs.data_ptr(),
                          partials=Scratch('se', B * nparts * (2 * C + 1) * 4), nparts=nparts, g_w1=gp(l1.weight), g_b1=gp(l1.bias),
                          g_w2=gp(l2.weight), g_b2=gp(l2.bias), g_ws=gp(cs.weight), g_bs=gp(cs.bias), dgap=dgap.data_ptr(),
                          dx=x.gview(), accumulate=acc, defer_param_grads=int(defer), **in_bwd)
                sb = self.bwd.add('scse_bwd', **kw)
                sg = self.bwd.add('scse_fc_grads', stream=1, **kw) if defer else None
                # round 6: with the input transform the kernel holds everything the producer layer's BatchNorm backward sums over - it takes
                # them too and that layer's bn_bwd loses its reduction pass (saltnet.h salt_scse_bwd_args.bnb_acc; SALT_SE_BNB=0: off)
                carry = (taken is not None and shards and getattr(x.buf, 'bn_train_out', None) == (x.c0, x.C) and acc == 0
                         and x.B * x.H * x.W < (1 << 31) and os.environ.get('SALT_SE_BIAS_FOLD', '1') != '0' and os.environ.get('SALT_SE_BNB', '1') != '0')
                if shards:
                    self._fin_slot('bwd', B * ((6 if carry else 2) * C + 1), (sb, 'acc'), *([(sg, 'acc')] if sg is not None else []))
                # x = relu(bn(conv)): its only gradient consumer is that layer's bn_bwd, which can add the channel-SE term dgap[b][c]
                # wherever it reads dL/dx - the broadcast-add pass over dx (read + write of the whole tensor) disappears
                if (getattr(x.buf, 'bn_train_out', None) == (x.c0, x.C) and acc == 0 and x.B * x.H * x.W < (1 << 31)
                        and os.environ.get('SALT_SE_BIAS_FOLD', '1') != '0'):
                    if getattr(x.buf, 'grad_bias', None) is not None:
                        raise SaltError('two pending gradient biases on %s' % x.buf.name)
                    self.bwd.set_fields(sb, skip_bcast=1)
                    x.buf.grad_bias = (x.c0, x.C, dgap) + ((sb, sg) if carry else ())
                    self._bias_bufs = getattr(self, '_bias_bufs', []) + [x.buf]
            self.tape.append(backward)
        return out
