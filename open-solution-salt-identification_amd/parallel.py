"""Data parallelism: one process per GPU, gradients averaged with RCCL over xGMI.

Replaces the reference's single-process ``nn.DataParallel`` (models.py:81-85: per-step parameter broadcast,
input scatter, logits gather to GPU0, loss on GPU0, gradient reduce to GPU0).  Here every rank owns a full
replica and 1/world of the minibatch; BatchNorm statistics stay per rank exactly as the reference's
per-replica BN does (no SyncBN anywhere in the reference); the only exchange is ONE gradient average per
step, and because gradients live in one flat buffer the all-reduce works on contiguous byte ranges:

  * buckets are contiguous ranges of the flat gradient buffer, formed from the END (the decoder's
    parameters get their gradients first) so a bucket is complete while the encoder backward still runs;
  * the backward program is executed in segments; after each segment an event is recorded and the
    bucket's all-reduce is issued on a side stream, overlapping RCCL with the remaining backward kernels;
  * xGMI is a point-to-point mesh (7 links x ~153 GB/s per GPU): RCCL's direct reduce-scatter/all-gather
    uses all links at once when the payload is large, so buckets are few and big (default 32 MB) rather
    than NCCL/NVSwitch-style 25 MB-by-habit; the 120.7 MB ResNet34 U-Net gradient is 5 collectives, the last
    of them (stem + first encoder stage, final only when backward ends and therefore not overlapped) <= 4 MB;
  * the 1/world average is folded into Adam's gradient scale (no extra pass).

CPU (gloo) is supported for the bucket planner / reducer so the N>1 logic is testable without GPUs.
"""
import os
import weakref

import torch
import torch.distributed as dist

DEFAULT_BUCKET_BYTES = 32 << 20
DEFAULT_TAIL_BYTES = 4 << 20
# RCCL's device kernels take ONE workgroup (= one CU) per channel.  The convolution kernels of this build are whole-CU workgroups
# (conv_ws / conv_ls / conv_wgrad_ls: 160 KB of LDS or > 256 registers per SIMD lane pair - DESIGN 4), so every channel RCCL opens is a
# CU the backward pass it overlaps cannot use, and a workgroup of ours waits for a CU RCCL holds.  RCCL's own default on this part is
# 64 channels for the 1-rank communicator (bench.py: rccl_info) - a quarter of the chip.  The 7 xGMI links of a GPU carry ~153 GB/s
# each and a channel moves ~20-25 GB/s, so 32 channels (1/8 of the CUs) already cover the links a ring / direct exchange can use at
# once; the ~120 MB of gradients per step are then ~0.4 ms of collectives underneath ~3.5 ms of backward.  OPT-IN (round 6):
# the cap is process-wide, applies to every communicator of the process and has never been measured on more than one rank, so RCCL is
# left alone unless SALT_RCCL_MAX_NCHANNELS=<n> asks for it (or the user sets NCCL_MAX_NCHANNELS, which always wins); 32 is the
# value the reasoning above suggests for an 8-GPU A/B.
DEFAULT_RCCL_MAX_NCHANNELS = 0
SUGGESTED_RCCL_MAX_NCHANNELS = 32


def configure_rccl_env():
    """SALT_RCCL_MAX_NCHANNELS=<n> (opt-in): export NCCL_MAX_NCHANNELS=<n> (unless the user set that variable) BEFORE the process group
    is created - RCCL reads it when the communicator comes up.  -> the value in effect, or None when RCCL's default is left alone."""
    if 'NCCL_MAX_NCHANNELS' in os.environ:
        return int(os.environ['NCCL_MAX_NCHANNELS'])
    n = int(os.environ.get('SALT_RCCL_MAX_NCHANNELS', str(DEFAULT_RCCL_MAX_NCHANNELS)))
    if n > 0:
        os.environ['NCCL_MAX_NCHANNELS'] = str(n)
        return n
    return None


def _cpulist(text):
    out = []
    for part in text.strip().split(','):
        if not part:
            continue
        a, _, b = part.partition('-')
        out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_node(index):
    """NUMA node of GPU ``index`` (sysfs, through the PCI address torch reports); None when the platform does not say"""
    try:
        pr = torch.cuda.get_device_properties(index)
        path = '/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open(path).read())
        return node if node >= 0 else None
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None


def pin_rank_threads(local_rank=0, local_world=1):
    """One process per GPU means one launch thread + one loader thread per GPU on the SAME host: pin this process (and every thread it
    starts later - torch's intra-op pool, the input pipeline's loader) to the cores of its GPU's NUMA node, the node's cores split evenly
    between the ranks whose GPUs share it (the reference's nn.DataParallel ran ONE process with a thread per replica, models.py:81-85,
    and num_workers loader processes, loaders.py:477-490).  SALT_NO_PIN=1: leave the affinity alone.
    -> {'numa_node', 'cpus' (count), 'mask' (cpulist string), 'ranks_on_node'} or None when nothing was changed."""
    if os.environ.get('SALT_NO_PIN') or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        ngpu = torch.cuda.device_count() if torch.cuda.is_available() else 0
        nodes = [gpu_numa_node(i) for i in range(ngpu)]
        node = nodes[local_rank] if local_rank < len(nodes) else None
        if node is not None:
            cpus = [c for c in _cpulist(open('/sys/devices/system/node/node%d/cpulist' % node).read()) if c in set(allowed)]
            peers = [r for r in range(local_world) if r < len(nodes) and nodes[r] == node]
        else:
            cpus, peers = allowed, list(range(local_world))
        if not cpus:
            return None
        k = peers.index(local_rank) if local_rank in peers else 0
        per = max(len(cpus) // max(len(peers), 1), 1)
        mine = cpus[k * per:(k + 1) * per] or cpus
        os.sched_setaffinity(0, mine)
        runs, a = [], None
        for c in mine + [None]:
            if a is None:
                a = b = c
            elif c is not None and c == b + 1:
                b = c
            else:
                runs.append('%d-%d' % (a, b) if b != a else '%d' % a)
                a = b = c
        return {'numa_node': node, 'cpus': len(mine), 'mask': ','.join(runs), 'ranks_on_node': len(peers)}
    except (OSError, ValueError):
        return None


def _destroy_events(handles):
    """salt_event_destroy on native event handles (ctypes.c_void_p) - weakref.finalize callback of a compiled net / drop_plans."""
    from ._abi import lib
    for h in handles:
        if h is not None and h.value:
            lib.salt_event_destroy(h)
            h.value = None


DEFAULT_FIRST_FRACTION = 0.25       # the first collective is issued by this fraction of the backward program (0: size rule only)
DEFAULT_FIRST_MIN_BYTES = 1 << 20


def plan_buckets(ready, total, bucket_bytes=DEFAULT_BUCKET_BYTES, tail_bytes=DEFAULT_TAIL_BYTES, first_pos=None, first_min_bytes=DEFAULT_FIRST_MIN_BYTES):
    """ready: list of (offset, numel, ready_index) per parameter (offsets ascending in forward order; ready_index =
    backward-program position after which that gradient is final).  Returns buckets in issue order:
    [(lo, hi, ready_index)] covering [0,total); every bucket but the remainder holds >= bucket_bytes.  The remainder (the first
    layers of the encoder: their gradients are final only when backward ends, so their collective is the one left exposed)
    is split once more so that the very last collective moves at most ``tail_bytes``.
    ``first_pos`` (round 6): the FIRST bucket is closed early - as soon as it holds ``first_min_bytes`` and the next parameter's
    gradient would only be final after backward-program position ``first_pos`` - so that the wire starts working by then instead of
    when 32 MB have piled up (the ResNet34 U-Net's decoder is 1 / 4 of its parameters: the size rule alone issued the first
    collective 46 % into backward)."""
    items = sorted(ready, key=lambda r: r[0])
    cuts = []                                    # lower bounds of the buckets, descending
    hi = total
    for i in range(len(items) - 1, -1, -1):
        off = items[i][0]
        early = (first_pos is not None and not cuts and i > 0 and (hi - off) * 4 >= first_min_bytes and items[i - 1][2] > first_pos)
        if (hi - off) * 4 >= bucket_bytes or early:
            cuts.append(off)
            hi = off
    if hi > 0:
        if tail_bytes and hi * 4 > tail_bytes:
            split = max((off for off, _, _ in items if 0 < off < hi and off * 4 <= tail_bytes), default=0)
            if split > 0:
                cuts.append(split)
        cuts.append(0)
    out, hi, m = [], total, 0
    for lo in cuts:
        m = max([m] + [r for off, _, r in items if lo <= off < hi])      # ready indices are non-decreasing in issue order
        out.append((lo, hi, m))
        hi = lo
    return out


def shard_batch(n, rank, world):
    """Contiguous per-rank slice of a global batch of n samples (reference scatter semantics: equal chunks)."""
    per = (n + world - 1) // world
    return slice(min(rank * per, n), min((rank + 1) * per, n))


class DataParallel:
    affinity = None                     # what pin_rank_threads did for this process (bench.py prints it)

    def __init__(self, rank=0, world=1, bucket_bytes=DEFAULT_BUCKET_BYTES):
        self.rank, self.world, self.bucket_bytes = rank, world, bucket_bytes
        self._comm_stream = None
        # per compiled net (weak keys: a freed and re-created CompiledNet can never meet another net's stale mark positions through
        # a recycled id()): bucket plan, the executor's marks (native events), torch events of the segmented form
        self._plans = weakref.WeakKeyDictionary()
        self._events = weakref.WeakKeyDictionary()
        self._marks = weakref.WeakKeyDictionary()
        self.measure = False            # bench.py: record an event pair around the final wait for the collectives
        self.exposed_events = []
        self.timeline = False           # bench.py / tools/dp_overhead.py: timing events per bucket (ready / all-reduce done) and at backward end
        self.timeline_events = []       # per step: (t0, [(ready, done)] per bucket, backward_end, wait_end)
        self.skip_collectives = False   # tools/dp_overhead.py: the segmented backward with its events but WITHOUT the collective calls (A/B)

    @classmethod
    def from_env(cls):
        if dist.is_available() and dist.is_initialized():
            return cls(dist.get_rank(), dist.get_world_size())
        return cls(0, 1)

    @staticmethod
    def init_process_group_from_env(backend=None):
        """torchrun-style bootstrap (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT)."""
        world = int(os.environ.get('WORLD_SIZE', '1'))
        if world > 1 and not dist.is_initialized():
            if torch.cuda.is_available():
                torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            if torch.cuda.is_available() and backend in (None, 'nccl'):
                configure_rccl_env()
            if torch.cuda.is_available():
                DataParallel.affinity = pin_rank_threads(int(os.environ.get('LOCAL_RANK', '0')), int(os.environ.get('LOCAL_WORLD_SIZE', str(world))))
            dist.init_process_group(backend or ('nccl' if torch.cuda.is_available() else 'gloo'))
        return DataParallel.from_env()

    # ------------------------------------------------------------------ replicas start identical
    def broadcast_parameters(self, model):
        if self.world == 1:
            return
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                self._broadcast(t.data, src=0)
        if getattr(model, '_engine', None) is not None:
            model._engine.touch()

    # ------------------------------------------------------------------ flat-buffer reducer (device agnostic)
    def allreduce_flat(self, flat, buckets):
        """Sum-reduce contiguous ranges of ``flat`` in place (the 1/world factor is applied by the optimizer)."""
        if not self._active():
            return
        works = [self._all_reduce(flat[lo:hi]) for lo, hi, _ in buckets]
        for w in works:
            if w is not None:
                w.wait()

    def _active(self):
        return self.world > 1 or bool(os.environ.get('SALT_FORCE_DP_PATH') and dist.is_initialized())

    def _all_reduce(self, t):
        """SUM all-reduce of one contiguous gradient range on the current stream (async work handle).  Tests substitute the
        collective (e.g. x2 = the sum over two ranks that hold the same batch) to exercise the plan without a second GPU.
        gloo + device tensors (two ranks on ONE GPU - RCCL refuses two ranks on one device - or a CPU-only fabric): the range is staged
        through the host ON THE CURRENT STREAM (the D2H copy waits for the bucket's events like a collective kernel would), summed by
        gloo, and copied back; no work handle - the caller's stream wait orders the optimizer behind the copy."""
        if t.is_cuda and dist.get_backend() == 'gloo':
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t.copy_(h)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)

    @staticmethod
    def _broadcast(t, src=0):
        if t.is_cuda and dist.get_backend() == 'gloo':
            h = t.cpu()
            dist.broadcast(h, src=src)
            t.copy_(h)
        else:
            dist.broadcast(t, src=src)

    def allreduce_gradients(self, eng, optimizer=None):
        """Autograd-bridge branch of _fit_loop: one SUM all-reduce of the whole flat gradient buffer after backward; the 1/world
        of the average goes into the optimizer's gradient scale exactly as in :meth:`backward`."""
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world
        if not self._active():
            return
        self.allreduce_flat(eng.grads, [(0, eng.n_live, 0)])

    # ------------------------------------------------------------------ rank-consistent trainer decisions
    def any_rank(self, flag):
        """True on every rank when ``flag`` is true on at least one (early stopping must end fit() everywhere in the same epoch,
        otherwise the remaining ranks block in the next all-reduce)."""
        if self.world == 1 or not dist.is_initialized():
            return bool(flag)
        dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor([1.0 if flag else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item() > 0)

    def broadcast_scalars(self, values, src=0):
        """Replace a list of python floats by rank ``src``'s (validation scores: the callbacks of every rank must decide on the
        same numbers - BatchNorm running statistics are per rank, so the per-rank scores differ slightly)."""
        if self.world == 1 or not dist.is_initialized():
            return [float(v) for v in values]
        dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=dev)
        dist.broadcast(t, src=src)
        return [float(v) for v in t.cpu()]

    def exposed_allreduce_ms(self):
        """Mean milliseconds per step the compute stream spent waiting for the collectives after backward finished (measure=True)."""
        if not self.exposed_events:
            return None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.exposed_events]
        return sum(ms) / len(ms)

    # ------------------------------------------------------------------ overlapped backward
    def backward(self, eng, net, optimizer=None):
        """Run net.bwd; with world > 1 run it in bucket segments with the all-reduce on a side stream."""
        if optimizer is not None:
            optimizer.grad_scale = 1.0 / self.world
        if not self._active():
            net.bwd.run(side=eng.side_stream)
            return
        key = net
        if key not in self._plans:
            frac = float(os.environ.get('SALT_DP_FIRST_FRACTION', str(DEFAULT_FIRST_FRACTION)))
            self._plans[key] = plan_buckets(net.g.grad_ready, eng.n_live, self.bucket_bytes,
                                            first_pos=int(frac * len(net.bwd)) if frac > 0 else None)
        if self.timeline or os.environ.get('SALT_DP_SEGMENTS'):
            if key not in self._events:
                # ONE event per bucket and queue, created once (round 4 allocated two torch.cuda.Event objects per bucket per step)
                self._events[key] = [(torch.cuda.Event(), torch.cuda.Event()) for _ in self._plans[key]]
            return self._backward_segments(eng, net, key)
        if key not in self._marks:
            self._marks[key] = self._make_marks(self._plans[key], len(net.bwd))
            weakref.finalize(net, _destroy_events, [h for pair in self._marks[key][4] for h in pair])
        # ---- ONE executor call for the whole backward program: the executor records each bucket's two events (compute queue, weight-
        # gradient queue) when it reaches the bucket's position - no cut of the program per bucket (round 4: + 2 % on one rank before
        # any wire time, most of it the per-segment flush / lost fork hand-off) - and the collectives are issued behind those events
        from ._abi import lib, check
        import ctypes
        pos, n, evm, evs, handles = self._marks[key]
        cur = torch.cuda.current_stream()
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        comm = self._comm_stream
        net.bwd.run(side=eng.side_stream, marks=(pos, n, evm, evs))
        works = []
        with torch.cuda.stream(comm):
            for (lo, hi, _), (hm, hs) in zip(self._plans[key], handles):
                check(lib.salt_stream_wait_event(ctypes.c_void_p(comm.cuda_stream), hm), 'stream_wait_event')
                check(lib.salt_stream_wait_event(ctypes.c_void_p(comm.cuda_stream), hs), 'stream_wait_event')
                works.append(None if self.skip_collectives else self._all_reduce(eng.grads[lo:hi]))
        if self.measure:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
        for w in works:
            if w is not None:
                w.wait()
        cur.wait_stream(comm)
        if self.measure:
            e1.record(cur)
            self.exposed_events.append((e0, e1))

    def drop_plans(self):
        """Forget every bucket plan (tools re-plan with another bucket size); the native events of the dropped marks are destroyed."""
        for m in list(self._marks.values()):
            _destroy_events([h for pair in m[4] for h in pair])
            del m[4][:]
        self._plans.clear(); self._events.clear(); self._marks.clear()

    def _make_marks(self, plan, n_ops):
        """ctypes arrays for salt_program_run_streams_marks: ascending positions + one native event per bucket and queue"""
        import ctypes
        from ._abi import lib, check
        n = len(plan)
        pos = (ctypes.c_int * n)(*[min(max(r, 0), n_ops) for _, _, r in plan])
        for a, b in zip(pos, list(pos)[1:]):
            assert a <= b, 'bucket positions must ascend'
        evm, evs = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
        handles = []
        for i in range(n):
            hm, hs = ctypes.c_void_p(), ctypes.c_void_p()
            check(lib.salt_event_create(ctypes.byref(hm)), 'event_create'); check(lib.salt_event_create(ctypes.byref(hs)), 'event_create')
            evm[i], evs[i] = hm.value, hs.value
            handles.append((hm, hs))
        return pos, n, evm, evs, handles

    def _backward_segments(self, eng, net, key):
        """round 4's form - the backward program cut into one executor call per bucket - kept for the per-bucket timing timeline
        (timeline=True needs timing events recorded from Python) and as an A/B (SALT_DP_SEGMENTS=1)"""
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        comm = self._comm_stream
        pos, works = 0, []
        n_ops = len(net.bwd)
        tl = None
        if self.timeline:
            tl = [torch.cuda.Event(enable_timing=True), [], None, None]
            tl[0].record(cur)
        for (lo, hi, ridx), (ev_plain, ev_side) in zip(self._plans[key], self._events[key]):
            ridx = min(max(ridx, pos), n_ops)
            if ridx > pos:
                # no join between segments: the main stream keeps running ahead of the weight-gradient stream; the bucket's
                # gradients come from both, so the communication stream waits for both
                net.bwd.run(begin=pos, end=ridx, side=eng.side_stream, join=False)
                pos = ridx
            ev = torch.cuda.Event(enable_timing=True) if tl is not None else ev_plain
            ev.record(cur)
            ev_side.record(eng.side_stream)
            with torch.cuda.stream(comm):
                comm.wait_event(ev)
                comm.wait_event(ev_side)
                w = None if self.skip_collectives else self._all_reduce(eng.grads[lo:hi])
                works.append(w)
                if tl is not None:
                    if w is not None:
                        w.wait()        # orders `comm` behind the collective (ProcessGroupNCCL runs it on its own stream)
                    done = torch.cuda.Event(enable_timing=True)
                    done.record(comm)
                    tl[1].append((ev, done, (hi - lo) * 4))
        if pos < n_ops:
            net.bwd.run(begin=pos, end=n_ops, side=eng.side_stream)
        if tl is not None:
            tl[2] = torch.cuda.Event(enable_timing=True)
            tl[2].record(cur)
        if self.measure:        # exposed communication = what the compute stream waits for after its last backward kernel
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
        for w in works:
            if w is not None:
                w.wait()        # the compute stream waits for RCCL before Adam reads the gradients
        cur.wait_stream(comm)   # (also orders a substituted collective that returned no work handle)
        if self.measure:
            e1.record(cur)
            self.exposed_events.append((e0, e1))
        if tl is not None:
            tl[3] = torch.cuda.Event(enable_timing=True)
            tl[3].record(cur)
            self.timeline_events.append(tl)

    def bucket_timeline(self):
        """Per bucket, averaged over the recorded steps (timeline=True), all in ms RELATIVE TO THE END OF BACKWARD on the compute stream
        (negative = before it): when its gradients were final (`ready_ms`), when its all-reduce had finished (`allreduce_done_ms`), its
        size; plus the backward span and the wait behind it.  A collective is fully overlapped when allreduce_done_ms <= 0."""
        if not self.timeline_events:
            return None
        torch.cuda.synchronize()
        n = len(self.timeline_events)
        nb = len(self.timeline_events[0][1])
        out = {'steps': n, 'backward_ms': 0.0, 'wait_after_backward_ms': 0.0, 'buckets': [{'mbytes': 0.0, 'ready_ms': 0.0, 'allreduce_done_ms': 0.0} for _ in range(nb)]}
        for t0, bs, t_end, t_wait in self.timeline_events:
            span = t0.elapsed_time(t_end)
            out['backward_ms'] += span / n
            out['wait_after_backward_ms'] += t_end.elapsed_time(t_wait) / n
            for j, (ready, done, nbytes) in enumerate(bs):
                out['buckets'][j]['mbytes'] = round(nbytes / 1e6, 2)
                out['buckets'][j]['ready_ms'] += (t0.elapsed_time(ready) - span) / n
                out['buckets'][j]['allreduce_done_ms'] += (t0.elapsed_time(done) - span) / n
        out['backward_ms'] = round(out['backward_ms'], 4); out['wait_after_backward_ms'] = round(out['wait_after_backward_ms'], 4)
        for b in out['buckets']:
            # all-reduce bus bandwidth (the nccl-tests convention: 2 (N - 1) / N x bytes / time) from "gradients final" to "collective done":
            # an UPPER bound on the collective's own duration (it may have queued behind the previous bucket's), so a LOWER bound on the rate
            dur = b['allreduce_done_ms'] - b['ready_ms']
            b['busbw_GBps_lower_bound'] = round(2.0 * (self.world - 1) / self.world * b['mbytes'] / dur, 1) if (dur > 0 and self.world > 1) else None
            b['ready_ms'] = round(b['ready_ms'], 4); b['allreduce_done_ms'] = round(b['allreduce_done_ms'], 4)
        return out
