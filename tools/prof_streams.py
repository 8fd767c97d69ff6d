#!/usr/bin/env python
"""Per-queue view of a training step from a rocprofv3 rocpd database: busy time per HIP stream (hardware queue), the kernels on
each, and the phases of the step (forward / loss / backward / optimizer) by wall time.
usage: python tools/prof_streams.py <db> [skip] [train_steps]
train_steps = warm-up + timed steps of the profiled bench.py run: only intervals between THOSE Adam launches are steps (what follows -
bench.py's per-operator roofline pass, the eval pass - also launches Adam-delimited work and used to inflate the per-step figures by ~12 %)."""
import sqlite3, sys
from collections import defaultdict
path = sys.argv[1]; skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ntrain = int(sys.argv[3]) if len(sys.argv) > 3 else 0
c = sqlite3.connect(path)
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if 'kernel_dispatch' in x][0]; ks = [x for x in t if 'kernel_symbol' in x][0]
cols = [r[1] for r in c.execute('pragma table_info(%s)' % kd)]
qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
rows = list(c.execute("select s.kernel_name, d.start, d.end, d.%s from %s d join %s s on d.kernel_id=s.id order by d.start" % (qcol, kd, ks)))
marks = [i for i, r in enumerate(rows) if ('adam_kernel' in r[0] or 'adam_pack_kernel' in r[0])]
if ntrain:
    marks = marks[:ntrain]
steps = [(marks[i] + 1, marks[i + 1] + 1) for i in range(skip, len(marks) - 1)]
n = len(steps)
perq = defaultdict(lambda: [0.0, 0, defaultdict(float)])
phase = defaultdict(float)
for b, e in steps:
    seg = rows[b:e]
    t0 = rows[b - 1][2]
    lov = [r for r in seg if 'lovasz' in r[0] or 'bce_dice' in r[0]]
    for name, s, en, q in seg:
        perq[q][0] += (en - s) / 1e3; perq[q][1] += 1
        perq[q][2][name.replace('_ZN12_GLOBAL__N_1', '').replace('.kd', '')[:48]] += (en - s) / 1e3
    if lov:
        phase['forward (step start -> loss start)'] += (lov[0][1] - t0) / 1e3
        phase['loss'] += (lov[-1][2] - lov[0][1]) / 1e3
        phase['backward + optimizer (loss end -> adam end)'] += (seg[-1][2] - lov[-1][2]) / 1e3
print('# %d steps' % n)
for k, v in phase.items():
    print('%-48s %9.1f us' % (k, v / n))
for q, (tot, cnt, names) in sorted(perq.items(), key=lambda kv: -kv[1][0]):
    print('queue %s: busy %.1f us/step, %.0f dispatches/step' % (q, tot / n, cnt / n))
    for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:12]:
        print('      %-50s %9.1f' % (k, v / n))
# the end of the step: how far the weight-gradient queue runs past the main queue's last backward kernel (what Adam waits for)
main_q = max(perq.items(), key=lambda kv: kv[1][0])[0]
tails = defaultdict(float)
for b, e in steps:
    seg = rows[b:e]
    adam = seg[-1]
    last_main = max(r[2] for r in seg[:-1] if r[3] == main_q)
    side = [r for r in seg[:-1] if r[3] != main_q]
    last_side = max(r[2] for r in side) if side else last_main
    tails['main queue: last kernel end -> adam start'] += (adam[1] - last_main) / 1e3
    tails['other queues: last kernel end -> adam start'] += (adam[1] - last_side) / 1e3
    tails['other queues run past the main queue by'] += (last_side - last_main) / 1e3
for k, v in tails.items():
    print('%-48s %9.1f us' % (k, v / n))
b, e = steps[-1]
seg = rows[b:e]
t_end = seg[-1][2]
print('# last 40 dispatches of the last step (us before the end of Adam): queue start end name')
for name, s, en, q in seg[-40:]:
    print('  q%-3s %8.1f %8.1f  %s' % (q, (s - t_end) / 1e3, (en - t_end) / 1e3, name.replace('_ZN12_GLOBAL__N_1', '').replace('.kd', '')[:60]))
