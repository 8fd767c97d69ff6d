#!/usr/bin/env python
"""Step timeline from a rocprofv3 rocpd database: wall span per optimizer step, GPU-busy union, idle gaps,
and per-kernel time per step (steps are delimited by the Adam kernel).
usage: python tools/prof_timeline.py <results.db> [skip_steps] [train_steps]
train_steps = warm-up + timed steps of the profiled bench.py run: only intervals between those Adam launches count as steps (the
roofline / eval passes behind them are not training steps)."""
import sqlite3
import sys
from collections import defaultdict


def main(path, skip=8, ntrain=0):
    c = sqlite3.connect(path)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if 'kernel_dispatch' in x][0]
    ks = [x for x in t if 'kernel_symbol' in x][0]
    rows = list(c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
    marks = [i for i, r in enumerate(rows) if ('adam_kernel' in r[0] or 'adam_pack_kernel' in r[0])]
    if ntrain:
        marks = marks[:ntrain]
    if len(marks) < skip + 2:
        skip = 0
    steps = [(marks[i] + 1, marks[i + 1] + 1) for i in range(skip, len(marks) - 1)]
    spans, busys, sums, counts = [], [], [], []
    per = defaultdict(lambda: [0, 0.0])
    gaps_hist = defaultdict(int)
    for b, e in steps:
        seg = rows[b:e]
        spans.append((seg[-1][2] - rows[b - 1][2]) / 1e3)
        cur_e = rows[b - 1][2]
        busy = 0.0
        for name, s, en in seg:
            if s > cur_e:
                gaps_hist[min(int((s - cur_e) / 1e3 // 2) * 2, 40)] += 1
            busy += max(0, en - max(s, cur_e))
            cur_e = max(cur_e, en)
            short = name.replace('_ZN12_GLOBAL__N_1', '').replace('.kd', '')[:70]
            per[short][0] += 1
            per[short][1] += (en - s) / 1e3
        busys.append(busy / 1e3)
        sums.append(sum(en - s for _, s, en in seg) / 1e3)
        counts.append(len(seg))
    n = len(steps)
    print('# %d steps: span %.1f us  busy-union %.1f us  kernel-sum %.1f us  dispatches %.0f' % (n, sum(spans) / n, sum(busys) / n, sum(sums) / n, sum(counts) / n))
    print('# idle per step %.1f us' % ((sum(spans) - sum(busys)) / n))
    print('# gap histogram (us bucket: count/step): ' + ' '.join('%d:%.1f' % (k, v / n) for k, v in sorted(gaps_hist.items())))
    print('%-72s %8s %10s %9s' % ('kernel', 'calls/st', 'us/step', 'avg_us'))
    for k, (cnt, tot) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print('%-72s %8.1f %10.1f %9.2f' % (k, cnt / n, tot / n, tot / cnt))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 0)
