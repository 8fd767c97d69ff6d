#!/usr/bin/env python
"""Per-class kernel time INSIDE the training step from a rocprofv3 rocpd database (both queues running, as the step executes them) -
the number next to bench.py's isolated per-launch roofline (report the in-step figure beside the isolated one).
usage: python tools/prof_instep.py <db> <skip> <train_steps> [commit]   -> JSON on stdout (committed as profiles/rNN_instep.json)"""
import json
import sqlite3
import sys
from collections import defaultdict

CLASSES = [('conv_wgrad', ('conv_wgrad',)), ('conv', ('conv_ws_kernel', 'conv_ls_kernel', 'conv1x1_ls_kernel', 'conv_thin_kernel', 'conv_mfma_kernel', 'conv_glds_kernel')),
           ('wgrad_reduce', ('wgrad_reduce',)), ('bn_bwd', ('bn_bwd',)), ('affine_act', ('affine_act',)), ('bilinear', ('bilinear',)),
           ('scse', ('scse', 'se_fc', 'gap_partial')), ('lovasz', ('lovasz',)), ('adam', ('adam_kernel', 'adam_pack_kernel')), ('pack', ('pack_batched',)),
           ('head', ('head1x1', 'head_bn')), ('hyper_stencil', ('hyper_stencil',))]


def main(path, skip, ntrain, commit):
    c = sqlite3.connect(path)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if 'kernel_dispatch' in x][0]
    ks = [x for x in t if 'kernel_symbol' in x][0]
    rows = list(c.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)))
    marks = [i for i, r in enumerate(rows) if ('adam_kernel' in r[0] or 'adam_pack_kernel' in r[0])][:ntrain]
    steps = [(marks[i] + 1, marks[i + 1] + 1) for i in range(skip, len(marks) - 1)]
    n = len(steps)
    agg = defaultdict(lambda: [0, 0.0])
    span = busy = 0.0
    for b, e in steps:
        seg = rows[b:e]
        cur = rows[b - 1][2]
        span += seg[-1][2] - rows[b - 1][2]
        for name, s, en in seg:
            busy += max(0, en - max(s, cur)); cur = max(cur, en)
            cls = 'other'
            for cname, keys in CLASSES:
                if any(k in name for k in keys):
                    cls = cname
                    break
            agg[cls][0] += 1; agg[cls][1] += (en - s) / 1e3
    out = {'source': 'rocprofv3 --kernel-trace of bench.py (bf16 r34_hyper batch 32), kernel time per training step with BOTH queues running',
           'commit': commit, 'steps': n, 'span_us_per_step': round(span / n / 1e3, 1), 'gpu_busy_union_us_per_step': round(busy / n / 1e3, 1),
           'classes': {k: {'launches_per_step': round(v[0] / n, 1), 'us_per_step': round(v[1] / n, 1)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else 'unrecorded')
