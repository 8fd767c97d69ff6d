#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc databases (FETCH_SIZE pass, WRITE_SIZE pass).

Counter units and corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE
tallies the 128-byte requests of wide coalesced reads at 64 bytes, so read bytes = FETCH_SIZE * 1024 * 2.  WRITE_SIZE is
uncalibrated in the guide; the Adam kernel (reads 4 and writes 3 fp32 arrays of known length) is printed as a calibration row."""
import json
import os
import sqlite3
import sys
from collections import defaultdict


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    tb = lambda s: [x for x in t if s in x][0]
    kd, ks, pe, pi = tb('kernel_dispatch'), tb('kernel_symbol'), tb('pmc_event'), tb('info_pmc')
    q = ("select s.kernel_name, d.id, sum(e.value) from %s e join %s p on e.pmc_id=p.id join %s d on e.event_id=d.event_id "
         "join %s s on d.kernel_id=s.id where p.name='%s' group by d.id order by d.start" % (pe, pi, kd, ks, counter))
    rows = list(c.execute(q))
    # keep the dispatches of complete optimizer steps after the first Adam launch (skips warm-up compile / first-touch effects)
    marks = [i for i, r in enumerate(rows) if ('adam_kernel' in r[0] or 'adam_pack_kernel' in r[0])]
    ntrain = int(os.environ.get('SALT_PMC_TRAIN_STEPS', '0'))          # warm-up + timed steps of the profiled run: what follows is not a training step
    if ntrain:
        marks = marks[:ntrain]
    lo, hi = marks[1] + 1, marks[-1] + 1
    steps = len(marks) - 2                                             # the intervals between marks[1] and marks[-1]: executed steps
    agg = defaultdict(lambda: [0, 0.0])
    for name, _, v in rows[lo:hi]:
        k = name.replace('_ZN12_GLOBAL__N_1', '').replace('.kd', '')
        if 'conv_wgrad' in k:                     # every weight-gradient kernel (conv_wgrad_ls_kernel, the fast / generic ones) is one class
            k = 'conv_wgrad_kernel'
        for base in ('conv_ws_kernel', 'conv1x1_ls_kernel', 'conv1x1_xs_kernel', 'conv_ls_kernel', 'conv_thin_kernel', 'conv_mfma_kernel', 'conv_glds_kernel', 'conv_wgrad_kernel', 'bn_bwd_reduce', 'bn_bwd_apply', 'bn_bwd_finalize', 'bn_finalize', 'affine_act',
                     'wgrad_reduce', 'adam_kernel', 'adam_pack_kernel', 'hyper_stencil', 'pad_fold', 'lovasz', 'bilinear_fwd', 'bilinear_bwd', 'pack_batched', 'head1x1', 'head_bn', 'scse', 'se_', 'gap_partial'):
            if base in k:
                k = base
                break
        agg[k][0] += 1
        agg[k][1] += v
    return steps, agg


def main(fetch_db, write_db):
    sf, f = per_kernel(fetch_db, 'FETCH_SIZE')
    sw, w = per_kernel(write_db, 'WRITE_SIZE')
    out = {'source': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) over bench.py bf16 r34_hyper batch 32',
           'corrections': 'bytes_read = FETCH_SIZE KiB * 1024 * 2 (gfx950 wide-read correction); bytes_written = WRITE_SIZE KiB * 1024',
           'commit': os.environ.get('SALT_COMMIT', 'unrecorded'),      # the gpurun snapshot has no .git: the caller passes `git rev-parse --short HEAD` (+dirty)
           'steps': sf, 'kernels': {}}
    tot_r = tot_w = 0.0
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, [0, 0])[1] * 2 + w.get(k, [0, 0])[1])):
        n = f.get(k, w.get(k))[0]
        rb = f.get(k, [0, 0.0])[1] * 1024 * 2
        wb = w.get(k, [0, 0.0])[1] * 1024
        tot_r += rb
        tot_w += wb
        out['kernels'][k] = {'launches_per_step': round(n / sf, 1), 'read_MB_per_launch': round(rb / n / 1e6, 3),
                             'write_MB_per_launch': round(wb / n / 1e6, 3), 'MB_per_step': round((rb + wb) / sf / 1e6, 1)}
    out['step_total_MB'] = {'read': round(tot_r / sf / 1e6, 1), 'write': round(tot_w / sf / 1e6, 1)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
