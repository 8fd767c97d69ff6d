#!/usr/bin/env python
"""Per-kernel SQ counters of the training step from rocprofv3 --pmc databases (one or more passes): what the waves of each kernel
family did with their cycles.  Counter meanings as MI355X_MICROARCH.md gives them (rocprofv3 PMC slots): SQ_WAVE_CYCLES, SQ_WAIT_ANY
(wave parked on s_waitcnt / barrier), SQ_WAIT_INST_ANY (issue stall), SQ_ACTIVE_INST_ANY count quad-cycles and are disjoint parts of
the wave cycles; SQ_VALU_MFMA_BUSY_CYCLES counts cycles the matrix pipe is busy; SQ_LDS_BANK_CONFLICT = extra LDS cycles of
SQ_LDS_IDX_ACTIVE.  Ratios only - absolute units differ per counter.
usage: python tools/pmc_sq.py <db> [<db> ...]"""
import json
import os
import sqlite3
import sys
from collections import defaultdict

FAMILIES = ('conv_wgrad_ls_kernel', 'conv_wgrad_fast', 'conv_ws_kernel', 'conv1x1_ls_kernel', 'conv_ls_kernel', 'conv_mfma_kernel', 'conv_glds_kernel',
            'bn_bwd_apply', 'affine_act', 'wgrad_reduce', 'adam_pack_kernel', 'hyper_stencil_fwd', 'hyper_stencil_bwd', 'scse_bwd1', 'scse_apply', 'gap_partial',
            'lovasz', 'bilinear_fwd', 'bilinear_bwd', 'head_bn')


def read(path):
    c = sqlite3.connect(path)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    tb = lambda s: [x for x in t if s in x][0]
    kd, ks, pe, pi = tb('kernel_dispatch'), tb('kernel_symbol'), tb('pmc_event'), tb('info_pmc')
    rows = list(c.execute("select s.kernel_name, d.id, p.name, sum(e.value), d.start from %s e join %s p on e.pmc_id=p.id join %s d on e.event_id=d.event_id "
                          "join %s s on d.kernel_id=s.id group by d.id, p.name order by d.start" % (pe, pi, kd, ks)))
    disp = {}
    for name, did, cname, v, start in rows:
        disp.setdefault(did, [name, start, {}])[2][cname] = v
    order = sorted(disp.values(), key=lambda r: r[1])
    marks = [i for i, r in enumerate(order) if 'adam_pack_kernel' in r[0] or 'adam_kernel' in r[0]]
    ntrain = int(os.environ.get('SALT_PMC_TRAIN_STEPS', '0'))
    if ntrain:
        marks = marks[:ntrain]
    lo, hi = marks[1] + 1, marks[-1] + 1
    steps = len(marks) - 2
    agg = defaultdict(lambda: [0, defaultdict(float)])
    for name, _, cs in order[lo:hi]:
        k = name.replace('_ZN12_GLOBAL__N_1', '').replace('.kd', '')
        for f in FAMILIES:
            if f in k:
                k = f
                break
        else:
            k = 'other'
        agg[k][0] += 1
        for cn, v in cs.items():
            agg[k][1][cn] += v
    return steps, agg


def main(paths):
    merged, steps = {}, None
    for p in paths:
        s, agg = read(p)
        steps = s if steps is None else min(steps, s)
        for k, (n, cs) in agg.items():
            m = merged.setdefault(k, {'launches_per_step': round(n / s, 1)})
            for cn, v in cs.items():
                m[cn] = v / n                       # per launch
    out = {'source': 'rocprofv3 --kernel-trace --pmc <SQ counters> (own passes, no other tracing) over bench.py bf16 r34_hyper batch 32',
           'commit': os.environ.get('SALT_COMMIT', 'unrecorded'), 'steps': steps, 'per_launch': {}}
    for k, m in sorted(merged.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0) * kv[1]['launches_per_step']):
        wc = m.get('SQ_WAVE_CYCLES')
        r = {'launches_per_step': m['launches_per_step']}
        for cn in sorted(m):
            if cn != 'launches_per_step':
                r[cn] = round(m[cn], 1)
        if wc:
            for cn, label in (('SQ_WAIT_ANY', 'wait_any_frac_of_wave_cycles'), ('SQ_WAIT_INST_ANY', 'issue_stall_frac_of_wave_cycles'),
                              ('SQ_ACTIVE_INST_ANY', 'active_frac_of_wave_cycles'), ('SQ_WAIT_INST_LDS', 'lds_issue_stall_frac_of_wave_cycles')):
                if cn in m:
                    r[label] = round(m[cn] / wc, 4)
        if m.get('SQ_LDS_IDX_ACTIVE'):
            r['lds_bank_conflict_frac_of_lds_active'] = round(m.get('SQ_LDS_BANK_CONFLICT', 0.0) / m['SQ_LDS_IDX_ACTIVE'], 4)
        if m.get('SQ_BUSY_CYCLES') and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
            r['mfma_busy_over_sq_busy_cycles'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_BUSY_CYCLES'], 4)
        out['per_launch'][k] = r
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1:])
