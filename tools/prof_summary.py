#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max / share.
usage: python tools/prof_summary.py gpurun_out/prof_x/r1_results.db [> profiles/rNN_x.txt]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if 'kernel_dispatch' in x][0]
    ks = [x for x in t if 'kernel_symbol' in x][0]
    q = ("select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, max(d.end-d.start)/1e3, "
         "max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
         "from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (kd, ks))
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows)
    print('# rocprofv3 --kernel-trace --stats  (%s)' % path)
    print('# total kernel time %.1f us over %d dispatches' % (tot, sum(r[1] for r in rows)))
    print('%-78s %7s %12s %10s %9s %10s %6s %5s %5s %5s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct', 'vgpr', 'agpr', 'sgpr', 'lds'))
    for r in rows:
        name = r[0].replace('_ZN12_GLOBAL__N_1', '').replace('.kd', '')
        print('%-78s %7d %12.1f %10.2f %9.2f %10.2f %5.1f%% %5d %5d %5d %7d' % (name[:78], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0))


if __name__ == '__main__':
    main(sys.argv[1])
