#!/usr/bin/env python
"""Dispatch-by-dispatch listing of one optimizer step from a rocprofv3 rocpd database (start offset, duration, kernel, grid, queue)."""
import sqlite3
import sys


def main(path, which=-3):
    c = sqlite3.connect(path)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if 'kernel_dispatch' in x][0]
    ks = [x for x in t if 'kernel_symbol' in x][0]
    rows = list(c.execute("select s.kernel_name,d.start,d.end,d.grid_size_x,d.grid_size_y,d.workgroup_size_x,d.queue_id from %s d join %s s "
                          "on d.kernel_id=s.id order by d.start" % (kd, ks)))
    marks = [i for i, r in enumerate(rows) if ('adam_kernel' in r[0] or 'adam_pack_kernel' in r[0])]
    b, e = marks[which] + 1, marks[which + 1] + 1
    t0 = rows[b][1]
    for r in rows[b:e]:
        n = r[0].replace('_ZN12_GLOBAL__N_1', '')[:44]
        print('%9.1f %8.1f %-44s g=%d,%d q=%s' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, n, r[3] // max(r[5], 1), r[4], r[6]))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else -3)       # which Adam-delimited interval (index into the Adam launches)
