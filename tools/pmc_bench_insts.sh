#!/bin/bash
# instruction mix of the step's kernels (one PMC pass).  usage: tools/pmc_bench_insts.sh <tag>
tag=${1:-insts}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_$tag
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $R/gpurun_out/pmc_$tag -o p -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-iou > $R/gpurun_out/pmc_$tag.log 2>&1
cd $R
python - <<'PY' $(find gpurun_out/pmc_$tag -name "*.db" | head -1) > gpurun_out/pmc_${tag}_summary.txt 2>&1
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
tb = lambda s: [x for x in t if s in x][0]
kd, ks, pe, pi = tb('kernel_dispatch'), tb('kernel_symbol'), tb('pmc_event'), tb('info_pmc')
q = ("select s.kernel_name, p.name, sum(e.value), count(distinct d.id) from %s e join %s p on e.pmc_id=p.id join %s d on e.event_id=d.event_id "
     "join %s s on d.kernel_id=s.id group by s.kernel_name, p.name" % (pe, pi, kd, ks))
res = {}
for name, cn, v, n in c.execute(q):
    res.setdefault(name, {})[cn] = (v, n)
for name, d in sorted(res.items(), key=lambda kv: -kv[1].get('SQ_INSTS_VALU', (0, 1))[0]):
    n = list(d.values())[0][1]
    w = d.get('SQ_WAVES', (1, 1))[0] / n
    print(name.replace('_ZN12_GLOBAL__N_1', '')[:60], 'dispatches', n, 'waves/launch %.0f' % w)
    print('   per wave: ' + '  '.join('%s %.0f' % (k.replace('SQ_INSTS_', ''), v[0] / n / max(w, 1)) for k, v in sorted(d.items()) if k != 'SQ_WAVES'))
PY
rm -rf gpurun_out/pmc_$tag
head -24 gpurun_out/pmc_${tag}_summary.txt
