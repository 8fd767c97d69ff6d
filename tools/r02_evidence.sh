#!/bin/bash
# Round-2 evidence in one call: bench line, rocprofv3 kernel stats / timeline / per-stream view of the headline config, PMC traffic,
# and the rocprof summaries of the vanilla fp32 (C1) and C4 inference configurations.   usage: tools/r02_evidence.sh <tag>
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tools/prof_run.sh ${tag}_c2 > /dev/null 2>&1
tools/prof_run.sh ${tag}_c1_vanilla_f32 --workload vanilla --dtype f32 > /dev/null 2>&1
tools/pmc_traffic.sh ${tag} > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${tag}_c4
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${tag}_c4 -o r1 -- python $R/tools/bench_c4.py --steps 4 --warmup 2 > $R/gpurun_out/prof_${tag}_c4.log 2>&1
cd $R
python tools/prof_summary.py $(find gpurun_out/prof_${tag}_c4 -name "*.db" | head -1) > gpurun_out/summary_${tag}_c4.txt
rm -rf gpurun_out/prof_${tag}_c4
cut -c1-300 gpurun_out/${tag}_bench.json
