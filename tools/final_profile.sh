#!/bin/bash
# End-of-round evidence: rocprofv3 --kernel-trace --stats of the DEFAULT bench command + the bench line itself.
# usage: tools/final_profile.sh <tag>      -> gpurun_out/<tag>_kernel_stats.txt, <tag>_bench.json, <tag>_timeline.txt
tag=${1:-r01_final}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o r1 -- python $R/bench.py > $R/gpurun_out/${tag}_bench_under_rocprof.json 2> $R/gpurun_out/prof_$tag.log
cd $R
f=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
python tools/prof_summary.py $f > gpurun_out/${tag}_kernel_stats.txt
python tools/prof_timeline.py $f > gpurun_out/${tag}_timeline.txt
find gpurun_out/prof_$tag -name "*stats*.csv" | head -3
rm -rf gpurun_out/prof_$tag
cut -c1-400 gpurun_out/${tag}_bench.json
