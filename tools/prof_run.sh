#!/bin/bash
# usage: tools/prof_run.sh <tag> [bench args...]   -> gpurun_out/timeline_<tag>.txt, summary_<tag>.txt, step_<tag>.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o r1 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-iou --no-configs "$@" > $R/gpurun_out/prof_$tag.log 2>&1
cd $R
f=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
python tools/prof_timeline.py $f 8 25 > gpurun_out/timeline_$tag.txt
python tools/prof_summary.py $f > gpurun_out/summary_$tag.txt
python tools/prof_step.py $f 20 > gpurun_out/step_$tag.txt; python tools/prof_streams.py $f 8 25 > gpurun_out/streams_$tag.txt
grep '"metric"' gpurun_out/prof_$tag.log | cut -c1-330
rm -rf gpurun_out/prof_$tag
