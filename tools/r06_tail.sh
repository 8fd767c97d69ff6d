#!/bin/bash
# per-queue view + end-of-step tail of C2 (rocprofv3 kernel trace); usage: tools/r06_tail.sh <tag> [ENV=...]
tag=${1:-tail}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out; cd $R; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $R/$O/prof_$tag
env "$@" rocprofv3 --kernel-trace --stats -d $R/$O/prof_$tag -o r1 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-iou --no-configs > $R/$O/prof_$tag.log 2>&1
cd $R
DB=$(find $O/prof_$tag -name "*.db" | head -1)
python tools/prof_streams.py $DB 8 25 > $O/${tag}_streams.txt
rm -rf $O/prof_$tag
tail -50 $O/${tag}_streams.txt
