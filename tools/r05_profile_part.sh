#!/bin/bash
# the rocprofv3 part of tools/r05_evidence.sh alone (C2 kernel stats / timeline / per-queue view / in-step class times + the two PMC passes)
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; O=gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/$O/prof_c2
rocprofv3 --kernel-trace --stats -d $R/$O/prof_c2 -o r1 -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-iou --no-configs > $R/$O/prof_c2.log 2>&1
cd $R
DB=$(find $O/prof_c2 -name "*.db" | head -1)
python tools/prof_summary.py $DB > $O/${tag}_c2_bf16_r34hyper_b32_kernel_stats.txt
python tools/prof_timeline.py $DB 8 25 > $O/${tag}_c2_bf16_r34hyper_b32_timeline.txt
python tools/prof_streams.py $DB 8 25 > $O/${tag}_c2_bf16_r34hyper_b32_streams.txt
python tools/prof_instep.py $DB 8 25 ${SALT_COMMIT:-unrecorded} > $O/${tag}_instep.json
rm -rf $O/prof_c2
bash tools/pmc_traffic.sh $tag > /dev/null 2>&1
cp $O/pmc_traffic_${tag}.json $O/${tag}_pmc_traffic.json
head -c 600 $O/${tag}_instep.json; echo; head -c 400 $O/${tag}_pmc_traffic.json; echo; head -5 $O/${tag}_c2_bf16_r34hyper_b32_streams.txt
