R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_c4
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_c4 -o r1 -- python $R/tools/bench_c4.py > $R/gpurun_out/prof_c4.log 2>&1
cd $R
f=$(find gpurun_out/prof_c4 -name "*.db" | head -1)
python tools/prof_summary.py $f > gpurun_out/summary_c4.txt
rm -rf gpurun_out/prof_c4
head -30 gpurun_out/summary_c4.txt
