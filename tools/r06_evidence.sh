#!/bin/bash
# Round-6 evidence in one gpurun call: parity counts of the BASELINE-config tests (incl. the new fp32 C2-shape eval test), the bench line (all
# configurations, fit_e2e, DP leg), rocprofv3 kernel stats / timeline / per-queue view / in-step class times of C2, kernel stats of C1 and C4,
# PMC traffic (separate --pmc passes), per-operator tables, DP-path overhead.
#   SALT_COMMIT=$(git rev-parse --short HEAD) tools/r06_evidence.sh r06
tag=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
O=gpurun_out
rm -f $O/${tag}_parity_counts.json
SALT_PARITY_COUNTS=$R/$O/${tag}_parity_counts.json timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_fused_step.py tests/test_gpu_baseline_configs.py \
  tests/test_gpu_convergence.py -m gpu -q -k "c1_shape or c2_shape or c4 or 20_steps" > $O/${tag}_parity_tests.log 2>&1
tail -3 $O/${tag}_parity_tests.log
python bench.py > $O/${tag}_c2_bf16_r34hyper_b32_bench.json 2> $O/${tag}_bench.err
cut -c1-400 $O/${tag}_c2_bf16_r34hyper_b32_bench.json
prof() {   # prof <name> <cmd...>: rocprofv3 --kernel-trace --stats -> db path in $DB
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/$O/prof_$1
  local n=$1; shift
  rocprofv3 --kernel-trace --stats -d $R/$O/prof_$n -o r1 -- "$@" > $R/$O/prof_$n.log 2>&1
  cd $R
  DB=$(find $O/prof_$n -name "*.db" | head -1)
}
prof c2 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-iou --no-configs
python tools/prof_summary.py $DB > $O/${tag}_c2_bf16_r34hyper_b32_kernel_stats.txt
python tools/prof_timeline.py $DB 8 25 > $O/${tag}_c2_bf16_r34hyper_b32_timeline.txt
python tools/prof_streams.py $DB 8 25 > $O/${tag}_c2_bf16_r34hyper_b32_streams.txt
python tools/prof_instep.py $DB 8 25 ${SALT_COMMIT:-unrecorded} > $O/${tag}_instep.json
rm -rf $O/prof_c2
prof c1 python $R/bench.py --workload vanilla --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline --no-iou --no-configs
python tools/prof_summary.py $DB > $O/${tag}_c1_vanilla_f32_b32_kernel_stats.txt
rm -rf $O/prof_c1
prof c4 python $R/tools/bench_c4.py
python tools/prof_summary.py $DB > $O/${tag}_c4_r152_256_b16_tta4_kernel_stats.txt
rm -rf $O/prof_c4
bash tools/pmc_traffic.sh $tag > /dev/null 2>&1
cp $O/pmc_traffic_${tag}.json $O/${tag}_pmc_traffic.json
python tools/op_profile.py --top 400 > $O/${tag}_c2_ops.txt 2>/dev/null
python tools/c4_ops.py --top 60 > $O/${tag}_c4_ops.txt 2>/dev/null
python tools/dp_overhead.py 30 > $O/${tag}_dp_overhead.json 2>/dev/null
python bench.py --host-contention 2>/dev/null | grep "^{" > $O/${tag}_host_contention.json
for r in 1 2; do for v in 0 1; do SALT_FWD_BN_FOLD=$v python tools/fwd_fold_probe.py 2>&1 | tail -1; done; done > $O/${tag}_fwd_fold_probe.txt
ls $O | grep "^${tag}_"
