#!/bin/bash
# Round-3 evidence in one call (gpurun): parity counts of the BASELINE-config tests, the default bench line, rocprofv3 kernel stats /
# timeline / per-stream view of C2, kernel stats of C1 and C4, PMC traffic (separate --pmc passes).  The caller passes the commit:
#   SALT_COMMIT=$(git rev-parse --short HEAD) tools/r03_evidence.sh r03
tag=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
rm -f gpurun_out/${tag}_parity_counts.json
SALT_PARITY_COUNTS=$R/gpurun_out/${tag}_parity_counts.json timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_fused_step.py tests/test_gpu_baseline_configs.py \
  -m gpu -q -k "c1_shape or c2_shape or c4" > gpurun_out/${tag}_parity_tests.log 2>&1
tail -3 gpurun_out/${tag}_parity_tests.log
bash tools/r02_evidence.sh ${tag}
python tools/op_profile.py --top 400 > gpurun_out/${tag}_ops.txt 2>/dev/null
ls gpurun_out | grep ${tag}
