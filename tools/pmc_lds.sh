#!/bin/bash
# LDS bank-conflict counters of the step per kernel family (one --pmc pass).  usage: tools/pmc_lds.sh <tag> [ENV=..]
tag=${1:-lds}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/pmc_lds_$tag; rm -rf $d
env "$@" timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $d -o p -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-iou --no-configs > $R/gpurun_out/pmc_lds_$tag.log 2>&1
cd $R
SALT_PMC_TRAIN_STEPS=9 python tools/pmc_sq.py $(find $d -name "*.db" | head -1) > gpurun_out/${tag}_pmc_lds.json
python -c "
import json; d=json.load(open('gpurun_out/${tag}_pmc_lds.json'))
for k,v in d['per_launch'].items():
    if 'conv' in k: print(k, v.get('SQ_LDS_IDX_ACTIVE'), v.get('SQ_LDS_BANK_CONFLICT'), v.get('lds_bank_conflict_frac_of_lds_active'))"
rm -rf $d
