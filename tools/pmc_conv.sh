#!/bin/bash
# rocprofv3 PMC pass over the conv micro-benchmark; summarises per-kernel counters from the rocpd database
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_$1; shift
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $OUT -o p -- python tools/conv_micro.py "$@" > $OUT.log 2>&1
python tools/pmc_summary.py $OUT/p_results.db >> $OUT.log 2>&1
grep -v "rocprofv3\|simple_timer\|amdgpu" $OUT.log | tail -12
