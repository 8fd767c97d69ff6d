#!/bin/bash
# rocprofv3 PMC passes over the weight-gradient micro-benchmark; summarises per-kernel counters from the rocpd database
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmcw
rm -rf $OUT*
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d ${OUT}1 -o p -- python tools/wgrad_micro.py "$@" > ${OUT}1.log 2>&1
python tools/pmc_summary.py ${OUT}1/p_results.db | grep -A12 "wgrad_fast\|conv_wgrad" > $OUT.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS -d ${OUT}2 -o p -- python tools/wgrad_micro.py "$@" > ${OUT}2.log 2>&1
python tools/pmc_summary.py ${OUT}2/p_results.db | grep -A12 "wgrad_fast\|conv_wgrad" >> $OUT.txt 2>&1
cat $OUT.txt
rm -rf ${OUT}1 ${OUT}2
