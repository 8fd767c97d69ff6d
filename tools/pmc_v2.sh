#!/bin/bash
# SQ counters of the conv micro-benchmark for the shipped library and ablation variants: tools/pmc_v2.sh "B Cin H W Cout" cfg [variants...]
cd /tmp; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
shape=$1; cfg=$2; shift; shift
V=open-solution-salt-identification_amd/csrc/_variants
for v in full "$@"; do
  OUT=gpurun_out/pmcv2_$v
  rm -rf $OUT
  lib=""; [ $v != full ] && lib=$R/$V/libsaltnet_hip.$v.so
  (cd /tmp && SALT_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $R/$OUT -o p -- python $R/tools/conv_micro.py $shape 3 1 bf16 10 $cfg > $R/$OUT.log 2>&1)
  echo "== $v"; python tools/pmc_summary.py $(find $OUT -name "*.db" | head -1) 2>&1 | grep -A8 conv_glds
  (cd /tmp && SALT_LIB=$lib rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $R/${OUT}b -o p -- python $R/tools/conv_micro.py $shape 3 1 bf16 10 $cfg > $R/${OUT}b.log 2>&1)
  python tools/pmc_summary.py $(find ${OUT}b -name "*.db" | head -1) 2>&1 | grep -A8 conv_glds
  rm -rf $OUT ${OUT}b
done
