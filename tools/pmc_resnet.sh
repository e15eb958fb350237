#!/bin/bash
# PMC passes over the wide-network chain (development aid).  usage: tools/pmc_resnet.sh OUTDIR
out=${1:-gpurun_out/pmc_resnet}; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $R/$out/sq_counters.txt
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/$out/p$i -o p -- python3 $R/tools/kbench.py resnetchain > $R/$out/p$i.log 2>&1
done
