#!/bin/bash
# TCC <-> fabric counters of tools/storebench's "pmc" selection, one rocprofv3 pass per group of four.
# usage (on the GPU box, from the repository root): bash tools/storebench_pmc.sh <output directory under gpurun_out/>
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_TAG_STALL_sum TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $grp -d $out/p$i -o p$i --output-format csv -- $GRAFT_REPO_ROOT/tools/bin/storebench pmc > $out/p$i.log 2>&1
  echo "pass $i done"
done
