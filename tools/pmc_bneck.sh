#!/bin/bash
# SQ counters of the fused bottleneck forward kernel and the three launches it replaces (three --pmc passes of tools/bench_bneck.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD"
P3="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INSTS_BRANCH"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rm -rf /tmp/pmcb_$i
  rocprofv3 --pmc $P -d /tmp/pmcb_$i -o p -- python $R/tools/bench_bneck.py > /tmp/pmcb_$i.log 2>&1
  DB=$(find /tmp/pmcb_$i -name '*_results.db' | head -1)
  python $R/tools/pmc_summary.py $DB | grep -A9 "bneck_fwd"
done
