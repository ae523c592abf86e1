#!/bin/bash
# L2 / LDS-side counters of one conv shape: tools/pmc_conv_mem.sh SHAPE [force_cfg] > out.txt   (companion of pmc_conv.sh)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
export TMPDIR=/tmp
S=$1; F=${2:-0}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE"
P3="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM"
P4="TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"
P5="TCC_REQ_sum TCC_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"
i=0
for P in "$P1" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rm -rf /tmp/pmcm_$i
  rocprofv3 --pmc $P -d /tmp/pmcm_$i -o p -- python $R/tools/prof_conv.py $S $F > /tmp/pmcm_$i.log 2>&1
  DB=$(find /tmp/pmcm_$i -name '*_results.db' | head -1)
  if [ -z "$DB" ]; then echo "pass $i failed:"; tail -5 /tmp/pmcm_$i.log; continue; fi
  python $R/tools/pmc_summary.py $DB | grep -A10 "conv_"
done
