#!/bin/bash
# PMC passes of the training step (separate rocprofv3 runs, --pmc only): HBM read / write requests and the SQ view of the kernels
# (MFMA busy cycles, wait / issue split, LDS bank conflicts).  TAG = output prefix under gpurun_out/
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
export TMPDIR=/tmp
TAG=${1:-r02}
CMD="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-prof --no-dsl"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $R/gpurun_out/pmc_${TAG}_$c -o p -- $CMD > $R/gpurun_out/pmc_${TAG}_$c.log 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  -d $R/gpurun_out/pmc_${TAG}_SQ -o p -- $CMD > $R/gpurun_out/pmc_${TAG}_SQ.log 2>&1
cd $R
# the algorithmic bytes per launch of the kernel classes come from an (instrumented) bench.py run's roofline object
python bench.py --no-cpu-baseline --no-dsl > gpurun_out/${TAG}_bench_prof.log 2>&1
F=$(find gpurun_out/pmc_${TAG}_FETCH_SIZE -name '*_results.db' | head -1)
W=$(find gpurun_out/pmc_${TAG}_WRITE_SIZE -name '*_results.db' | head -1)
S=$(find gpurun_out/pmc_${TAG}_SQ -name '*_results.db' | head -1)
python tools/pmc_traffic.py $F $W gpurun_out/${TAG}_traffic.json gpurun_out/${TAG}_bench_prof.log > gpurun_out/${TAG}_pmc_traffic.txt 2>&1
python tools/pmc_summary.py $S gpurun_out/${TAG}_pmc_sq.json > gpurun_out/${TAG}_pmc_sq.txt 2>&1
rm -rf gpurun_out/pmc_${TAG}_FETCH_SIZE gpurun_out/pmc_${TAG}_WRITE_SIZE gpurun_out/pmc_${TAG}_SQ
head -12 gpurun_out/${TAG}_pmc_traffic.txt
head -40 gpurun_out/${TAG}_pmc_sq.txt
