R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/g9
for rep in 1 2 3; do for v in rla_tail=0 rla_tail=1; do
  echo "[$v] $(DSL_TUNE=$v python tools/bench_dsl_variant.py 0 1 0 2>&1 | tail -1)"
done; done > gpurun_out/g9/rla_tail_ab.txt 2>&1
cat gpurun_out/g9/rla_tail_ab.txt
timeout 300 python -m pytest tests/test_rla_gpu.py -m gpu -q -x -k "tail" 2>&1 | tail -3
