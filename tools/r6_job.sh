R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/g10
for rep in 1 2 3; do for v in rla_tail=0 rla_tail=1; do
  echo "[$v] $(DSL_TUNE=$v python tools/bench_dsl_variant.py 0 1 0 2>&1 | tail -1 | sed 's/, .imgs_per_iter.*//')"
done; done > gpurun_out/g10/rla_tail_ab2.txt 2>&1
for v in rla_tail=0 rla_tail=1; do echo "[$v async refresh] $(DSL_TUNE=$v python tools/bench_dsl_variant.py 1 1 1 2>&1 | tail -1 | sed 's/, .imgs_per_iter.*//')"; done >> gpurun_out/g10/rla_tail_ab2.txt 2>&1
cat gpurun_out/g10/rla_tail_ab2.txt
