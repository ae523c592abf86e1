#!/bin/bash
# "What would the step gain if graph region X cost nothing?"  bench.py with one region's convolution / data-gradient launches left out of
# the op lists (tuning key `skip`, engine.OpList; WRONG results - only the clock is read), round-robin in one box.  The rows the component
# ablation (tools/step_ablation.sh) does not have: the trained backbone's and the FPN's / head's chains.
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env $1 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-prof --no-dsl 2>gpurun_out/ablation_stderr.log | grep '"value"' | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])"; }
for rep in 1 2; do
  echo "[baseline] $(run "X=1")"
  for t in fwd.l2 fwd.l3 fwd.l4 fwd.fpn fwd.head bwd.head bwd.fpn bwd.l4 bwd.l3 bwd.l2; do
    echo "[no $t] $(run "DSL_TUNE=skip=$t")"
  done
  echo "[no fwd.l2,l3,l4] $(run "DSL_TUNE=skip=fwd.l2+fwd.l3+fwd.l4")"
  echo "[no bwd.l2,l3,l4] $(run "DSL_TUNE=skip=bwd.l2+bwd.l3+bwd.l4")"
  echo "[no fwd+bwd l2] $(run "DSL_TUNE=skip=fwd.l2+bwd.l2")"
  echo "[no trained backbone convs at all] $(run "DSL_TUNE=skip=fwd.l2+fwd.l3+fwd.l4+bwd.l2+bwd.l3+bwd.l4")"
done
