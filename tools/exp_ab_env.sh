#!/bin/bash
# A/B of environment settings in one box, round-robin: tools/exp_ab_env.sh "A=1 B=2" "-" "C=3" ...  ("-" = no setting); three rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do for v in "$@"; do
  if [ "$v" = "-" ]; then e=""; else e="$v"; fi
  env $e python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/ab_env.log 2>&1
  echo "[$v] $(grep -h '"value"' gpurun_out/ab_env.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
