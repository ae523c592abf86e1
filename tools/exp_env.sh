#!/bin/bash
# A/B runs of bench.py under environment settings: tools/exp_env.sh "VAR=a VAR2=b" "VAR=c" ...   ("-" = defaults)
# prints img/s and ms/step per setting (30 timed steps, no instrumentation passes)
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then cfg=""; fi
  env $cfg python bench.py --no-prof --no-cpu-baseline --no-dsl --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline()); print('%-50s' % '$cfg', j['value'], j['ms_per_step'])"
done
