#!/bin/bash
# "What would the step gain if component X cost nothing?"  bench.py with one component's launches skipped through DSL_TUNE (WRONG results - only the
# clock is read), round-robin in one box.  Kinds: 3 GN fwd, 4 GN bwd, 2 / 14 / 19 weight gradients (single / group / multi), 18 dgrad packs.
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env $1 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-prof --no-dsl 2>/dev/null | grep '"value"' | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])"; }
for rep in 1 2; do
  echo "[baseline] $(run "X=1")"
  echo "[no GN forward (kind 3)] $(run "DSL_TUNE=lib.skip_kinds=0x8")"
  echo "[no GN backward (kind 4)] $(run "DSL_TUNE=lib.skip_kinds=0x10")"
  echo "[no GN at all] $(run "DSL_TUNE=lib.skip_kinds=0x18")"
  echo "[no weight gradients at all (2, 14, 19)] $(run "DSL_TUNE=lib.skip_kinds=0x84004")"
  echo "[no tower weight-gradient group (kind 14 on side stream 1)] $(run "DSL_TUNE=lib.skip_kinds=0x40004000")"
  echo "[no multi weight gradients (kind 19)] $(run "DSL_TUNE=lib.skip_kinds=0x80000")"
  echo "[no optimizer kernels] $(run "DSL_TUNE=skip=sgd")"
  echo "[no dgrad packs (kind 18)] $(run "DSL_TUNE=lib.skip_kinds=0x40000")"
  echo "[no layer1 convolutions in the prefix] $(run "DSL_TUNE=skip=prefix")"
  echo "[no optimizer, no packs, no layer1] $(run "DSL_TUNE=skip=sgd+prefix,lib.skip_kinds=0x40000")"
done
