#!/bin/bash
# The DSL iteration with the RLA_ResNet backbone (configs[2] as shipped), asynchronous teacher refresh: what each component costs
# (launches skipped through DSL_TUNE=lib.skip_kinds - results are wrong, only the clock is read).  Kinds: 17 RLA ops (avg pool, BN-tanh, record
# sums, BN post pass), 3 / 4 GroupNorm, 2 / 14 / 19 weight gradients, 18 packs, 6 sum2x2.
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env $1 timeout 300 python tools/bench_dsl_variant.py 1 1 $2 2>/dev/null | tail -1 | python -c "
import sys, re
l = sys.stdin.readline(); m = re.search(r\"'ms_per_iter[a-z_]*': ([0-9.]+)\", l); print(m.group(1) if m else l[:120])"; }
for rep in 1 2; do
  echo "[baseline async] $(run X=1 1)   [baseline sync] $(run X=1 0)"
  echo "[no RLA small ops (kind 17)] $(run DSL_TUNE=lib.skip_kinds=0x20000 1)"
  echo "[no GroupNorm] $(run DSL_TUNE=lib.skip_kinds=0x18 1)"
  echo "[no weight gradients] $(run DSL_TUNE=lib.skip_kinds=0x84004 1)"
  echo "[no multi weight gradients] $(run DSL_TUNE=lib.skip_kinds=0x80000 1)"
  echo "[no grouped weight gradients (14)] $(run DSL_TUNE=lib.skip_kinds=0x4000 1)"
  echo "[no single weight gradients (2)] $(run DSL_TUNE=lib.skip_kinds=0x4 1)"
  echo "[no packs] $(run DSL_TUNE=lib.skip_kinds=0x40000 1)"
done
