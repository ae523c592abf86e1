cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "patch" 2>&1 | tail -2
for rep in 1 2 3; do for v in 0 1; do
  export DSL_PATCH3=$v
  python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/ab_p3_${v}.log 2>&1
  echo "DSL_PATCH3=$v $(grep -h '"value"' gpurun_out/ab_p3_${v}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done | tee gpurun_out/s2_ab_patch3.log
