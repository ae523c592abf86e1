R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp
bash tools/exp_prof.sh r06 > gpurun_out/r06_exp_prof.log 2>&1
bash tools/exp_pmc.sh r06 > gpurun_out/r06_exp_pmc.log 2>&1
python bench.py > gpurun_out/r06_bench_full.log 2>&1
tail -3 gpurun_out/r06_exp_prof.log; head -5 gpurun_out/r06_pmc_traffic.txt; tail -c 300 gpurun_out/r06_bench_full.log
