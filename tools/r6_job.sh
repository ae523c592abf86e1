R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/g5; export TMPDIR=/tmp
python tools/fp8_probe.py 30 > gpurun_out/g5/fp8_probe.log 2>&1; tail -1 gpurun_out/g5/fp8_probe.log | cut -c1-200
python tools/fp8_probe.py 30 > gpurun_out/g5/fp8_probe2.log 2>&1; tail -1 gpurun_out/g5/fp8_probe2.log | cut -c1-200
timeout 300 python -m pytest tests/test_comm_proxy_gpu.py -m gpu -q > gpurun_out/g5/tests.log 2>&1; tail -4 gpurun_out/g5/tests.log
timeout 400 python bench.py > gpurun_out/g5/bench.log 2>&1
tail -1 gpurun_out/g5/bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step']); e=d['extra']; print('fp8', e['fp8_towers']['imgs_per_s'], e['fp8_towers']['ms_per_step']); print(e['comm_proxy']['ms_per_step'], e['comm_proxy']['cost_frac']); print(e['train_detector']['imgs_per_s']); print(e['dsl_iteration']['ms_per_iter'], e['dsl_iteration']['ms_per_iter_rla_backbone'])"
