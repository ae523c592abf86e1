"""The fp8-towers step alone (bench.fp8_step_timing) - for a rocprofv3 kernel trace of configs[4]'s slice:
   cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_fp8 -o fp8 -- python $R/tools/fp8_probe.py [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from dsl_amd import detectors  # noqa: E402,F401

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
b = bench.synth_batch(0, 2)
print(json.dumps(bench.fp8_step_timing(b, steps=steps, warm=4)))
