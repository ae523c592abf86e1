"""One variant of bench.py's dsl_iteration timing (refresh, rla, async as 0/1): python tools/bench_dsl_variant.py 1 0 0"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dsl_amd import detectors  # noqa: F401  (registers the model classes)
v = tuple(bool(int(a)) for a in sys.argv[1:4])
print(sys.argv[1:4], bench.dsl_iteration_timing(variants=(v,)))
