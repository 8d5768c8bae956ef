#!/usr/bin/env python3
"""Development aid: bench.py with another build of the device library (A/B of kernel variants in one GPU session).
    python scripts/ab_bench.py build_variants/libqdtrace_x.so [bench.py arguments]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import device  # noqa: E402

device.use_library(os.path.abspath(sys.argv[1]))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
