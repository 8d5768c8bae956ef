#!/usr/bin/env python3
"""Development aid: the pytest suite against another build of the device library (tests that spawn ranks load the
tree's own build in the children).
    python scripts/suite_with_lib.py build_variants/libqdtrace_x.so [pytest arguments]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from raytracing_amd import device
    device.use_library(os.path.abspath(sys.argv[1]))
    import pytest
    sys.exit(pytest.main(sys.argv[2:]))
