#!/usr/bin/env python
"""The pytest fuzz slice (tests/test_gpu_fuzz.py) with the same per-family seeds but a longer budget: visits the cases a faster box would reach.
    python tools/fuzz_deep.py [seconds per family, default 25] [seed base, default 2026]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import fuzz_gpu
from tests.test_gpu_fuzz import NAMES

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0
base = int(sys.argv[2]) if len(sys.argv) > 2 else 2026          # 2026 = the seeds tests/test_gpu_fuzz.py uses
fams = dict(fuzz_gpu.FAMILIES)
bad = 0
for i, name in enumerate(NAMES):
    fuzz_gpu.reseed(base + i)
    bad += not fuzz_gpu.run(name, fams[name], budget=budget)
sys.exit(1 if bad else 0)
