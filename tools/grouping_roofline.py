#!/usr/bin/env python
"""Print the K5 grouping micro-benchmark (bench.grouping_roofline) alone: python tools/grouping_roofline.py"""
import sys, os, json
sys.path.insert(0, os.getcwd())
import torch
import bench
print(json.dumps({k: (round(v["ms"],4), round(v["GBps"])) for k, v in bench.grouping_roofline().items()}))
