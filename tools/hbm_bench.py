#!/usr/bin/env python3
"""The HBM-bound north-star kernels at their BASELINE shapes: GB/s and fraction of the 8 TB/s peak (bench.py prints the same
table in its JSON line).   python tools/hbm_bench.py     (GPU box)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

print(json.dumps(bench.hbm_kernel_table(torch.device("cuda:0")), indent=1))
