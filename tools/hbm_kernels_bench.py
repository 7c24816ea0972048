#!/usr/bin/env python3
"""The HBM-bound kernels of the path at tracker sizes (bench.py's `hbm_kernels` leg) as a stand-alone program for rocprofv3:
   python3 tools/hbm_kernels_bench.py   -> prints the same table bench.py puts into its JSON line"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from busca_amd import _lib
import bench

ctx = _lib.Context(0)
print(json.dumps(bench.hbm_kernels(ctx, torch.device("cuda", 0)), indent=1))
