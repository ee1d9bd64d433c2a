"""Kernel times of BASELINE cfg 3 (synthetic_light_mask.yml networks): run under `rocprofv3 --kernel-trace --stats` to see what the
light-mask head's fp32-MFMA kernels cost next to the bf16x3 ones.
    rocprofv3 --kernel-trace --stats -d OUT -o cfg3 --output-format csv -- python scripts/ab/cfg3_kernels.py"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench

args = argparse.Namespace(fused_adam=1, dp_transport="auto")
w = bench.Workload(args, torch.device("cuda:0"), 0, 1, light=True)
r = w.run(1024, 1234, 2, 10, 3, timing=True, windows=3)
print("cfg3 step %.4f ms" % (r["dt"] / 10 * 1e3), flush=True)
