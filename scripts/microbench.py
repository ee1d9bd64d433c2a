#!/usr/bin/env python3
"""Kernel micro-benchmarks on one MI355X (HIP events on torch's current stream, which is the stream the C ABI gets)."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from i2sdf_amd.config import NetConfig, synthetic_conf
from i2sdf_amd.engine import RenderEngine


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=131072)
    a = ap.parse_args()
    cfg = NetConfig.from_conf(synthetic_conf())
    eng = RenderEngine(cfg)
    flat = eng.layout.init_flat(torch.Generator().manual_seed(0)).cuda()
    flat += torch.randn_like(flat) * 0.01
    eng.pack(flat)
    print("pack ms", timeit(lambda: eng.pack(flat)))
    for M in (a.M, 99328, 1 << 20):
        x = (torch.rand(M, 3, device="cuda") * 2 - 1) * 2.5
        ms = timeit(lambda: eng.sdf_forward(x))
        macs = 39 * 256 + 256 * 256 * 2 + 217 * 256 + 256 * 256 * 4 + 256
        print(f"sdf_forward(sdf only) M={M}: {ms:.3f} ms  {2*macs*M/ms/1e9:.1f} TFLOP/s (algorithmic)")
        ms = timeit(lambda: eng.sdf_forward(x, True))
        macs += 256 * 256
        print(f"sdf_forward(full)     M={M}: {ms:.3f} ms  {2*macs*M/ms/1e9:.1f} TFLOP/s (algorithmic)")


if __name__ == "__main__":
    main()
