#!/usr/bin/env python3
"""Round 6: per-entry-point times of a training step by batch width -- does a point range that fits the Infinity Cache run faster per point?
   python scripts/ab/r6_small.py [rays ...]      -> step time (chained, k = 2) and un-chained entry-point times, also scaled to 1024 rays"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from r4_time import make, batch, step_fn
from i2sdf_amd import I2SDFLoss, FusedAdam

SHORT = {"i2sdf_weight_grads": "wgrad", "i2sdf_sdf_backward": "sdf_bwd", "i2sdf_sdf_forward_grad": "sdf_fwdg", "i2sdf_sample_rays": "sampler",
         "i2sdf_rgb_forward": "rgb_f", "i2sdf_rgb_backward": "rgb_b"}


def main(rays):
    net, dev = make()
    net.force_iters = 2
    loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)
    opt = FusedAdam(net, lr=5e-4, eps=1e-15)
    for B in rays:
        inp, gt = batch(B, dev)
        step = step_fn(net, loss_fn, opt, inp, gt)
        step()
        eng = net._engine_for(dev)
        for parts in (2, 1) if B <= 640 else (2,):
            eng.set_parts(parts)
            for _ in range(3):
                step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30):
                step()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
            eng.use_chain = False
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            eng.start_timing()
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            kt = eng.stop_timing()
            eng.use_chain = True
            s = 1024.0 / B
            print(f"rays={B:5d} parts={parts} step {dt * 1e3:7.3f} ms ({dt * 1e3 * s:7.3f} per 1024 rays)  " +
                  " ".join(f"{SHORT[k]}={v[0] / 10:.3f}({v[0] / 10 * s:.3f})" for k, v in kt.items() if k in SHORT), flush=True)
        eng.set_parts(2)


if __name__ == "__main__":
    main([int(a) for a in sys.argv[1:]] or [160, 320, 512, 640, 1024, 2048])
