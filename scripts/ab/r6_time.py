#!/usr/bin/env python3
"""Round-6 A/B timing: ONE process per library build (I2SDF_LIB_PATH), step time (k = 2, chained) and the un-chained entry-point times:

  I2SDF_LIB_PATH=... python scripts/ab/r6_time.py [tag]          -> lines `tag step ...` / `tag entries ...`
  python scripts/ab/r6_time.py implicit                         -> implicit_network(x) on 2^20 points: 257 columns vs sdf-only vs i2sdf_sdf_grid"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from r4_time import make, batch, ev_time, step_fn
from i2sdf_amd import I2SDFLoss, FusedAdam


def main(tag):
    net, dev = make()
    net.force_iters = 2
    loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)
    opt = FusedAdam(net, lr=5e-4, eps=1e-15)
    inp, gt = batch(1024, dev)
    step = step_fn(net, loss_fn, opt, inp, gt)
    step()
    eng = net._engine_for(dev)
    for rnd in range(3):
        for _ in range(3):
            step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40):
            step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
        print(f"{tag} step round {rnd}: {dt * 1e3:7.3f} ms", flush=True)
    short = {"i2sdf_weight_grads": "wgrad", "i2sdf_sdf_backward": "sdf_bwd", "i2sdf_sdf_forward_grad": "sdf_fwdg", "i2sdf_sample_rays": "sampler",
             "i2sdf_rgb_forward": "rgb_f", "i2sdf_rgb_backward": "rgb_b"}
    for rnd in range(2):
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
        print(f"{tag} entries round {rnd}: " + " ".join(f"{short[k]}={v[0] / 10:.3f}" for k, v in kt.items() if k in short), flush=True)


def implicit():
    net, dev = make()
    net.eval()
    M = 1 << 20
    x = ((torch.rand(M, 3, device=dev) * 2 - 1) * 1.5).contiguous()
    eng = net._engine_for(dev)
    with torch.no_grad():
        for name, fn in (("implicit_network(x) -> (M,257)", lambda: net.implicit_network(x)),
                         ("engine.sdf_forward(x, want_features=True) (no concatenation)", lambda: eng.sdf_forward(x, want_features=True)),
                         ("get_sdf_vals(x) (sdf-only kernel)", lambda: net.implicit_network.get_sdf_vals(x)),
                         ("net.sdf_grid(x) (chunks of 2^20)", lambda: net.sdf_grid(x))):
            fn()
            med, best = ev_time(fn, rep=5, rounds=5)
            print(f"implicit: {name}: {med:7.3f} ms (best {best:7.3f}) per {M} points", flush=True)
        eng.set_sdf_forward_bf16x3(False)
        med, best = ev_time(lambda: net.implicit_network(x), rep=3, rounds=3)
        print(f"implicit: implicit_network(x), fp32-input MFMA kernel (the round-5 path): {med:7.3f} ms (best {best:7.3f})", flush=True)


def sampler_modes(tag):
    """the per-ray sampler kernels where they weigh most: the data-dependent loop (training step, natural k) and a 640x480 eval render"""
    net, dev = make()
    loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)
    opt = FusedAdam(net, lr=5e-4, eps=1e-15)
    inp, gt = batch(1024, dev)
    step = step_fn(net, loss_fn, opt, inp, gt)
    eng = net._engine_for(dev)
    cam, dirs, dnorm = eng.ray_setup(inp["uv"], inp["pose"], inp["intrinsics"])
    for force, name in ((2, "k2"), (0, "natural_k")):
        net.force_iters = force
        step()
        for rnd in range(3):
            for _ in range(3):
                step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40):
                step()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
            print(f"{tag} step {name} round {rnd}: {dt * 1e3:7.3f} ms iters={int(net.last_sampler_iters.item())}", flush=True)
        med, best = ev_time(lambda: eng.sample_rays(net._flat, cam, dirs, training=False, force_iters=force), rep=10, rounds=5)
        print(f"{tag} sample_rays {name}: {med * 1e3:7.1f} us (best {best * 1e3:7.1f})", flush=True)
    net.eval(); net.force_iters = 0
    H, W = 480, 640
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    uv = torch.stack([xs, ys], -1).reshape(1, -1, 2).float().to(dev)
    img_in = {"uv": uv, "intrinsics": inp["intrinsics"][:1], "pose": inp["pose"][:1]}
    net.render_image(img_in, 12000)
    for rnd in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        o = net.render_image(img_in, 12000)
        torch.cuda.synchronize()
        print(f"{tag} image round {rnd}: {time.perf_counter() - t0:6.3f} s  checksum {float(o['rgb_values'].double().sum()):.6f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sampler":
        sampler_modes(sys.argv[2] if len(sys.argv) > 2 else "lib")
    elif len(sys.argv) > 1 and sys.argv[1] == "implicit":
        implicit()
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else "lib")
