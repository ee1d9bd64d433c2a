#!/usr/bin/env python3
"""Round-6 experiment: the sampler's sdf-only passes with two split planes (I2SDF_OPT_SAMPLER_BF16X2) against the three-plane form,
in ONE process on one box (options toggled on the same engine):

  python scripts/ab/sampler_x2.py            depths (both forms, eval + training draws), entry-point times, step times (k = 2, natural k), image

What it reports: how far the chosen depths move (median / 99 % / max over all samples, iteration counts), how far the RENDERED outputs
move when the depths come from the two-plane passes (rgb / depth / normal max-norm relative), and the times."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from i2sdf_amd import I2SDFNetwork, I2SDFLoss, FusedAdam, synthetic_conf
from r4_time import make, batch, ev_time, step_fn


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    net, dev = make()
    eng = net._engine_for(dev)
    B = 1024
    inp, gt = batch(B, dev)
    cam, dirs, dnorm = eng.ray_setup(inp["uv"], inp["pose"], inp["intrinsics"])
    flat = net._flat
    # ---- depths: eval mode (deterministic), natural iteration count and k = 2
    for beta in (0.02, 0.1):
        with torch.no_grad():
            net.density.beta.fill_(beta)
        eng = net._engine_for(dev)
        for force in (0, 2):
            z = {}
            for x2 in (False, True):
                eng.set_sampler_bf16x2(x2)
                zz, ze, it = eng.sample_rays(flat, cam, dirs, training=False, force_iters=force)
                z[x2] = (zz.clone(), ze.clone(), int(it.item()))
            d = (z[True][0] - z[False][0]).abs()
            q = torch.quantile(d.flatten()[: 1 << 24].float(), torch.tensor([0.5, 0.99, 0.999], device=dev))
            print(f"depths beta={beta} force_iters={force}: iterations x3={z[False][2]} x2={z[True][2]}  |dz| median {q[0]:.2e} 99% {q[1]:.2e} 99.9% {q[2]:.2e} "
                  f"max {float(d.max()):.2e}", flush=True)
            # rendered outputs on the two depth sets (eval render: no draws)
            net.eval()
            outs = {}
            for x2 in (False, True):
                with torch.no_grad():
                    outs[x2] = net.render(inp, cam, dirs, dnorm, z[x2][0], z[x2][1])
            net.train()
            print("   rendered with those depths: " + "  ".join(f"{k} {rel(outs[True][k], outs[False][k]):.2e}" for k in ("rgb_values", "depth_values", "weight_sum", "normal_map")),
                  flush=True)
    with torch.no_grad():
        net.density.beta.fill_(0.02)
    # ---- entry point + pass times
    eng = net._engine_for(dev)
    for rnd in range(3):
        for x2 in (False, True):
            eng.set_sampler_bf16x2(x2)
            for force in (2, 5):
                med, best = ev_time(lambda: eng.sample_rays(flat, cam, dirs, training=False, force_iters=force), rep=10, rounds=5)
                print(f"sample_rays round {rnd} x2={int(x2)} k={force}: {med * 1e3:7.1f} us (best {best * 1e3:7.1f})", flush=True)
    # ---- training step
    loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)
    opt = FusedAdam(net, lr=5e-4, eps=1e-15)
    step = step_fn(net, loss_fn, opt, inp, gt)
    for force, name in ((2, "k2"), (0, "natural_k")):
        net.force_iters = force
        step()
        for rnd in range(3):
            for x2 in (False, True):
                eng.set_sampler_bf16x2(x2)
                for _ in range(3):
                    step()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(30):
                    step()
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
                print(f"step {name} round {rnd} x2={int(x2)}: {dt * 1e3:7.3f} ms  iters={int(net.last_sampler_iters.item())}", flush=True)
    # ---- full image (cfg 4)
    net.eval()
    net.force_iters = 0
    H, W = 480, 640
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    uv = torch.stack([xs, ys], -1).reshape(1, -1, 2).float().to(dev)
    img_in = {"uv": uv, "intrinsics": inp["intrinsics"][:1], "pose": inp["pose"][:1]}
    imgs = {}
    for rnd in range(2):
        for x2 in (False, True):
            eng.set_sampler_bf16x2(x2)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            o = net.render_image(img_in, 12000)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            imgs[x2] = o
            print(f"image round {rnd} x2={int(x2)}: {dt:6.3f} s  iters={net.last_sampler_iters.tolist()[:6]}...", flush=True)
    print("image x2 vs x3: " + "  ".join(f"{k} max {rel(imgs[True][k], imgs[False][k]):.2e} mean {float((imgs[True][k] - imgs[False][k]).abs().mean()):.2e}"
                                          for k in ("rgb_values", "depth_values", "normal_map")), flush=True)
    mse = float(((imgs[True]["rgb_values"] - imgs[False]["rgb_values"]) ** 2).mean())
    import math
    print(f"image PSNR(x2 render, x3 render) = {-10 * math.log10(max(mse, 1e-30)):.1f} dB", flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
