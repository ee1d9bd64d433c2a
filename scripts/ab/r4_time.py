#!/usr/bin/env python3
"""Round-4 A/B timing on the GPU box, everything in ONE process so that variants see the same box and clocks, interleaved and repeated:

  python scripts/ab/r4_time.py fwd                 sdf-only forward (one sampler pass: 131 072 points)
  python scripts/ab/r4_time.py step [rays ...] [c=PARTS ...]   training step per point-range count at the given ray counts
  python scripts/ab/r4_time.py entries             un-chained per-entry-point times of a step
(I2SDF_LIB_PATH selects an A/B build, scripts/ab/variant_build.sh)

Kernel times are HIP-event brackets around REP back-to-back launches (median of several); steps are wall clock around synchronised windows."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from i2sdf_amd import I2SDFNetwork, I2SDFLoss, FusedAdam, synthetic_conf


def make(light=False):
    dev = torch.device("cuda:0")
    conf = synthetic_conf(light)
    conf["use_normal"] = True
    torch.manual_seed(0)
    net = I2SDFNetwork(conf).to(dev).train()
    with torch.no_grad():
        net.density.beta.fill_(0.02)
    return net, dev


def batch(B, dev, seed=1000):
    g = torch.Generator().manual_seed(seed)
    K = torch.eye(4); K[0, 0] = K[1, 1] = 600.0; K[0, 2] = 320.0; K[1, 2] = 240.0
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.0, 0.0, -2.0])
    uv = torch.stack([torch.randint(0, 640, (B,), generator=g), torch.randint(0, 480, (B,), generator=g)], -1).float().reshape(B, 1, 2)
    inp = {"uv": uv.to(dev), "intrinsics": K.repeat(B, 1, 1).to(dev), "pose": pose.repeat(B, 1, 1).to(dev)}
    gt = {"rgb": torch.rand(B, 3, generator=g).to(dev), "depth": (torch.rand(B, generator=g) * 3).to(dev),
          "depth_mask": torch.ones(B, dtype=torch.bool, device=dev),
          "normal": torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1).to(dev),
          "normal_mask": torch.ones(B, dtype=torch.bool, device=dev)}
    return inp, gt


def ev_time(fn, rep=10, rounds=5):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(rep):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / rep)
    return sorted(ts)[len(ts) // 2], min(ts)


def mode_fwd():
    net, dev = make()
    eng = net._engine_for(dev)
    M = 1024 * 128
    x = ((torch.rand(M, 3, device=dev) * 2 - 1) * 2.5).contiguous()
    for rnd in range(3):
        eng.sdf_forward(x)
        med, best = ev_time(lambda: eng.sdf_forward(x))
        print(f"fwd round {rnd}: {med * 1e3:7.1f} us (best {best * 1e3:7.1f}) per {M} points, {2 * 459008 * M / med / 1e9:6.1f} TFLOP/s fp32-eq", flush=True)


def step_fn(net, loss_fn, opt, inp, gt):
    st = {"i": 0}

    def step():
        out = net(inp)
        l = loss_fn(out, gt, st["i"])["loss"]
        opt.zero_grad(set_to_none=True)
        l.backward()
        opt.step()
        st["i"] += 1
    return step


def mode_step(rays, combos=None):
    net, dev = make()
    net.force_iters = 2
    loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)
    opt = FusedAdam(net, lr=5e-4, eps=1e-15)
    combos = combos or [(2,), (0,)]
    for B in rays:
        inp, gt = batch(B, dev)
        step = step_fn(net, loss_fn, opt, inp, gt)
        step()
        eng = net._engine_for(dev)
        n = max(6, min(30, int(600 / max(B / 1024 * 6.3, 1))))
        for rnd in range(3):
            for (parts,) in combos:
                eng.set_parts(parts)
                for _ in range(3):
                    step()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(n):
                    step()
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
                print(f"step rays={B:5d} round {rnd} parts={parts}: {dt * 1e3:8.3f} ms  = {dt / B * 1e6:6.3f} us/ray  "
                      f"{B * (eng.n_z - 1) / dt / 1e6:6.2f} M ray-samples/s", flush=True)


def mode_entries():
    net, dev = make()
    net.force_iters = 2
    loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)
    opt = FusedAdam(net, lr=5e-4, eps=1e-15)
    inp, gt = batch(1024, dev)
    step = step_fn(net, loss_fn, opt, inp, gt)
    step()
    eng = net._engine_for(dev)
    for rnd in range(2):
        for _ in (0,):
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
            short = {"i2sdf_weight_grads": "wgrad", "i2sdf_sdf_backward": "sdf_bwd", "i2sdf_sdf_forward_grad": "sdf_fwdg", "i2sdf_sample_rays": "sampler",
                     "i2sdf_rgb_forward": "rgb_f", "i2sdf_rgb_backward": "rgb_b"}
            print(f"entries round {rnd}: " + " ".join(f"{short[k]}={v[0] / 10:.3f}" for k, v in kt.items() if k in short), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    if mode == "fwd":
        mode_fwd()
    elif mode == "step":
        rays = [int(a) for a in sys.argv[2:] if not a.startswith("c=")] or [1024]
        combos = [(int(a[2:]),) for a in sys.argv[2:] if a.startswith("c=")] or None
        mode_step(rays, combos)
    elif mode == "entries":
        mode_entries()
