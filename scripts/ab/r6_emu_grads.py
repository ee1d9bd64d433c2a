#!/usr/bin/env python3
"""Round 6, numerical gate by emulation (scripts/ab/knockout_build.py emu_*): parameter gradients of ONE full-size training step
(synthetic.yml nets, 1024 rays, k = 2, fixed seeds) per library build, and their distance to the fp32-equivalent production build.

   I2SDF_LIB_PATH=... I2SDF_WGRAD_BF16X2=0|1 python scripts/ab/r6_emu_grads.py dump TAG      -> gpurun_out/emu_grads_TAG.pt
   python scripts/ab/r6_emu_grads.py compare BASE TAG [TAG ...]                               -> per build: worst max-norm-relative distance to BASE over the
                                                                                                  parameter tensors (the bar of the parity tests is 1e-4 against fp64)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out")


def dump(tag):
    from r4_time import make, batch
    from i2sdf_amd import I2SDFLoss
    net, dev = make()
    net.force_iters = 2
    loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)
    inp, gt = batch(1024, dev)
    res = {}
    for rep in range(2):          # twice: the second must reproduce the first bit for bit (same seed)
        torch.manual_seed(1234)
        torch.cuda.manual_seed(1234)
        net.zero_grad(set_to_none=True)
        out = net(inp)
        l = loss_fn(out, gt, 0)["loss"]
        l.backward()
        g = {k: p.grad.detach().float().cpu().clone() for k, p in net.named_parameters() if p.grad is not None}
        if rep == 0:
            res = g
        else:
            same = all(torch.equal(res[k], g[k]) for k in g)
            print(f"{tag}: loss {float(l):.8f}, {len(g)} gradient tensors, reproducible: {same}", flush=True)
    torch.save(res, os.path.join(OUT, f"emu_grads_{tag}.pt"))


def compare(base, tags):
    b = torch.load(os.path.join(OUT, f"emu_grads_{base}.pt"))
    for t in tags:
        g = torch.load(os.path.join(OUT, f"emu_grads_{t}.pt"))
        rows = []
        for k in b:
            if b[k].numel() < 2:
                continue
            d = (g[k].double() - b[k].double()).abs().max().item()
            n = b[k].double().abs().max().item()
            rows.append((d / max(n, 1e-30), k, n))
        rows.sort(reverse=True)
        print(f"{t:>12s} vs {base}: worst {rows[0][0]:.3e} ({rows[0][1]}), then " + ", ".join(f"{r[0]:.2e} {r[1].split('.')[-2]}.{r[1].split('.')[-1][:6]}" for r in rows[1:4]) +
              f"; median over tensors {rows[len(rows) // 2][0]:.2e}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2])
    else:
        compare(sys.argv[2], sys.argv[3:])
