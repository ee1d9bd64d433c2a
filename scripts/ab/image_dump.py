#!/usr/bin/env python3
"""render the 640x480 bench view with the library selected by I2SDF_LIB_PATH and save rgb / depth / per-chunk iteration counts:
    python scripts/ab/image_dump.py OUT.npz          ;   python scripts/ab/image_dump.py --compare A.npz B.npz"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    print("iters", a["iters"].tolist(), b["iters"].tolist())
    for k in ("rgb", "depth", "wsum"):
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print(f"{k}: max {d.max():.3e} mean {d.mean():.3e} signed-sum {float((a[k].astype(np.float64) - b[k]).sum()):.3f}  #>1e-2: {int((d > 1e-2).sum())}  #>1e-4: {int((d > 1e-4).sum())} of {d.size}")
    d = np.abs(a["rgb"].astype(np.float64) - b["rgb"]).max(axis=1).reshape(480, 640)
    rows = np.nonzero((d > 1e-2).any(axis=1))[0]
    print("rows with a pixel off by > 1e-2:", rows[:20].tolist(), "..." if len(rows) > 20 else "", "first chunk boundaries (rows):", [round(12000 * i / 640, 1) for i in range(1, 6)])
    sys.exit(0)
import torch
from r4_time import make, batch
net, dev = make()
net.eval(); net.force_iters = 0
inp, _ = batch(4, dev)
H, W = 480, 640
ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
uv = torch.stack([xs, ys], -1).reshape(1, -1, 2).float().to(dev)
o = net.render_image({"uv": uv, "intrinsics": inp["intrinsics"][:1], "pose": inp["pose"][:1]}, 12000)
np.savez_compressed(sys.argv[1], rgb=o["rgb_values"].cpu().numpy(), depth=o["depth_values"].cpu().numpy(), wsum=o["weight_sum"].cpu().numpy(),
                    iters=net.last_sampler_iters.cpu().numpy())
print("saved", sys.argv[1], "checksum", float(o["rgb_values"].double().sum()))
