"""Which torch ops make up the glue of a training step: torch.profiler table of aten ops (count, device time) for 4 steps of
the bench workload.  python scripts/dev/glue_profile.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from i2sdf_amd import I2SDFNetwork, I2SDFLoss, synthetic_conf

dev = torch.device("cuda", 0)
conf = synthetic_conf(); conf["use_normal"] = True
torch.manual_seed(0)
net = I2SDFNetwork(conf).to(dev); net.train(); net.force_iters = 2
with torch.no_grad():
    net.density.beta.fill_(0.02)
loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)
opt = torch.optim.Adam(net.get_param_groups(5.0e-4), eps=1e-15)
B = 1024
g = torch.Generator().manual_seed(1000)
K = torch.eye(4); K[0, 0] = K[1, 1] = 600.0; K[0, 2] = 320.0; K[1, 2] = 240.0
pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.0, 0.0, -2.0])
uv = torch.stack([torch.randint(0, 640, (B,), generator=g), torch.randint(0, 480, (B,), generator=g)], -1).float().reshape(B, 1, 2)
inp = {"uv": uv.to(dev), "intrinsics": K.repeat(B, 1, 1).to(dev), "pose": pose.repeat(B, 1, 1).to(dev)}
gt = {"rgb": torch.rand(B, 3, generator=g).to(dev), "depth": (torch.rand(B, generator=g) * 3).to(dev),
      "depth_mask": torch.ones(B, dtype=torch.bool, device=dev),
      "normal": torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1).to(dev),
      "normal_mask": torch.ones(B, dtype=torch.bool, device=dev)}

def step(i):
    out = net(inp)
    losses = loss_fn(out, gt, i)
    opt.zero_grad(set_to_none=True)
    losses["loss"].backward()
    opt.step()

for i in range(3):
    step(i)
torch.cuda.synchronize()
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(N):
        step(3 + i)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, "self_device_time_total", None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0 and e.key.startswith("aten::"):
        rows.append((dt / N, e.count / N, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
tot = 0.0
for dt, cnt, k, shp in rows:
    tot += dt
    print(f"{dt:8.1f} us/step  x{cnt:5.1f}  {k:34s} {shp}")
print(f"total aten device time: {tot:.1f} us/step")
