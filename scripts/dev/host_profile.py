"""Host-side cost of a training step: cProfile over 30 steps of the bench workload (the GPU work is asynchronous, so the
profile shows where the Python thread spends its time while it runs ahead of / falls behind the device).
python scripts/dev/host_profile.py [fused]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from i2sdf_amd import I2SDFNetwork, I2SDFLoss, synthetic_conf

dev = torch.device("cuda", 0)
conf = synthetic_conf(); conf["use_normal"] = True
torch.manual_seed(0)
net = I2SDFNetwork(conf).to(dev); net.train(); net.force_iters = 2
with torch.no_grad():
    net.density.beta.fill_(0.02)
loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)
opt = torch.optim.Adam(net.get_param_groups(5.0e-4), eps=1e-15, **({"fused": True} if "fused" in sys.argv else {}))
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

for i in range(5):
    step(i)
torch.cuda.synchronize()
N = 30
# host-only time: how long the Python thread needs to ENQUEUE a step (device idle time excluded by not synchronising)
t0 = time.perf_counter()
for i in range(N):
    step(i)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"enqueue {t_enq / N * 1e3:.2f} ms/step, with device {t_all / N * 1e3:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
