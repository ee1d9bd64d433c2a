"""Can one training step (forward + loss + backward, no optimizer) be captured in a hipGraph through torch.cuda.graph?"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from i2sdf_amd import I2SDFNetwork, I2SDFLoss, synthetic_conf
dev = torch.device("cuda:0")
conf = synthetic_conf(); conf["use_normal"] = True
torch.manual_seed(0)
net = I2SDFNetwork(conf).to(dev).train(); net.force_iters = 2
with torch.no_grad(): net.density.beta.fill_(0.02)
loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)
B = 1024
g = torch.Generator().manual_seed(1)
K = torch.eye(4); K[0, 0] = K[1, 1] = 600.0; K[0, 2] = 320.0; K[1, 2] = 240.0
pose = torch.eye(4); pose[2, 3] = -2.0
uv = torch.stack([torch.randint(0, 640, (B,), generator=g), torch.randint(0, 480, (B,), generator=g)], -1).float().reshape(B, 1, 2)
inp = {"uv": uv.to(dev), "intrinsics": K.repeat(B, 1, 1).to(dev), "pose": pose.repeat(B, 1, 1).to(dev)}
gt = {"rgb": torch.rand(B, 3, generator=g).to(dev), "depth": (torch.rand(B, generator=g) * 3).to(dev), "depth_mask": torch.ones(B, dtype=torch.bool, device=dev),
      "normal": torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1).to(dev), "normal_mask": torch.ones(B, dtype=torch.bool, device=dev)}
params = [p for p in net.parameters()]
def step():
    out = net(inp)
    l = loss_fn(out, gt, 10)["loss"]
    for p in params: p.grad = None
    l.backward()
    return l
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print("eager fwd+loss+bwd: %.3f ms/step" % ((time.perf_counter() - t0) / 20 * 1e3))
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        lg = step()
    grads = [p.grad for p in params]
    gr.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): gr.replay()
    torch.cuda.synchronize(); print("graph replay: %.3f ms/step, loss %.6f, grads finite %s" % ((time.perf_counter() - t0) / 20 * 1e3, float(lg), all(torch.isfinite(x).all() for x in grads)))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:400])
