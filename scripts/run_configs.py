#!/usr/bin/env python3
"""Run the other BASELINE.json configurations at full size on one MI355X (timings for DESIGN.md; parity is in tests/):
  cfg 3  synthetic_light_mask.yml training step (1024 rays)          cfg 5  4096 rays/GPU training step (synthetic.yml)
  cfg 4  full-resolution 640x480 eval render in 12000-ray chunks (utils.split_input / merge_output protocol)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from i2sdf_amd import I2SDFNetwork, I2SDFLoss, synthetic_conf


def cam_batch(B, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    K = torch.eye(4); K[0, 0] = K[1, 1] = 600.0; K[0, 2] = 320.0; K[1, 2] = 240.0
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.0, 0.0, -2.0])
    uv = torch.stack([torch.randint(0, 640, (B,), generator=g), torch.randint(0, 480, (B,), generator=g)], -1).float().reshape(B, 1, 2)
    gt = {"rgb": torch.rand(B, 3, generator=g), "depth": torch.rand(B, generator=g) * 3, "depth_mask": torch.ones(B, dtype=torch.bool),
          "normal": torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1), "normal_mask": torch.ones(B, dtype=torch.bool),
          "light_mask": (torch.rand(B, 1, generator=g) > 0.5).float()}
    inp = {"uv": uv, "intrinsics": K.repeat(B, 1, 1), "pose": pose.repeat(B, 1, 1)}
    to = lambda d: {k: v.to(dev) for k, v in d.items()}
    return to(inp), to(gt)


def train_case(name, light, B, k):
    dev = torch.device("cuda:0")
    conf = synthetic_conf(light); conf["use_normal"] = True
    torch.manual_seed(0)
    net = I2SDFNetwork(conf).to(dev).train()
    with torch.no_grad():
        net.density.beta.fill_(0.02)
    net.force_iters = k
    loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05,
                        light_mask_weight=0.5 if light else 0.0)
    from i2sdf_amd import FusedAdam
    opt = FusedAdam(net, lr=5e-4, eps=1e-15)              # the production optimizer (one launch), as bench.py
    inp, gt = cam_batch(B, dev)

    def step(i):
        out = net(inp)
        l = loss_fn(out, gt, i)["loss"]
        opt.zero_grad(set_to_none=True); l.backward(); opt.step()
        return l
    for i in range(5):
        step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for i in range(n):
        l = step(5 + i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    ns = net._engine_for(dev).n_z - 1
    print(f"{name}: {B} rays x {ns} samples, k={k}: {dt*1e3:.2f} ms/step, {B*ns/dt/1e6:.2f} M ray-samples/s, sampler iterations {int(net.last_sampler_iters.item())}, "
          f"loss {float(l.detach()):.4f}, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)


def eval_image():
    dev = torch.device("cuda:0")
    conf = synthetic_conf(False); conf["use_normal"] = True
    torch.manual_seed(0)
    net = I2SDFNetwork(conf).to(dev).eval()
    with torch.no_grad():
        net.density.beta.fill_(0.02)
    H, W, chunk = 480, 640, 12000          # config/synthetic.yml: img_res, split_n_pixels
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    uv = torch.stack([xs, ys], -1).float().reshape(1, -1, 2).to(dev)
    K = torch.eye(4); K[0, 0] = K[1, 1] = 600.0; K[0, 2] = 320.0; K[1, 2] = 240.0
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.0, 0.0, -2.0])
    base = {"intrinsics": K.unsqueeze(0).to(dev), "pose": pose.unsqueeze(0).to(dev)}

    def render():
        res = []
        with torch.no_grad():
            for idx in torch.split(torch.arange(H * W, device=dev), chunk):     # utils.split_input (utils/__init__.py:35-47)
                d = dict(base); d["uv"] = torch.index_select(uv, 1, idx)
                res.append(net(d))
        return {k: torch.cat([r[k] for r in res], 0) for k in res[0]}           # utils.merge_output (:70-84)
    render()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = render()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    its = int(net.last_sampler_iters.item())
    print(f"cfg4 full-res eval 640x480: {dt*1e3:.1f} ms/image = {H*W/dt/1e6:.2f} M rays/s ({H*W*97/dt/1e6:.1f} M ray-samples/s), "
          f"{(H*W+chunk-1)//chunk} chunks of {chunk}, last-chunk sampler iters {its}, rgb mean {float(out['rgb_values'].mean()):.4f}, "
          f"finite {bool(torch.isfinite(out['normal_map']).all())}, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)


def dense128():
    """BASELINE.json metric convention 'dense-128': 128 shaded samples per ray, sampler bypassed (uniform depths)."""
    dev = torch.device("cuda:0")
    conf = synthetic_conf(False); conf["use_normal"] = True
    torch.manual_seed(0)
    net = I2SDFNetwork(conf).to(dev).train()
    with torch.no_grad():
        net.density.beta.fill_(0.02)
    loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)
    from i2sdf_amd import FusedAdam
    opt = FusedAdam(net, lr=5e-4, eps=1e-15)              # the production optimizer (one launch), as bench.py
    B, n = 1024, 128
    inp, gt = cam_batch(B, dev)
    eng = net._engine_for(dev)
    c, d, nrm = eng.ray_setup(inp["uv"], inp["pose"], inp["intrinsics"])
    z = torch.linspace(0.0, 6.0, n + 1, device=dev).repeat(B, 1).contiguous()      # n samples + z_max column
    z_eik = z[:, n // 2:n // 2 + 1].contiguous()

    def step(i):
        out = net.render(inp, c, d, nrm, z, z_eik)
        l = loss_fn(out, gt, i)["loss"]
        opt.zero_grad(set_to_none=True); l.backward(); opt.step()
        return l
    for i in range(2):
        step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(5):
        l = step(2 + i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"dense-128: {B} rays x {n} samples, sampler bypassed: {dt*1e3:.2f} ms/step, {B*n/dt/1e6:.2f} M ray-samples/s, loss {float(l.detach()):.4f}", flush=True)


def cpu_single_thread():
    """the CPU restatement additionally with one thread (bounded sample)."""
    import sys as _s
    _s.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from oracle import i2sdf_oracle as orc
    from helpers import camera_inputs, make_draws, make_gt
    torch.set_num_threads(1)
    ocfg = orc.synthetic_cfg(False); ocfg.use_normal = True
    sd = orc.init_params(ocfg, seed=0); sd["density.beta"] = torch.tensor(0.02)
    B = 16
    inp = camera_inputs(B, (0.0, 0.0, -2.0), seed=0); gt = make_gt(B)
    dr = make_draws(ocfg, B, n_row=256, seed=0)
    lc = orc.LossCfg(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)
    t0 = time.perf_counter()
    orc.training_step_grads(sd, ocfg, inp, gt, lc, dr, step=10, force_iters=2)
    dt = time.perf_counter() - t0
    print(f"CPU oracle, 1 thread: {B} rays x 97 samples, k=2, fwd+loss+bwd: {dt:.2f} s/step = {B*97/dt:.0f} ray-samples/s", flush=True)


if __name__ == "__main__":
    dense128()
    cpu_single_thread()
    train_case("cfg3 light-mask", True, 1024, 2)
    train_case("cfg5 4096 rays/GPU", False, 4096, 2)
    train_case("cfg2 k=1", False, 1024, 1)
    train_case("cfg2 k=5", False, 1024, 5)
    train_case("cfg2 natural k", False, 1024, 0)
    eval_image()
