#!/usr/bin/env python3
"""Headline benchmark: one training step of the I2-SDF render core on synthetic rays / random-weight networks.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): synthetic.yml networks (8x256 SDF + 4x256 radiance MLP, 800 955 parameters,
reference init), 1024 rays per GPU, N_samples 64 -> 97 shaded samples per ray, camera (ii) of BASELINE.md (t=(0,0,-2),
beta=0.02, looking at the init sphere), sampler iteration count fixed to k=2 (it is data dependent).
A step = ray set-up -> error-bounded sampler (k SDF-MLP passes over 128 samples/ray) -> SDF MLP with d sdf/dx ->
radiance MLP -> density/compositing -> I2SDFLoss -> backward (double backward through the SDF MLP, all parameter
gradients) -> [N>1: one flat all-reduce] -> Adam step.  Inputs are resident in HBM before the timed region.
`value` = rays x 97 x N / step time (whole job).  fp32 throughout (fp32 MFMA: the 1e-4 parity bar excludes bf16).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=1024, help="rays per GPU")
    ap.add_argument("--sampler-iters", type=int, default=2, help="fixed sampler iterations k (0 = data dependent)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL; gloo only to smoke-test the N>1 path)")
    ap.add_argument("--share-gpu", action="store_true", help="testing only: all ranks use cuda:0 (needs --backend gloo)")
    ap.add_argument("--cpu-rays", type=int, default=512)
    ap.add_argument("--bf16x3", type=int, default=-1,
                    help="bit mask of the kernels that run in bf16x3 split arithmetic (1 sampler forward, 2 weight gradients, "
                         "4 training forward, 8 SDF backward, 16 radiance net); -1 = the engine's default (all available), 0 = plain fp32 MFMA everywhere")
    ap.add_argument("--profile-kernels", action="store_true", default=True)
    return ap.parse_args()


def flops_per_point(cfg):
    """Algorithmic dense-contraction FLOPs (2 x MACs) per point for each kernel family."""
    sdf, rgb = cfg.sdf.dims, cfg.rgb.dims
    F = cfg.feature_size
    mac_fwd_hidden = sum(o * i for o, i in sdf[:-1])
    mac_fwd = mac_fwd_hidden + sdf[-1][0] * sdf[-1][1]
    mac_igrad = mac_fwd_hidden + sdf[-1][1]                      # reverse chain; last layer contributes row 0 only
    mac_rgb = sum(o * i for o, i in rgb)
    return {
        "sdf_forward": 2 * (mac_fwd_hidden + sdf[-1][1]),          # sampler: sdf row only
        "sdf_forward_grad": 2 * (mac_fwd + mac_igrad),
        "rgb_forward": 2 * mac_rgb,
        "rgb_backward": 2 * (mac_rgb - rgb[0][0] * (rgb[0][1] - F)),   # input grad for the feature columns only
        "sdf_backward": 2 * (mac_igrad + mac_fwd - sdf[0][0] * sdf[0][1]),
        "wgrad_sdf": 2 * (mac_igrad + mac_fwd),
        "wgrad_rgb": 2 * mac_rgb,
    }


def bytes_per_point(cfg):
    """Algorithmic (compulsory) HBM bytes per point of the per-point saved-tensor traffic, fp32 (DESIGN.md "Data layout"):
    every tensor counted once per kernel that must read or write it."""
    H, F = cfg.sdf.hidden, cfg.feature_size
    nh = cfg.sdf.n_lin - 1                         # hidden activations h_1..h_{L-1} of the SDF net
    nr = cfg.rgb.n_lin - 1
    row, frow = 4 * H, 4 * F
    return {
        "i2sdf_sdf_forward_grad": nh * row * 3 + frow + 160 + 28,      # write h, write abar, re-read h (chain), feature, PE, sdf/grad
        "i2sdf_sdf_backward": nh * row * (2 + 2 + 2 + 1) + frow + 200,  # sweep 1: read h, abar, write G(hbar), G2; sweep 2: read h, G2, write G(a)
        "i2sdf_weight_grads": nh * row * 4 + row + frow + nr * 2 * 4 * cfg.rgb.hidden + frow + 288,   # A, A', B, B' per SDF layer; rgb G(a), r
        "i2sdf_rgb_forward": frow + nr * 4 * cfg.rgb.hidden + 128 + 12,
        "i2sdf_rgb_backward": nr * 4 * cfg.rgb.hidden * 2 + frow + 40,
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev_index = 0 if (world == 1 or args.share_gpu) else local_rank
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    from i2sdf_amd import I2SDFNetwork, I2SDFLoss, NetConfig, synthetic_conf
    from i2sdf_amd import dist as i2dist

    conf = synthetic_conf()
    conf["use_normal"] = True
    torch.manual_seed(0)                                  # identical initial weights on every rank
    net = I2SDFNetwork(conf).to(dev)
    with torch.no_grad():
        net.density.beta.fill_(0.02)
    net.train()
    net.force_iters = args.sampler_iters
    loss_fn = I2SDFLoss(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)   # synthetic.yml:15-23
    opt = torch.optim.Adam(net.get_param_groups(5.0e-4), eps=1e-15)   # model/trainer/recon.py:203
    if world > 1:
        i2dist.attach_data_parallel(net)

    B = args.rays
    g = torch.Generator().manual_seed(1000 + rank)        # each rank draws its own rays (ray-sharded data parallelism)
    W_, H_ = 640, 480
    K = torch.eye(4); K[0, 0] = K[1, 1] = 600.0; K[0, 2] = W_ / 2; K[1, 2] = H_ / 2
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.0, 0.0, -2.0])
    uv = torch.stack([torch.randint(0, W_, (B,), generator=g), torch.randint(0, H_, (B,), generator=g)], -1).float().reshape(B, 1, 2)
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
        return losses["loss"]

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step(0)                                   # builds the engine (and the flat parameter buffer) the way a trainer would
    eng = net._engine_for(dev)
    if args.bf16x3 >= 0:
        eng.set_sdf_forward_bf16x3(bool(args.bf16x3 & 1))
        eng.set_wgrad_bf16x3(bool(args.bf16x3 & 2))
        eng.set_train_forward_bf16x3(bool(args.bf16x3 & 4))
        eng.set_sdf_backward_bf16x3(bool(args.bf16x3 & 8))
        eng.set_rgb_bf16x3(bool(args.bf16x3 & 16))
    for i in range(args.warmup):
        step(i)
    fence()
    eng.start_timing()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    ktimes = eng.stop_timing()
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    iters = int(net.last_sampler_iters.item())
    n_shaded = eng.n_z - 1
    ms = dt / args.steps * 1e3
    value = B * n_shaded * world / (dt / args.steps)

    result = None
    if rank == 0:
        cfg = net.cfg
        fp = flops_per_point(cfg)
        M_main, M_sdf = B * n_shaded, B * n_shaded + 3 * B
        launch_flops = {
            # the sampler entry point = k SDF-MLP passes over 128 samples/ray + the per-ray Algorithm-1 kernels (counted as 0 FLOP)
            "i2sdf_sample_rays": fp["sdf_forward"] * B * cfg.sampler.N_samples_eval * max(iters, 1),
            "i2sdf_sdf_forward_grad": fp["sdf_forward_grad"] * M_sdf,
            "i2sdf_rgb_forward": fp["rgb_forward"] * M_main,
            "i2sdf_rgb_backward": fp["rgb_backward"] * M_main,
            "i2sdf_sdf_backward": fp["sdf_backward"] * M_sdf,
            "i2sdf_weight_grads": fp["wgrad_sdf"] * M_sdf + fp["wgrad_rgb"] * M_main,
        }
        kern = {}
        for name, (tot_ms, cnt) in ktimes.items():
            kern[name] = {"ms_per_step": tot_ms / args.steps, "launches_per_step": cnt / args.steps}
            key = name
            if key in launch_flops and cnt:
                kern[name]["tflops"] = launch_flops[key] / (tot_ms / cnt * 1e-3) / 1e12
        # dominant = the kernel family with the largest share of the step
        mfma_names = [n for n in kern if "tflops" in kern[n]]
        dom = max(mfma_names, key=lambda n: kern[n]["ms_per_step"]) if mfma_names else None
        PEAK = 157.3            # TFLOP/s, fp32-input MFMA on MI355X (MI355X_MICROARCH.md)
        PEAK_X3 = 2500.0 / 6    # bf16 dense MFMA peak / six bf16 MFMAs per fp32 product block (csrc/x3.h): fp32-equivalent TFLOP/s
        x3 = {"i2sdf_sample_rays": eng.sdf_forward_bf16x3, "i2sdf_sdf_forward_grad": eng.train_forward_bf16x3,
              "i2sdf_sdf_backward": eng.sdf_backward_bf16x3, "i2sdf_weight_grads": eng.wgrad_bf16x3,
              "i2sdf_rgb_forward": eng.rgb_bf16x3, "i2sdf_rgb_backward": eng.rgb_bf16x3}
        roof = None
        if dom:
            # the roof that binds the dominant entry point: time at the arithmetic peak vs time at the HBM peak (8 TB/s) for its
            # algorithmic FLOPs / bytes; `achieved` and `peak` are reported in the unit of the binding roof
            ach = kern[dom]["tflops"]
            peak = PEAK_X3 if x3.get(dom) else PEAK
            npts = {"i2sdf_rgb_forward": M_main, "i2sdf_rgb_backward": M_main}.get(dom, M_sdf)
            alg_bytes = bytes_per_point(cfg).get(dom, 0) * npts
            t_launch = kern[dom]["ms_per_step"] / max(kern[dom]["launches_per_step"], 1e-9) * 1e-3
            t_mfma, t_hbm = launch_flops[dom] / (peak * 1e12), alg_bytes / 8.0e12
            common = {"kernel": dom, "arithmetic": "bf16x3 split (fp32-equivalent)" if x3.get(dom) else "f32 MFMA",
                      "traffic": profiled_traffic(dom), "tflops": round(ach, 2), "frac_of_arithmetic_peak": round(ach / peak, 4),
                      "algorithmic_bytes": int(alg_bytes), "frac_vs_fp32_mfma_peak": round(ach / PEAK, 4),
                      "all_mfma_kernels_tflops": round(sum(launch_flops[n] * kern[n]["launches_per_step"] for n in mfma_names)
                                                       / (sum(kern[n]["ms_per_step"] for n in mfma_names) * 1e-3) / 1e12, 2)}
            if t_hbm > t_mfma:
                gbs = alg_bytes / t_launch / 1e9
                roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4)}
            else:
                roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4)}
            roof.update(common)
        total_flops = sum(launch_flops.values())
        result = {
            "metric": "ray-samples/sec (fwd+bwd)", "value": round(value, 1), "unit": "ray-samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if not any(x3.values()) else "f32 (bf16x3 split MFMA: fp32 operands as 3 bf16 terms, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "synthetic.yml nets (8x256 SDF + 4x256 radiance, 800955 params), training step incl. sampler, loss, backward, Adam",
                       "rays_per_gpu": B, "shaded_samples_per_ray": n_shaded, "sampler_iters": iters, "sampler_samples_per_iter": cfg.sampler.N_samples_eval,
                       "camera": "t=(0,0,-2), R=I, f=600, beta=0.02", "parallelism": f"dp{world} (ray-sharded, 1 flat grad all-reduce)"},
            "rays_per_s": round(B * world / (dt / args.steps), 1),
            "step_tflops": round(total_flops * world / (dt / args.steps) / 1e12, 2),
            "frac_fp32_mfma_roofline_whole_step": round(total_flops / (dt / args.steps) / 1e12 / PEAK, 4),   # per GPU (weak scaling)
            "final_loss": float(loss.item()),
            "roofline": roof, "kernels": {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in kern.items()},
        }
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(args, iters, n_shaded)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def profiled_traffic(entry):
    """HBM bytes per launch of an entry point from the committed PMC summaries (profiles/r1_pmc_{fetch,write}_summary.csv:
    FETCH_SIZE / WRITE_SIZE in KB per dispatch, separate --pmc passes; FETCH doubled as MI355X_MICROARCH.md prescribes for
    16-B-per-lane loads on gfx950).  None when the summaries are absent or do not cover the entry point's kernels."""
    import csv
    pat = {"i2sdf_weight_grads": ("wgrad", "wn_backward"), "i2sdf_sdf_backward": ("sdf_bwd",), "i2sdf_sdf_forward_grad": ("sdf_train_fwd", "sdf_igrad"),
           "i2sdf_sample_rays": ("sdf_fwd", "sampler_")}.get(entry)
    here = os.path.dirname(os.path.abspath(__file__))
    try:
        tot = 0.0
        for tag, col, mult in (("fetch", "FETCH_SIZE_per_dispatch", 2.0), ("write", "WRITE_SIZE_per_dispatch", 1.0)):
            rows = list(csv.DictReader(open(os.path.join(here, "profiles", f"r1_pmc_{tag}_summary.csv"))))
            steps = max([int(r["dispatches"]) for r in rows if "wn_backward" in r["kernel"]] or [0])
            if not pat or steps == 0:
                return None
            tot += sum(float(r[col]) * 1024.0 * mult * int(r["dispatches"]) for r in rows if any(q in r["kernel"] for q in pat)) / steps
        return round(tot) if tot > 0 else None
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(args, iters, n_shaded):
    """The CPU oracle (a port of the reference's PyTorch path, validated against it) timed on this node's host cores on
    a bounded sample of the same workload: same networks, same camera, same fixed k, fewer rays."""
    import torch
    from oracle import i2sdf_oracle as orc
    # torch's CPU GEMMs on 256-wide layers stop scaling (and then collapse) beyond a few tens of threads: use at most 32
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    ocfg = orc.synthetic_cfg(False)
    ocfg.use_normal = True
    sd = orc.init_params(ocfg, seed=0)
    sd["density.beta"] = torch.tensor(0.02)
    Bc = args.cpu_rays
    g = torch.Generator().manual_seed(7)
    K = torch.eye(4); K[0, 0] = K[1, 1] = 600.0; K[0, 2] = 320.0; K[1, 2] = 240.0
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.0, 0.0, -2.0])
    uv = torch.stack([torch.randint(0, 640, (Bc,), generator=g), torch.randint(0, 480, (Bc,), generator=g)], -1).float().reshape(Bc, 1, 2)
    inp = {"uv": uv, "intrinsics": K.repeat(Bc, 1, 1), "pose": pose.repeat(Bc, 1, 1)}
    gt = {"rgb": torch.rand(Bc, 3, generator=g), "depth": torch.rand(Bc, generator=g) * 3, "depth_mask": torch.ones(Bc, dtype=torch.bool),
          "normal": torch.nn.functional.normalize(torch.randn(Bc, 3, generator=g), dim=1), "normal_mask": torch.ones(Bc, dtype=torch.bool)}
    sc = ocfg.sampler
    R = ocfg.scene_bounding_sphere
    dr = orc.Draws(strat_u=torch.rand(Bc, sc.N_samples_eval, generator=g), cdf_u=torch.rand(Bc, sc.N_samples, generator=g),
                   extra_idx=torch.randperm(sc.N_samples_eval * max(iters, 1), generator=g)[: sc.N_samples_extra],
                   eik_idx=torch.randint(n_shaded + 1, (Bc,), generator=g), eik_pts=(torch.rand(Bc, 3, generator=g) * 2 - 1) * R,
                   nbr_off=(torch.rand(Bc, 3, generator=g) * 2 - 1) * 0.005)
    lc = orc.LossCfg(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=150000, depth_weight=0.1, normal_weight=0.05)
    times = []
    for i in range(4):
        t0 = time.perf_counter()
        orc.training_step_grads(sd, ocfg, inp, gt, lc, dr, step=10, force_iters=iters or None)
        times.append(time.perf_counter() - t0)
        if sum(times) > 40.0 and len(times) >= 2:      # keep the default run within minutes on any host
            break
    med = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": round(Bc * n_shaded / med, 1), "unit": "ray-samples/s", "cores": cores, "kind": "port",
            "sample": f"{Bc} rays x {n_shaded} shaded samples, same nets/camera/k={iters}, fwd+loss+bwd (no optimizer), torch CPU fp32 "
                      f"{cores} threads (host has {os.cpu_count()}), median of {len(times) - 1} after 1 warm-up, {med:.2f} s/step"}


if __name__ == "__main__":
    main()
