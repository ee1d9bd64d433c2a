"""Worker of tests/test_gpu_dp_equivalence.py: rank r of a 2-rank gloo group, both ranks on cuda:0 (one GPU box), the REAL
I2SDFNetwork with attach_data_parallel(equivalent=True).  Rank 0 then repeats the step alone on the concatenated batch and
writes the comparison to a file."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def full_batch(Bh, seed=5):
    """2*Bh rays of camera (ii): the first half looks at the image corner (misses the sphere), the second half at the centre."""
    from i2sdf_amd import synthetic_conf
    g = torch.Generator().manual_seed(seed)
    B = 2 * Bh
    W, H = 640, 480
    K = torch.eye(4); K[0, 0] = K[1, 1] = 600.0; K[0, 2], K[1, 2] = W / 2, H / 2
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.0, 0.0, -2.0])
    uv_corner = torch.stack([torch.randint(0, 50, (Bh,), generator=g), torch.randint(0, 50, (Bh,), generator=g)], -1)
    uv_centre = torch.stack([torch.randint(260, 380, (Bh,), generator=g), torch.randint(180, 300, (Bh,), generator=g)], -1)
    uv = torch.cat([uv_corner, uv_centre]).float().reshape(B, 1, 2)
    inp = {"uv": uv, "intrinsics": K.repeat(B, 1, 1), "pose": pose.repeat(B, 1, 1)}
    gt = {"rgb": torch.rand(B, 3, generator=g), "depth": torch.rand(B, generator=g) * 3,
          "depth_mask": torch.cat([torch.rand(Bh, generator=g) > 0.7, torch.rand(Bh, generator=g) > 0.2]),       # very different counts per rank
          "normal": torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1),
          "normal_mask": torch.cat([torch.rand(Bh, generator=g) > 0.1, torch.rand(Bh, generator=g) > 0.6])}
    sc = synthetic_conf()["ray_sampler"]
    n_z = sc["N_samples"] + sc["N_samples_extra"] + 2
    draws = {"strat_u": torch.rand(B, sc["N_samples_eval"], generator=g), "cdf_u": torch.rand(B, sc["N_samples"], generator=g),
             "eik_idx": torch.randint(n_z, (B,), generator=g), "eik_pts": (torch.rand(B, 3, generator=g) * 2 - 1) * 3.0,
             "nbr_off": (torch.rand(B, 3, generator=g) * 2 - 1) * 0.005}
    return inp, gt, draws


def make_net():
    from i2sdf_amd import I2SDFNetwork, synthetic_conf
    conf = dict(synthetic_conf())
    conf["use_normal"] = True
    torch.manual_seed(0)
    net = I2SDFNetwork(conf)
    with torch.no_grad():
        net.density.beta.fill_(0.02)
    return net.cuda().train()


def loss_of(net, loss_fn, inp, gt, draws, extra=None):
    d = {k: v.cuda() for k, v in draws.items()}
    if extra is not None:
        d["extra_idx"] = extra
    out = net({k: v.cuda() for k, v in inp.items()}, draws=d)
    losses = loss_fn(out, {k: v.cuda() for k, v in gt.items()}, 10)
    net.zero_grad()
    losses["loss"].backward()
    return out, losses


def worker(rank, world, port, Bh, result_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from i2sdf_amd import I2SDFLoss
    from i2sdf_amd import dist as i2dist
    inp, gt, draws = full_batch(Bh)
    sl = slice(rank * Bh, (rank + 1) * Bh)
    cut = lambda d: {k: v[sl] for k, v in d.items()}
    kw = dict(eikonal_weight=0.1, smooth_weight=0.01, smooth_iter=None, depth_weight=0.1, normal_weight=0.05)
    net = make_net()
    i2dist.attach_data_parallel(net, equivalent=True)
    loss_fn = i2dist.attach_loss(I2SDFLoss(**kw), net)
    torch.manual_seed(100 + rank)              # different RNG per rank: the shared randperm columns must come from rank 0
    out, losses = loss_of(net, loss_fn, cut(inp), cut(gt), cut(draws))
    iters = int(net.last_sampler_iters.item())
    extra = net.last_extra_idx.clone()
    # what would each rank have done alone (per-rank flag)?  -> shows the global OR matters for this batch
    net_solo = make_net()
    with torch.no_grad():
        net_solo({k: v.cuda() for k, v in cut(inp).items()}, draws={k: v.cuda() for k, v in cut(draws).items()})
    solo_iters = int(net_solo.last_sampler_iters.item())
    gath = [None] * world
    dist.all_gather_object(gath, {"iters": iters, "solo_iters": solo_iters, "extra": extra.cpu(), "loss": float(losses["loss"]),
                                  "rgb": out["rgb_values"].detach().cpu(), "xcalls": net.dp_state.xchg.calls})
    if rank == 0:
        ref = make_net()
        out1, losses1 = loss_of(ref, I2SDFLoss(**kw), inp, gt, draws, extra=extra)
        res = {"iters": [g["iters"] for g in gath], "solo_iters": [g["solo_iters"] for g in gath], "ref_iters": int(ref.last_sampler_iters.item()),
               "extra_equal": bool(torch.equal(gath[0]["extra"], gath[1]["extra"])), "xcalls": [g["xcalls"] for g in gath],
               "loss_dp_mean": sum(g["loss"] for g in gath) / world, "loss_ref": float(losses1["loss"]),
               "rgb_err": float((torch.cat([g["rgb"] for g in gath]) - out1["rgb_values"].detach().cpu()).abs().max()),
               "grad_err": {}}
        for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
            den = float(q.grad.abs().max())
            res["grad_err"][n] = float((p.grad - q.grad).abs().max()) / den if den > 0 else float(p.grad.abs().max())
        torch.save(res, result_path)
    # ---- multi-GPU full-image inference (N3): whole chunks dealt to the ranks, rows all-gathered into image order
    ev = make_net().eval()
    Wi, Hi, chunk = 32, 24, 100
    K = torch.eye(4); K[0, 0] = K[1, 1] = 30.0; K[0, 2], K[1, 2] = Wi / 2, Hi / 2
    pose = torch.eye(4); pose[:3, 3] = torch.tensor([0.0, 0.0, -2.0])
    ys, xs = torch.meshgrid(torch.arange(Hi), torch.arange(Wi), indexing="ij")
    img = {"uv": torch.stack([xs, ys], -1).float().reshape(1, -1, 2).cuda(), "intrinsics": K.unsqueeze(0).cuda(), "pose": pose.unsqueeze(0).cuda()}
    merged = i2dist.render_image(ev, img, chunk)
    # the ATTACHED net after its equivalent=True step (the exchange hook is installed on its plan): eval renders must stay rank-local.
    # 7 chunks over 2 ranks = 4 + 3 chunks: a per-iteration collective inside the eval sampler would hang here (or OR the flags of
    # unrelated chunks); same weights as `ev` (no optimizer step was taken), so the image must be the single-process image bit for bit
    calls0 = net.dp_state.xchg.calls
    net.eval()
    merged_att = i2dist.render_image(net, img, 110)
    if rank == 0:                                 # rank-0-only validation: forward + loss in eval mode, no collective may be entered
        loss_fn.eval()
        with torch.no_grad():
            vo = net({k: v.cuda() for k, v in cut(inp).items()})
        v_loss = float(loss_fn(vo, {k: v.cuda() for k, v in cut(gt).items() if k in ("rgb", "depth", "depth_mask")}, 10)["loss"])
        loss_fn.train()
    eval_calls = net.dp_state.xchg.calls - calls0
    net.train()
    if rank == 0:
        single = ev.render_image(img, chunk)
        single110 = ev.render_image(img, 110)
        res = torch.load(result_path)
        res["image_equal"] = all(bool(torch.equal(merged[k], single[k])) for k in single)
        res["image_rows"] = int(merged["rgb_values"].shape[0])
        res["attached_image_equal"] = all(bool(torch.equal(merged_att[k], single110[k])) for k in single110)
        res["eval_exchange_calls"] = int(eval_calls)
        res["rank0_validation_loss"] = v_loss
        torch.save(res, result_path)
    # ---- multi-GPU SDF volume (N4): contiguous slabs of the flat output, all-gathered
    from i2sdf_amd.grid import aligned_axes
    ax = aligned_axes(None, 7, torch.tensor([-0.9, -0.4, -1.1]).numpy(), torch.tensor([0.8, 0.5, 1.2]).numpy())
    vol = i2dist.sdf_volume(ev, ax, chunk=300)
    if rank == 0:
        res = torch.load(result_path)
        res["volume_equal"] = bool(torch.equal(vol, ev.sdf_volume(ax))) and tuple(vol.shape) == ax.shape_volume
        torch.save(res, result_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
