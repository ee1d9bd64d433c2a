#!/usr/bin/env python3
"""Design probe for tests/test_gpu_psnr_ensemble.py (GPU): how far apart do MEMBERS OF ONE ARM land?  Runs the production path only
(it is 8x cheaper than the eager restatement) with 8 members per variant -- members differ by 1e-6 weight noise and by their draw seeds,
exactly as in the test (the test now uses variant I) -- and prints mean +- std of the tail PSNR over several windows and of the held-out PSNR, per variant of
teacher / batch size / step count.  The variant with the smallest spread that still trains by > 5 dB is the one the test uses: the test's
standard error is std / sqrt(16).

    python scripts/ab/ensemble_probe.py [variant ...]
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from oracle import i2sdf_oracle as orc
import test_gpu_psnr_ensemble as T


def teacher(ocfg, sd0, kind):
    if kind == "r075":
        return T._teacher_weights(ocfg)
    sd = T._teacher_weights(ocfg)
    if kind == "r060":
        b = sd["implicit_network.lin8.bias"].clone(); b[0] = -0.6
        sd["implicit_network.lin8.bias"] = b
    elif kind == "samegeo":                      # the student's own geometry net: only the colours have to be learnt
        for k, v in sd0.items():
            if k.startswith("implicit_network."):
                sd[k] = v.clone()
    return sd


def run(name, kind, B, steps, decay=1.0, members=8, lr=5e-4):
    from i2sdf_amd import I2SDFNetwork, I2SDFLoss, FusedAdam, synthetic_conf
    dev = torch.device("cuda:0")
    conf = dict(synthetic_conf(False)); conf["use_normal"] = True
    ocfg = orc.synthetic_cfg(False); ocfg.use_normal = True
    sd0 = orc.init_params(ocfg, seed=11); sd0["density.beta"] = torch.tensor(0.05)
    tnet = I2SDFNetwork(conf); tnet.load_state_dict(teacher(ocfg, sd0, kind)); tnet = tnet.to(dev).eval()
    batches = []
    for step in range(steps):
        inp = T._rays(step, dev, n=B)
        batches.append((inp, T._targets(tnet, inp)))
    vin = T._rays(1_000_000, dev, n=2048); vgt = T._targets(tnet, vin)
    net = I2SDFNetwork(conf).to(dev).train()
    eng = net._engine_for(dev)
    loss_fn = I2SDFLoss(**T.LKW)
    curves, held = [], []
    for s in range(members):
        net.load_state_dict(T._member_init(sd0, s)); net.train()
        opt = FusedAdam(net, lr=lr, eps=1e-15)
        # the reference's scheduler (model/trainer/recon.py:204-206: ExponentialLR, total decay `sched_decay_rate` = 0.1 over the run), compressed
        sched = torch.optim.lr_scheduler.ExponentialLR(opt, decay ** (1.0 / steps))
        ps = []
        for step in range(steps):
            inp, gt = batches[step]
            out = net(inp, draws=eng.training_draws(B, 7_000_000 + 100_003 * s + step, dev, net.scene_bounding_sphere, want_eik=True))
            l = loss_fn(out, gt, step)["loss"]
            opt.zero_grad(set_to_none=True); l.backward(); opt.step(); sched.step()
            ps.append(orc.get_psnr(out["rgb_values"].detach(), gt["rgb"]))
        curves.append(torch.stack(ps).cpu())
        net.eval()
        with torch.no_grad():
            held.append(float(orc.get_psnr(net(vin)["rgb_values"], vgt["rgb"])))
    C = torch.stack(curves)                      # (members, steps)
    ms = lambda x: f"{float(x.mean()):7.3f} +- {float(x.std()):.3f}"
    wins = [(a, a + 50) for a in range(50, steps, 100) if a + 50 <= steps]
    if (steps - 50, steps) not in wins:
        wins.append((steps - 50, steps))
    print(f"== {name}: teacher {kind}, {B} rays, {steps} steps, {members} members, lr {lr} x {decay} over the run: step 0 {ms(C[:, 0])} dB", flush=True)
    for a, b in wins:
        print(f"   mean PSNR of steps {a:4d}..{b - 1:4d}: {ms(C[:, a:b].mean(1))} dB over members (per-step std across members {float(C[:, a:b].std(0).mean()):.3f})")
    h = torch.tensor(held)
    print(f"   held-out (2048 rays): {ms(h)} dB   -> SE with 16 members: tail {float(C[:, -50:].mean(1).std()) / 4:.3f}, held-out {float(h.std()) / 4:.3f} dB", flush=True)


VARIANTS = {
    "A": ("r075", 256, 300), "B": ("r060", 256, 300), "C": ("samegeo", 256, 300), "D": ("r075", 256, 600),
    "E": ("samegeo", 1024, 300), "F": ("samegeo", 256, 600),
    "G": ("samegeo", 1024, 300, 0.1), "H": ("samegeo", 1024, 300, 0.01), "I": ("r075", 1024, 300, 0.01), "J": ("samegeo", 256, 300, 0.01),
    "K": ("r075", 256, 300, 0.1), "L": ("r075", 1024, 400, 0.003),
}

if __name__ == "__main__":
    for v in (sys.argv[1:] or list(VARIANTS)):
        run(v, *VARIANTS[v])
