#!/usr/bin/env python3
"""debug: the saved tensors and parameter gradients of one SDF forward / backward with packed 24-bit records against fp32 storage (one process)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import i2sdf_oracle as orc
from test_gpu_train_forward import make_engine
from i2sdf_amd.config import synthetic_conf

M = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ocfg, conf = orc.synthetic_cfg(False), synthetic_conf(False)
sd = orc.perturb_params(orc.init_params(ocfg, seed=11), 0.05, seed=12)
eng = make_engine(conf, sd)
flat = eng.layout.flat_from_state_dict(sd).cuda()
g = torch.Generator().manual_seed(5)
x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 1.5).cuda()
F = 256
sw, fw_ = torch.randn(M, generator=g).cuda(), (torch.randn(M, F, generator=g) * 0.1).cuda()
res = {}
for mode in (0, 1):
    eng.set_saves24(bool(mode))
    fwd = eng.sdf_forward_grad(points=x)
    n = fwd["grad"]; nn = n.norm(dim=1, keepdim=True); nbar = 2 * (nn - 1) * n / nn
    Mp = fwd["Mp"]
    fbar = torch.zeros(Mp, F, device="cuda"); fbar[:M] = fw_
    bw = eng.sdf_backward(fwd, sbar=sw, fbar=fbar, m_fbar=M, nbar=nbar)
    gflat = torch.zeros_like(flat)
    eng.weight_grads(flat, gflat, fwd, bw, M_main=M, fbar=fbar)
    torch.cuda.synchronize()
    print("mode", mode, "saves24 points", eng.saves24_points(M, Mp), "of", Mp, "parts", eng.parts, "x2", eng.wgrad_bf16x2)
    res[mode] = {"abars": eng.saved_pm("abars", fwd["abars"], M).cpu(), "gus": eng.saved_pm("gus", bw["gus"], M).cpu(), "gas": eng.saved_pm("gas", bw["gas"], M).cpu(),
                 "grads": eng.layout.state_dict_from_flat(gflat.cpu())}
for name in ("abars", "gus", "gas"):
    a, b = res[0][name], res[1][name]
    for l in range(a.shape[0]):
        d = (a[l].double() - b[l].double()).abs().max().item(); s = a[l].double().abs().max().item()
        print(f"{name}[{l}]: max |diff| / max |ref| = {d / max(s, 1e-300):.3e}   (max ref {s:.3e})")
for k in res[0]["grads"]:
    a, b = res[0]["grads"][k].double(), res[1]["grads"][k].double()
    if a.numel() < 2 or not k.startswith("implicit"): continue
    print(f"grad {k}: {(a - b).abs().max().item() / max(a.abs().max().item(), 1e-300):.3e}")
b0, b1 = res[0]["grads"]["implicit_network.lin0.bias"].double(), res[1]["grads"]["implicit_network.lin0.bias"].double()
want = res[1]["gas"][0].double().sum(0)
print("bias lin0: fp32-storage run vs column sums of decoded gas[0]:", ((b0 - want).abs().max() / want.abs().max()).item())
print("bias lin0: packed run vs the same:", ((b1 - want).abs().max() / want.abs().max()).item())
torch.set_printoptions(precision=4, linewidth=200)
print("want[:32]", want[:32].float()); print("got [:32]", b1[:32].float())
print("ratio[:64]", (b1 / want)[:64].float())
# try permutations: got[n] == want[perm(n)]?
for n in range(16):
    j = int((want - b1[n]).abs().argmin()); print(n, "->", j, float(b1[n]), float(want[j]))
