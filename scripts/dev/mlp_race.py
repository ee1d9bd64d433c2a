"""Bitwise run-to-run reproducibility of the MLP kernels' outputs (same inputs, 25 repetitions): a missing DMA wait shows up here."""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import i2sdf_oracle as orc
from test_gpu_train_forward import make_engine
from i2sdf_amd.config import synthetic_conf
ocfg = orc.synthetic_cfg(False)
sd = orc.perturb_params(orc.init_params(ocfg, seed=13), 0.05, seed=14)
g = torch.Generator().manual_seed(6)
B, n = 1024, 97
M = B * n + 3072
x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 1.5).cuda()
dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1).cuda()
cw = torch.randn(B * n, 3, generator=g).cuda()
nb = torch.randn(M, 3, generator=g).cuda(); sb = torch.randn(M, generator=g).cuda()
eng = make_engine(synthetic_conf(False), sd)
ref = None
for rep in range(25):
    fwd = eng.sdf_forward_grad(points=x)
    rgb_h, rs, pev = eng.rgb_forward(dirs, n, fwd["feat"], B * n)
    gar, ga_last, fbar = eng.rgb_backward(rgb_h, cw, rs, B * n)
    bw = eng.sdf_backward(fwd, sbar=sb, fbar=fbar, m_fbar=B * n, nbar=nb)
    sdfs = eng.sdf_forward(x)
    torch.cuda.synchronize()
    cur = {"sdf": fwd["sdf"], "feat": fwd["feat"][:M], "grad": fwd["grad"], "hs": fwd["hs"][:, :M], "abars": fwd["abars"][:, :M], "rgb": rgb_h, "rs": rs[:, :B * n],
           "gar": gar[:, :B * n], "fbar": fbar[:B * n], "gus": bw["gus"][1:, :M], "gas": bw["gas"][:, :M], "sdf_only": sdfs}
    if ref is None:
        ref = {k: v.clone() for k, v in cur.items()}
        continue
    bad = {k: int((cur[k] != ref[k]).sum()) for k in cur if not torch.equal(cur[k], ref[k])}
    if bad: print("rep", rep, "mismatches", bad)
print("done")
