import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import i2sdf_oracle as orc
from test_gpu_train_forward import make_engine
from i2sdf_amd.config import synthetic_conf
ocfg = orc.synthetic_cfg(False)
sd = orc.perturb_params(orc.init_params(ocfg, seed=13), 0.05, seed=14)
g = torch.Generator().manual_seed(6)
B, n = 4700, 7
M = B * n
x = (torch.rand(M, 3, generator=g) * 2 - 1)
dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1)
cw = torch.randn(M, 3, generator=g)
eng = make_engine(synthetic_conf(False), sd)
flat = eng.layout.flat_from_state_dict(sd).cuda()
fwd = eng.sdf_forward_grad(points=x.cuda())
rgb_h, rs, pev = eng.rgb_forward(dirs.cuda(), n, fwd["feat"], M)
gar, ga_last, fbar = eng.rgb_backward(rgb_h, cw.cuda(), rs, M)
nb = torch.randn(M, 3, generator=g).cuda()
bw = eng.sdf_backward(fwd, sbar=torch.randn(M, generator=g).cuda(), fbar=fbar, m_fbar=M - 29, nbar=nb)
ref = None
names = [(nme, off, shp) for nme, off, shp in eng.layout.entries]
for rep in range(40):
    gflat = torch.zeros_like(flat)
    eng.weight_grads(flat, gflat, fwd, bw, M_main=M - 29, fbar=fbar, rgb_fw={"pev": pev, "rs": rs}, rgb_bw={"gar": gar, "ga_last": ga_last})
    torch.cuda.synchronize()
    if ref is None: ref = gflat.clone(); continue
    bad = (gflat != ref).nonzero().flatten()
    if bad.numel():
        where = {}
        for i in bad.tolist()[:2000]:
            for nme, off, shp in names:
                sz = 1
                for q in shp: sz *= q
                if off <= i < off + sz: where[nme] = where.get(nme, 0) + 1; break
        print("rep", rep, "mismatching entries", bad.numel(), where, "max abs diff", float((gflat - ref).abs().max()))
print("done")
