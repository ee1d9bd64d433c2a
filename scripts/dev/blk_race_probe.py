"""Where do saved tensors differ between runs when the blocked layout is on? (diagnostic for test_tail_overlap_changes_nothing)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from oracle import i2sdf_oracle as orc
from i2sdf_amd.config import synthetic_conf, NetConfig
from i2sdf_amd.engine import RenderEngine
ocfg = orc.synthetic_cfg(False)
sd = orc.perturb_params(orc.init_params(ocfg, seed=21), 0.05, seed=22)
eng = RenderEngine(NetConfig.from_conf(dict(synthetic_conf(False))))
eng.pack(eng.layout.flat_from_state_dict(sd).cuda())
g = torch.Generator().manual_seed(8)
B, n = 420, 97
M = B * n + 3 * B
x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 1.5).cuda()
nb, sb = torch.randn(M, 3, generator=g).cuda(), torch.randn(M, generator=g).cuda()
fb = torch.randn(B * n, 256, generator=g).cuda()
bulk = 256 * 128
def run(poison):
    fwd = eng.sdf_forward_grad(points=x)
    if poison:
        pass
    bw = eng.sdf_backward(fwd, sbar=sb, fbar=fb, m_fbar=B * n, nbar=nb)
    torch.cuda.synchronize()
    return {"hs": fwd["hs"].clone(), "abars": fwd["abars"].clone(), "gus": bw["gus"][1:].clone(), "gas": bw["gas"].clone(), "grad": fwd["grad"].clone()}
for overlap in (False, True):
    eng.set_tail_overlap(overlap)
    ref = run(False)
    for rep in range(12):
        cur = run(False)
        for k in cur:
            a, b = cur[k], ref[k]
            if k != "grad":
                a, b = a[:, :M], b[:, :M]
            ne = (a != b)
            if ne.any():
                idx = ne.nonzero()
                if k == "grad":
                    print(f"overlap={overlap} rep={rep} {k}: {int(ne.sum())} differ rows {int(idx[:,0].min())}..{int(idx[:,0].max())}")
                else:
                    rows = idx[:, 1]
                    print(f"overlap={overlap} rep={rep} {k}: {int(ne.sum())} differ; layers {sorted(set(idx[:,0].tolist()))} rows {int(rows.min())}..{int(rows.max())} "
                          f"(<bulk: {int((rows < bulk).sum())}, >=bulk: {int((rows >= bulk).sum())}) cols {int(idx[:,2].min())}..{int(idx[:,2].max())} "
                          f"nan cur {int(torch.isnan(a).sum())} ref {int(torch.isnan(b).sum())} sample {a[ne][:3].tolist()} vs {b[ne][:3].tolist()}")
print("done; blocked prefix", eng.blocked_points(0, M, eng.pad_rows(M)))
