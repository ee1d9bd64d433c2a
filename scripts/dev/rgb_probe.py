import os, sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import i2sdf_oracle as orc
from helpers import rel_max
from test_gpu_train_forward import make_engine
from i2sdf_amd.config import synthetic_conf
ocfg = orc.synthetic_cfg(False)
sd = orc.perturb_params(orc.init_params(ocfg, seed=13), 0.05, seed=14)
g = torch.Generator().manual_seed(6)
for Bn in ((320, 97), (400, 97), (31, 9), (243, 128)):
    B, n = Bn; M, F = B * n, 256
    x = (torch.rand(M, 3, generator=g) * 2 - 1)
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, generator=g), dim=1)
    cw = torch.randn(M, 3, generator=g)
    res = {}
    for mode in (True, False):
        conf = dict(synthetic_conf(False)); conf["bf16x3"] = mode
        eng = make_engine(conf, sd)
        flat = eng.layout.flat_from_state_dict(sd).cuda()
        fwd = eng.sdf_forward_grad(points=x.cuda())
        rgb_h, rs, pev = eng.rgb_forward(dirs.cuda(), n, fwd["feat"], M)
        gar, ga_last, fbar = eng.rgb_backward(rgb_h, cw.cuda(), rs, M)
        bw = eng.sdf_backward(fwd, sbar=None, fbar=fbar, m_fbar=M, nbar=None)
        gflat = torch.zeros_like(flat)
        eng.weight_grads(flat, gflat, fwd, bw, M_main=M, fbar=fbar, rgb_fw={"pev": pev, "rs": rs}, rgb_bw={"gar": gar, "ga_last": ga_last})
        res[mode] = dict(rgb=rgb_h.cpu(), rs=rs[:, :M].cpu(), gar=gar[:, :M].cpu(), fbar=fbar[:M].cpu(), g=eng.layout.state_dict_from_flat(gflat.cpu()))
        # reference bias sums straight from gar
        res[mode]["bsum"] = [gar[l, :M].double().sum(0).cpu() for l in range(gar.shape[0])]
    a, b = res[True], res[False]
    print(B, n, "rgb", f"{rel_max(a['rgb'], b['rgb']):.1e}", "rs", f"{rel_max(a['rs'], b['rs']):.1e}", "gar", f"{rel_max(a['gar'], b['gar']):.1e}", "fbar", f"{rel_max(a['fbar'], b['fbar']):.1e}")
    for mode in (True, False):
        r = res[mode]
        errs = [f"{rel_max(r['g'][f'rendering_network.lin{l}.bias'], r['bsum'][l]):.1e}" for l in range(4)]
        print("   bf16x3", mode, "bias grad vs column sums of gar:", errs)
