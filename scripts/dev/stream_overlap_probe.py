"""Feasibility: can the idle CUs of a kernel's last (partial) round be filled by the next kernel via stream priorities?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from i2sdf_amd.config import NetConfig, synthetic_conf
from i2sdf_amd.engine import RenderEngine

cfg = NetConfig.from_conf(synthetic_conf())
eng = RenderEngine(cfg)
flat = eng.layout.init_flat(torch.Generator().manual_seed(0)).cuda()
eng.pack(flat)
M = 102400
Mb = 768 * 128
x = (torch.rand(M, 3, device="cuda") * 2 - 1) * 2
dirs = torch.nn.functional.normalize(torch.randn(M, 3, device="cuda"), dim=1)
hi = torch.cuda.Stream(priority=-1)
lo = torch.cuda.Stream(priority=0)


def serial():
    fw = eng.sdf_forward_grad(points=x)
    eng.rgb_forward(dirs, 1, fw["feat"], M)


def overlapped():
    cur = torch.cuda.current_stream()
    hi.wait_stream(cur); lo.wait_stream(cur)
    with torch.cuda.stream(hi):
        fa = eng.sdf_forward_grad(points=x[:Mb])
        eng.rgb_forward(dirs[:Mb], 1, fa["feat"], Mb)
    with torch.cuda.stream(lo):
        fb = eng.sdf_forward_grad(points=x[Mb:])
        eng.rgb_forward(dirs[Mb:], 1, fb["feat"], M - Mb)
    cur.wait_stream(hi); cur.wait_stream(lo)
    return fa, fb


def timeit(fn, n=10):
    for _ in range(3):
        r = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("serial     ms", timeit(serial))
print("overlapped ms", timeit(overlapped))
print("serial     ms", timeit(serial))
