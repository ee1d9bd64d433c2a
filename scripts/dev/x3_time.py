"""Time the sdf-only forward (131072 points = one sampler pass at 1024 rays) in fp32-MFMA and bf16x3 mode."""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import i2sdf_oracle as orc
from i2sdf_amd.config import NetConfig, synthetic_conf
from i2sdf_amd.engine import RenderEngine
ocfg = orc.synthetic_cfg(False)
sd = orc.perturb_params(orc.init_params(ocfg, seed=3), 0.05, seed=4)
eng = RenderEngine(NetConfig.from_conf(synthetic_conf(False)))
eng.pack(eng.layout.flat_from_state_dict(sd).cuda())
M = 131072
x = ((torch.rand(M, 3) * 2 - 1) * 2.5).cuda()
import os
for mode in ((True, True, True) if os.environ.get('ONLY_X3') else (False, True, False, True)):
    eng.set_sdf_forward_bf16x3(mode)
    for _ in range(3): eng.sdf_forward(x)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): eng.sdf_forward(x)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"bf16x3={mode}: {ms:.4f} ms  ({M * 0.918e6 / ms / 1e9:.1f} TFLOP/s fp32-equivalent)")
