"""Knock-out / A-B timing with the weights held fixed: bench.py's training step WITHOUT the optimizer update, so a library variant that
computes wrong weight gradients (a timing-only knock-out build, scripts/ab/variant_build.sh) still sees the same points,
the same sampler decisions and the same operand values in every step.  Prints the step time and the per-entry-point times.
    I2SDF_LIB_PATH=.../libi2sdf_NAME.so I2SDF_PARTS=2 python scripts/ab/fixed_weights_time.py [steps]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
args = argparse.Namespace(fused_adam=1, dp_transport="auto")
w = bench.Workload(args, torch.device("cuda:0"), 0, 1)
w.opt.step = lambda: None
r = w.run(1024, 1234, 2, steps, 5, timing=True, windows=5, profile_steps=10)
short = {'i2sdf_weight_grads': 'wgrad', 'i2sdf_sdf_backward': 'sdf_bwd', 'i2sdf_sdf_forward_grad': 'sdf_fwdg', 'i2sdf_sample_rays': 'sampler',
         'i2sdf_rgb_forward': 'rgb_f', 'i2sdf_rgb_backward': 'rgb_b'}
kt = r["ktimes"] or {}
print(os.path.basename(os.environ.get("I2SDF_LIB_PATH", "in-tree")), "step %.4f ms" % (r["dt"] / steps * 1e3),
      " ".join("%s=%.3f" % (short[k], kt[k][0] / r["prof_steps"]) for k in short if k in kt), flush=True)
