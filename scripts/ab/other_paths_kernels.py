"""Kernel times of the paths beside the headline step (run under `rocprofv3 --kernel-trace --stats`): MODE = image (one 640x480 eval
render through i2sdf_render_image) or natural (training step with the data-dependent sampler loop) -- a check that no kernel of these paths is an outlier the headline profile cannot show.
    rocprofv3 --kernel-trace --stats -d OUT -o NAME --output-format csv -- python scripts/ab/other_paths_kernels.py MODE"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import bench

mode = sys.argv[1]
args = argparse.Namespace(fused_adam=1, dp_transport="auto")
dev = torch.device("cuda:0")
w = bench.Workload(args, dev, 0, 1)
if mode == "image":
    r = bench.full_image(w, dev, 97, reps=1)
    print("image", r["s_per_image"], flush=True)
elif mode == "natural":
    r = w.run(1024, 1234, 0, 10, 3, timing=True, windows=2)
    print("natural-k step %.4f ms" % (r["dt"] / 10 * 1e3), flush=True)
