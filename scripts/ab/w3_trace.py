"""Per-workgroup timeline of wgrad3p in one production training step (development tool; needs a library built with -DW3_TRACE=1:
scripts/ab/variant_build.sh trace "-DW3_TRACE=1" wgrad.hip).   I2SDF_LIB_PATH=.../libi2sdf_trace.so python scripts/ab/w3_trace.py"""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch
import bench

args = argparse.Namespace(fused_adam=1, dp_transport="auto")
w = bench.Workload(args, torch.device("cuda:0"), 0, 1)
w.opt.step = lambda: None
chain = int(os.environ.get("CHAIN", "1"))
lib = ctypes.CDLL(os.environ["I2SDF_LIB_PATH"])
lib.i2sdf_debug_w3_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
r = w.run(1024, 1234, 2, 5, 5)
eng = r["eng"]
eng.use_chain = bool(chain)
w.run(1024, 1234, 2, 2, 2)
torch.cuda.synchronize()
lib.i2sdf_debug_w3_trace(None, 0)
w.run(1024, 1234, 2, 1, 0) if False else None
# exactly one step
inp, gt = w.inputs(1024, 1234)
out = w.net(inp); losses = w.loss_fn(out, gt, 0); w.opt.zero_grad(set_to_none=True); losses["loss"].backward()
torch.cuda.synchronize()
buf = np.zeros((32768, 8), dtype=np.uint64)
n = lib.i2sdf_debug_w3_trace(buf.ctypes.data_as(ctypes.c_void_p), 32768)
rec = buf[:n].astype(np.int64)
t0 = rec[:, 0].min()
us = lambda x: (x - t0) / 100.0
start, pro, loop_end, end = us(rec[:, 0]), us(rec[:, 1]), us(rec[:, 2]), us(rec[:, 3])
task, chunk = rec[:, 4] >> 32, rec[:, 4] & 0xffffffff
nst0, njobs = rec[:, 5], rec[:, 7] >> 16
plain, blk = (rec[:, 7] >> 8) & 1, rec[:, 7] & 3
cu = ((rec[:, 6] >> 32) << 8) | ((rec[:, 6] >> 8) & 0xff)
span = end.max()
print(f"parts {eng.parts} chain {chain}: {n} workgroups, first start -> last end {span:.1f} us; sum of workgroup times / (256 CUs x span) = {np.sum(end - start) / (256 * span):.3f}; "
      f"{len(np.unique(cu))} distinct CUs")
dur = end - start
for nj in sorted(set(njobs.tolist())):
    m = njobs == nj
    full = m & (nst0 == 64)
    print(f"  tasks with {nj} job(s): {m.sum()} workgroups ({full.sum()} with 64-stage jobs): duration median {np.median(dur[full]):.1f} us (p10 {np.percentile(dur[full], 10):.1f}, p90 {np.percentile(dur[full], 90):.1f}); "
          f"prologue {np.median((pro - start)[full]):.2f} us, stages {np.median((loop_end - pro)[full]):.1f} us = {np.median((loop_end - pro)[full]) / (64 * nj):.3f} us/stage, epilogue {np.median((end - loop_end)[full]):.2f} us; "
          f"plain {plain[m].mean():.2f} blk {np.bincount(blk[m], minlength=4).tolist()}")
# concurrency over time
edges = np.linspace(0, span, 41)
conc = [(np.minimum(end, b) - np.maximum(start, a)).clip(min=0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
print("  workgroups in flight per 1/40 of the span:", " ".join(f"{c:.0f}" for c in conc))
# per-CU idle gaps
gaps = []
for c in np.unique(cu):
    m = cu == c
    o = np.argsort(start[m]); s_, e_ = start[m][o], end[m][o]
    gaps.extend((s_[1:] - e_[:-1]).tolist())
gaps = np.array(gaps)
print(f"  gap between consecutive workgroups on one CU: median {np.median(gaps):.2f} us, p90 {np.percentile(gaps, 90):.2f} us (negative = overlapped)")
order = np.argsort(start)
print("  task of the workgroups in start order (every 16th):", " ".join(str(int(t)) for t in task[order][::16]))
