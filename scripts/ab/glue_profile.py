"""Which torch-side glue launches (copies, fills, cats) does one training step issue, and from where?  torch.profiler over 3 steps of
bench.py's workload; prints the aten ops that launch device work other than the library's kernels, with their Python call sites."""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from torch.profiler import profile, ProfilerActivity
import bench

args = argparse.Namespace(fused_adam=1, dp_transport="auto")
w = bench.Workload(args, torch.device("cuda:0"), 0, 1)
w.run(1024, 1234, 2, 3, 3)
inp, gt = w.inputs(1024, 1234)
def step():
    out = w.net(inp); losses = w.loss_fn(out, gt, w.step_no); w.opt.zero_grad(set_to_none=True); losses["loss"].backward(); w.opt.step(); w.step_no += 1
step(); torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
ev = prof.events()
# device-side launches grouped by kernel name
from collections import Counter, defaultdict
names = Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        names[e.name[:60]] += 1
print("device activities per step:")
for k, v in names.most_common():
    print(f"  {v / N:6.1f}  {k}")
# leaf aten ops (no aten children) that are not pure allocation / view bookkeeping, with their call sites in this repository
NOLAUNCH = ("aten::empty", "aten::view", "aten::reshape", "aten::as_strided", "aten::select", "aten::slice", "aten::detach", "aten::alias",
            "aten::unsqueeze", "aten::squeeze", "aten::expand", "aten::t", "aten::transpose", "aten::permute", "aten::_unsafe_view", "aten::item",
            "aten::_local_scalar_dense", "aten::resize_", "aten::set_", "aten::lift_fresh", "aten::result_type", "aten::is_nonzero", "aten::unbind",
            "aten::narrow", "aten::empty_like", "aten::empty_strided", "aten::to", "aten::_to_copy", "aten::contiguous", "aten::clone",
            "aten::zeros", "aten::ones", "aten::full", "aten::zeros_like", "aten::ones_like", "aten::new_empty", "aten::new_zeros", "aten::flatten")
sites = defaultdict(lambda: [0, Counter()])
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::") or e.name in NOLAUNCH:
        continue
    if any(c.name.startswith("aten::") and c.name not in NOLAUNCH for c in e.cpu_children):
        continue
    st = [f for f in (e.stack or []) if ("i2sdf_amd" in f or "bench.py" in f or "glue_profile" in f)]
    key = (e.name, str(e.input_shapes)[:70])
    sites[key][0] += 1; sites[key][1][st[0][-100:] if st else "?"] += 1
print("leaf aten ops (count per step, shapes, call sites):")
for k, (n, c) in sorted(sites.items(), key=lambda kv: -kv[1][0]):
    print(f"  {n / N:5.1f}  {k[0]:28s} {k[1]}")
    for s_, m in c.most_common(4):
        print(f"         {m / N:4.1f} x {s_}")
