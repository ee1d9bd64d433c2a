"""Device timeline of one production training step from a rocprofv3 --kernel-trace run of bench.py: wall time of the step, time during
which NO kernel is running (the gaps, with the kernel that follows each), and per kernel name the summed duration.
    python scripts/ab/timeline_gaps.py            (on the GPU box; starts rocprofv3 itself)"""
import csv, glob, os, subprocess, sys, tempfile
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.join(here, "..", "..")
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    cmd = ["rocprofv3", "--kernel-trace", "-d", d, "-o", "tl", "--output-format", "csv", "--", sys.executable, os.path.join(root, "bench.py"),
           "--steps", "6", "--warmup", "3", "--no-cpu-baseline", "--no-extras", "--scaling", "weak", "--windows", "1", "--profile-steps", "0", "--no-live-traffic"]
    subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=400, check=True)
    f = glob.glob(os.path.join(d, "**", "tl_kernel_trace.csv"), recursive=True)[0]
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:44]
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
for which in (-3, -2):
    a, b = adam[which], adam[which + 1]
    seg = rows[a + 1:b + 1]
    t0, t1 = rows[a][1], seg[-1][1]
    cur, idle = t0, []
    for s, e, n in seg:
        if s > cur:
            idle.append((s - cur, short(n), (cur - t0) / 1e3))
        cur = max(cur, e)
    print(f"step: {(t1 - t0) / 1e3:.1f} us from the end of one adam_kernel to the end of the next, {len(seg)} kernels; no kernel running for "
          f"{sum(x[0] for x in idle) / 1e3:.1f} us in {len(idle)} gaps")
    for g, n, at in sorted(idle, reverse=True)[:14]:
        print(f"    {g / 1e3:6.1f} us before {n:44s} at +{at:7.1f} us")
seq = [(short(n), (s - t0) / 1e3, (e - s) / 1e3) for s, e, n in seg]
print("kernel sequence of the last analysed step (start us, duration us):")
for n, s, dur in seq:
    print(f"   +{s:8.1f} {dur:8.1f}  {n}")
