"""Clock, matrix-pipe occupancy and parked-wave share per kernel of one library variant (fixed weights, scripts/ab/fixed_weights_time.py):
one rocprofv3 --kernel-trace --pmc pass (counters only).   python scripts/ab/pmc_variant.py [kernel-substring ...]
    I2SDF_LIB_PATH selects the variant; prints one line per matching kernel."""
import csv, os, subprocess, sys, tempfile
here = os.path.dirname(os.path.abspath(__file__))
pats = sys.argv[1:] or ["wgrad3p"]
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE", "-d", d, "-o", "v",
           "--output-format", "csv", "--", sys.executable, os.path.join(here, "fixed_weights_time.py"), "6"]
    subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
    agg, cnt, seen, dur = {}, {}, set(), {}
    for row in csv.DictReader(open(os.path.join(d, "v_counter_collection.csv"))):
        k = row["Kernel_Name"]
        a = agg.setdefault(k, {})
        a[row["Counter_Name"]] = a.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        if row["Dispatch_Id"] not in seen:
            seen.add(row["Dispatch_Id"]); cnt[k] = cnt.get(k, 0) + 1
    for row in csv.DictReader(open(os.path.join(d, "v_kernel_trace.csv"))):
        dur[row["Kernel_Name"]] = dur.get(row["Kernel_Name"], 0.0) + (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    lib = os.path.basename(os.environ.get("I2SDF_LIB_PATH", "in-tree"))
    for k, a in agg.items():
        if not any(p in k for p in pats):
            continue
        us = dur[k] / cnt[k] / 1e3
        g = a["GRBM_GUI_ACTIVE"]
        print(f"{lib} {k[:70]}: {us:.1f} us/launch (serialised by the counters), clock {g / cnt[k] / 8 / (us * 1e3):.3f} GHz, "
              f"mfma_busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (128 * g):.3f}, busy cycles/launch {a['SQ_VALU_MFMA_BUSY_CYCLES'] / cnt[k] / 1024:.0f} per SIMD, "
              f"parked {a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES']:.3f}", flush=True)
