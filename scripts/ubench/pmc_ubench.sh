#!/bin/bash
# clock and matrix-pipe occupancy of every mode of scripts/ubench/mfma_stage (one rocprofv3 --pmc pass, counters only); run on the GPU box
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ub_pmc
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -d /tmp/ub_pmc -o ub --output-format csv -- $GRAFT_REPO_ROOT/scripts/ubench/mfma_stage > /tmp/ub_pmc.log 2>&1
python3 - <<'PY'
import csv, glob, re
cc = glob.glob('/tmp/ub_pmc/**/ub_counter_collection.csv', recursive=True)[0]
kt = glob.glob('/tmp/ub_pmc/**/ub_kernel_trace.csv', recursive=True)[0]
agg, cnt, seen, dur = {}, {}, set(), {}
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"]; a = agg.setdefault(k, {})
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"]); cnt[k] = cnt.get(k, 0) + 1
for r in csv.DictReader(open(kt)):
    dur[r["Kernel_Name"]] = dur.get(r["Kernel_Name"], 0.0) + int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, a in agg.items():
    m = re.search(r"<(\d+)>", k); us = dur[k] / cnt[k] / 1e3; g = a["GRBM_GUI_ACTIVE"]
    mf = 1024 * 4 * 512 * 96 * 32768.0
    print(f"mode {m.group(1) if m else k[:30]:>4}: {us:8.1f} us  clock {g / cnt[k] / 8 / (us * 1e3):.3f} GHz  mfma_busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (128 * g):.3f}  {mf / (us * 1e-6) / 1e12:7.1f} TFLOP/s")
PY
