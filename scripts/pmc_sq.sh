#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d gpurun_out/pmc2 -o sq --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc2/sq.log 2>&1
python - <<'PY'
import csv,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter(); seen=set()
for r in csv.DictReader(open("gpurun_out/pmc2/sq_counter_collection.csv")):
    k=r["Kernel_Name"]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); n[k]+=1
dur=collections.defaultdict(float)
for r in csv.DictReader(open("gpurun_out/pmc2/sq_kernel_trace.csv")):
    dur[r["Kernel_Name"]]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
rows=[]
for k in agg:
    a=agg[k]; d=dur[k]/n[k]
    if a["SQ_VALU_MFMA_BUSY_CYCLES"]==0: continue
    rows.append((dur[k], k[:48], n[k], d/1e3, a["GRBM_GUI_ACTIVE"]/n[k]/8/d, a["SQ_VALU_MFMA_BUSY_CYCLES"]/a["GRBM_GUI_ACTIVE"]/128, a["SQ_WAIT_ANY"]/a["SQ_WAVE_CYCLES"], a["SQ_WAIT_INST_ANY"]/a["SQ_WAVE_CYCLES"], a["SQ_ACTIVE_INST_VALU"]/a["SQ_WAVE_CYCLES"]))
for r in sorted(rows, reverse=True):
    print("%-48s n=%2d %8.1f us clk %.2f GHz mfma_util %.3f wait_any %.3f wait_inst %.3f valu %.3f" % r[1:])
PY
import os
