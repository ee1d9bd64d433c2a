#!/bin/bash
# Refresh profiles/: rocprofv3 kernel stats of the default bench command + PMC passes (counters only, separate runs).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r4}
BARGS="--no-cpu-baseline --no-extras --scaling weak --windows 1 --profile-steps 0 --no-live-traffic"     # the headline workload only (same kernels and launch shapes as the timed windows of the default command)
O=gpurun_out/profiles_$R
mkdir -p $O
PASSES=${PASSES:-"stats sq fetch write"}      # e.g. PASSES=sq for a quick look at one kernel's issue counters
has() { [[ " $PASSES " == *" $1 "* ]]; }
has stats && rocprofv3 --kernel-trace --stats -d $O -o ${R}_bench --output-format csv -- python bench.py --steps 5 --warmup 2 $BARGS > $O/${R}_bench_under_rocprof.log 2>&1
run() { rocprofv3 --kernel-trace --pmc $2 -d $O -o ${R}_$1 --output-format csv -- python bench.py --steps 2 --warmup 1 $BARGS > $O/${R}_$1.log 2>&1; }
has sq && run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
has fetch && run fetch "FETCH_SIZE"
has write && run write "WRITE_SIZE"
python - <<PY
import csv, glob, collections, os
R="$R"; O="$O"
for tag in ("sq","fetch","write"):
    f=f"{O}/{R}_{tag}_counter_collection.csv"
    if not os.path.exists(f): print("missing", f); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); seen=set()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen: seen.add(r["Dispatch_Id"]); cnt[k]+=1
    dur=collections.defaultdict(float)
    for r in csv.DictReader(open(f"{O}/{R}_{tag}_kernel_trace.csv")):
        dur[r["Kernel_Name"][:70]]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    with open(f"{O}/{R}_pmc_{tag}_summary.csv","w") as o:
        names=sorted({c for k in agg for c in agg[k]})
        o.write("kernel,dispatches,avg_duration_us,"+",".join(n+"_per_dispatch" for n in names)+"\n")
        for k in sorted(agg, key=lambda k:-dur[k]):
            o.write(f'"{k}",{cnt[k]},{dur[k]/cnt[k]/1e3:.1f},'+",".join(f"{agg[k].get(c,0)/cnt[k]:.0f}" for c in names)+"\n")
    for g in glob.glob(f"{O}/{R}_{tag}_counter_collection.csv")+glob.glob(f"{O}/{R}_{tag}_kernel_trace.csv")+glob.glob(f"{O}/{R}_{tag}_agent_info.csv"): os.remove(g)
for g in glob.glob(f"{O}/{R}_bench_kernel_trace.csv")+glob.glob(f"{O}/{R}_bench_agent_info.csv"): os.remove(g)
PY
python -c "import sys; sys.path.insert(0, \".\"); import bench; open(\"$O/${R}_source_hash.txt\", \"w\").write(bench.source_hash() + \"  sha256[:16] over i2sdf_amd/csrc/*.h* *.cpp + include/i2sdf.h at the time of these passes (bench.py: profiled_traffic)\\n\")"
ls $O
