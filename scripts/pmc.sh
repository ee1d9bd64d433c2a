#!/bin/bash
# PMC passes for the bench (counters only: no sys/hip trace alongside, as gpurun requires)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
run() { rocprofv3 --kernel-trace --pmc $2 -d gpurun_out/pmc -o $1 --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc/$1.log 2>&1; }
run sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
ls gpurun_out/pmc
python - <<'PY'
import csv, glob, collections, os
for tag in ("sq","fetch","write","lds"):
    f=glob.glob(f"gpurun_out/pmc/{tag}_counter_collection.csv")
    if not f: print("missing", tag); continue
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    seen=set()
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
        key=(r["Dispatch_Id"])
        if key not in seen: seen.add(key); cnt[k]+=1
    with open(f"gpurun_out/pmc/{tag}_summary.csv","w") as o:
        names=sorted({c for k in agg for c in agg[k]})
        o.write("kernel,dispatches,"+",".join(names)+"\n")
        for k in sorted(agg, key=lambda k:-sum(agg[k].values())):
            o.write(f'"{k}",{cnt[k]},'+",".join(f"{agg[k].get(c,0):.0f}" for c in names)+"\n")
    os.remove(f[0])
PY
