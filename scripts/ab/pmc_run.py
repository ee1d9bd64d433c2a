#!/usr/bin/env python3
"""One rocprofv3 --kernel-trace --pmc pass (counters only) over a command; per-kernel summary of the kernels matching the patterns.
    python scripts/ab/pmc_run.py SET pattern [pattern ...] -- command ...
SET: icache (instruction cache requests / hits / misses, fetches in flight), sq (clock, matrix-pipe occupancy, parked / issue-stalled wave share), lds (LDS instruction counts, bank conflicts, LDS issue stalls),
     mem (FETCH_SIZE) or memw (WRITE_SIZE)."""
import csv, os, subprocess, sys, tempfile
SETS = {"sq": ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"],
        "lds": ["SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU"],
        "mem": ["FETCH_SIZE", "GRBM_GUI_ACTIVE"], "memw": ["WRITE_SIZE"],
        "l2": ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum", "GRBM_GUI_ACTIVE"],
        "vmem": ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_MFMA", "SQ_INST_CYCLES_VMEM", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES"],
        "icache": ["SQC_ICACHE_REQ", "SQC_ICACHE_HITS", "SQC_ICACHE_MISSES", "SQC_ICACHE_MISSES_DUPLICATE", "SQ_IFETCH", "SQ_IFETCH_LEVEL", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES"]}
which = sys.argv[1]
sep = sys.argv.index("--")
pats, cmd = sys.argv[2:sep], sys.argv[sep + 1:]
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    full = ["rocprofv3", "--kernel-trace", "--pmc"] + SETS[which] + ["-d", d, "-o", "v", "--output-format", "csv", "--"] + cmd
    r = subprocess.run(full, env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, text=True)
    if r.returncode != 0 or not os.path.exists(os.path.join(d, "v_counter_collection.csv")):
        print(f"rocprofv3 failed for set {which} (rc {r.returncode}): " + r.stdout[-600:].replace("\n", " | "), flush=True)
        sys.exit(0)
    agg, cnt, seen, dur = {}, {}, set(), {}
    for row in csv.DictReader(open(os.path.join(d, "v_counter_collection.csv"))):
        k = row["Kernel_Name"]
        a = agg.setdefault(k, {})
        a[row["Counter_Name"]] = a.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        if row["Dispatch_Id"] not in seen:
            seen.add(row["Dispatch_Id"]); cnt[k] = cnt.get(k, 0) + 1
    for row in csv.DictReader(open(os.path.join(d, "v_kernel_trace.csv"))):
        dur[row["Kernel_Name"]] = dur.get(row["Kernel_Name"], 0.0) + (int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, a in sorted(agg.items(), key=lambda kv: -dur.get(kv[0], 0)):
        if pats and not any(p in k for p in pats):
            continue
        n = cnt[k]
        us = dur[k] / n / 1e3
        line = f"{k[:64]}: n={n} {us:.1f} us"
        g = a.get("GRBM_GUI_ACTIVE", 0.0)
        if g:
            line += f", clock {g / n / 8 / (us * 1e3):.3f} GHz"
        if which == "sq":
            wc = a["SQ_WAVE_CYCLES"]
            line += (f", mfma_busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (128 * g):.3f}, parked {a['SQ_WAIT_ANY'] / wc:.3f}, issue-stalled {a['SQ_WAIT_INST_ANY'] / wc:.3f}, "
                     f"issuing {a['SQ_ACTIVE_INST_ANY'] / wc:.3f} (valu {a['SQ_ACTIVE_INST_VALU'] / wc:.3f}), wave-cycles/launch {wc / n:.3e}")
        elif which == "lds":
            wc = a["SQ_WAVE_CYCLES"]
            line += (f", lds insts/launch {a['SQ_INSTS_LDS'] / n:.3e}, valu insts/launch {a['SQ_INSTS_VALU'] / n:.3e}, lds active {a['SQ_ACTIVE_INST_LDS'] / wc:.3f}, "
                     f"lds issue-stall {a['SQ_WAIT_INST_LDS'] / wc:.3f}, bank conflict / idx active {a['SQ_LDS_BANK_CONFLICT'] / max(a['SQ_LDS_IDX_ACTIVE'], 1):.4f}, "
                     f"idx active / cycles {a['SQ_LDS_IDX_ACTIVE'] / (32 * g):.3f}")
        elif which == "icache":
            req = max(a["SQC_ICACHE_REQ"], 1.0)
            line += (f", icache req/launch {a['SQC_ICACHE_REQ'] / n:.3e}, hit {a['SQC_ICACHE_HITS'] / req:.4f}, miss {a['SQC_ICACHE_MISSES'] / req:.4f}, dup-miss "
                     f"{a['SQC_ICACHE_MISSES_DUPLICATE'] / req:.4f}, ifetch/launch {a['SQ_IFETCH'] / n:.3e}, ifetch level / wave-cycles {a['SQ_IFETCH_LEVEL'] / a['SQ_WAVE_CYCLES']:.3f}")
        elif which in ("l2", "vmem"):
            line += ", " + ", ".join(f"{c} {a.get(c, 0.0) / n:.4e}" for c in SETS[which] if c != "GRBM_GUI_ACTIVE")
            if which == "l2" and a.get("TCC_REQ_sum"):
                line += f", L2 hit rate {a.get('TCC_HIT_sum', 0.0) / max(a.get('TCC_HIT_sum', 0.0) + a.get('TCC_MISS_sum', 0.0), 1.0):.3f}"
        elif which == "mem":
            line += f", fetch {a['FETCH_SIZE'] * 2048 / n / 1e6:.1f} MB/launch (x2 corrected) = {a['FETCH_SIZE'] * 2048 / n / us / 1e6:.2f} TB/s"
        elif which == "memw":
            line += f", write {a['WRITE_SIZE'] * 1024 / n / 1e6:.1f} MB/launch = {a['WRITE_SIZE'] * 1024 / n / us / 1e6:.2f} TB/s"
        print(line, flush=True)
