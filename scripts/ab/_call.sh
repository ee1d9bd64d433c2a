cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
T0=$(date +%s); timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3; echo "suite wall $(( $(date +%s) - T0 )) s"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash scripts/profile_round.sh r4 2>&1 | tail -3
timeout 600 python scripts/ab/timeline_gaps.py > gpurun_out/profiles_r4/r4_step_timeline.txt 2>&1; head -3 gpurun_out/profiles_r4/r4_step_timeline.txt
T0=$(date +%s); python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r4_bench_stderr.log | tail -1 > gpurun_out/profiles_r4/r4_bench_line.json; echo "bench wall $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d=json.load(open("gpurun_out/profiles_r4/r4_bench_line.json"))
print("headline", d["ms_per_step"], d["value"], d["windows_ms_per_step"])
for k in ("dense128","wgrad_bf16x3","k1","k5","natural_k","rays4096","cfg3","strong"):
    if k in d: print(k, d[k]["ms_per_step"], d[k]["value"], d[k].get("us_per_ray"))
print("cfg4_image", d["cfg4_image"]["s_per_image"], d["cfg4_image"]["value"])
print("roofline", {k:v for k,v in d["roofline"].items() if k in ("kernel","achieved","frac","launch_ms","traffic","peak")})
print("entry points", d["roofline"]["entry_points"])
print("whole step frac", d["frac_bf16x3_mfma_roofline_whole_step"], "step_tflops", d["step_tflops"], "hbm", d.get("step_hbm_bytes"))
print("clocks", d.get("clocks_ghz"))
print("cpu", d["cpu_baseline"]["value"], "eager", d["eager_rocm_baseline"].get("value"))
PY
} 2>&1 | tee gpurun_out/r4_call34.log
