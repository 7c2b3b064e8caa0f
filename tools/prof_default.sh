cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_default_kt -- $CMD > gpurun_out/prof_default_kt.log 2>&1
grep "^{" gpurun_out/prof_default_kt.log | tail -1 > gpurun_out/prof_default_bench.json
cp $(find gpurun_out/prof_default_kt -name "*_kernel_stats.csv" | head -1) gpurun_out/prof_default_kernel_stats.csv
find gpurun_out/prof_default_kt -type f -size +2M -delete
python3 - <<'PY'
import csv, json
j = json.load(open("gpurun_out/prof_default_bench.json"))
print("bench (HIP events, same process):", {k: round(v["ms"], 3) for k, v in j["kernels"].items()}, "walk:", j["config"]["fwd_grid_walk"], "pairs/s", round(j["value"], 2))
for r in csv.DictReader(open("gpurun_out/prof_default_kernel_stats.csv")):
    if "k_tall_diag" in r["Name"]:
        print(f'{r["Calls"]:>4s} calls  avg {float(r["AverageNs"]) / 1e6:8.3f} ms  {r["Percentage"]:>6s} %  {r["Name"][:150]}')
PY
