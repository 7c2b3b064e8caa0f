# round 5, final validation of the tree: the whole GPU suite, smoke(), the default bench, and the rocprofv3 kernel trace (+ --stats) of the DEFAULT bench
# command (the program directly after `--`); summaries -> gpurun_out/ (copied into profiles/ by hand)
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 120 > gpurun_out/pytest_gpu_r05_final.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_gpu_r05_final.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/bench_r05_final.json 2> gpurun_out/bench_r05_final.err; echo "bench rc $?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/bench_r05_final.json"))
print("bench:", round(j["value"], 3), "pairs/s", round(j["ms_per_step"], 3), "ms/step; roofline", {k: j["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic")}, "cpu", j.get("cpu_baseline", {}).get("value"), "placement_none", (j.get("placement_none") or {}).get("value"))
PY
ROUND=r05 bash tools/prof_default.sh 2>&1 | tail -8
