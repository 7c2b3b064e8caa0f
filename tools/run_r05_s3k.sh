# round 5, session 3: the whole GPU suite, smoke, default bench on the tree with the under-aligned kernels
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 120 > gpurun_out/pytest_gpu_s3k.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_gpu_s3k.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/bench_s3k.json 2> gpurun_out/bench_s3k.err; echo "bench rc $?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/bench_s3k.json"))
print("bench:", round(j["value"], 3), "pairs/s", round(j["ms_per_step"], 3), "ms/step; roofline", {k: j["roofline"][k] for k in ("kernel", "achieved", "frac", "traffic")}, "cpu", j.get("cpu_baseline", {}).get("value"))
PY
