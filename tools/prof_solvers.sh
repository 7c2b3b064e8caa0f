# kernel trace of the three solver loops behind the ABI (LSQR one pass / CGLS two passes / CG through the fused A'A) in ONE process; summary -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_solvers_kt
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --lsqr 12 --cgls 12 --cgnr 12"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_solvers_kt -- $CMD > gpurun_out/prof_solvers_kt.log 2>&1
grep "^{" gpurun_out/prof_solvers_kt.log | tail -1 > gpurun_out/prof_solvers_bench.json
cp $(find gpurun_out/prof_solvers_kt -name "*_kernel_stats.csv" | head -1) gpurun_out/prof_solvers_kernel_stats.csv
find gpurun_out/prof_solvers_kt -type f -size +2M -delete
python3 - <<'PY'
import csv, json
j = json.load(open("gpurun_out/prof_solvers_bench.json"))
print("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --lsqr 12 --cgls 12 --cgnr 12   (1024 x 256^3 Float32, one MI355X)")
for k in ("lsqr", "cgls", "cgnr"):
    print(f"# bench line, {k}: {j[k]['ms_per_iteration']:.2f} ms per iteration, {j[k]['GBps']:.0f} GB/s of its algorithmic bytes, rel. error vs x_true {j[k]['rel_err_vs_x_true']:.1e}")
n = 256 ** 3 * 4
N = 1024
algo = {"k_tall_diag_bidiag": (3 * N + 2) * n, "k_tall_diag_adj<float, 1, 4, 4, 4, true, 1": (N + 2) * n, "k_tall_diag_adj<float, 1, 4, 4, 4, true, 0": (2 * N + 1) * n / 2,
        "k_tall_diag_fwd": (2 * N + 1) * n}
print("| calls | avg ms | % | algorithmic GB/s | kernel |")
print("|---|---|---|---|---|")
for r in csv.DictReader(open("gpurun_out/prof_solvers_kernel_stats.csv")):
    if float(r["Percentage"]) < 0.3:
        continue
    avg = float(r["AverageNs"]) / 1e6
    gb = ""
    for key, b in algo.items():
        if key in r["Name"]:
            gb = f"{b / avg / 1e6:.0f}"
    print(f'| {r["Calls"]} | {avg:.3f} | {r["Percentage"]} | {gb} | `{r["Name"][:120]}` |')
PY
