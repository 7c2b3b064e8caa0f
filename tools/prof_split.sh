# kernel trace + PMC (FETCH_SIZE, WRITE_SIZE in their own passes) of the split-row walk (tools/prof_split.py); summary -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python3 tools/prof_split.py 4096"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps_kt -- $CMD > gpurun_out/ps_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/ps_fetch -- $CMD > gpurun_out/ps_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/ps_write -- $CMD > gpurun_out/ps_write.log 2>&1
python3 - <<'PY' > gpurun_out/ps_summary.md
import csv, glob, collections
print("# rocprofv3 of `python3 tools/prof_split.py 4096`: the split-row walk on 65536 x 4096 Float32 (1 GiB), 20 calls each")
print()
print([ln for ln in open("gpurun_out/ps_kt.log").read().splitlines() if " Float32, parts " in ln][-1])
print()
print("| kernel | calls | avg ms | HBM traffic per launch (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes) |")
print("|---|---|---|---|")
import re
def key(full):                      # "void (anonymous namespace)::k_x<float, 1, ...>(args)" -> "k_x<float, 1, ...>"
    m = re.search(r"(k_\w+(?:<[^>]*>)?)", full)
    return m.group(1) if m else full[:80]
def counters(d, name):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                agg[key(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fe, wr = counters("gpurun_out/ps_fetch", "FETCH_SIZE"), counters("gpurun_out/ps_write", "WRITE_SIZE")
f = glob.glob("gpurun_out/ps_kt/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    nm = key(r["Name"])
    if any(k in nm for k in ("k_tall_diag", "k_fold_parts", "k_sum_partials")):
        t = 2 * 1024 * fe.get(nm, 0) + 1024 * wr.get(nm, 0)
        print(f"| `{nm}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.4f} | {t / 1e6:.1f} MB |")
PY
find gpurun_out/ps_kt gpurun_out/ps_fetch gpurun_out/ps_write -type f -size +2M -delete
cat gpurun_out/ps_summary.md
