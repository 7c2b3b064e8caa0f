# kernel trace + PMC (FETCH_SIZE, WRITE_SIZE in their own passes) of the batched dense kernels (tools/prof_dense.py); summary -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python3 tools/prof_dense.py 512"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pdn_kt -- $CMD > gpurun_out/pdn_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pdn_fetch -- $CMD > gpurun_out/pdn_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pdn_write -- $CMD > gpurun_out/pdn_write.log 2>&1
python3 - <<'PY' > gpurun_out/pdn_summary.md
import csv, glob, collections, re
print("# rocprofv3 of `python3 tools/prof_dense.py 512`: batched GEMV kernels on a tall operator of 1024 dense 512 x 512 Float32 children (1 GiB), 20 calls each way")
print()
print([ln for ln in open("gpurun_out/pdn_kt.log").read().splitlines() if "dense children" in ln][-1])
print()
print("| kernel | calls | avg ms | HBM traffic per launch (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes) |")
print("|---|---|---|---|")
def key(full):
    m = re.search(r"(k_\w+(?:<[^>]*>)?)", full)
    return m.group(1) if m else full[:80]
def counters(d, name):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                agg[key(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}
fe, wr = counters("gpurun_out/pdn_fetch", "FETCH_SIZE"), counters("gpurun_out/pdn_write", "WRITE_SIZE")
f = glob.glob("gpurun_out/pdn_kt/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    nm = key(r["Name"])
    if any(k in nm for k in ("k_gemv", "k_fold", "k_sum_chunks")):
        t = 2 * 1024 * fe.get(nm, 0) + 1024 * wr.get(nm, 0)
        print(f"| `{nm}` | {r['Calls']} | {float(r['AverageNs']) / 1e6:.4f} | {t / 1e6:.1f} MB |")
PY
find gpurun_out/pdn_kt gpurun_out/pdn_fetch gpurun_out/pdn_write -type f -size +2M -delete
cat gpurun_out/pdn_summary.md
