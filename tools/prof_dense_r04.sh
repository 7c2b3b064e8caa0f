# round 4: kernel trace + PMC (FETCH_SIZE, WRITE_SIZE in their own passes) of the batched dense kernels on tall operators of
# 1 GiB of K x K Float32 children, K = 512, 256, 128 (tools/prof_dense.py); summary -> gpurun_out/pdn4_summary.md
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
: > gpurun_out/pdn4_summary.md
for K in 512 256 128; do
echo "K=$K kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pdn4_kt_$K -- python3 tools/prof_dense.py $K > gpurun_out/pdn4_kt_$K.log 2>&1 &&
echo "K=$K fetch" &&
rocprofv3 --kernel-include-regex 'k_gemv|k_fold|k_sum' --pmc FETCH_SIZE --output-format csv -d gpurun_out/pdn4_fetch_$K -- python3 tools/prof_dense.py $K > gpurun_out/pdn4_fetch_$K.log 2>&1 &&
echo "K=$K write" &&
rocprofv3 --kernel-include-regex 'k_gemv|k_fold|k_sum' --pmc WRITE_SIZE --output-format csv -d gpurun_out/pdn4_write_$K -- python3 tools/prof_dense.py $K > gpurun_out/pdn4_write_$K.log 2>&1 &&
K=$K python3 - <<'PY' >> gpurun_out/pdn4_summary.md
import csv, glob, collections, re, os
K = os.environ["K"]
print(f"# rocprofv3 of `python3 tools/prof_dense.py {K}`: batched dense kernels, 20 calls each way")
print()
line = [ln for ln in open(f"gpurun_out/pdn4_kt_{K}.log").read().splitlines() if "dense children" in ln][-1]
alg = int(line.split()[-1])
print(line)
print()
print("| kernel | calls | avg ms | algorithmic GB/s over the matrices | HBM traffic per launch (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes) | traffic / algorithmic |")
print("|---|---|---|---|---|---|")
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
fe, wr = counters(f"gpurun_out/pdn4_fetch_{K}", "FETCH_SIZE"), counters(f"gpurun_out/pdn4_write_{K}", "WRITE_SIZE")
f = glob.glob(f"gpurun_out/pdn4_kt_{K}/**/*_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    nm = key(r["Name"])
    if any(k in nm for k in ("k_gemv", "k_fold", "k_sum_chunks")):
        t = 2 * 1024 * fe.get(nm, 0) + 1024 * wr.get(nm, 0)
        ms = float(r['AverageNs']) / 1e6
        big = "k_gemv" in nm
        print(f"| `{nm}` | {r['Calls']} | {ms:.4f} | {alg / ms / 1e6:.0f} | {t / 1e6:.1f} MB | {t / alg:.3f} |" if big else f"| `{nm}` | {r['Calls']} | {ms:.4f} | | {t / 1e6:.1f} MB | |")
print()
PY
done
find gpurun_out -path "*pdn4_*" -type f -size +2M -delete
cat gpurun_out/pdn4_summary.md
