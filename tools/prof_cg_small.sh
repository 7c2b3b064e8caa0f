# kernel trace of the small-operator solver loops (tools/prof_cg_small.py): per-kernel durations and the gaps between consecutive kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
: > gpurun_out/pcg_summary.md
for W in cgnr cgls lsqr; do
echo "profiling $W"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pcg_$W -- python3 tools/prof_cg_small.py $W > gpurun_out/pcg_$W.log 2>&1 &&
W=$W python3 - <<'PY' >> gpurun_out/pcg_summary.md
import csv, glob, os, re, collections
W = os.environ["W"]
f = glob.glob(f"gpurun_out/pcg_{W}/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def key(full):
    m = re.search(r"(k_\w+)", full)
    return m.group(1) if m else full[:40]
# steady state: the last 40 % of the dispatches (graph replays of the second solve)
tail = rows[int(0.6 * len(rows)):]
dur = collections.defaultdict(list)
gaps = []
for a, b in zip(tail, tail[1:]):
    gaps.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for r in tail:
    dur[key(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(f"## {W}: {open(f'gpurun_out/pcg_{W}.log').read().strip().splitlines()[-1]}; steady state = the last {len(tail)} of {len(rows)} dispatches")
print()
print("| kernel | dispatches | median us | mean us |")
print("|---|---|---|---|")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"| `{k}` | {len(v)} | {v2[len(v2)//2] / 1e3:.2f} | {sum(v) / len(v) / 1e3:.2f} |")
g = sorted(gaps)
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
print()
print(f"gap between consecutive kernels: median {g[len(g)//2] / 1e3:.2f} us, mean {sum(g) / len(g) / 1e3:.2f} us, 90th percentile {g[int(0.9 * len(g))] / 1e3:.2f} us; "
      f"kernels {sum(sum(v) for v in dur.values()) / 1e3:.0f} us + gaps {sum(gaps) / 1e3:.0f} us = span {span / 1e3:.0f} us")
print()
PY
done
find gpurun_out -path "*pcg_*" -type f -size +4M -delete
cat gpurun_out/pcg_summary.md
