# round 5: a wider look at ONE tool's kernels through the issue / memory-pipeline counters, one rocprofv3 --pmc pass per group (no trace in the same run,
# the program directly after `--`):   TAG=gt REGEX='k_general_tile|k_grid_tile' bash tools/prof_pmc_groups.sh tools/bench_grid.py 32 32 128
# Output: gpurun_out/<TAG>_pmc.md -- per kernel instantiation the average of every counter.
# (Six groups.  A seventh, the TA_* / TD_* counters, never returned on this pool -- the run sat silent for seven minutes and was killed -- and is not collected.)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${TAG:-pmc}
REGEX=${REGEX:-k_}
O=gpurun_out/pg_$TAG
rm -rf ${O}_*
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH SQ_WAIT_ANY" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  echo "[$TAG] group $i: $grp"
  rocprofv3 --kernel-include-regex "$REGEX" --pmc $grp --output-format csv -d ${O}_g$i -- python3 "$@" > ${O}_g$i.log 2>&1 || { echo "group $i failed"; tail -3 ${O}_g$i.log; }
done
python3 - "$TAG" "$REGEX" "$*" <<'PY' > gpurun_out/${TAG}_pmc.md
import collections, csv, glob, os, re, sys
tag, rx, cmd = sys.argv[1], re.compile(sys.argv[2]), sys.argv[3]
def key(full):
    m = re.search(r"(k_\w+(?:<[^()]*>)?)", full)
    return (m.group(1) if m else full[:100]).replace("(anonymous namespace)::", "")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pg_{tag}_g*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if rx.search(r["Kernel_Name"]):
            agg[key(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({n for k in agg for n in agg[k]})
kernels = sorted(agg)
print(f"# counters of `python3 {cmd}` (one --pmc pass per group; averages per launch)\n")
print("| counter | " + " | ".join(f"`{k}`" for k in kernels) + " |")
print("|---|" + "---|" * len(kernels))
for n in names:
    print(f"| {n} | " + " | ".join((f"{sum(agg[k][n]) / len(agg[k][n]):,.0f}" if agg[k][n] else "-") for k in kernels) + " |")
PY
find gpurun_out -path "*pg_${TAG}_g*" -type f -size +2M -delete 2>/dev/null
cat gpurun_out/${TAG}_pmc.md
