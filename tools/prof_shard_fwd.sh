# round 4 (late): kernel trace + PMC (FETCH_SIZE, WRITE_SIZE in their own passes) of the tall forward at a rank's row count (NROW, default 128) with the walk pinned to
# the row-concurrent candidate 7 and to the column-band candidate WALK (default 8): durations and HBM traffic per launch.  Output: gpurun_out/shard_fwd_*.txt
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
NROW=${NROW:-128}
for W in 7 ${WALK:-8}; do
  CMD="python3 bench.py --nblocks $NROW --steps 5 --warmup 2 --no-cpu-baseline --placement none --fwd-walk $W"
  echo "walk $W: kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sf_kt_$W -- $CMD > gpurun_out/sf_kt_$W.log 2>&1
  echo "walk $W: fetch"; rocprofv3 --kernel-include-regex 'k_tall_diag_fwd' --pmc FETCH_SIZE --output-format csv -d gpurun_out/sf_fetch_$W -- $CMD > gpurun_out/sf_fetch_$W.log 2>&1
  echo "walk $W: write"; rocprofv3 --kernel-include-regex 'k_tall_diag_fwd' --pmc WRITE_SIZE --output-format csv -d gpurun_out/sf_write_$W -- $CMD > gpurun_out/sf_write_$W.log 2>&1
done
python3 - <<'PY'
import csv, glob, os
nrow = int(os.environ.get("NROW", "128"))
alg = (2 * nrow + 1) * 256 ** 3 * 4
out = open("gpurun_out/shard_fwd_summary.txt", "w")
def P(*a):
    print(*a); print(*a, file=out)
P(f"# rocprofv3 of `python3 bench.py --nblocks {nrow} --steps 5 --warmup 2 --no-cpu-baseline --placement none --fwd-walk W`: the tall forward at a rank's row count;")
P(f"# algorithmic bytes per launch {alg}; traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB -> bytes per launch (gfx950 corrections of MI355X_MICROARCH.md), separate passes")
for W in ("7", os.environ.get("WALK", "8")):
    ks = glob.glob(f"gpurun_out/sf_kt_{W}/**/*_kernel_stats.csv", recursive=True)
    avg = calls = None
    for r in csv.DictReader(open(ks[0])):
        if "k_tall_diag_fwd" in r["Name"]:
            avg, calls, name = float(r["AverageNs"]) / 1e6, int(r["Calls"]), r["Name"][:70]
    def counter(kind, cname):
        fs = glob.glob(f"gpurun_out/sf_{kind}_{W}/**/*_counter_collection.csv", recursive=True)
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == cname and "k_tall_diag_fwd" in r["Kernel_Name"]]
        return sum(vals) / len(vals)
    fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
    traffic = (2 * fetch + write) * 1024
    P(f"walk {W}: {calls} launches, avg {avg:.3f} ms = {alg / avg / 1e6:.0f} GB/s over the algorithmic bytes; HBM traffic {traffic / 1e6:.1f} MB = x{traffic / alg:.3f}  [{name}]")
PY
find gpurun_out/sf_* -type f -size +2M -delete
