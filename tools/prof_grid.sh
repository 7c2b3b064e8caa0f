# PMC traffic + kernel trace of the grid kernels on an M x K grid (tools/bench_grid.py, default route only): FETCH_SIZE and WRITE_SIZE in passes of their own
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS="${GRID_ARGS:-32 32 128}"
export GRID_ROUTES="${GRID_ROUTES:-1:1}"
rm -rf gpurun_out/pg_kt gpurun_out/pg_fetch gpurun_out/pg_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pg_kt -- python3 tools/bench_grid.py $ARGS > gpurun_out/pg_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pg_fetch -- python3 tools/bench_grid.py $ARGS > gpurun_out/pg_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pg_write -- python3 tools/bench_grid.py $ARGS > gpurun_out/pg_write.log 2>&1
python3 - "$ARGS" <<'PY'
import csv, glob, collections, sys
M, K, edge = (int(v) for v in sys.argv[1].split())
n = edge ** 3 * 4
algo = {"fwd": (M * K + K + 2 * M) * n, "adj": (M * K + M + K) * n}
def agg(d, name):
    a = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and ("k_grid_" in r["Kernel_Name"] or "k_block_" in r["Kernel_Name"]):
                kn = r["Kernel_Name"]
                which = "adj" if ("true>" in kn or "_adj_" in kn) else "fwd"
                a[(kn.split("<")[0].split("(")[0][-24:].strip(), which)].append(float(r["Counter_Value"]))
    return a
f, w = agg("gpurun_out/pg_fetch", "FETCH_SIZE"), agg("gpurun_out/pg_write", "WRITE_SIZE")
print(f"# {M} x {K} grid of {edge}^3 Float32 diagonal blocks; PMC per launch, (2*FETCH_SIZE + WRITE_SIZE) * 1024 B (gfx950 corrections)")
for key in sorted(f):
    ff = sum(f[key]) / len(f[key])
    ww = sum(w.get(key, [0])) / max(1, len(w.get(key, [0])))
    tr = (2 * ff + ww) * 1024
    print(f"{key[0]:24s} {key[1]}: {len(f[key]):3d} launches  FETCH {ff:12.0f} KiB  WRITE {ww:12.0f} KiB  traffic {tr / 1e9:8.3f} GB = {tr / algo[key[1]]:.3f} x the unique bytes ({algo[key[1]] / 1e9:.3f} GB)")
ks = glob.glob("gpurun_out/pg_kt/**/*_kernel_stats.csv", recursive=True)
if ks:
    print("# rocprofv3 --kernel-trace --stats of the same command:")
    for r in csv.DictReader(open(ks[0])):
        if "k_grid_" in r["Name"] or "k_block_" in r["Name"]:
            which = "adj" if ("true>" in r["Name"] or "_adj_" in r["Name"]) else "fwd"
            avg = float(r["AverageNs"]) / 1e6
            print(f'{r["Calls"]:>4s} calls  avg {avg:8.3f} ms  {algo[which] / avg / 1e6:8.1f} GB/s algorithmic  {r["Percentage"]:>6s} %  {r["Name"][:110]}')
PY
find gpurun_out/pg_fetch gpurun_out/pg_write gpurun_out/pg_kt -type f -size +2M -delete
