# PMC traffic of the general kernels on an M x K grid (tools/bench_grid.py): FETCH_SIZE and WRITE_SIZE in passes of their own
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS="${GRID_ARGS:-32 32 128}"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pg_fetch -- python3 tools/bench_grid.py $ARGS > gpurun_out/pg_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pg_write -- python3 tools/bench_grid.py $ARGS > gpurun_out/pg_write.log 2>&1
python3 - <<'PY'
import csv, glob, collections
def agg(d, name):
    a = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "k_block_" in r["Kernel_Name"]:
                a[r["Kernel_Name"].split("<")[0].split("(")[0][-28:]].append(float(r["Counter_Value"]))
    return a
f, w = agg("gpurun_out/pg_fetch", "FETCH_SIZE"), agg("gpurun_out/pg_write", "WRITE_SIZE")
for k in f:
    # launches come in the order general_xcd = 1, 0, 2 (8 launches each: 2 warm + 6 timed)
    vals_f, vals_w = f[k], w.get(k, [0] * len(f[k]))
    for lab, lo in (("xcd=1", 0), ("xcd=0", 8), ("xcd=2", 16)):
        ff = sum(vals_f[lo:lo + 8]) / 8
        ww = sum(vals_w[lo:lo + 8]) / 8 if len(vals_w) >= lo + 8 else 0
        print(f"{k:28s} {lab}: FETCH {ff:12.0f} KiB  WRITE {ww:12.0f} KiB  traffic (2*FETCH+WRITE)*1024 = {(2 * ff + ww) * 1024 / 1e9:8.3f} GB")
PY
find gpurun_out/pg_fetch gpurun_out/pg_write -type f -size +2M -delete
