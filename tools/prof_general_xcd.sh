cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 tools/ab_general_xcd.py > gpurun_out/ab_general_xcd.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_xcd_fetch -- python3 tools/ab_general_xcd.py > gpurun_out/prof_xcd_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_xcd_write -- python3 tools/ab_general_xcd.py > gpurun_out/prof_xcd_write.log 2>&1
python3 - <<'PY'
import csv, glob
def vals(d, name):
    f = glob.glob(f"gpurun_out/{d}/**/*_counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "k_block_fwd_general_vec" in r["Kernel_Name"] and r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]
fe, wr = vals("prof_xcd_fetch", "FETCH_SIZE"), vals("prof_xcd_write", "WRITE_SIZE")
out = open("gpurun_out/ab_general_xcd.txt", "a")
for k, name in ((0, "XCD-aware   "), (3, "tile-fastest")):
    f, w = sum(fe[k:k + 3]) / 3, sum(wr[k:k + 3]) / 3
    line = f"{name}: FETCH_SIZE {f:,.0f} KiB, WRITE_SIZE {w:,.0f} KiB per launch -> HBM traffic (2*FETCH + WRITE)*1024 = {(2 * f + w) * 1024 / 1e9:.2f} GB"
    print(line); out.write(line + "\n")
PY
find gpurun_out/prof_xcd_fetch gpurun_out/prof_xcd_write -type f -size +1M -delete
cat gpurun_out/ab_general_xcd.txt
