set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --lsqr 10"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_lsqr_kt -- $CMD > gpurun_out/prof_lsqr_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_lsqr_fetch -- $CMD > gpurun_out/prof_lsqr_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_lsqr_write -- $CMD > gpurun_out/prof_lsqr_write.log 2>&1
python3 tools/prof_summary.py --round ${ROUND:-r02} --tag _lsqr --kt gpurun_out/prof_lsqr_kt --fetch gpurun_out/prof_lsqr_fetch --write gpurun_out/prof_lsqr_write --merge --cmd "$CMD" > gpurun_out/prof_lsqr_summary.txt 2>&1
cp profiles/rocprof_${ROUND:-r02}_lsqr_summary.md profiles/rocprof_${ROUND:-r02}_lsqr_kernel_stats.csv profiles/traffic_latest.json gpurun_out/
# keep the merged output small: drop the raw traces
find gpurun_out/prof_lsqr_kt gpurun_out/prof_lsqr_fetch gpurun_out/prof_lsqr_write -type f -size +2M -delete
cat gpurun_out/prof_lsqr_summary.txt | head -30
