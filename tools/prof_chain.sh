# kernel trace + PMC (FETCH_SIZE, WRITE_SIZE in their own passes) of the LSQR loop with the CHAINED one-pass step forced
# (--tune step_chain=1): does handing the ordered sum from chunk to chunk cost HBM traffic?  summaries -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --lsqr 10 --tune step_chain=1"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc_kt -- $CMD > gpurun_out/pc_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pc_fetch -- $CMD > gpurun_out/pc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pc_write -- $CMD > gpurun_out/pc_write.log 2>&1
python3 tools/prof_summary.py --round ${ROUND:-r02} --tag _lsqr_chain --kt gpurun_out/pc_kt --fetch gpurun_out/pc_fetch --write gpurun_out/pc_write --adj-launches 2 --merge --cmd "$CMD" > gpurun_out/pc_summary.txt 2>&1
cp profiles/rocprof_${ROUND:-r02}_lsqr_chain_summary.md profiles/rocprof_${ROUND:-r02}_lsqr_chain_kernel_stats.csv profiles/traffic_latest.json gpurun_out/
grep "^{" gpurun_out/pc_kt.log | tail -1 > gpurun_out/pc_bench.json
find gpurun_out/pc_kt gpurun_out/pc_fetch gpurun_out/pc_write -type f -size +2M -delete
head -14 gpurun_out/pc_summary.txt
