# kernel trace + PMC (FETCH_SIZE, WRITE_SIZE in their own passes) of the DEFAULT bench command; summaries -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CMD="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pd_kt -- $CMD > gpurun_out/pd_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pd_fetch -- $CMD > gpurun_out/pd_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pd_write -- $CMD > gpurun_out/pd_write.log 2>&1
python3 tools/prof_summary.py --round ${ROUND:-r02} --tag _default_pmc --kt gpurun_out/pd_kt --fetch gpurun_out/pd_fetch --write gpurun_out/pd_write --merge --adj-launches 2 --cmd "$CMD" > gpurun_out/pd_summary.txt 2>&1
cp profiles/rocprof_${ROUND:-r02}_default_pmc_summary.md profiles/traffic_latest.json gpurun_out/
grep "^{" gpurun_out/pd_kt.log | tail -1 > gpurun_out/pd_bench.json
find gpurun_out/pd_kt gpurun_out/pd_fetch gpurun_out/pd_write -type f -size +2M -delete
head -14 gpurun_out/pd_summary.txt
