# round 4: kernel trace + PMC (FETCH_SIZE, WRITE_SIZE in their own passes) of ONE instantiation of the tall forward: the bench with the
# grid walk pinned to candidate WALK (default 7, the one the headline runs), no placement probe, 3 timed steps -- so that the counters
# are an average over launches of the kernel the roofline names and of nothing else.  Summary -> profiles/rocprof_r04_walk<WALK>_pmc_summary.md,
# traffic -> profiles/traffic_latest.json key k_tall_diag_fwd@walk<WALK> (round $ROUND, default r04); copies -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
WALK=${WALK:-7}
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --placement none --fwd-walk $WALK"
echo "kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pw_kt -- $CMD > gpurun_out/pw_kt.log 2>&1
echo "fetch"; rocprofv3 --kernel-include-regex 'k_tall_diag' --pmc FETCH_SIZE --output-format csv -d gpurun_out/pw_fetch -- $CMD > gpurun_out/pw_fetch.log 2>&1
echo "write"; rocprofv3 --kernel-include-regex 'k_tall_diag' --pmc WRITE_SIZE --output-format csv -d gpurun_out/pw_write -- $CMD > gpurun_out/pw_write.log 2>&1
python3 tools/prof_summary.py --round ${ROUND:-r04} --tag _walk${WALK}_pmc --walk $WALK --kt gpurun_out/pw_kt --fetch gpurun_out/pw_fetch --write gpurun_out/pw_write --merge --adj-launches 2 --cmd "$CMD" > gpurun_out/pw_summary.txt 2>&1
cp profiles/rocprof_${ROUND:-r04}_walk${WALK}_pmc_summary.md profiles/rocprof_${ROUND:-r04}_walk${WALK}_pmc_kernel_stats.csv profiles/traffic_latest.json gpurun_out/
grep "^{" gpurun_out/pw_kt.log | tail -1 > gpurun_out/pw_bench.json
find gpurun_out/pw_kt gpurun_out/pw_fetch gpurun_out/pw_write -type f -size +2M -delete
head -16 gpurun_out/pw_summary.txt
