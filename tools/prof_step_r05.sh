# round 5: kernel trace + PMC (FETCH_SIZE, WRITE_SIZE in their own passes) of the ONE-PASS LSQR STEP, one instantiation at a time: the LSQR loop of
# bench.py with the step pinned to the plain walk (step_chain=0) and to the chained row chunks (step_chain=1).  Summaries ->
# profiles/rocprof_r05_step_{plain,chain}_summary.md, traffic -> profiles/traffic_latest.json (keys k_tall_diag_bidiag[_chain], round r05); copies -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for WHICH in plain chain; do
  if [ $WHICH = plain ]; then T="step_chain=0"; else T="step_chain=1"; fi
  CMD="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --placement none --lsqr 10 --tune $T"
  echo "[$WHICH] kernel trace"; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps_${WHICH}_kt -- $CMD > gpurun_out/ps_${WHICH}_kt.log 2>&1 &&
  echo "[$WHICH] fetch" && rocprofv3 --kernel-include-regex 'k_tall_diag_bidiag' --pmc FETCH_SIZE --output-format csv -d gpurun_out/ps_${WHICH}_fetch -- $CMD > gpurun_out/ps_${WHICH}_fetch.log 2>&1 &&
  echo "[$WHICH] write" && rocprofv3 --kernel-include-regex 'k_tall_diag_bidiag' --pmc WRITE_SIZE --output-format csv -d gpurun_out/ps_${WHICH}_write -- $CMD > gpurun_out/ps_${WHICH}_write.log 2>&1
  python3 tools/prof_summary.py --round r05 --tag _step_${WHICH} --kt gpurun_out/ps_${WHICH}_kt --fetch gpurun_out/ps_${WHICH}_fetch --write gpurun_out/ps_${WHICH}_write --merge --adj-launches 2 --cmd "$CMD" > gpurun_out/ps_${WHICH}_summary.txt 2>&1
  cp profiles/rocprof_r05_step_${WHICH}_summary.md profiles/rocprof_r05_step_${WHICH}_kernel_stats.csv profiles/traffic_latest.json gpurun_out/
  grep "^{" gpurun_out/ps_${WHICH}_kt.log | tail -1 > gpurun_out/ps_${WHICH}_bench.json
  find gpurun_out/ps_${WHICH}_kt gpurun_out/ps_${WHICH}_fetch gpurun_out/ps_${WHICH}_write -type f -size +2M -delete
  head -12 gpurun_out/ps_${WHICH}_summary.txt
done
