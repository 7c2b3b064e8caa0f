# Round-2 profiles of the default bench command and of the LSQR loop: rocprofv3 kernel trace (+ --stats) and, in passes of their
# own, the FETCH_SIZE / WRITE_SIZE counters.  The forward's grid walk is pinned for the counter passes (the lazy autotune would
# mix candidate shapes into the per-kernel averages): --tune fwd_wg=512,fwd_unroll=4,fwd_group=8,fwd_order=1 = candidate 4, the
# walk the un-profiled runs of this round settled on.  Summaries -> profiles/ (tools/prof_summary.py), raw traces are dropped.
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export ROUND=r02
CMD="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p2_kt -- $CMD > gpurun_out/p2_kt.log 2>&1
grep "^{" gpurun_out/p2_kt.log | tail -1 > gpurun_out/p2_bench.json
python3 tools/prof_summary.py --round r02 --tag _default --kt gpurun_out/p2_kt --adj-launches 2 --merge --cmd "$CMD" > gpurun_out/p2_default_summary.txt 2>&1
PIN="--tune fwd_wg=512,fwd_unroll=4,fwd_group=8,fwd_order=1"
CMD2="python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline $PIN"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p2_pin_kt -- $CMD2 > gpurun_out/p2_pin_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p2_fetch -- $CMD2 > gpurun_out/p2_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p2_write -- $CMD2 > gpurun_out/p2_write.log 2>&1
python3 tools/prof_summary.py --round r02 --tag _default_pmc --kt gpurun_out/p2_pin_kt --fetch gpurun_out/p2_fetch --write gpurun_out/p2_write --walk 1 --merge --adj-launches 2 --cmd "$CMD2" > gpurun_out/p2_pmc_summary.txt 2>&1
CMD3="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --lsqr 10"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/p2_lsqr_kt -- $CMD3 > gpurun_out/p2_lsqr_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/p2_lsqr_fetch -- $CMD3 > gpurun_out/p2_lsqr_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/p2_lsqr_write -- $CMD3 > gpurun_out/p2_lsqr_write.log 2>&1
python3 tools/prof_summary.py --round r02 --tag _lsqr --kt gpurun_out/p2_lsqr_kt --fetch gpurun_out/p2_lsqr_fetch --write gpurun_out/p2_lsqr_write --adj-launches 2 --merge --cmd "$CMD3" > gpurun_out/p2_lsqr_summary.txt 2>&1
mkdir -p gpurun_out/profiles_r02 && cp profiles/rocprof_r02_* profiles/traffic_latest.json gpurun_out/profiles_r02/
find gpurun_out/p2_kt gpurun_out/p2_pin_kt gpurun_out/p2_fetch gpurun_out/p2_write gpurun_out/p2_lsqr_kt gpurun_out/p2_lsqr_fetch gpurun_out/p2_lsqr_write -type f -size +1M -delete
head -12 gpurun_out/p2_default_summary.txt; head -14 gpurun_out/p2_pmc_summary.txt; head -16 gpurun_out/p2_lsqr_summary.txt
