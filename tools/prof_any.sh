# round 5: kernel trace + PMC (FETCH_SIZE, WRITE_SIZE, L2 hit/miss -- each in a pass of its own) of ANY tool of this directory that prints
# "ALGO <kernel regex> <bytes>" lines for the kernels it times (tools/bench_grid_mixed.py, tools/bench_jetsum.py, tools/bench_cgnr_sizes.py ...):
#   TAG=gt32 REGEX='k_general_tile' bash tools/prof_any.sh tools/bench_grid_mixed.py 32 32 128
# The program stands directly after `--` (python3 <tool> <args>), counters never share a run with a trace.  Summary -> gpurun_out/<TAG>_summary.md
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${TAG:-any}
REGEX=${REGEX:-k_}
O=gpurun_out/pa_$TAG
rm -rf ${O}_kt ${O}_fetch ${O}_write ${O}_l2
echo "[$TAG] kernel trace: python3 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d ${O}_kt -- python3 "$@" > ${O}_kt.log 2>&1 &&
echo "[$TAG] fetch" &&
rocprofv3 --kernel-include-regex "$REGEX" --pmc FETCH_SIZE --output-format csv -d ${O}_fetch -- python3 "$@" > ${O}_fetch.log 2>&1 &&
echo "[$TAG] write" &&
rocprofv3 --kernel-include-regex "$REGEX" --pmc WRITE_SIZE --output-format csv -d ${O}_write -- python3 "$@" > ${O}_write.log 2>&1 &&
echo "[$TAG] l2" &&
rocprofv3 --kernel-include-regex "$REGEX" --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d ${O}_l2 -- python3 "$@" > ${O}_l2.log 2>&1
python3 tools/prof_any_summary.py --tag $TAG --regex "$REGEX" --cmd "python3 $*" > gpurun_out/${TAG}_summary.md 2> gpurun_out/${TAG}_summary.err
find ${O}_kt ${O}_fetch ${O}_write ${O}_l2 -type f -size +2M -delete 2>/dev/null
cat gpurun_out/${TAG}_summary.md
