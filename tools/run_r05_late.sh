# late round 5: regression check of the paths the block-sparse work touched only indirectly (route tests, general kernels' signatures, the block table's DENSE entries)
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/bench_suite.py > gpurun_out/bench_suite_late.txt 2>&1; echo "suite rc $?"
for a in "32 32 128" "16 16 256" "64 4 128"; do timeout -k 10 200 python tools/bench_grid_mixed.py $a 2>&1 | grep "mixed grid"; done > gpurun_out/bench_grid_mixed_late.txt
for a in "32 32 128" "16 16 256" "64 4 256"; do timeout -k 10 200 python tools/bench_grid.py $a 2>&1 | tail -4; done > gpurun_out/bench_grid_late.txt
timeout -k 10 300 python tools/bench_dense_blocks.py > gpurun_out/bench_dense_blocks_late.txt 2>&1; echo "dense rc $?"
timeout -k 10 300 python tools/bench_dense_mixed.py 8 8 384 > gpurun_out/bench_dense_mixed_late.txt 2>&1; timeout -k 10 300 python tools/bench_dense_mixed.py 16 16 384 >> gpurun_out/bench_dense_mixed_late.txt 2>&1
timeout -k 10 300 python tools/bench_mixed_rows.py 256 256 > gpurun_out/bench_mixed_rows_late.txt 2>&1; echo "mixed rows rc $?"
