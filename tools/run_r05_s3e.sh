# round 5, session 3: grids off the pack grid -- new tests + the suites of the general kernels + the cliff hunt + grid regression benches
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_tall_unaligned.py tests/test_gpu_blockop.py tests/test_gpu_grid_sparse.py tests/test_gpu_random_differential.py tests/test_gpu_reference_suite.py tests/test_gpu_nonlinear.py tests/test_gpu_dense_blocks.py tests/test_gpu_small_loop.py -x -q --timeout 120 > gpurun_out/pytest_gpu_s3e.txt 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/pytest_gpu_s3e.txt
timeout -k 10 400 python tools/cliff_hunt2.py 512 > gpurun_out/cliff_hunt2_e.txt 2>&1; echo "cliff2 rc $?"; cat gpurun_out/cliff_hunt2_e.txt
for a in "32 32 128" "16 16 256" "64 4 128"; do timeout -k 10 200 python tools/bench_grid_mixed.py $a 2>&1 | grep "mixed grid"; done > gpurun_out/bench_grid_mixed_s3e.txt; cat gpurun_out/bench_grid_mixed_s3e.txt
