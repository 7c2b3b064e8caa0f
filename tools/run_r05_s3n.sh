# round 5, session 3: dense children of odd dimensions -- the adjoint's column kernels on under-aligned packs
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_dense_blocks.py tests/test_gpu_dense_lists.py tests/test_gpu_small_loop.py tests/test_gpu_reference_suite.py tests/test_gpu_random_differential.py -x -q -m gpu --timeout 120 > gpurun_out/pytest_gpu_s3n.txt 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/pytest_gpu_s3n.txt
timeout -k 10 600 python tools/bench_dense_odd.py 512 > gpurun_out/bench_dense_odd_b.txt 2>&1; echo rc $?; cat gpurun_out/bench_dense_odd_b.txt
