# round 5, session 3: the under-aligned tall route -- its tests, then the second cliff hunt again
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_tall_unaligned.py tests/test_gpu_blockop.py tests/test_gpu_grid_sparse.py -x -q --timeout 120 > gpurun_out/pytest_gpu_s3b.txt 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/pytest_gpu_s3b.txt
timeout -k 10 400 python tools/cliff_hunt2.py 512 > gpurun_out/cliff_hunt2_b.txt 2>&1; echo "cliff2 rc $?"; grep -i "odd\|wide\|unaligned" gpurun_out/cliff_hunt2_b.txt
