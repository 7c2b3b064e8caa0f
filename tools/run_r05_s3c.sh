# round 5, session 3: the fused solver passes on rows off the pack grid -- new tests + the suites of the kernels touched
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_tall_unaligned.py tests/test_gpu_lsqr.py tests/test_gpu_cgls.py tests/test_gpu_mixed_rows.py tests/test_gpu_step_chain.py tests/test_gpu_split_rows.py -x -q --timeout 120 > gpurun_out/pytest_gpu_s3c.txt 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/pytest_gpu_s3c.txt
