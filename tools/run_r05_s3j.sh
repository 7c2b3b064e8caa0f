# round 5, session 3: temporal accesses on rows off the 16-byte grid -- tests + bench
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_tall_unaligned.py tests/test_gpu_mixed_rows.py tests/test_gpu_lsqr.py tests/test_gpu_cgls.py tests/test_gpu_blockop.py -x -q -m gpu --timeout 120 > gpurun_out/pytest_gpu_s3j.txt 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/pytest_gpu_s3j.txt
( timeout -k 10 300 python tools/bench_unaligned.py 1024 101 && timeout -k 10 300 python tools/bench_unaligned.py 256 255 && timeout -k 10 300 python tools/bench_unaligned.py 512 127 float64 ) > gpurun_out/bench_unaligned_j.txt 2>&1; echo "rc $?"; cat gpurun_out/bench_unaligned_j.txt
