# round 5, session 3: fused JetSum off the pack grid + regression of the aligned JetSum kernels
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_gpu_tall_unaligned.py tests/test_gpu_mixed_rows.py tests/test_scalar_types.py tests/test_gpu_known_answers.py tests/test_gpu_split_rows.py -x -q -m gpu --timeout 120 > gpurun_out/pytest_gpu_s3i.txt 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/pytest_gpu_s3i.txt
for a in "16 32 256" "11 32 256" "8 32 256" "3 32 256"; do timeout -k 10 200 python tools/bench_jetsum.py $a 2>&1 | grep "JetSum"; done > gpurun_out/bench_jetsum_s3i.txt; cat gpurun_out/bench_jetsum_s3i.txt
