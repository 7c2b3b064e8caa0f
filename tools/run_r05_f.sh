# round 5, sixth GPU call: the chained step with 32-row chunks as the default -- step / LSQR / CGLS tests, fuzz of the tall operators through every
# step mode, every BASELINE config in one process, the mixed-rows bench, then the ranged (pipelined) form
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_step_chain.py tests/test_gpu_lsqr.py tests/test_gpu_cgls.py tests/test_gpu_graphs.py tests/test_gpu_mixed_rows.py tests/test_gpu_fullsize.py tests/test_gpu_team_hygiene.py tests/test_gpu_known_answers.py -x -q > gpurun_out/pytest_gpu_r05_f.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_gpu_r05_f.txt
timeout -k 10 300 python tools/fuzz_tall.py 6000 70000 > gpurun_out/fuzz_r05_f.txt 2>&1; tail -1 gpurun_out/fuzz_r05_f.txt
timeout -k 10 900 python tools/bench_configs.py > gpurun_out/bench_configs_r05.txt 2>&1; echo "configs rc $?"; cat gpurun_out/bench_configs_r05.txt
timeout -k 10 300 python tools/bench_mixed_rows.py 256 256 > gpurun_out/bench_mixed_rows_r05.txt 2>&1; cat gpurun_out/bench_mixed_rows_r05.txt | tail -8
timeout -k 10 300 python tools/ab_step_ranged.py > gpurun_out/ab_r05_step_ranged.txt 2>&1; tail -6 gpurun_out/ab_r05_step_ranged.txt
