# round 5: the chained step's 32-row chunks -- tests, soak beside a busy stream (ranged at a rank's shard size and whole at full size), the configs again
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_step_chain.py tests/test_gpu_lsqr.py tests/test_gpu_cgls.py tests/test_gpu_graphs.py tests/test_gpu_team_hygiene.py tests/test_gpu_fullsize.py -x -q > gpurun_out/pytest_gpu_r05_g.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_gpu_r05_g.txt
{ timeout -k 10 300 python tools/soak_step_chain.py 128 1200 --ranged --beside | tail -2; timeout -k 10 300 python tools/soak_step_chain.py 1024 100 --beside | tail -2; } > gpurun_out/soak_r05_step_chain32.txt 2>&1; cat gpurun_out/soak_r05_step_chain32.txt
