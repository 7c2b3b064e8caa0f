# round 5, session 3, first call: the GPU suite on the tree as restored, the second cliff hunt, and baselines of the two kernels furthest below the roofline
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q --timeout 120 > gpurun_out/pytest_gpu_s3a.txt 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/pytest_gpu_s3a.txt
timeout -k 10 400 python tools/cliff_hunt2.py 512 > gpurun_out/cliff_hunt2.txt 2>&1; echo "cliff2 rc $?"; tail -3 gpurun_out/cliff_hunt2.txt
timeout -k 10 200 python tools/cliff_hunt.py 1024 > gpurun_out/cliff_hunt1.txt 2>&1; echo "cliff1 rc $?"
