# round 5, third GPU call: suite on the rewritten sum kernels, A/B of the sum kernels against the previous build, step bands, nt sweep
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r05_c.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_gpu_r05_c.txt
export SUM_GROUPS=16,16
for k in 16 11 8 3; do
  bash tools/ab_cmd.sh build/libjetship_base.so "python tools/bench_jetsum.py $k 32 256" JetSum >> gpurun_out/ab_r05_jetsum.txt 2>&1
done
bash tools/ab_cmd.sh build/libjetship_base.so "python tools/bench_jetsum.py 16 16 256" JetSum >> gpurun_out/ab_r05_jetsum.txt 2>&1
cat gpurun_out/ab_r05_jetsum.txt
timeout -k 10 600 python tools/exp_step_band.py > gpurun_out/exp_r05_step_band.txt 2>&1; echo "step band rc $?"; tail -6 gpurun_out/exp_r05_step_band.txt
timeout -k 10 900 python tools/exp_nt_small.py > gpurun_out/exp_r05_nt_small.txt 2>&1; echo "nt rc $?"; cat gpurun_out/exp_r05_nt_small.txt
