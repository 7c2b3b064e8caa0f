# round 5, fifth GPU call: step structure experiments, the dense A/B (pre-flat build against this one), rocprofv3 + PMC of the one-pass step
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python tools/exp_step_pipe.py > gpurun_out/exp_r05_step_pipe.txt 2>&1; echo "step pipe rc $?"; cat gpurun_out/exp_r05_step_pipe.txt | tail -6
bash tools/ab_cmd.sh build/libjetship_r4.so "python tools/bench_dense_blocks.py 1024" "children" > gpurun_out/ab_r05_dense_tall.txt 2>&1
bash tools/ab_cmd.sh build/libjetship_r4.so "python tools/bench_dense_blocks.py 1024 wide" "children" > gpurun_out/ab_r05_dense_wide.txt 2>&1
cat gpurun_out/ab_r05_dense_tall.txt gpurun_out/ab_r05_dense_wide.txt
bash tools/prof_step_r05.sh > gpurun_out/prof_step_r05.log 2>&1; echo "prof rc $?"; tail -30 gpurun_out/prof_step_r05.log
