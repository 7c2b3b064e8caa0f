# round 5, fourth GPU call: the suite and the fuzz campaigns on the SPLIT library (jh_blockop.hip by kernel family, pruned adjoint shapes, rewritten
# JetSum kernels, dense kernels without flat accesses), the dense A/B against the pre-flat build, and the default bench
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r05_d.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_gpu_r05_d.txt
{ timeout -k 10 300 python tools/fuzz_tall.py 6000 50000; timeout -k 10 300 python tools/fuzz_grid.py 3000 50000; timeout -k 10 300 python tools/fuzz_dense.py 3000 50000; timeout -k 10 400 python tools/fuzz_differential.py 6000 50000; } > gpurun_out/fuzz_r05_d.txt 2>&1; echo "fuzz rc $?"; cat gpurun_out/fuzz_r05_d.txt | tail -8
bash tools/ab_cmd.sh build/libjetship_r4.so "python tools/bench_dense_blocks.py 1024" "x" > gpurun_out/ab_r05_dense_tall.txt 2>&1
bash tools/ab_cmd.sh build/libjetship_r4.so "python tools/bench_dense_blocks.py 1024 wide" "x" > gpurun_out/ab_r05_dense_wide.txt 2>&1
cat gpurun_out/ab_r05_dense_tall.txt gpurun_out/ab_r05_dense_wide.txt
timeout -k 10 600 python bench.py > gpurun_out/bench_r05_d.json 2> gpurun_out/bench_r05_d.err; echo "bench rc $?"; tail -c 1500 gpurun_out/bench_r05_d.json
