# round 5, second GPU call: the whole -m gpu suite on the new tree, then the step-band / general-tile-shape / wide-JetSum measurements
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r05_b.txt 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/pytest_gpu_r05_b.txt
timeout -k 10 600 python tools/exp_step_band.py > gpurun_out/exp_r05_step_band.txt 2>&1; echo "step band rc $?"; tail -6 gpurun_out/exp_r05_step_band.txt
for shape in "32 32 128" "16 16 256" "64 4 128"; do
  GENERAL_TILE=1,42,8,1,42,8 timeout -k 10 300 python tools/bench_grid_mixed.py $shape >> gpurun_out/exp_r05_general_tile_shapes.txt 2>&1
done
grep -v ALGO gpurun_out/exp_r05_general_tile_shapes.txt
for k in 16 11 3; do
  SUM_GROUPS=16,16 SUM_SCALE=wide timeout -k 10 300 python tools/bench_jetsum.py $k 32 256 >> gpurun_out/bench_jetsum_wide_r05.txt 2>&1
  SUM_GROUPS=16,16 timeout -k 10 300 python tools/bench_jetsum.py $k 32 256 >> gpurun_out/bench_jetsum_wide_r05.txt 2>&1
done
grep -v ALGO gpurun_out/bench_jetsum_wide_r05.txt
