# round 5, first GPU call: rocprofv3 kernel trace + PMC of k_general_tile (mixed grids) and k_tall_sum_fwd/_adj (JetSum), the two kernels the
# round-4 verdict found lowest and without counter evidence.  bash tools/prof_r05_a.sh  -> gpurun_out/{gt32,gt64x4,gt16,js16,js11}_summary.md
export GENERAL_TILE=1,1 SUM_GROUPS=16,16
TAG=gt32 REGEX='k_general_tile' bash tools/prof_any.sh tools/bench_grid_mixed.py 32 32 128 &&
TAG=gt64x4 REGEX='k_general_tile' bash tools/prof_any.sh tools/bench_grid_mixed.py 64 4 128 &&
TAG=gt16 REGEX='k_general_tile' bash tools/prof_any.sh tools/bench_grid_mixed.py 16 16 256 &&
TAG=js16 REGEX='k_tall_sum' bash tools/prof_any.sh tools/bench_jetsum.py 16 32 256 &&
TAG=js11 REGEX='k_tall_sum' bash tools/prof_any.sh tools/bench_jetsum.py 11 32 256
