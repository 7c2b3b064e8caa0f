cd $GRAFT_REPO_ROOT
export GENERAL_LIST=2,2
TAG=sparse32_four REGEX='k_general_tile' bash tools/prof_any.sh tools/bench_grid_sparse.py 32 32 128 diag > /dev/null 2>&1
export GENERAL_LIST=3,3
TAG=sparse32_line REGEX='k_general_tile' bash tools/prof_any.sh tools/bench_grid_sparse.py 32 32 128 bidiag > /dev/null 2>&1
TAG=sparse16_line REGEX='k_general_tile' bash tools/prof_any.sh tools/bench_grid_sparse.py 16 16 256 diag > /dev/null 2>&1
for t in sparse32_four sparse32_line sparse16_line; do grep "^| \`" gpurun_out/${t}_summary.md; done
