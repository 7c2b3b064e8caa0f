# late round 5: rocprofv3 kernel trace + PMC traffic of the step-list walks and of the dense children's list kernels (the program directly after `--`)
cd $GRAFT_REPO_ROOT
export GENERAL_LIST=2,2
TAG=sparse32_four REGEX='k_general_tile' bash tools/prof_any.sh tools/bench_grid_sparse.py 32 32 128 diag > /dev/null 2>&1
export GENERAL_LIST=3,3
TAG=sparse32_line REGEX='k_general_tile' bash tools/prof_any.sh tools/bench_grid_sparse.py 32 32 128 bidiag > /dev/null 2>&1
TAG=sparse16_line REGEX='k_general_tile' bash tools/prof_any.sh tools/bench_grid_sparse.py 16 16 256 diag > /dev/null 2>&1
export DENSE_LIST=1,1
TAG=dense64 REGEX='k_gemv' bash tools/prof_any.sh tools/bench_dense_blockdiag.py 64 1024 > /dev/null 2>&1
TAG=dense256 REGEX='k_gemv' bash tools/prof_any.sh tools/bench_dense_blockdiag.py 256 512 1 > /dev/null 2>&1
TAG=dense16 REGEX='k_gemv' bash tools/prof_any.sh tools/bench_dense_blockdiag.py 16 4096 > /dev/null 2>&1
for t in sparse32_four sparse32_line sparse16_line dense64 dense256 dense16; do cat gpurun_out/${t}_summary.md; echo; done
