# round 5, last session: the fuzz campaigns on the final tree (general kernels now take odd lengths on under-aligned packs; dense adjoint column kernels too)
cd $GRAFT_REPO_ROOT
for t in "fuzz_differential.py 4000" "fuzz_grid.py 1500" "fuzz_grid_sparse.py 1000" "fuzz_dense.py 1500" "fuzz_dense_sparse.py 800"; do
  timeout -k 10 500 python tools/$t > gpurun_out/fuzz_s3_$(echo $t | cut -d. -f1).txt 2>&1; echo "$t rc $?"; tail -1 gpurun_out/fuzz_s3_$(echo $t | cut -d. -f1).txt
done
