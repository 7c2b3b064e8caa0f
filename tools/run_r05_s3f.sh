# round 5, session 3: regression check of the MIXED tall kernels on ALIGNED operators after the under-aligned helpers went in
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python tools/bench_mixed_rows.py 256 256 > gpurun_out/bench_mixed_rows_s3f.txt 2>&1; echo "rc $?"; cat gpurun_out/bench_mixed_rows_s3f.txt
timeout -k 10 400 python tools/bench_mixed_rows.py 1024 128 > gpurun_out/bench_mixed_rows_s3f_b.txt 2>&1; echo "rc $?"; cat gpurun_out/bench_mixed_rows_s3f_b.txt
