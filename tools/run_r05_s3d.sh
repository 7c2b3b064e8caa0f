# round 5, session 3: odd block lengths, before / after
cd $GRAFT_REPO_ROOT
( timeout -k 10 300 python tools/bench_unaligned.py 1024 101 && timeout -k 10 300 python tools/bench_unaligned.py 256 255 && SEPARATE=1 timeout -k 10 300 python tools/bench_unaligned.py 256 255 && timeout -k 10 300 python tools/bench_unaligned.py 512 127 float64 && timeout -k 10 300 python tools/bench_unaligned.py 512 127 complex64 ) > gpurun_out/bench_unaligned.txt 2>&1; echo "rc $?"; cat gpurun_out/bench_unaligned.txt
