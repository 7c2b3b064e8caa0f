# late round 5: block-sparse operators of dense children, list route (dense_list=1) against round 3's grid route (dense_list=0) and -- for small children --
# against the one-launch loop (SMALL_LOOP_MAX_KIB huge)
cd $GRAFT_REPO_ROOT
export DENSE_LIST=1,0,1,0
for a in "64 1024" "64 1024 1" "16 4096" "8 2048 1" "4 8192" "256 512 1" "128 1024 1"; do timeout -k 10 200 python tools/bench_dense_blockdiag.py $a 2>&1 | grep dense_list; done
echo "# small children: the list route (first line of each pair) against the one-launch loop (second)"
export DENSE_LIST=1
for a in "256 256" "512 128" "64 128" "16 256" "32 64" "8 128 1" "128 32" "512 64"; do
  SMALL_LOOP_MAX_KIB=0 timeout -k 10 200 python tools/bench_dense_blockdiag.py $a 2>&1 | grep dense_list
  SMALL_LOOP_MAX_KIB=1000000000 timeout -k 10 200 python tools/bench_dense_blockdiag.py $a 2>&1 | grep dense_list | sed "s/dense_list=1 rl=[0-9]*/one-launch loop  /"
done
