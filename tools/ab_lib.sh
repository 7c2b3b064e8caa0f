# A/B of two builds of libjetship.so inside ONE gpurun call (boxes differ by more than most changes):
#   bash tools/ab_lib.sh build/old_csrc/libjetship_old.so [bench flags]
# alternates old / new processes of bench.py and prints forward ms, adjoint ms, pairs/s (+ LSQR ms/iteration with --lsqr K)
OLD=$1; shift
for i in 1 2 3; do
for which in old new; do
if [ $which = old ]; then export JETSHIP_LIB=$PWD/$OLD; else unset JETSHIP_LIB; fi
python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$which', j['config']['fwd_grid_walk'], 'fwd', round(j['kernels']['forward']['ms'],2), 'adj', round(j['kernels']['adjoint']['ms'],2), 'pairs/s', round(j['value'],2), 'lsqr ms/it', round(j.get('lsqr',{}).get('ms_per_iteration',0),2))"
done
done
