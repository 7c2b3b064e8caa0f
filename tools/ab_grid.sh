# A/B of two builds of libjetship.so on the grid benchmark inside ONE gpurun call:  bash tools/ab_grid.sh build/libjetship.so "32 32 128"
OLD=$1; ARGS=$2
for i in 1 2; do
for which in old new; do
if [ $which = old ]; then export JETSHIP_LIB=$PWD/$OLD; else unset JETSHIP_LIB; fi
echo "$which: $(python tools/bench_grid.py $ARGS 2>/dev/null | grep 'xcd=1')"
done
done
