# A/B of two builds of libjetship.so on ANY tool inside ONE gpurun call:  bash tools/ab_cmd.sh build/libjetship.so "python tools/bench_mixed_rows.py 256 256" [grep pattern]
OLD=$1; CMD=$2; PAT=${3:-.}
for i in 1 2; do
for which in old new; do
if [ $which = old ]; then export JETSHIP_LIB=$PWD/$OLD; else unset JETSHIP_LIB; fi
$CMD 2>/dev/null | grep "$PAT" | sed "s/^/$which: /"
done
done
