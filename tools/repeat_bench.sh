# repeatability of bench.py across processes on one box:  bash tools/repeat_bench.sh
mkdir -p gpurun_out
for i in 1 2 3 4 5 6 7 8 9 10; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print(j['config']['fwd_grid_walk'], round(j['kernels']['forward']['ms'],2), round(j['kernels']['adjoint']['ms'],2), round(j['value'],2))"
done
