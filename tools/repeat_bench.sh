# repeatability of bench.py across processes on one box (the forward's grid walk is autotuned per operator)
mkdir -p gpurun_out
for i in 1 2 3 4 5; do
for at in 1 0; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --tune autotune=$at 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('autotune=$at', round(j['kernels']['forward']['ms'],2), round(j['kernels']['adjoint']['ms'],2), round(j['value'],2))"
done
done
