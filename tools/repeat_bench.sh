mkdir -p gpurun_out
for i in 1 2 3 4; do
for cfg in "fwd_order=1,fwd_group=2,fwd_unroll=1,fwd_wg=512" "fwd_order=8,fwd_group=2,fwd_unroll=1,fwd_wg=512" "fwd_order=32,fwd_group=2,fwd_unroll=1,fwd_wg=512" "fwd_order=4,fwd_group=8,fwd_unroll=4,fwd_wg=256" "fwd_order=16,fwd_group=8,fwd_unroll=4,fwd_wg=256" "fwd_order=2,fwd_group=16,fwd_unroll=8,fwd_wg=1024" "fwd_order=8,fwd_group=16,fwd_unroll=8,fwd_wg=1024"; do
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --tune $cfg 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$cfg', round(j['kernels']['forward']['ms'],2), round(j['kernels']['adjoint']['ms'],2), round(j['value'],2))"
done
done
