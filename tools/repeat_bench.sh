# one-shot-like shapes of the tall forward (G = 1 row per workgroup) vs the defaults
mkdir -p gpurun_out
for i in 1 2; do
for cfg in "autotune=1" "autotune=0,fwd_order=0,fwd_group=1,fwd_unroll=2,fwd_wg=256" "autotune=0,fwd_order=0,fwd_group=1,fwd_unroll=4,fwd_wg=256" "autotune=0,fwd_order=0,fwd_group=1,fwd_unroll=1,fwd_wg=512" "autotune=0,fwd_order=0,fwd_group=1,fwd_unroll=2,fwd_wg=512" "autotune=0,fwd_order=0,fwd_group=1,fwd_unroll=4,fwd_wg=1024" "autotune=0,fwd_order=0,fwd_group=2,fwd_unroll=2,fwd_wg=256"; do
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --tune $cfg 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('$cfg', round(j['kernels']['forward']['ms'],2), round(j['kernels']['adjoint']['ms'],2), round(j['value'],2))"
done
done
