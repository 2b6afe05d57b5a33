#!/bin/bash
mkdir -p gpurun_out/r4e
timeout 420 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --maxfail=8 -k "labels or mrf or stress or golden or one_shot or sweep_loop or region or sharded_path_equals or two_ranks or logical or texrecon or config5_shape or real_like" > gpurun_out/r4e/t1.log 2>&1
echo "t1 rc=$?"; tail -6 gpurun_out/r4e/t1.log
for cfg in 3 2; do for k in 4 5; do
  MVS_MRF_KERNEL=$k timeout 200 python bench.py --config $cfg --steps 6 --warmup 2 --no-cpu-baseline --no-dropin --no-real-like --no-shuffled --no-traffic --no-parity > gpurun_out/r4e/ab_c${cfg}_k$k.json 2> gpurun_out/r4e/ab_c${cfg}_k$k.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4e/ab_c${cfg}_k$k.json').read().strip().splitlines()[-1])
print('cfg ${cfg} kernel $k: ms_per_step %.3f sweeps %d sweep_ms %.4f mrf_sweep %.3f setup %.3f'%(d['ms_per_step'], d['config']['sweeps'], d['roofline']['sweep_ms'], d['stages']['mrf_sweep']['ms_per_step'], d['stages']['mrf_setup']['ms_per_step']))
PY
done; done
timeout 500 python bench.py --steps 10 --warmup 3 > gpurun_out/r4e/bench_full.json 2> gpurun_out/r4e/bench_full.err
echo "bench rc=$?"; tail -2 gpurun_out/r4e/bench_full.err
