mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 600 python -m pytest tests -q -x -m gpu -p no:cacheprovider --timeout=280 -k "failing_rank or integration_tu" ) > gpurun_out/r05b_pytest.log 2>&1; tail -8 gpurun_out/r05b_pytest.log
( MVS_BENCH_ONE_GPU=1 timeout 300 python bench.py --gpus 2 --config 2 --steps 2 --warmup 1 ) > gpurun_out/r05b_inproc2.json 2> gpurun_out/r05b_inproc2.err; tail -3 gpurun_out/r05b_inproc2.err; python -c "
import json; d=json.load(open('gpurun_out/r05b_inproc2.json')); print({k:d[k] for k in ('n_gpus','ms_per_step','parity_checked','launch')}, d['halo'], d['parity'])"
V=mvs-texturing_amd/csrc/variants
( timeout 500 python scripts/sweep_probe.py --config 3 --sweeps 20 --rounds 2 base=mvs-texturing_amd/csrc/libmvs_viewsel.so exp5=$V/libmvs_viewsel_exp5.so exp6=$V/libmvs_viewsel_exp6.so exp7=$V/libmvs_viewsel_exp7.so exp8=$V/libmvs_viewsel_exp8.so exp1=$V/libmvs_viewsel_exp1.so exp2=$V/libmvs_viewsel_exp2.so ) > gpurun_out/r05_sweep_probe.json 2> gpurun_out/r05b_probe.err; tail -2 gpurun_out/r05b_probe.err; cat gpurun_out/r05_sweep_probe.json
( timeout 400 python scripts/ab_libs.py --config 3 --rounds 3 --steps 3 base=mvs-texturing_amd/csrc/libmvs_viewsel.so prefetch=$V/libmvs_viewsel_rayprefetch.so ) 2>&1 | tail -4 | cut -c1-900
python scripts/h2d_time.py 2>&1 | tail -3
