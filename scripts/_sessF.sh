mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
V=mvs-texturing_amd/csrc/variants
( timeout 400 python scripts/sweep_probe.py --config 3 --sweeps 20 --rounds 2 base=mvs-texturing_amd/csrc/libmvs_viewsel.so exp9=$V/libmvs_viewsel_exp9.so exp10=$V/libmvs_viewsel_exp10.so exp11=$V/libmvs_viewsel_exp11.so ) > gpurun_out/r05_sweep_probe2.json 2> gpurun_out/r05f_probe.err; tail -2 gpurun_out/r05f_probe.err; cat gpurun_out/r05_sweep_probe2.json
( time timeout 1500 python -m pytest tests -q -x -m gpu -p no:cacheprovider --timeout=900 --durations=5 ) > gpurun_out/r05_gpu_tests.log 2>&1; tail -10 gpurun_out/r05_gpu_tests.log
TAG=r05 STEPS="bench prof pmc" BENCH_STEPS=20 bash scripts/gpu_round.sh 2>&1 | tail -42 | cut -c1-700
