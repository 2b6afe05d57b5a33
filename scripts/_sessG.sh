mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 600 python scripts/upload_stress.py --iters 30 ) > gpurun_out/r05_upload_stress.json 2> gpurun_out/r05g_stress.err; tail -2 gpurun_out/r05g_stress.err; cat gpurun_out/r05_upload_stress.json
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time timeout 1500 python -m pytest tests -q -x -m gpu -p no:cacheprovider --timeout=900 --durations=5 ) > gpurun_out/r05_gpu_tests.log 2>&1; tail -8 gpurun_out/r05_gpu_tests.log
TAG=r05 STEPS="bench prof" BENCH_STEPS=20 bash scripts/gpu_round.sh 2>&1 | tail -30 | cut -c1-500
