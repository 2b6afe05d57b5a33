mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( time timeout 1500 python -m pytest tests -q -x -m gpu -p no:cacheprovider --timeout=900 --durations=8 ) > gpurun_out/r05_gpu_tests.log 2>&1; tail -14 gpurun_out/r05_gpu_tests.log
TAG=r05 STEPS="bench prof" BENCH_STEPS=20 bash scripts/gpu_round.sh 2>&1 | tail -40
