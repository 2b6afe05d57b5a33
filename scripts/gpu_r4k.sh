#!/bin/bash
mkdir -p gpurun_out/r4k
export PYTHONFAULTHANDLER=1
timeout 1000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --maxfail=6 --durations=6 > gpurun_out/r4k/test.log 2> gpurun_out/r4k/test.err
echo "suite rc=$?"; tail -12 gpurun_out/r4k/test.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r4k/bench_full.json 2> gpurun_out/r4k/bench_full.err
echo "bench rc=$?"; tail -2 gpurun_out/r4k/bench_full.err
