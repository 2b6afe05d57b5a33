#!/bin/bash
mkdir -p gpurun_out/r4d
export PYTHONFAULTHANDLER=1 AMD_LOG_LEVEL=1
free -g > gpurun_out/r4d/mem.txt; nproc >> gpurun_out/r4d/mem.txt; ulimit -a >> gpurun_out/r4d/mem.txt
timeout 1000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --maxfail=6 --durations=12 > gpurun_out/r4d/test.log 2> gpurun_out/r4d/test.err
echo "suite rc=$?"; tail -25 gpurun_out/r4d/test.log; tail -c 1500 gpurun_out/r4d/test.err
unset AMD_LOG_LEVEL
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dropin --no-real-like --no-shuffled > gpurun_out/r4d/bench.json 2> gpurun_out/r4d/bench.err
echo "bench rc=$?"
