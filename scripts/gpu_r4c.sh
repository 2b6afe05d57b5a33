#!/bin/bash
mkdir -p gpurun_out/r4c
export PYTHONFAULTHANDLER=1 AMD_LOG_LEVEL=1
timeout 500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "config5_one or word_walk or real_like" > gpurun_out/r4c/t1.log 2>&1
echo "t1 rc=$?"; tail -3 gpurun_out/r4c/t1.log
unset AMD_LOG_LEVEL
timeout 200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "real_like" > gpurun_out/r4c/t2.log 2>&1
echo "t2 rc=$?"; tail -3 gpurun_out/r4c/t2.log
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --maxfail=6 -k "undistortion or region or two_ranks or config2 or config3 or degenerate or empty or row_f or bench_contract or soup" > gpurun_out/r4c/t3.log 2>&1
echo "t3 rc=$?"; tail -5 gpurun_out/r4c/t3.log
