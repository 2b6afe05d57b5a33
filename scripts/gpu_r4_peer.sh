#!/bin/bash
mkdir -p gpurun_out/r4p
export PYTHONFAULTHANDLER=1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -x -k "sharded or shard or config4" > gpurun_out/r4p/test.log 2>&1
echo "tests rc=$?"; tail -8 gpurun_out/r4p/test.log
timeout 420 python scripts/transport_time.py --parts 2,4,8 > gpurun_out/r4p/transport.json 2> gpurun_out/r4p/transport.err
echo "transport rc=$?"; tail -12 gpurun_out/r4p/transport.err
