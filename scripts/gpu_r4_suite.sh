#!/bin/bash
mkdir -p gpurun_out/r4s
export PYTHONFAULTHANDLER=1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --maxfail=5 --durations=5 > gpurun_out/r4s/test.log 2> gpurun_out/r4s/test.err
echo "suite rc=$?"; tail -10 gpurun_out/r4s/test.log
timeout 420 python bench.py --steps 10 --warmup 3 > gpurun_out/r4s/bench_c3.json 2> gpurun_out/r4s/bench_c3.err
echo "bench rc=$?"; tail -2 gpurun_out/r4s/bench_c3.err
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r4s/bench_c3.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "parity_checked", "dropin_ms", "h2d_ms", "h2d_first_ms", "h2d_GBps", "pcie_inclusive_value")}, d["roofline"]["frac"], d["shuffled"]["ms_per_step"], d["real_like"]["ms_per_step"], d["dropin"].get("dropin_core_ms"))
except Exception as e: print("unreadable:", e)
P
