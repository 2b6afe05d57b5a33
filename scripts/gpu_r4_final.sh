#!/bin/bash
# round 4, last GPU session: the evidence files of profiles/r04_* (each step under its own timeout)
export PYTHONFAULTHANDLER=1
TAG=r04 STEPS="prof pmc" bash scripts/gpu_round.sh
echo "== bench config 3 (the default workload, 10 steps)"
timeout 420 python bench.py --steps 10 --warmup 3 > gpurun_out/r04_bench_c3.json 2> gpurun_out/r04_bench_c3.err; echo "bench rc=$?"; tail -2 gpurun_out/r04_bench_c3.err
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r04_bench_c3.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "parity_checked", "dropin_ms")}, d["roofline"]["frac"], d["pre_path"], d["shuffled"]["ms_per_step"], d["real_like"]["ms_per_step"], d["dropin"].get("dropin_core_ms"))
except Exception as e: print("unreadable:", e)
P
TAG=r04 STEPS="c2 pmcreal" bash scripts/gpu_round.sh
echo "== C++ adapter tests"
timeout 240 python -m pytest tests/test_cpp_api.py -q -m gpu -p no:cacheprovider 2>&1 | tail -3
echo "== config 5 in full, 8 and 4 logical ranks, oracle windows"
timeout 480 python scripts/config5_full.py --also 4 --oracle-window 100000 > gpurun_out/r04_config5_full_p8_logical.json 2> gpurun_out/r04_config5_full.err; echo "c5 rc=$?"; tail -3 gpurun_out/r04_config5_full.err
python - <<'P'
import json
try:
    d = json.load(open("gpurun_out/r04_config5_full_p8_logical.json"))
    print({k: v for k, v in d.items() if not isinstance(v, (list, dict))}); print(d.get("oracle_windows"))
except Exception as e: print("unreadable:", e)
P
