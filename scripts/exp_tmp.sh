mkdir -p gpurun_out
echo "== pytest"; timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
S='import sys,json; d=json.loads(sys.stdin.read()); s=d["stages"]; print(round(d["ms_per_step"],2), "sweeps", d["config"]["sweeps"], "energy", d["config"]["energy"], {k: round(v["ms_per_step"],2) for k,v in s.items()})'
echo "== base"; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$S"
