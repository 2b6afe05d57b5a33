# quick GPU check: a few parity tests + one bench line summary.  Environment does not travel through gpurun: pass the knobs inside
# the command string, e.g. gpurun -- 'QB_K="ray or golden" QB_ARGS="--no-real-like" bash scripts/quick_bench.sh'
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -q -m gpu -x -k "${QB_K:-labels_bit_exact or stress or golden or config2_equals or random_instances}") > gpurun_out/t1.log 2>&1; tail -4 gpurun_out/t1.log
(timeout 900 python bench.py --no-traffic --no-cpu-baseline ${QB_ARGS}) > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -2 gpurun_out/bench.err
python - <<EOF
import json
d=json.loads(open("gpurun_out/bench.log").read().strip().splitlines()[-1])
print("ms/step", round(d["ms_per_step"],2), "sweeps", d["config"]["sweeps"], "E", d["config"]["energy"], "parity", d.get("parity_checked"))
print({k:round(v["ms_per_step"],3) for k,v in d["stages"].items()})
print("sweep_ms", d["roofline"]["sweep_ms"], "frac", d["roofline"]["frac"])
EOF
