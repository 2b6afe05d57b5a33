#!/bin/bash
# one GPU-box session.  Steps are selected by STEPS="smoke tests bench cliffs pmc prof c2 real" (default: all of the
# first line); every step has its own timeout and writes under gpurun_out/ (tag: $TAG, default r05).
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=${TAG:-r06}
STEPS=${STEPS:-"smoke tests bench cliffs pmc prof"}
has() { [[ " $STEPS " == *" $1 "* ]]; }
if has smoke; then echo "== smoke + quick bench (config 2)"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  ( time timeout 600 python bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline ) > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err; tail -2 gpurun_out/${TAG}_bench_c2.err; head -c 1500 gpurun_out/${TAG}_bench_c2.json; echo; fi
if has tests; then echo "== pytest -m gpu ${PYTEST_K:+-k $PYTEST_K}"; ( time timeout ${TEST_TIMEOUT:-1000} python -m pytest tests -q -x -m gpu -p no:cacheprovider ${PYTEST_K:+-k "$PYTEST_K"} ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1; tail -15 gpurun_out/${TAG}_pytest_gpu.log; fi
if has bench; then echo "== bench config 3"; ( time timeout 600 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 ) > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err; tail -4 gpurun_out/${TAG}_bench_c3.err; python - <<P
import json
try:
    d = json.load(open("gpurun_out/${TAG}_bench_c3.json"))
    print({k: d[k] for k in ("value", "ms_per_step", "parity_checked") if k in d}); print("roofline", {k: d["roofline"][k] for k in ("frac", "sweep_ms", "traffic")})
    print("path", {k: d["roofline_path"][k] for k in ("frac", "B_dc", "B_mrf", "N_ray_nodes", "N_ray_tris")})
    for k, v in d["roofline_path"]["stages"].items(): print("  ", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
    print("stages", {k: round(v["ms_per_step"], 3) for k, v in d["stages"].items()})
    r = d.get("real_like", {}); print("real_like", {k: r.get(k) for k in ("ms_per_step", "parity_checked", "footprints_lane_group", "footprints_rewalked", "error")}, r.get("parity")); print("parity", d.get("parity")); print("cpu", d.get("cpu_baseline"))
except Exception as e: print("bench output unreadable:", e)
P
fi
if has cliffs; then echo "== solver cliffs"; ( time timeout 900 python scripts/cliff_time.py --config ${CLIFF_CONFIG:-3} ) > gpurun_out/${TAG}_cliffs_c${CLIFF_CONFIG:-3}.json 2> gpurun_out/${TAG}_cliffs.err; tail -3 gpurun_out/${TAG}_cliffs.err; python - <<P
import json
try:
    for c in json.load(open("gpurun_out/${TAG}_cliffs_c${CLIFF_CONFIG:-3}.json"))["cases"]: print(c["case"], "| mrf_ms", round(c["mrf_ms"], 2), "sweeps", c["sweeps"], {k: round(v, 2) for k, v in c["stages_ms"].items()})
except Exception as e: print("cliffs output unreadable:", e)
P
fi
if has pmc; then echo "== SQ counters (config 3)"; ( time timeout 500 python scripts/pmc_sq.py --config 3 --out gpurun_out/${TAG}_pmc_sq_c3.json ) 2>&1 | tail -14; fi
if has pmcreal; then echo "== SQ counters (real-like)"; ( time timeout 300 python scripts/pmc_sq.py --config real --groups 2 --out gpurun_out/${TAG}_pmc_sq_real.json ) 2>&1 | tail -12; fi
if has prof; then
  echo "== rocprofv3 kernel trace"
  REPO=$PWD; cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_${TAG} -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-parity --no-real-like --no-shuffled --no-dropin > $REPO/gpurun_out/${TAG}_prof_bench.log 2>&1
  cd $REPO; f=$(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f gpurun_out/${TAG}_bench_c3_kernel_stats.csv && head -12 $f | cut -c1-160
  rm -rf gpurun_out/prof_${TAG}
fi
if [ -n "$EXTRA_CMD" ]; then echo "== extra: $EXTRA_CMD"; bash -c "$EXTRA_CMD" 2>&1 | tail -${EXTRA_TAIL:-30}; fi
if has c2; then echo "== bench config 2"; ( time timeout 900 python bench.py --config 2 --steps 5 --warmup 2 ) > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err; tail -2 gpurun_out/${TAG}_bench_c2.err; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_c2.json')); print({k: d[k] for k in ('value','ms_per_step','parity_checked')}, 'frac', d['roofline']['frac'], 'dropin', d.get('dropin',{}).get('dropin_ms')); print({k: round(v['ms_per_step'],3) for k,v in d['stages'].items()})"; fi
if has c5; then echo "== bench config 5 (one rank's share)"; ( time timeout 1200 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-traffic ) > gpurun_out/${TAG}_bench_c5_reduced.json 2> gpurun_out/${TAG}_bench_c5.err; tail -2 gpurun_out/${TAG}_bench_c5.err; python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench_c5_reduced.json')); print({k: d[k] for k in ('value','ms_per_step','parity_checked')}); print({k: round(v['ms_per_step'],3) for k,v in d['stages'].items()})"; fi
