#!/bin/bash
# one GPU-box session: parity tests, bench, rocprof kernel trace.  Outputs under gpurun_out/.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
if [ -z "$SKIP_TESTS" ]; then echo "== pytest -m gpu ${PYTEST_K:+-k $PYTEST_K}"; ( time timeout 1200 python -m pytest tests -q -m gpu ${PYTEST_K:+-k "$PYTEST_K"} ) > gpurun_out/pytest_gpu.log 2>&1; tail -5 gpurun_out/pytest_gpu.log; fi
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
if [ -n "$DO_DIAG" ]; then echo "== diag $DO_DIAG"; timeout 900 python tests/tools/gpu_diag.py $DO_DIAG > gpurun_out/diag.log 2>&1; grep -v "^   " gpurun_out/diag.log | tail -20; fi
echo "== bench config ${BENCH_CONFIG:-3}"; ( time timeout 1200 python bench.py --config ${BENCH_CONFIG:-3} --steps 3 --warmup 1 ) > gpurun_out/bench.log 2> gpurun_out/bench.err; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.log
if [ -n "$DO_MODE0" ]; then echo "== bench ray_mode 0"; MVS_RAY_MODE=0 timeout 1200 python bench.py --config ${BENCH_CONFIG:-3} --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-parity --no-real-like 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stages']['dc_rays'])"; fi
if [ -n "$DO_VARIANTS" ]; then
  IFS=';' read -ra VARS <<< "$DO_VARIANTS"
  for v in "${VARS[@]}"; do echo "== bench variant: $v"; env $v timeout 1200 python bench.py --config ${BENCH_CONFIG:-3} --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-parity --no-real-like 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); s=d['stages']; print(round(d['ms_per_step'],1), 'sweep', round(s['mrf_sweep']['ms_per_step'],1), 'icm', round(s['mrf_icm']['ms_per_step'],1), 'setup', round(s['mrf_setup']['ms_per_step'],1), 'rays', round(s['dc_rays']['ms_per_step'],1))"; done
fi
if [ -n "$DO_PROF" ]; then
  echo "== rocprofv3 kernel trace"
  REPO=$PWD; cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof -o bench -- python $REPO/bench.py --config ${BENCH_CONFIG:-3} --steps 2 --warmup 1 --no-cpu-baseline --no-traffic --no-parity --no-real-like > $REPO/gpurun_out/prof_bench.log 2>&1
  cd $REPO; find gpurun_out/prof -type f | head
fi
if [ -n "$DO_PMC" ]; then
  echo "== rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE separately)"
  REPO=$PWD; cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --pmc $c --output-format csv -d $REPO/gpurun_out/pmc_$c -o pmc -- python $REPO/bench.py --config ${BENCH_CONFIG:-3} --steps 1 --warmup 0 --no-cpu-baseline --no-traffic --no-parity --no-real-like > $REPO/gpurun_out/pmc_$c.log 2>&1
  done
  cd $REPO; find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -type f | head; du -sh gpurun_out
fi
