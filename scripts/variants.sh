#!/bin/bash
# A/B of bench.py under environment variants on one box: bash scripts/variants.sh "A=1;B=2 C=3;..."  (first entry may be empty = baseline)
IFS=';' read -ra VARS <<< "$1"
for v in "${VARS[@]}"; do
  echo "== variant: ${v:-baseline}"
  env $v timeout 600 python bench.py --config ${BENCH_CONFIG:-3} --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-traffic --no-parity --no-real-like 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['stages']
print(round(d['ms_per_step'],2), 'sweep', round(s['mrf_sweep']['ms_per_step'],2), 'sweep_ms', round(d['roofline']['sweep_ms'],4), 'rays', round(s['dc_rays']['ms_per_step'],2), 'prep', round(s['dc_prep']['ms_per_step'],3), 'info', round(s['dc_face_info']['ms_per_step'],2), 'cull', round(s['dc_cull']['ms_per_step'],2), 'setup', round(s['mrf_setup']['ms_per_step'],2))"
done
