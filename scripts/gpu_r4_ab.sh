#!/bin/bash
mkdir -p gpurun_out/r4ab
V=mvs-texturing_amd/csrc/variants
timeout 400 python scripts/ab_libs.py --config 3 --rounds 3 --steps 3 base=mvs-texturing_amd/csrc/libmvs_viewsel.so ray2=$V/libmvs_viewsel_ray2.so ray4=$V/libmvs_viewsel_ray4.so ray8=$V/libmvs_viewsel_ray8.so > gpurun_out/r4ab/ab.txt 2>&1
echo "ab rc=$?"; grep -v amdgpu.ids gpurun_out/r4ab/ab.txt | tail -5
for v in base ray4 ray8; do
  lib=$V/libmvs_viewsel_$v.so; [ $v = base ] && lib=mvs-texturing_amd/csrc/libmvs_viewsel.so
  MVS_VIEWSEL_LIB=$PWD/$lib timeout 120 python scripts/rank_share_time.py --config 3 --parts 8 --reps 3 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$v', 'whole rays', round(d['whole']['stages_ms']['dc_rays'],3), 'share rays', round(d['share']['stages_ms']['dc_rays'],3), 'share total', round(d['share']['total_ms'],3))"
done
