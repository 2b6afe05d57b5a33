#!/bin/bash
mkdir -p gpurun_out/r4f
V=mvs-texturing_amd/csrc/variants
timeout 300 python scripts/sweep_probe.py --config 3 --rounds 1 --kernel 4 base= m0=$V/libmvs_viewsel_m0.so nonl=$V/libmvs_viewsel_m1.so noin=$V/libmvs_viewsel_m2.so nonl_noin=$V/libmvs_viewsel_m3.so norec=$V/libmvs_viewsel_m4.so nostore=$V/libmvs_viewsel_m8.so noold=$V/libmvs_viewsel_m16.so > gpurun_out/r4f/probe_c3.json 2> gpurun_out/r4f/probe_c3.err
cat gpurun_out/r4f/probe_c3.json
timeout 60 python scripts/sweep_probe.py --config 3 --rounds 1 --kernel 5 k5= > gpurun_out/r4f/probe_k5.json 2> gpurun_out/r4f/probe_k5.err
cat gpurun_out/r4f/probe_k5.json
timeout 300 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "golden or labels_bit or mrf_random or mrf_mixed or sweep_loop or stress" > gpurun_out/r4f/t1.log 2>&1
echo "t1 rc=$?"; tail -4 gpurun_out/r4f/t1.log
