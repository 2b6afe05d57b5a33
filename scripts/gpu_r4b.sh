#!/bin/bash
# round 4, run b: the sweep-kernel probes, then the GPU suite after the layout change (every step under its own timeout)
mkdir -p gpurun_out/r4b
V=mvs-texturing_amd/csrc/variants
timeout 240 python scripts/sweep_probe.py --config 3 --rounds 1 base= exp1=$V/libmvs_viewsel_exp1.so exp2=$V/libmvs_viewsel_exp2.so exp3=$V/libmvs_viewsel_exp3.so exp4=$V/libmvs_viewsel_exp4.so > gpurun_out/r4b/probe_c3.json 2> gpurun_out/r4b/probe_c3.err
timeout 120 python scripts/sweep_probe.py --config 2 --rounds 1 base= exp1=$V/libmvs_viewsel_exp1.so exp2=$V/libmvs_viewsel_exp2.so exp3=$V/libmvs_viewsel_exp3.so > gpurun_out/r4b/probe_c2.json 2> gpurun_out/r4b/probe_c2.err
cat gpurun_out/r4b/probe_c3.json gpurun_out/r4b/probe_c2.json
timeout 800 python -m pytest tests -m gpu -q --tb=short --maxfail=8 -p no:cacheprovider > gpurun_out/r4b/test.log 2>&1
tail -4 gpurun_out/r4b/test.log
