mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
V=mvs-texturing_amd/csrc/variants
( timeout 500 python scripts/sweep_probe.py --config 3 --sweeps 20 --rounds 2 base=mvs-texturing_amd/csrc/libmvs_viewsel.so exp5=$V/libmvs_viewsel_exp5.so exp6=$V/libmvs_viewsel_exp6.so exp7=$V/libmvs_viewsel_exp7.so exp8=$V/libmvs_viewsel_exp8.so exp1=$V/libmvs_viewsel_exp1.so exp2=$V/libmvs_viewsel_exp2.so ) > gpurun_out/r05_sweep_probe.json 2> gpurun_out/r05c_probe.err; tail -2 gpurun_out/r05c_probe.err; cat gpurun_out/r05_sweep_probe.json
( timeout 400 python scripts/ab_libs.py --config 3 --rounds 3 --steps 3 base=mvs-texturing_amd/csrc/libmvs_viewsel.so prefetch=$V/libmvs_viewsel_rayprefetch.so ) 2>&1 | tail -4 | cut -c1-900
