mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
( timeout 400 python scripts/rank_share_mrf.py --config 3 --parts 8 ) > gpurun_out/r05_rank_share_mrf_c3.json 2> gpurun_out/r05e_mrf.err; tail -3 gpurun_out/r05e_mrf.err; cat gpurun_out/r05_rank_share_mrf_c3.json
( timeout 300 python scripts/rank_share_time.py --config 3 --parts 8 --reps 3 ) > gpurun_out/r05_rank_share_dc_c3.json 2> gpurun_out/r05e_dc.err; tail -2 gpurun_out/r05e_dc.err; cat gpurun_out/r05_rank_share_dc_c3.json
( timeout 500 python scripts/transport_time.py --config 3 --parts 2,4,8 --reps 3 ) > gpurun_out/r05_transport_c3.json 2> gpurun_out/r05e_tr.err; tail -7 gpurun_out/r05e_tr.err
( time timeout 400 python bench.py --config 2 --steps 20 --warmup 2 --no-cpu-baseline ) > gpurun_out/r05_bench_c2.json 2> gpurun_out/r05e_c2.err; tail -2 gpurun_out/r05e_c2.err; python -c "
import json; d=json.load(open('gpurun_out/r05_bench_c2.json')); print({k: d[k] for k in ('value','ms_per_step','parity_checked')}, 'frac', d['roofline']['frac']); print({k: round(v['ms_per_step'],3) for k,v in d['stages'].items()})"
( MVS_BENCH_ONE_GPU=1 timeout 400 python bench.py --gpus 8 --steps 3 --warmup 1 ) > gpurun_out/r05_bench_c3_inproc8_one_gpu.json 2> gpurun_out/r05e_in8.err; tail -2 gpurun_out/r05e_in8.err; python -c "
import json; d=json.load(open('gpurun_out/r05_bench_c3_inproc8_one_gpu.json')); print({k: d[k] for k in ('n_gpus','ms_per_step','parity_checked')}, d['halo'], d['per_rank_ms_per_step'])"
