mkdir -p gpurun_out
S='import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); s=d["stages"]; print(d["n_gpus"], round(d["ms_per_step"],2), "sweeps", d["config"]["sweeps"], "energy", d["config"]["energy"], {k: (round(v["ms_per_step"],2), v["launches_per_step"]) for k,v in s.items()})'
export MVS_BENCH_ONE_GPU=1
for cfg in 2 3; do
echo "== 2 ranks gloo one GPU config $cfg"; ( time timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --config $cfg --steps 2 --warmup 1 --backend gloo 2>gpurun_out/mg_err_$cfg.log ) 2>&1 | python -c "$S"
tail -3 gpurun_out/mg_err_$cfg.log
done
