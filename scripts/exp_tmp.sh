mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
  d=$(echo $set | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/sq_$d -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/sq_$d.log 2>&1
  f=$(find $R/gpurun_out/sq_$d -name "*counter_collection.csv" | head -1)
  echo "== $set"; python $R/scripts/pmc_summary.py $f "ray_packet2"
  rm -rf $R/gpurun_out/sq_$d
done
