cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$(echo $c | tr ' ' '_')
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/probe_$d -o pmc -- python $R/scripts/sweep_probe.py --sweeps 5 "${PROBE_VARIANTS:-base:}" > $R/gpurun_out/probe_$d.log 2>&1
  f=$(find $R/gpurun_out/probe_$d -name "*counter_collection.csv" | head -1)
  python $R/scripts/pmc_summary.py $f "mrf_sweep4" --groups ${PROBE_GROUPS:-1}
done
tail -4 $R/gpurun_out/probe_FETCH_SIZE.log
