# PMC probe of the sweep kernel: each counter group in its own pass (rocprofv3 --pmc; never combined with traces)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
i=0
if [ -n "$PROBE_SHORT" ]; then SETS=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"); else SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "GRBM_GUI_ACTIVE"); fi
for c in "${SETS[@]}"; do
  i=$((i+1)); d=p$i
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/probe2_$d -o pmc -- python $R/scripts/sweep_probe.py --sweeps 4 "${PROBE_VARIANTS:-base:}" > $R/gpurun_out/probe2_$d.log 2>&1
  f=$(find $R/gpurun_out/probe2_$d -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/scripts/pmc_summary.py $f "${PROBE_KERNEL:-mrf_sweep4}" --groups ${PROBE_GROUPS:-1}; else echo "pass failed: $c"; tail -2 $R/gpurun_out/probe2_$d.log; fi
  rm -rf $R/gpurun_out/probe2_$d
done
tail -3 $R/gpurun_out/probe2_p1.log
