cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/trace_r06 -o t -- python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-traffic --no-parity --no-real-like --no-shuffled --no-dropin > $REPO/gpurun_out/trace.log 2>&1
cd $REPO; f=$(find gpurun_out/trace_r06 -name "*kernel_trace.csv" | head -1); echo $f; python - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step: find last occurrence of mrf_sortkey / first kernel of mrf setup: print from last 'mrf_size_kernel' backwards?  print the last 400 kernels compactly
out=[]
prev=None
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    out.append((r['Kernel_Name'][:70], (e-s)/1e3, (s-prev)/1e3 if prev else 0)); prev=e
open('gpurun_out/trace_r06_compact.txt','w').write('\n'.join('%-70s %9.1f us  gap %8.1f' % o for o in out))
print(len(out))
P
rm -rf gpurun_out/trace_r06
