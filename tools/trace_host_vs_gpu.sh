# When does the HOST issue the launches of a proof's front, and when do they start on the GPU?  (GPU box; kernel + HIP API + copy traces, no counters)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/trace_host; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace --memory-copy-trace --output-format csv -d $O/t -o t -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-untiled --no-standalone > $O/log 2>&1
ls -la $O/t | head
python3 - <<'PY'
import csv, glob
O='gpurun_out/trace_host/t'
kt=[r for f in glob.glob(O+'/**/*kernel_trace.csv',recursive=True) for r in csv.DictReader(open(f))]
api=[r for f in glob.glob(O+'/**/*hip_api_trace.csv',recursive=True) for r in csv.DictReader(open(f))]
mc=[r for f in glob.glob(O+'/**/*memory_copy_trace.csv',recursive=True) for r in csv.DictReader(open(f))]
print(len(kt),'kernels',len(api),'api calls',len(mc),'copies')
if api: print(api[0].keys())
if mc: print(mc[0].keys())
kt.sort(key=lambda r:int(r['Start_Timestamp']))
sp=[r for r in kt if 'spmv_binned' in r['Kernel_Name']]
t0=int(sp[4]['Start_Timestamp'])      # the 5th evaluation of the run: inside the pipelined loop (the last ones are the one-proof-at-a-time leg)
corr={r.get('Correlation_Id'):r for r in api}
print('t = 0: GPU start of the evaluation kernel of the 5th proof of the run (pipelined loop)')
for r in kt:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if -50e6 < s-t0 < 15e6 and (e-s>2e5 or 'spmv' in r['Kernel_Name'] or 'gather' in r['Kernel_Name']):
        a=corr.get(r.get('Correlation_Id'))
        host=(int(a['Start_Timestamp'])-t0)/1e6 if a else float('nan')
        print('%9.2f %8.2f  host issue %9.2f  q%s  %s'%((s-t0)/1e6,(e-s)/1e6,host,r.get('Queue_Id','?'),r['Kernel_Name'].split('(')[0][-50:]))
for r in mc:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    if -200e6 < s-t0 < 15e6 and int(r.get('Bytes',r.get('Size',0)) or 0) > 1e6:
        print('copy %9.2f %8.2f ms  %s bytes  %s'%((s-t0)/1e6,(e-s)/1e6,r.get('Bytes',r.get('Size')),r.get('Direction','')))
PY
find $O -name "*.csv" -size +1M -delete
