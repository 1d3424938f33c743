# PMC passes over the evaluation of a, b, c of the benchmark's system alone (GPU box; separate passes, no other trace domains)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/spmv_pmc; mkdir -p $O
export ONLY=whole
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 tools/spmv_rollup_probe.py > $O/kt.log 2>&1
timeout 300 rocprofv3 --pmc VALUBusy MemUnitBusy VALUUtilization --output-format csv -d $O/d1 -o d -- python3 tools/spmv_rollup_probe.py > $O/d1.log 2>&1
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/d2 -o d -- python3 tools/spmv_rollup_probe.py > $O/d2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/d3 -o d -- python3 tools/spmv_rollup_probe.py > $O/d3.log 2>&1      # FETCH_SIZE and WRITE_SIZE together exceed the counter hardware: rocprofv3 aborts and then waits -- one per pass, under a timeout
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/d4 -o d -- python3 tools/spmv_rollup_probe.py > $O/d4.log 2>&1
python3 - <<'PY'
import csv,glob,collections
for d in ('d1','d2','d3','d4'):
    for f in glob.glob('gpurun_out/spmv_pmc/%s/**/*counter_collection.csv'%d, recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][-40:]
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items():
            if 'spmv' in k:
                print(d,k,{c:(sum(x)/len(x)) for c,x in v.items()}, 'launches', len(next(iter(v.values()))))
for f in glob.glob('gpurun_out/spmv_pmc/kt/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'spmv' in r['Name']: print(r['Name'].split('(')[0][-40:], r['Calls'], r['AverageNs'])
PY
find $O -name "*kernel_trace.csv" -delete
