# Kernel timeline of ONE standalone 2^20 G1 multiplication (BASELINE configs[1]) -> where its 2.5 ms go.   -> gpurun_out/msm20/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
LOG=${1:-20}; O=gpurun_out/msm$LOG; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 tools/msm_small_trace.py $LOG 20 > $O/run.log 2>&1
tail -1 $O/run.log
python3 - $O/kt <<'PY'
import csv, glob, os, re, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), re.sub(r'\(.*$', '', r['Kernel_Name']).replace('void ', '').replace('fk::', '')[:60], r.get('Queue_Id', '?')))
rows.sort()
i0 = [i for i, r in enumerate(rows) if r[2].startswith('msm_digits_kernel')][-1]
t0 = rows[i0][0]
print('last multiplication, one line per launch: start (us), duration (us), gap before (us), queue, kernel')
prev_end = t0
for s, e, n, q in rows[i0:]:
    print('%9.1f %9.1f %8.1f  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, n))
    prev_end = max(prev_end, e)
print('span %.1f us' % ((prev_end - t0) / 1e3))
PY
find $O -name "*kernel_trace.csv" -delete
