# kernel traces of the pipelined loop with 4 (default) and 8 hardware queues: the boundary between two pipelined proofs
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/trace_hwq; rm -rf $O; mkdir -p $O
F="--steps 6 --warmup 2 --no-cpu-baseline --no-untiled --no-standalone --no-other-sizes"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/q4 -o k -- python3 bench.py $F > $O/q4.log 2>&1
python3 tools/trace_window.py $O/q4 14 50 45 0.2 > $O/q4_boundary.txt 2>&1
GPU_MAX_HW_QUEUES=8 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/q8 -o k -- python3 bench.py $F > $O/q8.log 2>&1
python3 tools/trace_window.py $O/q8 14 50 45 0.2 > $O/q8_boundary.txt 2>&1
python3 - <<'PY'
import csv, glob
for q in ('q4', 'q8'):
    for f in glob.glob('gpurun_out/trace_hwq/%s/**/*memory_copy_trace.csv' % q, recursive=True):
        rows = list(csv.DictReader(open(f)))
        big = [r for r in rows if int(r.get('Size', r.get('size', 0)) or 0) > 500000000]
        print(q, f.split('/')[-1], len(rows), 'copies,', len(big), 'large;', rows[0].keys() if rows else '')
        for r in big[-6:]:
            print('   ', r)
PY
find $O -name "*kernel_trace.csv" -delete; find $O -name "*memory_copy_trace.csv" -size +5M -delete
