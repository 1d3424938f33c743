# same box, same process settings: `value` on the explicit system out of the Parameters image (default) against the tiled headline of
# rounds 1-4, then a kernel trace of the default form's pipelined loop (boundary between two proofs)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/ab_forms; rm -rf $O; mkdir -p $O
F="--steps 12 --warmup 4 --no-cpu-baseline --no-other-sizes --no-standalone"
python3 bench.py $F > $O/explicit_a.log 2>&1
python3 bench.py --tiled-headline $F > $O/tiled_a.log 2>&1
python3 bench.py $F > $O/explicit_b.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-untiled --no-standalone --no-other-sizes > $O/kt.log 2>&1
python3 tools/trace_gantt.py $O/kt 0.25 > $O/kt_gantt.txt 2>&1
python3 tools/trace_window.py $O/kt ${WIN_K:-10} ${WIN_BEFORE:-50} 35 0.2 > $O/kt_boundary.txt 2>&1
python3 tools/trace_union.py $O/kt auto > $O/kt_union.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.csv" -size +20M -delete
