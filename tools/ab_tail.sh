# same box: the G2 tail on a high-priority stream (default) against FK_G2_TAIL_PRIORITY=0 (on its lane, rounds 1-4)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/ab_tail; rm -rf $O; mkdir -p $O
F="--steps 12 --warmup 4 --no-cpu-baseline --no-other-sizes --no-standalone --no-untiled"
for i in a b; do
  python3 bench.py $F > $O/prio_$i.log 2>&1
  FK_G2_TAIL_PRIORITY=0 python3 bench.py $F > $O/lane_$i.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-untiled --no-standalone --no-other-sizes > $O/kt.log 2>&1
python3 tools/trace_gantt.py $O/kt 0.25 > $O/kt_gantt.txt 2>&1
python3 tools/trace_union.py $O/kt auto > $O/kt_union.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.csv" -size +20M -delete
