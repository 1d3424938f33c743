set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/trace_shared; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-untiled --no-standalone > $O/kt.log 2>&1
python3 tools/trace_union.py $O/kt 11 > $O/kt_union.txt 2>&1; python3 tools/trace_gantt.py $O/kt 0.25 > $O/kt_gantt.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
cat $O/kt_gantt.txt | head -70
