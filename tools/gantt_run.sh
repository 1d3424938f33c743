# kernel timeline of one proof (GPU box): tools/trace_gantt.py over a short traced bench run
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/gantt; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o k -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
python3 tools/trace_gantt.py $O/kt 0.25 > $O/gantt.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
tail -5 $O/gantt.txt
