# one kernel trace of the bench's pipelined loop (GPU box): per-kernel union durations, the last proof's timeline and the boundary
# between two pipelined proofs
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/trace_once; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-untiled --no-standalone > $O/kt.log 2>&1
python3 tools/trace_gantt.py $O/kt 0.25 > $O/kt_gantt.txt 2>&1
# (the last evaluations of a bench run belong to its one-proof-at-a-time leg: the 10th last one lies in the pipelined loop)
python3 tools/trace_window.py $O/kt ${WIN_K:-10} ${WIN_BEFORE:-50} 35 0.2 > $O/kt_boundary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
cat $O/kt_boundary.txt
