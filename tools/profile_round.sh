# Measurement pass of a round (run on the GPU box through gpurun): GPU tests, bench line, kernel trace, PMC passes and their summaries.
# Counter passes are separate runs and never combined with other trace domains; the program after `--` is python3 itself.
# Usage: bash tools/profile_round.sh [tag] [notest]      -> gpurun_out/<tag>/   (copy what is to be judged into profiles/)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r06}; O=gpurun_out/$TAG; mkdir -p $O
# the profiled legs prove ONE system only (the default one): every proof in a pass has the same size, so the number of proofs in a
# pass follows from its dispatch counts (tools/pmc_summary.py derives and cross-checks it)
B="bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-untiled --no-standalone --no-other-sizes --no-preflight --measure-traffic off"   # (never a profiler inside a profiler)
if [ "${2:-}" != "notest" ]; then
  python3 -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
fi
python3 bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -c 1500 $O/bench.log; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-untiled --no-standalone --no-other-sizes --no-preflight --measure-traffic off > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $B > $O/write.log 2>&1
rocprofv3 --pmc VALUBusy VALUUtilization MemUnitBusy --output-format csv -d $O/derived -o d -- python3 $B > $O/derived.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o s -- python3 $B > $O/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/gcal -o g -- tools/mulbench/gathercal > $O/gathercal.log 2>&1
# summaries: the proof count of a pass is derived from its dispatches; points and rows come from the pass's own bench line
python3 - $O <<'PY'
import json, subprocess, sys
O = sys.argv[1]
line = [l for l in open(O + '/fetch.log') if l.startswith('{"metric"')][-1]
c = json.loads(line)['config']
pts = c['msm_points']
g1 = pts['h'] + pts['l'] + pts['a'] + pts['b_g1']
subprocess.check_call([sys.executable, 'tools/pmc_summary.py', 'traffic', O + '/fetch', O + '/write', str(c['log2_domain']), str(g1), O + '/pmc_traffic.json', 'auto',
                       'rollup1024', O + '/gcal', str(c['rows'])])
subprocess.check_call([sys.executable, 'tools/pmc_summary.py', 'valu', O + '/sq', O + '/derived', O + '/pmc_valu_busy.json'])
PY
# the kernel trace itself is tens of MB: keep the statistics, compute the busy fraction first
python3 tools/trace_busy.py $O/kt > $O/kt_busy.txt 2>&1; cat $O/kt_busy.txt
# (the last evaluations of the traced run belong to its one-proof-at-a-time legs: the 14th last one lies in the pipelined loop)
python3 tools/trace_window.py $O/kt ${WIN_K:-14} ${WIN_BEFORE:-50} 40 0.2 > $O/kt_boundary.txt 2>&1
python3 tools/trace_union.py $O/kt auto > $O/kt_union.txt 2>&1; python3 tools/trace_gantt.py $O/kt 0.25 > $O/kt_gantt.txt 2>&1
find $O -name "*kernel_trace.csv" -delete
find $O -name "*.csv" -size +20M -delete
ls -la $O | head -40
