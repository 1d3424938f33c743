# Per-kernel times of rank 0-of-8's share with and without fixed-base levels on its (4 M-point) shard arrays: why does the merged
# form lose at shard sizes?  (tools/shard_levels_sweep.sh has the totals.)  -> gpurun_out/shard_trace/
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/shard_trace; mkdir -p $O
export FK_LIB_VARIANT=exp
for cfg in "24 3" "18 3"; do
  set -- $cfg
  export FK_MSM_PRE_MIN_LOG2=$1 FK_MSM_PRE_DC=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/min$1_dc$2 -o k -- python3 tools/rank_budget.py --copies 1741 --ranks 8 --reps 5 > $O/min$1_dc$2.log 2>&1
  f=$(find $O/min$1_dc$2 -name "*kernel_stats.csv" | head -1)
  echo "== FK_MSM_PRE_MIN_LOG2=$1 FK_MSM_PRE_DC=$2"; grep "^W = " $O/min$1_dc$2.log
  python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    name = re.sub(r'\(.*$', '', r['Name']).replace('void ', '').replace('fk::', '')[:70]
    print('%-72s calls %5s  total %9.2f ms  avg %8.3f ms  %5.1f%%' % (name, r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e6, float(r['Percentage'])))
PY
  find $O -name "*kernel_trace.csv" -delete
done
