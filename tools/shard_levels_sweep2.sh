# second sweep: levels on shard-sized arrays with a shorter serial chain in the single-set bucket reduction
# (FK_MSM_RED_L buckets per lane: the merged form has ONE bucket set, 2^19 buckets / 64 per lane = 32 workgroups -- the kernel stats of
# tools/shard_levels_trace.sh show its reduction at 2x the W-set form's) and wider windows.  Experiment library.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/shard_levels2; mkdir -p $O; rm -f $O/*.log
export FK_LIB_VARIANT=exp
for cfg in "24 3 0" "18 3 8" "18 3 16" "18 4 8" "18 5 8" "18 5 16"; do
  set -- $cfg
  FK_MSM_PRE_MIN_LOG2=$1 FK_MSM_PRE_DC=$2 FK_MSM_RED_L=$3 python3 tools/rank_budget.py --copies 1741 --ranks 4,8 --reps 5 > $O/min$1_dc$2_L$3.log 2>&1
  echo "== FK_MSM_PRE_MIN_LOG2=$1 FK_MSM_PRE_DC=$2 FK_MSM_RED_L=$3 (rc=$?)"; grep "^W = " $O/min$1_dc$2_L$3.log
done
