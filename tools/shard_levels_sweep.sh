# Fixed-base levels on SHARD-sized arrays: the per-rank share of the N-GPU prover (tools/rank_budget.py: rank 0 of
# N alone on this GPU) with the levels kept on shards below 2^24 points (FK_MSM_PRE_MIN_LOG2 lowered) and the merged form's window
# rule varied (FK_MSM_PRE_DC: c = base rule + dc), against the release rule (no levels below 2^24).  Experiment library.
# Usage: bash tools/shard_levels_sweep.sh [copies] [ranks]   -> gpurun_out/shard_levels/
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/shard_levels; mkdir -p $O; rm -f $O/*.log
COPIES=${1:-1741}; RANKS=${2:-2,4,8}
export FK_LIB_VARIANT=exp
for cfg in "24 3" "18 3" "18 2" "18 1" "18 0"; do
  set -- $cfg
  FK_MSM_PRE_MIN_LOG2=$1 FK_MSM_PRE_DC=$2 python3 tools/rank_budget.py --copies $COPIES --ranks $RANKS > $O/min$1_dc$2.log 2>&1
  echo "== FK_MSM_PRE_MIN_LOG2=$1 FK_MSM_PRE_DC=$2 (rc=$?)"; grep "^W = " $O/min$1_dc$2.log
done
