# Fixed-base levels BELOW 2^24 points on one GPU, with a short serial chain in the single-set bucket reduction (FK_MSM_RED_L): round 1
# measured the merged form slower there (2^22: 29.9 vs 25.8 ms) -- with 64 buckets per reduction lane, i.e. 32 workgroups for the
# one bucket set.  Experiment library; every figure is bench.py's ms_per_step (host-witness pipeline).  -> gpurun_out/small_levels/
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/small_levels; mkdir -p $O; rm -f $O/*.log
export FK_LIB_VARIANT=exp
run() {  # tag, env..., -- bench args
  tag=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-untiled --no-standalone --no-other-sizes "$@" > $O/$tag.log 2>&1
  python3 - "$O/$tag.log" "$tag" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        j = json.loads(l); print('%-40s ms_per_step %8.2f  dev-resident %8.2f  levels %s' % (sys.argv[2], j['ms_per_step'], j['device_resident_ms_per_step'], j['config']['msm_fixed_base_levels']))
PY
}
for wl in "syn20 --workload synthetic --log2n 20" "syn22 --workload synthetic --log2n 22" "syn23 --workload synthetic --log2n 23" "roll64 --copies 64" "roll256 --copies 256" "roll512 --copies 512"; do
  set -- $wl; name=$1; shift
  run ${name}_release FK_NOP=1 -- "$@"
  run ${name}_lev_L8 FK_MSM_PRE_MIN_LOG2=18 FK_MSM_RED_L=8 -- "$@"
  run ${name}_lev_L16 FK_MSM_PRE_MIN_LOG2=18 FK_MSM_RED_L=16 -- "$@"
  run ${name}_nolev_L8 FK_MSM_RED_L=8 -- "$@"
done
