# Experiment (GPU box): how the MSM lanes should share the GPU on the 1024-transaction workload, and what repeated witness
# values cost.  Prints one line per configuration.  Usage: bash tools/lanes_probe.sh  (through gpurun)
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/lanes; mkdir -p $O
run() {  # name, env...
  name=$1; shift
  env "$@" CIRCUIT=rollup COPIES=1024 TILED=1 WORKERS=128 python3 tools/eddsa_batch_probe.py > $O/$name.log 2>&1
  echo "$name: $(grep 'witness resident' $O/$name.log)"
}
run d1024_default DISTINCT=1024 ZCACHE=/tmp/z1024.npz
run d1024_lanes1 DISTINCT=1024 ZCACHE=/tmp/z1024.npz FK_MSM_LANES=1
run d1024_lanes3 DISTINCT=1024 ZCACHE=/tmp/z1024.npz FK_MSM_LANES=3
run d1024_sortalone DISTINCT=1024 ZCACHE=/tmp/z1024.npz FK_MSM_SORT_ALONE=1
run d1024_sortalone_l3 DISTINCT=1024 ZCACHE=/tmp/z1024.npz FK_MSM_SORT_ALONE=1 FK_MSM_LANES=3
run d3_default DISTINCT=3
run d64_default DISTINCT=64
run d128_default DISTINCT=128
run d3_oldmany DISTINCT=3 FK_MSM_OVER_MANY=400000
