# Experiment (GPU box): 9 x 29-bit limb accumulator on / off, lane counts, few distinct witnesses
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/l29; mkdir -p $O
python -m pytest tests/test_gpu_msm.py tests/test_gpu_precompute.py tests/test_gpu_prove.py tests/test_gpu_setup.py -q -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
run() {  # name, env...
  name=$1; shift
  env "$@" CIRCUIT=rollup COPIES=1024 TILED=1 WORKERS=128 python3 tools/eddsa_batch_probe.py > $O/$name.log 2>&1
  echo "$name: $(grep 'witness resident' $O/$name.log)"
}
run d64_l29 DISTINCT=64 ZCACHE=/tmp/z64.npz
run d64_l32 DISTINCT=64 ZCACHE=/tmp/z64.npz FK_MSM_LIMB29=0
run d64_l29_lanes3 DISTINCT=64 ZCACHE=/tmp/z64.npz FK_MSM_LANES=3
run d8_l29 DISTINCT=8
run d16_l29 DISTINCT=16
run d32_l29 DISTINCT=32
