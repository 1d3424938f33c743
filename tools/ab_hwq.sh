# same box: HIP's default of 4 hardware queues against GPU_MAX_HW_QUEUES=8 (seven streams per context: main, copy, auxiliary, four lanes)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/ab_hwq; rm -rf $O; mkdir -p $O
F="--steps 12 --warmup 4 --no-cpu-baseline --no-other-sizes --no-standalone --no-untiled"
for i in a b; do
  python3 bench.py $F > $O/q4_$i.log 2>&1
  GPU_MAX_HW_QUEUES=8 python3 bench.py $F > $O/q8_$i.log 2>&1
done
GPU_MAX_HW_QUEUES=6 python3 bench.py $F > $O/q6_a.log 2>&1
