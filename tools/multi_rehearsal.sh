# Rehearsal of `bench.py --gpus N` on a ONE-GPU box: N ranks share GPU 0 over gloo (FK_BENCH_SAME_DEVICE=1) -- the multi-GPU code
# path end to end (sharded keys, balanced / distributed schedule, witness slots, all-gather, replica leg), not a measurement.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/multi; mkdir -p $O
run() { # name, nproc, env..., -- args
  name=$1; np=$2; shift 2
  env FK_BENCH_SAME_DEVICE=1 "$@" python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $np --master-addr 127.0.0.1 --master-port 29577 \
      bench.py --gpus $np --backend gloo --copies 64 --steps 2 --warmup 1 --no-cpu-baseline > $O/$name.log 2>&1
  echo "$name rc=$?: $(grep '^{' $O/$name.log | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print(j['n_gpus'], j['config']['parallelism'], 'ms', round(j['ms_per_step'],1), 'replica', j.get('replica_proofs_per_sec'))" 2>/dev/null)"
  tail -2 $O/$name.log | cut -c1-200 | grep -v '^{' | head -2
}
run w2_balanced 2 FK_X=0
run w2_distq 2 FK_DIST_QUOTIENT=1
run w3_balanced 3 FK_X=0
run w4_distq 4 FK_X=0
run w4_balanced 4 FK_DIST_QUOTIENT=0
