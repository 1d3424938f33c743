set -u
cd "$GRAFT_REPO_ROOT"
for L in 20 22 24; do for cfg in "FK_MSM_LANES=2" "FK_MSM_LANES=3"; do
  env $cfg python3 bench.py --workload synthetic --log2n $L --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('2^$L [$cfg]', round(j['ms_per_step'],2), round(j['device_resident_ms_per_step'],2))"
done; done
for cfg in "FK_MSM_LANES=2" "FK_MSM_LANES=3"; do
  env $cfg python3 bench.py --copies 64 --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('rollup64 [$cfg]', round(j['ms_per_step'],2), round(j['device_resident_ms_per_step'],2))"
done
