# A/B of environment settings at the smaller sizes (GPU box).  Usage: bash tools/ab_small.sh "ENV=.." "ENV=.." ...
set -u
cd "$GRAFT_REPO_ROOT"
for L in 20 22 24; do for cfg in "$@"; do
  env $cfg python3 bench.py --workload synthetic --log2n $L --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('2^$L [$cfg]', round(j['ms_per_step'],2), round(j['device_resident_ms_per_step'],2))"
done; done
for C in 64 256; do for cfg in "$@"; do
  env $cfg python3 bench.py --copies $C --steps 12 --warmup 3 --no-cpu-baseline 2>&1 | grep "^{" | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('rollup$C [$cfg]', round(j['ms_per_step'],2), round(j['device_resident_ms_per_step'],2))"
done; done
