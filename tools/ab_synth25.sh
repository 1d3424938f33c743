# A/B of environment settings on the synthetic 2^25 shape (round 1's workload).  Usage: bash tools/ab_synth25.sh "ENV=.." ...
set -u
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do for cfg in "$@"; do
  env $cfg python3 bench.py --workload synthetic --log2n 25 --steps 6 --warmup 2 --no-cpu-baseline 2>&1 | grep "^{" | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('synthetic 2^25 round $round [$cfg]', round(j['ms_per_step'],2), round(j['device_resident_ms_per_step'],2))"
done; done
