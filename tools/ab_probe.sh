# A/B on ONE box (GPU boxes differ by 2-3 %): alternate configurations, two rounds each.  Usage: bash tools/ab_probe.sh "ENV1=.. ENV2=.." "ENV=.." ...
set -u
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do for cfg in "$@"; do
  env $cfg python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | grep "^{" | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('round $round [$cfg]', round(j['ms_per_step'],2), round(j['device_resident_ms_per_step'],2))"
done; done
