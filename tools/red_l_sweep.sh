# Buckets per lane of the single-set (merged) bucket reduction at the benchmark size: FK_MSM_RED_L_MERGED = 64 (rounds 1-3), 32, 16, 8.
# The reduction of B2 and of H is the latency-bound tail a proof ends on.  Experiment library, same box, alternating.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/red_l; mkdir -p $O; rm -f $O/*.log
export FK_LIB_VARIANT=exp
for rep in 1 2; do
  for L in 64 16 32 8; do
    FK_MSM_RED_L_MERGED=$L python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-untiled --no-standalone --no-other-sizes > $O/L${L}_$rep.log 2>&1
    python3 - "$O/L${L}_$rep.log" "L=$L rep=$rep" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        j = json.loads(l); print('%-16s ms_per_step %8.2f  dev-resident %8.2f  latency %8.2f' % (sys.argv[2], j['ms_per_step'], j['device_resident_ms_per_step'], j['latency_ms_per_proof']))
PY
  done
done
