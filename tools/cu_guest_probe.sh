#!/bin/bash
# see tools/cu_guest_probe.py.  Legs: (1) the guest ALONE on the whole chip (no mask), (2) the guest alone on the set-aside units, (3) the guest on the
# set-aside units while bench.py proves on the others (its accumulations masked), (4) the same with the host NOT masked (what a guest costs without a partition)
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/cu_guest; mkdir -p $O; rm -f $O/* /tmp/fk_guest_ready
HOSTARGS="--steps 40 --warmup 3 --no-cpu-baseline --no-untiled --no-standalone --no-other-sizes --no-preflight --measure-traffic off"
echo "== (1) guest alone, whole chip"; FK_LIB_VARIANT=exp GUEST_SECONDS=6 python3 tools/cu_guest_probe.py 2>&1 | tail -3
echo "== (2) guest alone on the set-aside units"; rm -f /tmp/fk_guest_ready; FK_LIB_VARIANT=exp FK_CU_SPLIT=4 FK_CU_SPLIT_MODE=2 FK_CU_SPLIT_ALL=1 GUEST_SECONDS=8 python3 tools/cu_guest_probe.py 2>&1 | tail -3
for hostmask in 1 0; do
  echo "== guest on the set-aside units, host proving (host masked: $hostmask)"
  rm -f /tmp/fk_guest_ready
  FK_LIB_VARIANT=exp FK_CU_SPLIT=4 FK_CU_SPLIT_MODE=2 FK_CU_SPLIT_ALL=1 GUEST_SECONDS=150 python3 tools/cu_guest_probe.py > $O/guest_$hostmask.log 2>&1 &
  GP=$!
  for i in $(seq 1 120); do [ -f /tmp/fk_guest_ready ] && break; sleep 1; done
  if [ $hostmask = 1 ]; then FK_LIB_VARIANT=exp FK_S1_THIN=0 FK_CU_SPLIT=4 FK_CU_SPLIT_MODE=2 python3 bench.py $HOSTARGS > $O/host_$hostmask.log 2>&1
  else FK_LIB_VARIANT=exp FK_S1_THIN=0 python3 bench.py $HOSTARGS > $O/host_$hostmask.log 2>&1; fi
  echo "host rc=$?"
  kill $GP 2>/dev/null; wait $GP 2>/dev/null
  python3 - $O/host_$hostmask.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if l:
    j=json.loads(l[-1]); k=j['kernel_ms_per_step']
    print('  host: ms_per_step %.1f  acc_g1 %.1f  acc_g2 %.1f  ntt %.1f' % (j['ms_per_step'], k['msm_accumulate_g1'], k['msm_accumulate_g2'], k['ntt_passes']))
else: print('  host: NO LINE', open(sys.argv[1]).read()[-300:])
PY
  tail -4 $O/guest_$hostmask.log
done
