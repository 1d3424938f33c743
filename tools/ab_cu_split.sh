#!/bin/bash
# Round 6: what does an accumulation lose when it may use only (8 - k) / 8 of the compute units?  Experiment build, lane streams (sorts, accumulations,
# tails) created with a CU mask (FK_CU_SPLIT=k, FK_CU_SPLIT_MODE=0 / 1); the main stream (evaluation, transforms) keeps the whole chip.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/ab_cu; mkdir -p $O; rm -f $O/*.log
ARGS="--steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-untiled --no-standalone --no-other-sizes --no-preflight --measure-traffic off"
run() { local name=$1; shift; env "$@" python3 bench.py $ARGS > $O/$name.log 2>&1; echo "$name rc=$?"; }
run k0 FK_LIB_VARIANT=exp FK_S1_THIN=0
for k in 1 2; do for m in 0 1; do run k${k}_mode$m FK_LIB_VARIANT=exp FK_S1_THIN=0 FK_CU_SPLIT=$k FK_CU_SPLIT_MODE=$m; done; done
run k0_again FK_LIB_VARIANT=exp FK_S1_THIN=0
python3 - <<'PY' | tee gpurun_out/ab_cu/summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_cu/*.log')):
    got=False
    for l in open(f):
        if l.startswith('{"metric"'):
            j=json.loads(l); k=j['kernel_ms_per_step']; got=True
            print('%-12s ms_per_step %7.2f  dev-resident %7.2f  acc_g1 %6.1f (sum %6.1f)  acc_g2 %6.1f  ntt %5.1f  digest %s' % (
                f.split('/')[-1][:-4], j['ms_per_step'], j['device_resident_ms_per_step'], k['msm_accumulate_g1'], k['msm_accumulate_g1_sum_of_side_by_side_launches'], k['msm_accumulate_g2'], k['ntt_passes'],
                (j.get('oracle_digest_check') or {}).get('equal')))
    if not got: print(f.split('/')[-1], 'NO LINE:', open(f).read()[-300:].replace('\n',' | '))
PY
