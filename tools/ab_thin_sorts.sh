#!/bin/bash
# Round 6, VERDICT r5 item 1: sort work BESIDE the G2 accumulation.  Experiment build: make EXP=1 EXTRA="-DFK_S1_NT=256 -DFK_S1_LEAN"
# (s2_scatter1_thin_kernel on a register diet: 32 VGPRs, 256 lanes -- one wave of it fits the 32 registers per SIMD lane that the G2
# accumulation's two 240-register waves leave).  Legs, same box, the explicit 2^25 system out of the Parameters image:
#   prod            the production library
#   thin0_early0    experiment library, production scatter (1024 lanes), production queueing      (= prod, other binary)
#   thin1_early0    lean scatter, production queueing            -> does H's first sort pass stop crawling underneath the accumulations?
#   thin1_early1    lean scatter + the next proof's witness sorts queued before the wait for H's accumulation (FK_PROVE_EARLY_SORTS)
#   thin0_early1    production scatter + early sorts (round 5's experiment, for reference)
# KILL CRITERION (written before the run): the lean scatter stays only if thin1_early1 or thin1_early0 beats prod by >= 5 ms per step on this box
# with the same proof_sha256; a G2 accumulation that slows by more than the step gains is the expected way to fail.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/ab_thin; mkdir -p $O; rm -f $O/*.log
ARGS="--steps ${STEPS:-12} --warmup 4 --no-cpu-baseline --no-untiled --no-standalone --no-other-sizes --no-preflight --measure-traffic off"
run() { # name, env...
  local name=$1; shift
  env "$@" python3 bench.py $ARGS > $O/$name.log 2>&1; echo "$name rc=$?"
}
for rep in 1 2; do
  run prod_$rep FK_DUMMY=1
  run thin0_early0_$rep FK_LIB_VARIANT=exp FK_S1_THIN=0 FK_PROVE_EARLY_SORTS=0
  run thin1_early0_$rep FK_LIB_VARIANT=exp FK_S1_THIN=1 FK_PROVE_EARLY_SORTS=0
  run thin1_early1_$rep FK_LIB_VARIANT=exp FK_S1_THIN=1 FK_PROVE_EARLY_SORTS=1
  [ $rep = 1 ] && run thin0_early1_$rep FK_LIB_VARIANT=exp FK_S1_THIN=0 FK_PROVE_EARLY_SORTS=1
done
python3 - <<'PY' | tee gpurun_out/ab_thin/summary.txt
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_thin/*.log')):
    got=False
    for l in open(f):
        if l.startswith('{"metric"'):
            j=json.loads(l); k=j['kernel_ms_per_step']; got=True
            print('%-18s ms_per_step %7.2f  dev-resident %7.2f  latency %7.2f  acc_g1 %6.1f  acc_g2 %6.1f  ntt %5.1f  sha %s  digest %s' % (
                f.split('/')[-1][:-4], j['ms_per_step'], j['device_resident_ms_per_step'], j['latency_ms_per_proof'], k['msm_accumulate_g1'], k['msm_accumulate_g2'], k['ntt_passes'],
                j['proof_sha256'][0][:8], (j.get('oracle_digest_check') or {}).get('equal')))
    if not got:
        print(f.split('/')[-1], 'NO LINE:', open(f).read()[-400:].replace('\n',' | '))
PY
# one kernel trace of the most promising leg: where do the sort passes sit now?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
FK_LIB_VARIANT=exp FK_S1_THIN=1 FK_PROVE_EARLY_SORTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-untiled --no-standalone --no-other-sizes --no-preflight --measure-traffic off > $O/kt.log 2>&1
python3 tools/trace_union.py $O/kt auto > $O/kt_union.txt 2>&1
python3 tools/trace_window.py $O/kt ${WIN_K:-14} ${WIN_BEFORE:-160} 60 0.2 > $O/kt_boundary.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.csv" -size +20M -delete
head -60 $O/kt_union.txt
