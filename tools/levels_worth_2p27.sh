#!/bin/bash
# Round 6, VERDICT r5 item 6 (halved fixed-base levels through the G1 endomorphism): what are the levels of ONE array worth at 2^27 on one GPU?
# The synthetic 2^27 system proved (a) as the loader plans it -- levels for `a` only, the other arrays' 11 further copies do not fit beside the 48 GiB
# key and the proof scratch -- and (b) with FK_MSM_PRECOMP=0 (no levels at all).  X = (b) - (a).  With halved levels (5 further copies instead of 11:
# 20 GiB per 2^26-point G1 array) the same HBM would cover `a` and ONE more array of that size, so the whole lever is bounded by 2 X minus the
# endomorphism's own price (one Fq product per gathered point of six of the twelve levels, ~5 % of those additions).
# KILL CRITERION (written before the run): 2 X < 5 % of (a).
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/levels27; mkdir -p $O
A="--workload synthetic --log2n 27 --steps 4 --warmup 2 --no-cpu-baseline --no-standalone --no-other-sizes --no-preflight --measure-traffic off"
python3 bench.py $A > $O/planned.log 2>&1; echo "planned rc=$?"
FK_MSM_PRECOMP=0 python3 bench.py $A > $O/nolevels.log 2>&1; echo "nolevels rc=$?"
python3 - <<'PY'
import json
r={}
for k in ('planned','nolevels'):
    for l in open('gpurun_out/levels27/%s.log' % k):
        if l.startswith('{"metric"'):
            j=json.loads(l); r[k]=j
            print('%-9s ms_per_step %8.2f  dev-resident %8.2f  levels %s  acc_g1 %.1f acc_g2 %.1f' % (k, j['ms_per_step'], j['device_resident_ms_per_step'], j['config']['msm_fixed_base_levels'],
                  j['kernel_ms_per_step']['msm_accumulate_g1'], j['kernel_ms_per_step']['msm_accumulate_g2']))
            lp=j['config'].get('levels_plan')
            if lp: print('          planner: ' + ', '.join('%s %s%.1f GiB' % (a, '' if v['levels'] else 'LEFT OUT ', abs(v['GiB'])) for a, v in lp.items()))
if len(r)==2:
    x=r['nolevels']['ms_per_step']-r['planned']['ms_per_step']
    print('X = %.2f ms = %.2f %% of the planned proof; bound of the halved-levels lever 2 X = %.2f %%  -> %s' % (x, 100*x/r['planned']['ms_per_step'], 200*x/r['planned']['ms_per_step'],
          'KILLED (< 5 %)' if 200*x/r['planned']['ms_per_step'] < 5 else 'worth building'))
PY
