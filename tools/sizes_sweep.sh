# ms per proof (pipelined loop, host witness) of the synthetic shape at several sizes + two rollup batch sizes: regression sweep
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/sweep
for lg in ${SWEEP_LOGS:-20 22 24 26 27}; do
  python3 bench.py --workload synthetic --log2n $lg --steps 8 --warmup 2 --no-cpu-baseline --no-standalone --no-other-sizes > gpurun_out/sweep/syn$lg.log 2>&1
done
for c in 64 256; do
  python3 bench.py --copies $c --steps 8 --warmup 2 --no-cpu-baseline --no-standalone --no-untiled --no-other-sizes > gpurun_out/sweep/roll$c.log 2>&1
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/sweep/*.log')):
    ok=False
    for l in open(f):
        if l.startswith('{"metric"'):
            j=json.loads(l); ok=True
            print('%-12s ms_per_step %8.2f  dev-resident %8.2f  proofs/s %7.2f  levels %s' % (f.split('/')[-1], j['ms_per_step'], j['device_resident_ms_per_step'], j['value'], j['config']['msm_fixed_base_levels']))
            lp = j['config'].get('levels_plan')
            if lp and any(v['GiB'] < 0 for v in lp.values()):
                print('             planner: ' + ', '.join('%s %s%.1f GiB' % (k, '' if v['levels'] else 'LEFT OUT ', abs(v['GiB'])) for k, v in lp.items()))
    if not ok: print(f, 'FAILED', open(f).read()[-300:])
PY
