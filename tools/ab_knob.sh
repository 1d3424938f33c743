# same-box A/B of one tuning knob of the experiment build: bash tools/ab_knob.sh FK_SOME_KNOB [bench args]
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/*.log
K=$1; shift
export FK_LIB_VARIANT=exp
for rep in 1 2; do
  for v in 1 0; do
    env $K=$v python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-untiled --no-standalone "$@" > gpurun_out/ab/${K}_${v}_$rep.log 2>&1
    echo "$K=$v rep=$rep rc=$?"
  done
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/*.log')):
    for l in open(f):
        if l.startswith('{"metric"'):
            j=json.loads(l); k=j['kernel_ms_per_step']
            print('%s  ms_per_step %.2f  dev-resident %.2f  acc_g1 %.1f  acc_g2 %.1f  ntt %.1f' % (f.split('/')[-1], j['ms_per_step'], j['device_resident_ms_per_step'], k['msm_accumulate_g1'], k['msm_accumulate_g2'], k['ntt_passes']))
PY
