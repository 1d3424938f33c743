# A/B of the two libraries: production (fused Fq2 mulsub) vs experiment build compiled without it
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/ab; rm -f gpurun_out/ab/*.log
for rep in 1 2; do
  for v in prod exp; do
    if [ $v = exp ]; then export FK_LIB_VARIANT=exp; else unset FK_LIB_VARIANT; fi
    python3 bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-untiled --no-standalone > gpurun_out/ab/lib_${v}_$rep.log 2>&1
    echo "$v rep=$rep rc=$?"
  done
done
unset FK_LIB_VARIANT
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/lib_*.log')):
    for l in open(f):
        if l.startswith('{"metric"'):
            j=json.loads(l); k=j['kernel_ms_per_step']
            print('%s  ms_per_step %.2f  dev-resident %.2f  acc_g1 %.1f  acc_g2 %.1f  ntt %.1f' % (f.split('/')[-1], j['ms_per_step'], j['device_resident_ms_per_step'], k['msm_accumulate_g1'], k['msm_accumulate_g2'], k['ntt_passes']))
PY
