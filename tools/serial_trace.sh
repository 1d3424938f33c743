# Standalone cost of every kernel of a proof (GPU box): one lane, quotient first -- nothing overlaps -- under rocprofv3 --kernel-trace --stats.
# The knobs it sets are tuning knobs: they exist in the experiment build only (make -C fawkes-crypto_amd/csrc EXP=1 -> libfawkes_hip_exp.so).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/serial; mkdir -p $O
export FK_LIB_VARIANT=exp FK_MSM_PRE_DC=${PRE_DC:-3} FK_MSM_LANES=1 FK_PROVE_WITNESS_FIRST=0 FK_PROVE_SORTS_FIRST=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-untiled --no-standalone > $O/kt.log 2>&1
grep '^{' $O/kt.log | python3 -c "
import json,sys
for l in sys.stdin:
    j=json.loads(l); print('serial: ms_per_step', j['ms_per_step'], j['kernel_ms_per_step'])"
find $O -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv,re
rows=list(csv.DictReader(open('gpurun_out/serial/kt/k_kernel_stats.csv')))
def short(n):
    n=re.sub(r'\(.*$','',n).replace('void ','').replace('fk::','')
    return n.replace('Fp<FqParams, true>','Fq').replace('Fq2T<Fq >','Fq2').replace('Fp<FqParams, false>','FqC').replace('Fp<FrParams, true>','Fr')
N=11
tot=0
for r in rows[:30]:
    ms=int(r['TotalDurationNs'])/1e6
    k=short(r['Name'])
    if 'level' in k or 'fixed_base' in k or 'csc' in k or 'calib' in k: continue
    tot+=ms/N
    print('%-55s %6s %9.2f ms/proof'%(k[:55], r['Calls'], ms/N))
print('sum %.1f'%tot)
PY
