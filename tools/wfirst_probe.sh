# Experiment (GPU box): witness multiplications begun before / after the quotient, by domain size (synthetic workload)
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/wfirst; mkdir -p $O
for L in 20 22 23 24; do for W in 0 1; do
  FK_PROVE_WITNESS_FIRST=$W python3 bench.py --workload synthetic --log2n $L --steps 10 --warmup 3 --no-cpu-baseline > $O/l${L}_w${W}.log 2>&1
  echo "log2n=$L witness_first=$W: $(python3 -c "
import json
l=[x for x in open('$O/l${L}_w${W}.log') if x.startswith('{')]
j=json.loads(l[0]) if l else {}
print(j.get('ms_per_step'), j.get('device_resident_ms_per_step'))
")"
done; done
for C in 64 256; do for W in 0 1; do
  FK_PROVE_WITNESS_FIRST=$W python3 bench.py --copies $C --steps 10 --warmup 3 --no-cpu-baseline > $O/c${C}_w${W}.log 2>&1
  echo "rollup copies=$C witness_first=$W: $(python3 -c "
import json
l=[x for x in open('$O/c${C}_w${W}.log') if x.startswith('{')]
j=json.loads(l[0]) if l else {}
print(j.get('ms_per_step'), j.get('device_resident_ms_per_step'), j.get('config',{}).get('log2_constraints'))
")"
done; done
