# Experiment (GPU box): CU-masked sort streams (FK_MSM_CU_SPLIT) x witness multiplications begun before the quotient
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/cusplit; mkdir -p $O
for S in 0 1 2 3; do for W in 0 1; do
  FK_MSM_CU_SPLIT=$S FK_PROVE_WITNESS_FIRST=$W python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/s${S}_w${W}.log 2>&1
  echo "split=$S witness_first=$W: $(python3 -c "
import json,sys
l=[x for x in open('$O/s${S}_w${W}.log') if x.startswith('{')]
j=json.loads(l[0]) if l else {}
print(j.get('ms_per_step'), j.get('device_resident_ms_per_step'), j.get('kernel_ms_per_step'))
")"
done; done
