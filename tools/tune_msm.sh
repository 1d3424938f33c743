# tuning aid: bench.py at several sizes under environment overrides
# (FK_MSM_C_SMALL, FK_MSM_C_DELTA, FK_MSM_CAP_SIGMA, FK_MSM_RED_L), e.g.  FK_MSM_C_DELTA=4 bash tools/tune_msm.sh "24 25"
sizes=${1:-"20 22 24 25"}
for L in $sizes; do
  r=$(timeout 300 python bench.py --log2n $L --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],2), {k:round(v,2) for k,v in j['kernel_ms_per_step'].items() if 'GB' not in k})")
  echo "2^$L -> $r"
done
