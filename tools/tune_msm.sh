# tuning aid: bench.py at several sizes under environment overrides (FK_MSM_C_SMALL, FK_MSM_C_DELTA, FK_MSM_CAP_SIGMA)
run() { # sizes, env...
  sizes=$1; shift
  for L in $sizes; do
    r=$(env "$@" timeout 300 python bench.py --log2n $L --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print(round(j['ms_per_step'],2), {k:round(v,2) for k,v in j['kernel_ms_per_step'].items() if 'GB' not in k})")
    echo "$* L=$L -> $r"
  done
}
if [ -n "$TUNE_SET" ]; then
  run "24 25" FK_MSM_C_DELTA=3
  run "24 25" FK_MSM_C_DELTA=4
  run "20 22" FK_MSM_C_SMALL=18
  run "20 22" FK_MSM_C_SMALL=16
else
  run "20 22 24 25" FK_X=0
fi
