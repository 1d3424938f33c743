#!/bin/bash
# experiment (round 5): 11 windows of <= 24 bits (2^23 buckets, merged levels) instead of 12 of <= 22 (2^21) at the benchmark size.
# Needs the experiment build with the wider first sort pass:  make EXP=1 EXTRA=-DFK_S2_MAX_HI=2048
mkdir -p gpurun_out
out=gpurun_out/ab_window24.log
: > $out
run() {
  echo "== $*" >> $out
  env FK_LIB_VARIANT=exp "$@" timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-sizes --no-standalone --no-untiled 2>>$out.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k: d.get(k) for k in ('value','ms_per_step','latency_ms_per_proof','device_resident_ms_per_step','proof_sha256','msm_fixed_base_levels')}))" >> $out
}
run FK_MSM_LB=11
run FK_MSM_C_MAX=24 FK_MSM_PRE_DC=5
run FK_MSM_LB=11
run FK_MSM_C_MAX=24 FK_MSM_PRE_DC=5
cat $out; tail -5 $out.err
