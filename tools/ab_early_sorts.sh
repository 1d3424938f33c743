#!/bin/bash
# experiment (round 5): the next proof's witness sorts queued before the wait for H's accumulation (each lane orders them behind its own work)
# instead of with the rest of the early front.  Needs the experiment build (make EXP=1).
mkdir -p gpurun_out
out=gpurun_out/ab_early_sorts.log
: > $out
for rep in 1 2; do for v in 0 1; do
  echo "== FK_PROVE_EARLY_SORTS=$v (rep $rep)" >> $out
  FK_LIB_VARIANT=exp FK_PROVE_EARLY_SORTS=$v timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-sizes --no-standalone --no-untiled 2>>$out.err | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k: d.get(k) for k in ('ms_per_step','device_resident_ms_per_step','latency_ms_per_proof','proof_sha256')}), 'tiled', (d.get('tiled') or {}).get('ms_per_step'))" >> $out
done; done
cat $out; grep -v amdgpu.ids $out.err | tail -5
