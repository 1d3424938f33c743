#!/bin/bash
# A/B of the chunked witness hand-over of fk_prove_r1cs on ONE box: the same bench command with FK_PROVE_CHUNKED_UPLOAD=0 and =1
# (latency_ms_per_proof is the figure; value / tiled must not move).  Output: gpurun_out/ab_chunked.log
mkdir -p gpurun_out
out=gpurun_out/ab_chunked.log
: > $out
for rep in 1; do
  for v in 0 1; do
    echo "== FK_PROVE_CHUNKED_UPLOAD=$v (rep $rep)" >> $out
    FK_PROVE_CHUNKED_UPLOAD=$v timeout 900 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-other-sizes --no-standalone --no-untiled 2>>$out.err | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({k: d.get(k) for k in ('value','ms_per_step','latency_ms_per_proof','device_resident_ms_per_step')}), d.get('tiled',{}).get('ms_per_step') if isinstance(d.get('tiled'),dict) else None)" >> $out
  done
done
cat $out
