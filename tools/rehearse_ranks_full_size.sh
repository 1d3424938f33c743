#!/bin/bash
# bench.py --gpus N at the FULL benchmark size with every rank-process on ONE GPU over gloo (a rehearsal of the N-rank path through the shared
# Parameters image, not a measurement): rc, proof bytes against the committed oracle digests, the preflight and load blocks.
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/rehearse; mkdir -p $O
for n in ${RANKS:-2 4 8}; do
  FK_BENCH_SAME_DEVICE=1 timeout 1500 python3 bench.py --gpus $n --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --no-replicas > $O/n$n.log 2>&1
  echo "ranks=$n rc=$?"
  python3 - $O/n$n.log <<'PY'
import json,sys
lines=[l for l in open(sys.argv[1]) if l.startswith('{"metric"')]
if not lines:
    print('  NO LINE:', open(sys.argv[1]).read()[-600:].replace('\n',' | ')); sys.exit(0)
j=json.loads(lines[-1])
print('  n_gpus %d  ms_per_step %.1f  matrix_form %r  sha %s  digest %s' % (j['n_gpus'], j['ms_per_step'], j['config']['matrix_form'], [x[:8] for x in j['proof_sha256']], j.get('oracle_digest_check')))
print('  load:', {k: j['load'].get(k) for k in ('image_bytes','image_file','decode_seconds','key_read_checked_seconds','load_parameters_seconds_rank0','one_rank_at_a_time')})
pf=j['preflight']; print('  preflight:', pf['decision'], 'control', pf['control_plane'], 'library ok', (pf.get('library') or {}).get('ok'), 'seconds', pf['seconds'])
sp=j.get('single_process_multi_gpu'); print('  one-call leg:', {k: sp.get(k) for k in ('ms_per_step','matrix_form','transport','error')} if sp else None)
print('  legs:', j['legs'])
PY
done
