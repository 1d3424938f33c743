#!/usr/bin/env python3
"""perf probe (G1 MSM + NTT at 2^16..2^22) and, with argv[1]=='g2', a G2 MSM debug run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
T0 = time.time()
def log(*a):
    print('[%7.2fs]' % (time.time() - T0), *a, flush=True)
import numpy as np
import fawkes_crypto_amd as fk
ctx = fk.Context(0)
log('context')
if len(sys.argv) > 1 and sys.argv[1] == 'g2':
    import c_oracle as co
    from helpers import rand_fr_mont, g2_bases
    rng = np.random.default_rng(1)
    for n in (1, 50):
        bases, sc = g2_bases(n, 3), rand_fr_mont(rng, n)
        log('calling msm_g2 n=%d' % n)
        o = ctx.msm_g2(bases, sc)
        log('msm_g2 n=%d' % n, o.tobytes() == co.msm_g2(bases, sc).tobytes())
    sys.exit(0)
for lg in (16, 20, 22):
    n = 1 << lg
    d = ctx.dev_alloc(n * 32)
    ctx.gen_scalars_dev(d, n, 5, 0)
    ctx.ntt_dev(d, lg); ctx.sync()
    t = time.time()
    for _ in range(5): ctx.ntt_dev(d, lg)
    ctx.sync(); dt = (time.time() - t) / 5
    log('ntt 2^%d: %.3f ms  (%.1f GB/s algorithmic at 64 B/elt/transform)' % (lg, dt * 1e3, n * 64 / dt / 1e9))
    ctx.dev_free(d)
for lg in (16, 20, 22):
    n = 1 << lg
    db, ds = ctx.dev_alloc(n * 64), ctx.dev_alloc(n * 32)
    t = time.time(); ctx.gen_points_g1_dev(db, n, 7); log('gen_points_g1 2^%d %.3fs' % (lg, time.time() - t))
    for kind in (0, 1):
        ctx.gen_scalars_dev(ds, n, 11, kind)
        ctx.msm_g1_dev(db, ds, n)
        ctx.stats_reset()
        t = time.time()
        reps = 3
        for _ in range(reps): out = ctx.msm_g1_dev(db, ds, n)
        dt = (time.time() - t) / reps
        st = ctx.stats()
        log('msm_g1 2^%d kind=%d: %.2f ms  (%.1f M scalar-muls/s; accumulate kernel %.2f ms/launch)' % (
            lg, kind, dt * 1e3, n / dt / 1e6, st['acc_g1']['ms'] / max(st['acc_g1']['launches'], 1)))
    ctx.dev_free(db); ctx.dev_free(ds)
log('done')
