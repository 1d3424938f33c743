#!/usr/bin/env python3
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fawkes_crypto_amd as fk
ctx = fk.Context(0)
for lg in (20, 25):
    n = 1 << lg
    d = ctx.dev_alloc(n * 32)
    ctx.gen_scalars_dev(d, n, 5, 0)
    ctx.ntt_dev(d, lg); ctx.sync()
    t = time.time()
    for _ in range(5): ctx.ntt_dev(d, lg)
    ctx.sync(); dt = (time.time() - t) / 5
    print('FK_NTT_THREADS=%s ntt 2^%d: %.3f ms' % (os.environ.get('FK_NTT_THREADS', 'default'), lg, dt * 1e3), flush=True)
    ctx.dev_free(d)
