#!/usr/bin/env python3
"""Step-by-step GPU probe with flushed timestamps (diagnostics; run under a short timeout)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
T0 = time.time()
def log(*a):
    print('[%7.2fs]' % (time.time() - T0), *a, flush=True)
import numpy as np
log('numpy imported')
import fawkes_crypto_amd as fk
log('package imported')
ctx = fk.Context(0)
log('context created')
import c_oracle as co, bn254_ref as ref, fixtures as fx
from helpers import rand_fr_mont, g1_bases, g2_bases
co.lib(); log('oracle loaded')
rng = np.random.default_rng(1)
a, b = rand_fr_mont(rng, 1000), rand_fr_mont(rng, 1000)
log('inputs made')
o = ctx.fr_mul_batch(a, b); log('fr_mul_batch done', np.array_equal(o, co.fe_mul_batch(co.FR, a, b)))
for k in (0, 1, 4, 10, 12):
    v = rand_fr_mont(rng, 1 << k)
    t = time.time(); o = ctx.ntt(v); dt = time.time() - t
    log('ntt 2^%d' % k, np.array_equal(o, co.fr_ntt(v)), '%.3fs' % dt)
for n in (1, 12, 100, 1000):
    bases, sc = g1_bases(n, 3), rand_fr_mont(rng, n)
    t = time.time(); o = ctx.msm_g1(bases, sc); dt = time.time() - t
    log('msm_g1 n=%d' % n, o.tobytes() == co.msm_g1(bases, sc).tobytes(), '%.3fs' % dt)
for n in (1, 50):
    bases, sc = g2_bases(n, 3), rand_fr_mont(rng, n)
    t = time.time(); o = ctx.msm_g2(bases, sc); dt = time.time() - t
    log('msm_g2 n=%d' % n, o.tobytes() == co.msm_g2(bases, sc).tobytes(), '%.3fs' % dt)
log('probe done')
