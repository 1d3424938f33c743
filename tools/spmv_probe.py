#!/usr/bin/env python3
"""SpMV (a = Az, b = Bz, c = Cz) timing on a dense-LC constraint system: COPIES eddsa signature checks tiled into one
system (133 matrix terms per gate on average, rows of up to 512 terms).  Tuning aid; uses the oracle-side circuit builder."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import fawkes_circuit as fc, fixtures as fx, bn254_ref as ref
from helpers import r1cs_product
import fawkes_crypto_amd as fk
cs = fc.eddsa_circuit(123456789, 987654321, 555)[0]
one = fx.r1cs_to_csr(cs.r1cs())
z1 = fx.witness_mont(cs.z_in, cs.z_aux)
ctx = fk.Context(0)


def run(copies, tiled):
    ni = one.num_input
    z = np.ascontiguousarray(np.concatenate([z1[:1]] + [z1[1:ni]] * copies + [z1[ni:]] * copies))
    batch = None if tiled else r1cs_product(fx.tile_r1cs(one, copies))
    dr = ctx.load_r1cs(r1cs_product(one), copies=copies) if tiled else ctx.load_r1cs(batch)
    info = dr.info()
    rows = info['rows']; m = 1
    while m < rows: m *= 2
    d = [ctx.dev_alloc(m * 32) for _ in range(3)]
    d_z = ctx.dev_alloc(z.nbytes); ctx.upload(d_z, z)
    ctx.r1cs_eval_dev(dr, d_z, *d); ctx.sync()
    t = time.time()
    for _ in range(10): ctx.r1cs_eval_dev(dr, d_z, *d)
    ctx.sync()
    dt = (time.time() - t) / 10
    nnz = sum(info['nnz'])
    print('copies %d %s: rows %d (m = 2^%d), nnz %d, SpMV %.3f ms = %.1f G terms/s' % (copies, 'tiled' if tiled else 'replicated', rows, m.bit_length() - 1, nnz,
          dt * 1e3, nnz / dt / 1e9), flush=True)
    if not tiled:
        want = fk.api.synthesize(batch, z)
        for k in range(3):
            assert np.array_equal(ctx.download(d[k], rows * 32, np.uint64).reshape(-1, 4), want[k]), k
        print('  matches the host synthesis')
    for p_ in d + [d_z]: ctx.dev_free(p_)
    dr.free()


for spec in os.environ.get('RUNS', '64r,64t,256r,256t,1024t,4096t').split(','):
    run(int(spec[:-1]), spec[-1] == 't')
