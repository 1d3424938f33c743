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
copies = int(os.environ.get('COPIES', '64'))
cs = fc.eddsa_circuit(123456789, 987654321, 555)[0]
one = fx.r1cs_to_csr(cs.r1cs())
batch = fx.tile_r1cs(one, copies)
z = fx.tile_witness([cs.z_in] * copies, [cs.z_aux] * copies)
ctx = fk.Context(0)
dr = ctx.load_r1cs(r1cs_product(batch))
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
print('copies %d: rows %d (m = 2^%d), nnz %d, SpMV %.3f ms = %.1f G terms/s, %.0f GB/s at 40 B/term' % (copies, rows, m.bit_length() - 1, nnz, dt * 1e3, nnz / dt / 1e9, nnz * 40 / dt / 1e9), flush=True)
want = fk.api.synthesize(r1cs_product(batch), z)
for k in range(3):
    got = ctx.download(d[k], rows * 32, np.uint64).reshape(-1, 4)
    assert np.array_equal(got, want[k]), k
print('matches the host synthesis')
