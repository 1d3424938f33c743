#!/usr/bin/env python3
"""Where the evaluation of a, b, c of the benchmark's system goes: the rollup instance tiled 1024 times, evaluated whole and
with parts of it blanked (one matrix only; only the rows of one length range), 10 repetitions each (GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import fawkes_crypto_amd as fk  # noqa: E402

COPIES = int(os.environ.get('COPIES', 1024))
ctx = fk.Context(0)
r1cs, zs = bench.load_rollup_instance()
z = bench.tile_witness(zs, r1cs.num_input, COPIES)
d_z = ctx.dev_alloc(z.nbytes); ctx.upload(d_z, z)
rows = COPIES * r1cs.num_gates + 1 + COPIES * (r1cs.num_input - 1)
m = 1 << max(rows - 1, 1).bit_length()
d = [ctx.dev_alloc(m * 32) for _ in range(3)]


def keep(mat, lo, hi):
    """the matrix with only the rows whose length is in [lo, hi)"""
    ptr, col, val = mat
    ln = np.diff(ptr.astype(np.int64))
    sel = (ln >= lo) & (ln < hi)
    nl = np.where(sel, ln, 0)
    nptr = np.concatenate([[0], np.cumsum(nl)]).astype(np.uint64)
    mask = np.repeat(sel, ln)
    return nptr, col[mask], (None if val is None else val[mask])


def run(name, mats):
    sys_ = fk.R1cs(r1cs.num_input, r1cs.num_aux, *mats)
    dr = ctx.load_r1cs(sys_, copies=COPIES)
    nnz = sum(dr.info()['nnz'])
    ctx.r1cs_eval_dev(dr, d_z, *d); ctx.sync()
    t = time.perf_counter()
    for _ in range(10):
        ctx.r1cs_eval_dev(dr, d_z, *d)
    ctx.sync()
    dt = (time.perf_counter() - t) / 10
    print('%-34s nnz %11d  %7.3f ms  %6.1f G terms/s' % (name, nnz, dt * 1e3, nnz / dt / 1e9), flush=True)
    dr.free()


A, B, C = r1cs.mats
E = lambda mt: keep(mt, 1 << 30, 1 << 31)        # noqa: E731  (an empty matrix of the same height)
run('whole system', (A, B, C))
if os.environ.get('ONLY') == 'whole':
    sys.exit(0)
run('A only', (A, E(B), E(C)))
run('B only', (E(A), B, E(C)))
run('C only', (E(A), E(B), C))
run('nothing (rows written as zero)', (E(A), E(B), E(C)))
for lo, hi in ((1, 2), (2, 4), (4, 32), (32, 64), (64, 128), (128, 1025)):
    run('rows of length [%d, %d)' % (lo, hi), tuple(keep(mt, lo, hi) for mt in (A, B, C)))
