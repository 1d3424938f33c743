#!/usr/bin/env python3
"""Evaluation of a, b, c of the benchmark's system with every term explicit in HBM (no tiling shortcut: what a circuit loaded
from a Parameters file looks like), alone on the GPU: 10 repetitions; beside it the tiled form (GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import fawkes_crypto_amd as fk  # noqa: E402

COPIES = int(os.environ.get('COPIES', 1024))
SEQ = os.environ.get('FK_LIB_VARIANT') == 'exp'
if SEQ:
    os.environ['FK_SPMV_SEQ'] = '0'
ctx = fk.Context(0)
r1cs, zs = bench.load_rollup_instance()
z = bench.tile_witness(zs, r1cs.num_input, COPIES)
d_z = ctx.dev_alloc(z.nbytes); ctx.upload(d_z, z)
rows = COPIES * r1cs.num_gates + 1 + COPIES * (r1cs.num_input - 1)
m = 1 << max(rows - 1, 1).bit_length()
d = [ctx.dev_alloc(m * 32) for _ in range(3)]
e = [ctx.dev_alloc(m * 32) for _ in range(3)]


def timed(dr, out):
    ctx.r1cs_eval_dev(dr, d_z, *out); ctx.sync()
    t = time.perf_counter()
    for _ in range(10):
        ctx.r1cs_eval_dev(dr, d_z, *out)
    ctx.sync()
    return (time.perf_counter() - t) / 10


dt = ctx.load_r1cs(r1cs, copies=COPIES)
t_t = timed(dt, d)
nnz = sum(dt.info()['nnz'])
print('tiled    %.3f ms  %.1f G terms/s' % (t_t * 1e3, nnz / t_t / 1e9), flush=True)
t0 = time.perf_counter()
u_in, u_aux, u_mats, u_table = bench.materialise_rollup(COPIES)
if SEQ:
    os.environ['FK_SPMV_SEQ'] = '1'         # (read at load: the permuted layout's pointers are built beside the class lists)
du = ctx.load_r1cs_coded(u_in, u_aux, u_mats, u_table)
if SEQ:
    os.environ['FK_SPMV_SEQ'] = '0'
print('untiled system built and loaded in %.1f s' % (time.perf_counter() - t0), flush=True)
t_u = timed(du, e)
print('untiled  %.3f ms  %.1f G terms/s' % (t_u * 1e3, nnz / t_u / 1e9), flush=True)
same = all(np.array_equal(ctx.download(d[k], rows * 32, np.uint64), ctx.download(e[k], rows * 32, np.uint64)) for k in range(3))
print('same a, b, c:', same)
if os.environ.get('FK_LIB_VARIANT') == 'exp':
    # upper bound of a 4-bytes-per-term form (TIMING ONLY: the kernel derives the coefficient index from the column instead of loading it)
    os.environ['FK_SPMV_FAKE4'] = '1'
    print('NOTE: tune() reads FK_SPMV_FAKE4 at every launch in the experiment build')
    t_f = timed(du, e)
    os.environ['FK_SPMV_FAKE4'] = '0'
    os.environ['FK_SPMV_SEQ'] = '1'
    t_s = timed(du, e)
    os.environ['FK_SPMV_SEQ'] = '0'
    print('untiled, terms read where a layout permuted into class-list order would hold them (timing only)  %.3f ms  -> at most %.3f ms to gain' % (t_s * 1e3, (t_u - t_s) * 1e3))
    print('untiled, 4 bytes per term streamed (timing only)  %.3f ms  %.1f G terms/s  -> at most %.3f ms to gain' % (t_f * 1e3, nnz / t_f / 1e9, (t_u - t_f) * 1e3))
