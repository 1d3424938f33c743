#!/usr/bin/env python3
"""Round 6, "memory-bound work on compute units of its own": how fast does the evaluation of a, b, c (the front's latency-bound half) run when it may
use ONLY the units a CU mask sets aside (FK_CU_SPLIT=4, mode 2: four of the 32 units of every XCD = 32 of 256), alone and while ANOTHER process proves on
the other 224?  Guest role (this script): an experiment-build context whose every stream carries the set-aside mask (FK_CU_SPLIT_ALL=1) evaluates the
explicit 1741-transaction system in a loop and prints every evaluation's time.  The host role is bench.py with FK_CU_SPLIT=4 FK_CU_SPLIT_MODE=2 (its
accumulations masked to the other units), started by tools/cu_guest_probe.sh once the guest is ready."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import fawkes_crypto_amd as fk  # noqa: E402

COPIES = int(os.environ.get('COPIES', 1741))
SECONDS = float(os.environ.get('GUEST_SECONDS', 120))
READY = os.environ.get('GUEST_READY_FILE', '/tmp/fk_guest_ready')
ctx = fk.Context(0)
r1cs, zs = bench.load_rollup_instance()
z = bench.tile_witness(zs, r1cs.num_input, COPIES)
d_z = ctx.dev_alloc(z.nbytes); ctx.upload(d_z, z)
rows = COPIES * r1cs.num_gates + 1 + COPIES * (r1cs.num_input - 1)
m = 1 << max(rows - 1, 1).bit_length()
d = [ctx.dev_alloc(m * 32) for _ in range(3)]
u_in, u_aux, u_mats, u_table = bench.materialise_rollup(COPIES)
du = ctx.load_r1cs_coded(u_in, u_aux, u_mats, u_table)
del u_mats
ctx.r1cs_eval_dev(du, d_z, *d); ctx.sync()
open(READY, 'w').write('ready')
print('guest ready: masks %s' % {k: os.environ.get(k) for k in ('FK_CU_SPLIT', 'FK_CU_SPLIT_MODE', 'FK_CU_SPLIT_ALL')}, flush=True)
t_end = time.time() + SECONDS
times = []
while time.time() < t_end:
    t = time.perf_counter()
    ctx.r1cs_eval_dev(du, d_z, *d); ctx.sync()
    times.append((time.time(), (time.perf_counter() - t) * 1e3))
ts = np.array([x[1] for x in times])
print('guest: %d evaluations, ms min %.1f  median %.1f  mean %.1f  max %.1f' % (len(ts), ts.min(), np.median(ts), ts.mean(), ts.max()))
# by time: first 10 s (likely alone) against the rest
t0 = times[0][0]
for lo, hi in ((0, 8), (8, 1e9)):
    sel = np.array([x[1] for x in times if lo <= x[0] - t0 < hi])
    if len(sel):
        print('  %4.0f s .. : %d evaluations, median %.1f ms, mean %.1f ms' % (lo, len(sel), np.median(sel), sel.mean()))
