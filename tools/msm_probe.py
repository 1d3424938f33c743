#!/usr/bin/env python3
"""dense 2^25 G1 MSM at window size $FKC (tuning aid)"""
import os, time, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fawkes_crypto_amd as fk
ctx = fk.Context(0); n = (1 << int(os.environ.get("FKLOG", "25"))) - int(os.environ.get("FKMINUS", "0"))
db, ds = ctx.dev_alloc(n * 64), ctx.dev_alloc(n * 32)
ctx.gen_points_g1_dev(db, n, 7); ctx.gen_scalars_dev(ds, n, 11, int(os.environ.get("FKKIND", "0")))
c = int(os.environ.get("FKC", "0")); ctx.set_window_bits(c)
ctx.msm_g1_dev(db, ds, n); ctx.stats_reset()
t = time.time()
for _ in range(3): out = ctx.msm_g1_dev(db, ds, n)
dt = (time.time() - t) / 3; st = ctx.stats()
print("c=%d 2^%d G1 MSM: %.2f ms total, accumulate %.1f ms, result %s" % (c, n.bit_length() - 1, dt * 1e3, st["acc_g1"]["ms"] / 3, out[:8].tobytes().hex()), flush=True)
