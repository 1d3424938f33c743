#!/usr/bin/env python3
"""Standalone G1 multiplication of 2^LOG points (BASELINE configs[1]: 2^20), repeated: for a kernel trace of ONE small multiplication.
    rocprofv3 --kernel-trace --stats -- python3 tools/msm_small_trace.py [log2n] [reps] [kind]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fawkes_crypto_amd as fk
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
kind = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx = fk.Context(0)
n = 1 << log_n
d_b, d_s = ctx.dev_alloc(n * 64), ctx.dev_alloc(n * 32)
ctx.gen_points_g1_dev(d_b, n, 11); ctx.gen_scalars_dev(d_s, n, 13, kind)
for _ in range(5):
    ctx.msm_g1_dev(d_b, d_s, n)
ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.msm_g1_dev(d_b, d_s, n)
ctx.sync()
print('2^%d G1 multiplication: %.3f ms each' % (log_n, (time.perf_counter() - t0) / reps * 1e3))
