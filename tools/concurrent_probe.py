#!/usr/bin/env python3
"""Throughput of T host threads, each with its own context (streams + scratch), proving concurrently on ONE GPU from the same
resident key and constraint system -- how a proving service would keep the device busy across the gaps of a single proof
(H's sort right after the quotient, the tail of the last multiplication).  Uses bench.py's workload builder."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import fawkes_crypto_amd as fk
log2n = int(os.environ.get('LOG2N', '25')); T = int(os.environ.get('THREADS', '2')); K = int(os.environ.get('PROOFS', '4'))
ctxs = [fk.Context(0) for _ in range(T)]
c0 = ctxs[0]
r1cs, z = bench.build_workload(c0, fk, log2n, seed=1)
tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
key, vk = c0.setup(r1cs, **tox)
dr = c0.load_r1cs(r1cs)
r, s = bench.mont(11), bench.mont(22)
d_z = [c.dev_alloc(z.nbytes) for c in ctxs]
for c, d in zip(ctxs, d_z):
    c.upload(d, z)
# a key / constraint system made through one context is plain device memory: any context on the device can prove from it
def prove(c, d):
    return c.prove_witness_dev(key, dr, d, r, s).tobytes()
want = prove(c0, d_z[0])
for c, d in zip(ctxs, d_z):
    assert prove(c, d) == want          # warm every context's scratch
t0 = time.perf_counter()
for _ in range(K):
    assert prove(c0, d_z[0]) == want
t1 = (time.perf_counter() - t0) / K
res = [None] * T
def work(i):
    res[i] = [prove(ctxs[i], d_z[i]) for _ in range(K)]
th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
t0 = time.perf_counter()
for t in th: t.start()
for t in th: t.join()
dt = time.perf_counter() - t0
assert all(p == want for rr in res for p in rr)
print('2^%d: one context %.2f ms per proof (%.2f proofs/s); %d contexts concurrently: %d proofs in %.1f ms = %.2f ms per proof (%.2f proofs/s), all identical' %
      (log2n, t1 * 1e3, 1 / t1, T, T * K, dt * 1e3, dt / (T * K) * 1e3, T * K / dt), flush=True)
