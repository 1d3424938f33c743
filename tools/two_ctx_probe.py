#!/usr/bin/env python3
"""Experiment (GPU box): how much would overlapping CONSECUTIVE proofs buy?  Two contexts on one GPU share one key and one
resident constraint system; two host threads prove in a loop (ctypes releases the GIL).  Aggregate proofs/s against one
context alone says what a proof pipeline two deep could reach at best."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import fawkes_crypto_amd as fk

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
torch.cuda.set_device(0)
ctx = [fk.Context(0), fk.Context(0)]
r1cs, zs = bench.load_rollup_instance()
num_input = 1 + copies * (r1cs.num_input - 1)
z = bench.tile_witness(zs, r1cs.num_input, copies)
dr = ctx[0].load_r1cs(r1cs, copies=copies)
tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
key, vk = ctx[0].setup(r1cs, copies=copies, **tox)
r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
d_z = [torch.from_numpy(z.view(np.uint8).reshape(-1)).cuda() for _ in range(2)]
torch.cuda.synchronize()
ref = ctx[0].prove_witness_dev(key, dr, d_z[0].data_ptr(), r, s)
for c in (0, 1):
    for _ in range(2):
        p = ctx[c].prove_witness_dev(key, dr, d_z[c].data_ptr(), r, s)
        assert bytes(p) == bytes(ref), 'context %d: different proof' % c
t0 = time.time()
for _ in range(K):
    ctx[0].prove_witness_dev(key, dr, d_z[0].data_ptr(), r, s)
one = (time.time() - t0) / K
out = [None, None]
def loop(c):
    for _ in range(K):
        out[c] = ctx[c].prove_witness_dev(key, dr, d_z[c].data_ptr(), r, s)
th = [threading.Thread(target=loop, args=(c,)) for c in (0, 1)]
t0 = time.time()
for t in th: t.start()
for t in th: t.join()
two = (time.time() - t0) / (2 * K)
ok = all(bytes(o) == bytes(ref) for o in out)
print('copies=%d  one context: %.2f ms per proof;  two contexts interleaved: %.2f ms per proof (aggregate), proofs equal: %s' % (copies, one * 1e3, two * 1e3, ok))
