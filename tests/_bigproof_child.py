"""Child of tests/test_gpu_big_oracle.py::test_whole_proof_2p21_sorts_first_default_levels: proves a 100-transaction rollup-style system
(1 927 201 rows, domain 2^21; l and a carry their fixed-base levels by the DEFAULT threshold, b does not) under the environment it was
started with -- FK_PROVE_SORTS_FIRST is read once per process -- and prints the proof bytes of the direct call and of the two-slot
pipeline.  No oracle code runs here; the parent compares with the C oracle's proof."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402  (data loading helpers only; nothing of the oracle)
import fawkes_crypto_amd as fk  # noqa: E402

copies = int(sys.argv[1])
ctx = fk.Context(0)
r1cs, zs = bench.load_rollup_instance()
z = bench.tile_witness(zs, r1cs.num_input, copies)
dr = ctx.load_r1cs(r1cs, copies=copies)
tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
key, vk = ctx.setup(r1cs, copies=copies, **tox)
r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
direct = bytes(ctx.prove_witness(key, dr, z, r, s)).hex()
zp = [ctx.host_alloc(z.shape) for _ in range(2)]
zp[0][:] = z; zp[1][:] = z
tk = ctx.prove_witness_submit(key, dr, zp[0], r, s)
got = []
for i in range(3):
    nxt = ctx.prove_witness_submit(key, dr, zp[(i + 1) & 1], r, s)
    got.append(bytes(ctx.prove_witness_wait(tk)).hex()); tk = nxt
got.append(bytes(ctx.prove_witness_wait(tk)).hex())
assert all(g == direct for g in got), 'pipelined proofs differ from the direct one'
print('PROOF', direct, key.precomputed())
