"""Child of tests/test_gpu_switches.py: proves a small tiled system (4 rollup-style transactions of the committed fixture) under the
environment it was started with and prints the proof bytes.  The switches are read once per process, hence the subprocess."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402  (data loading helpers only; nothing of the oracle)
import fawkes_crypto_amd as fk  # noqa: E402

copies = 4
ctx = fk.Context(0)
r1cs, zs = bench.load_rollup_instance()
z = bench.tile_witness(zs[:3], r1cs.num_input, copies)
dr = ctx.load_r1cs(r1cs, copies=copies)
tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
key, vk = ctx.setup(r1cs, copies=copies, **tox)
r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
out = []
for _ in range(2):          # twice: the second proof runs on warm lane buffers (no growth, different queueing)
    out.append(bytes(ctx.prove_witness(key, dr, z, r, s)).hex())
assert out[0] == out[1], 'two proofs of the same witness differ'
print('PROOF', out[0], key.precomputed(), 'roctx=%d' % fk.load_library().fk_roctx_active())
