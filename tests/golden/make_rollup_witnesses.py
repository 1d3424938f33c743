#!/usr/bin/env python3
"""Writes tests/golden/_generated/rollup_tx_witnesses.npy: N (default 32) DISTINCT satisfying witnesses of the rollup-style
transaction whose constraint system is the committed fixture rollup_tx_instance.npz -- the input data of
`bench.py --workload rollup1024` and tests/test_gpu_fullsize.py.

Why a generated file: a real 1024-transaction batch holds 1024 different transactions, and repeated witness values change
the work of a Pippenger MSM (equal scalars meet in the same buckets: with the fixture's 3 witnesses tiled 341 times each a
proof takes 274 ms, with 32 or more distinct ones 216 ms, profiles/r02_distinct_witness_probe.log).  32 witnesses are
20 MB of incompressible field elements -- too large to commit, so the file is produced by `__graft_entry__.build()` in the
build container (it is git-ignored but travels to the GPU box with the snapshot, like the built .so files).  The circuit
builder (oracle/fawkes_circuit.py) is test infrastructure and is only ever run HERE, never by bench.py, which loads the
data file -- or falls back to the fixture's three witnesses, and says so in its JSON line, when the file is absent.

Witness k (k >= 3; 0..2 are the fixture's) = tests/golden/make_rollup_tx_fixture.py: tx(k).  Run: python tests/golden/make_rollup_witnesses.py [N]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, '..', '..')
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests'), HERE):
    sys.path.insert(0, p)
OUT = os.path.join(HERE, '_generated', 'rollup_tx_witnesses.npy')


def one(k):
    import fixtures as fx
    import make_rollup_tx_fixture as mf
    t = mf.tx(k)
    assert t.satisfied()
    return fx.witness_mont(t.z_in, t.z_aux)


def main(n=32):
    if os.path.exists(OUT) and np.load(OUT, mmap_mode='r').shape[0] >= n:
        return OUT
    import multiprocessing as mp
    workers = max(1, min(os.cpu_count() or 1, n))
    with mp.get_context('fork').Pool(workers) as pool:
        zs = pool.map(one, range(n))
    fix = np.load(os.path.join(HERE, 'rollup_tx_instance.npz'))['z']
    for k in range(min(3, n)):
        assert np.array_equal(zs[k], fix[k]), 'generator drifted from the committed fixture'
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.save(OUT, np.stack(zs))
    return OUT


if __name__ == '__main__':
    print(main(int(sys.argv[1]) if len(sys.argv) > 1 else 32))
