"""Host-side pieces of bench.py that need no GPU: the workload's dimensions from the data fixture (what the preflight sizes its collectives with),
the roofline block's two readings of "the kernel's duration", and the comparison of timed proofs with the committed oracle digests."""
import argparse
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _args(**kw):
    d = dict(workload='rollup1024', copies=1741, log2n=25)
    d.update(kw)
    return argparse.Namespace(**d)


def test_workload_dims_of_the_benchmark_and_of_the_synthetic_family():
    import bench
    nv, m = bench.workload_dims(_args())
    assert m == 1 << 25 and nv == 3483 + 33597818            # BENCH_r05.json: num_input 3483, num_aux 33 597 818
    nv, m = bench.workload_dims(_args(copies=217))
    assert m == 1 << 22
    assert bench.workload_dims(_args(workload='synthetic', log2n=14)) == (1 << 14, 1 << 14)


def test_roofline_block_union_and_per_launch_readings():
    import bench
    # four launches of 30 ms each that ran side by side in pairs: union 60 ms, sum 120 ms; 1e9 units of 96 bytes
    b = bench.hbm_block('k', 1e9, 96, 60.0, 120.0, 4, 'scalar_mul')
    assert b['bound'] == 'hbm' and b['peak'] == 8000.0 and b['unit'] == 'GB/s'
    assert abs(b['achieved'] - 96e9 / 0.060 / 1e9) < 1e-6 and abs(b['frac'] - b['achieved'] / 8000.0) < 1e-12
    assert abs(b['achieved_per_launch'] - 96e9 / 0.120 / 1e9) < 1e-6 and b['frac_per_launch'] == b['achieved_per_launch'] / 8000.0
    assert b['union_ms_per_launch'] == 15.0 and b['avg_launch_ms'] == 30.0 and b['launches'] == 4
    assert b['algorithmic_bytes_per_scalar_mul'] == 96 and b['traffic'] is None and b['traffic_ratio'] is None
    z = bench.hbm_block('k', 0, 96, 0.0, 0.0, 0, 'scalar_mul')
    assert z['achieved'] == 0.0 and z['frac_per_launch'] == 0.0


def test_timed_proofs_are_compared_with_the_committed_oracle_bytes():
    import bench
    path = os.path.join(ROOT, 'tests', 'golden', 'fullsize_digests.json')
    doc = json.load(open(path))
    e = doc['entries']['rollup1741']
    assert e['rows'] == 33552553 and e['log2_domain'] == 25 and all(e['gpu_equal']) and all(e['pairing']) and len(e['proofs']) == 2
    assert all(len(bytes.fromhex(p)) == 256 for v in doc['entries'].values() for p in v['proofs'])
    assert set(doc['entries']) >= {'rollup1741', 'eddsa4096', 'rollup1024', 'rollup1853', 'synthetic2p27'} and doc['entries']['rollup1853']['log2_domain'] == 26
    assert doc['entries']['synthetic2p27']['rows'] == 1 << 27 and all(doc['entries']['synthetic2p27']['gpu_equal'])
    inst_zs = bench.load_rollup_instance()[1]
    import hashlib
    same_set = hashlib.sha256(np.ascontiguousarray(inst_zs).tobytes()).hexdigest() == e['witness_set_sha256']
    want = [bytes.fromhex(p) for p in e['proofs']]
    got = bench.check_digest(1741, True, inst_zs, want)
    if same_set:           # the generated 32-witness set is present (build() makes it): the entry applies
        assert got['applies'] is True and got['equal'] is True and got['proofs_compared'] == 2
        bad = [want[0], bytes(255) + b'\x01']
        with pytest.raises(AssertionError, match='differ from the oracle'):
            bench.check_digest(1741, True, inst_zs, bad)
    else:
        assert got['applies'] is False
    assert bench.check_digest(5, True, inst_zs, want) is None          # no entry for that size
    assert bench.check_digest(None, False, None, want) is None         # synthetic workload
    other = bench.check_digest(1741, True, inst_zs[:3], want)           # another witness set: the entry does not apply, and says why
    assert other['applies'] is False and 'another set' in other['why']
