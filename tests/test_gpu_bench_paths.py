"""bench.py end to end at a small size (subprocesses): the default single-GPU path with its cpu_baseline leg, and the
multi-GPU code path rehearsed with ONE rank over real RCCL (FK_BENCH_REHEARSE=1: process group 'nccl', all_to_all_single /
all_gather on device tensors, distributed quotient and balanced schedule)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args):
    env = dict(os.environ, **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1', *args],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    assert out.stdout.strip().splitlines()[-1] == lines[0], 'the JSON line must be the last line on stdout: ' + out.stdout[-500:]
    return json.loads(lines[0])


def _check_line(j):
    assert j['unit'] == 'proofs/s' and j['n_gpus'] == 1 and j['value'] > 0 and j['proof_verified_by_pairing_check'] is True
    assert j['roofline']['bound'] == 'hbm' and 0 < j['roofline']['frac'] < 1
    rv = j['roofline_valu']
    assert rv['mixed_additions'] > 0 and 0 < rv['frac'] < 1 and rv['peak'] > rv['multiplier_alone'] > 0
    cb = j['cpu_baseline']
    assert cb['kind'] == 'port' and 1 <= cb['cores'] <= (os.cpu_count() or 1) and cb['value'] > 0
    assert cb['single_thread']['cores'] == 1 and cb['single_thread']['value'] > 0
    assert j['device_resident_ms_per_step'] > 0
    assert j['config']['workload'] and 'model' not in j['config']


def test_bench_default_workload_small():
    """the default workload (tiled rollup-style transactions from the committed fixture, host-witness pipeline) at 6 copies"""
    j = _run({}, '--copies', '6', '--cpu-copies', '2')
    _check_line(j)
    assert 'rollup-style transactions' in j['config']['workload'] and j['config']['num_input'] == 1 + 6 * 2
    assert j['config']['witness_bytes_per_proof'] == (j['config']['num_input'] + j['config']['num_aux']) * 32


def test_bench_synthetic_workload_small():
    j = _run({}, '--workload', 'synthetic', '--log2n', '14', '--cpu-log2n', '12')
    _check_line(j)


@pytest.mark.parametrize('dist_q', ['1', '0'])
def test_bench_multi_gpu_path_rehearsed_over_rccl(dist_q):
    env = {'FK_BENCH_REHEARSE': '1', 'FK_DIST_QUOTIENT': dist_q, 'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0',
           'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29541'}
    j = _run(env, '--workload', 'synthetic', '--log2n', '14')
    assert j['proof_verified_by_pairing_check'] is True
    assert ('distributed-quotient' in j['config']['parallelism']) == (dist_q == '1')


def test_bench_default_workload_multi_gpu_path_rehearsed():
    env = {'FK_BENCH_REHEARSE': '1', 'FK_DIST_QUOTIENT': '1', 'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0',
           'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29542'}
    j = _run(env, '--copies', '5')
    assert j['proof_verified_by_pairing_check'] is True and 'distributed-quotient' in j['config']['parallelism']


def test_bench_long_linear_combinations():
    j = _run({}, '--workload', 'synthetic', '--log2n', '14', '--lc-terms', '8', '--cpu-log2n', '12')
    assert j['proof_verified_by_pairing_check'] is True and j['config']['nnz'][0] > 4 * (1 << 14) * 0.4
