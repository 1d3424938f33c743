"""Every tuning / schedule switch of the library must leave the proof bytes unchanged: the alternative code paths (canonical
instead of lazily reduced arithmetic, the scatter kernel shapes, the schedules, lane counts, fixed-base levels at small sizes)
are run on a small tiled system in subprocesses -- the switches are read once per process -- and compared with the default
path, which tests/test_gpu_tiled.py pins against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWITCHES = [
    {'FK_MSM_LAZY': '0'}, {'FK_NTT_LAZY': '0'},
    {'FK_PROVE_SORTS_FIRST': '1'}, {'FK_PROVE_SORTS_FIRST': '1', 'FK_PROVE_ACC_AFTER_NTT': '0'},
    {'FK_PROVE_SORTS_FIRST': '1', 'FK_PROVE_SPMV_AFTER_SORTS': '1'}, {'FK_PROVE_SORTS_FIRST': '0'},
    {'FK_PROVE_Z_EARLY': '0'}, {'FK_PROVE_WITNESS_FIRST': '0'},
    {'FK_MSM_SORT_NT1': '0'}, {'FK_MSM_SORT_NT1': '256', 'FK_MSM_SORT_NT2': '256'}, {'FK_MSM_SORT_NT1': '512', 'FK_MSM_SORT_NT2': '1024'},
    {'FK_MSM_LB': '12'}, {'FK_MSM_LB': '11', 'FK_MSM_C_SMALL': '20'},
    {'FK_MSM_PRE_MIN_LOG2': '8'}, {'FK_MSM_PRE_MIN_LOG2': '8', 'FK_PROVE_SORTS_FIRST': '1'},
    {'FK_MSM_PRE_MIN_LOG2': '8', 'FK_PROVE_SORTS_FIRST': '1', 'FK_MSM_H_PRIO': '1', 'FK_MSM_UNDER_NT1': '512', 'FK_MSM_UNDER_NT2': '512'},
    {'FK_MSM_LANES': '1'}, {'FK_MSM_LANES': '2'}, {'FK_MSM_LIMB29': '1'}, {'FK_MSM_CU_SPLIT': '1'}, {'FK_MSM_SORT_ALONE': '1'},
    {'FK_MSM_PRECOMP': '0'}, {'FK_MSM_RED_HIER': '1'}, {'FK_MSM_RED_HIER': '1', 'FK_MSM_PRE_MIN_LOG2': '8'}, {'FK_UPLOAD_DEFER': '0'},
    {'FK_PROVE_SORTS_FIRST': '1', 'FK_PROVE_G2_FIRST': '1'}, {'FK_MSM_RED_L_G2': '16'}, {'FK_PROVE_SORTS_FIRST': '1', 'FK_PROVE_H_SORT_FIRST': '1'},
]


def _child(env):
    e = dict(os.environ)
    for k in list(e):
        if k.startswith(('FK_MSM_', 'FK_NTT_', 'FK_PROVE_')): del e[k]
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_switch_child.py')], env=e, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (env, out.stderr[-2000:])
    line = [l for l in out.stdout.splitlines() if l.startswith('PROOF ')]
    assert len(line) == 1, out.stdout[-1000:]
    return line[0].split(' ', 2)


@pytest.fixture(scope='module')
def reference():
    return _child({})[1]


@pytest.mark.parametrize('env', SWITCHES, ids=lambda e: ' '.join('%s=%s' % kv for kv in e.items()))
def test_switch_leaves_the_proof_unchanged(env, reference):
    _, proof, levels = _child(env)
    assert proof == reference
    if 'FK_MSM_PRE_MIN_LOG2' in env:
        assert any(v for v in eval(levels).values()), 'the fixed-base levels were expected to be in use: ' + levels
