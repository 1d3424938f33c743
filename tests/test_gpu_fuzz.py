"""A short leg of the differential fuzzer (tools/fuzz_parity.py: random multiplications, transforms, quotients and whole proofs
of random -- ragged, degenerate, unsatisfied, tiled, sharded -- constraint systems through randomly chosen entry points, each
result compared with the oracle's bytes).  The long campaigns are run by hand and logged under profiles/."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_leg_is_clean():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_parity.py'), '--seconds', '25', '--seed', '7'],
                       capture_output=True, text=True, timeout=600)
    tail = p.stdout[-3000:] + p.stderr[-3000:]
    assert p.returncode == 0, tail
    summary = [l for l in p.stdout.splitlines() if l.startswith('fuzz:')]
    assert summary and ' 0 failures' in summary[0], tail
    cases = int(summary[0].split()[1])
    assert cases >= 20, tail           # the leg really ran cases of several kinds
