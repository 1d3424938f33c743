"""The host-witness pipeline (fk_prove_r1cs_submit / _wait) with DIFFERENT witnesses in flight: with a ticket outstanding the
upload of the next witness is queued by the running proof (deferred upload), so the two slots, the deferral and the event
ordering must never mix witnesses up -- every pipelined proof must equal the proof of the same witness computed directly.
Also: the union-of-intervals kernel statistic (fk_stats_get 5 / 6)."""
import numpy as np
import pytest

import fixtures as fx
from helpers import r1cs_product, TOXIC

pytestmark = pytest.mark.gpu
TOX = {k: fx.mont_fr(v) for k, v in TOXIC.items()}


def _system(ctx, seed, gates=3000, num_input=3, num_aux=2500):
    cs, z, _, _ = fx.fast_r1cs(seed, gates, num_input, num_aux)
    prod = r1cs_product(cs)
    dr = ctx.load_r1cs(prod)
    key, _ = ctx.setup(prod, **TOX)
    return cs, prod, dr, key, z


def test_pipeline_with_distinct_witnesses(ctx):
    cs, prod, dr, key, z0 = _system(ctx, 4242)
    r, s = fx.mont_fr(0x1234567), fx.mont_fr(0x7654321)
    nv = z0.shape[0]
    # distinct "witnesses": the proof of ANY assignment is well defined (an unsatisfying one simply does not verify), so
    # perturbing the aux part gives different inputs that exercise the slots without needing a second solver run
    zs = []
    rng = np.random.default_rng(7)
    for i in range(5):
        z = z0.copy()
        if i:
            idx = rng.integers(cs.num_input, nv, size=50)
            z[idx] = np.array([fx.mont_fr(int(v)) for v in rng.integers(1, 1 << 60, size=50)], dtype=np.uint64).reshape(50, 4)
        zs.append(np.ascontiguousarray(z))
    direct = [ctx.prove_witness(key, dr, z, r, s).tobytes() for z in zs]
    assert len(set(direct)) == len(direct), 'the perturbed witnesses must give different proofs'
    pins = [ctx.host_alloc((nv, 4)) for _ in range(2)]
    got = []
    pins[0][:] = zs[0]
    ticket = ctx.prove_witness_submit(key, dr, pins[0], r, s)
    for k in range(len(zs)):
        nxt = None
        if k + 1 < len(zs):
            pins[(k + 1) & 1][:] = zs[k + 1]                    # its previous occupant's ticket has been waited for
            nxt = ctx.prove_witness_submit(key, dr, pins[(k + 1) & 1], r, s)      # deferred: a ticket is outstanding
        got.append(ctx.prove_witness_wait(ticket).tobytes())
        ticket = nxt
    assert got == direct
    for p in pins:
        ctx.host_free(p)


@pytest.mark.parametrize('sorts_first', ['1', '0'])
def test_early_front_with_distinct_witnesses(sorts_first):
    """the early front -- `_wait(k)` queues proof k + 1's evaluation, gathers and witness sorts out of the other
    slot -- is on by default only from 2^25 on, where every test and the bench used to put the SAME witness into both slots.  Here it
    is forced at a small size (FK_PROVE_SORTS_FIRST=1, read once per process -> subprocess) with five different witnesses through
    submit / wait, plus the abandon path; FK_PROVE_SORTS_FIRST=0 runs the same sequence on the queue-everything schedule."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if not k.startswith(('FK_MSM_', 'FK_PROVE_', 'FK_SPMV_', 'FK_NTT_'))}
    env['FK_PROVE_SORTS_FIRST'] = sorts_first
    out = subprocess.run([sys.executable, os.path.join(root, 'tests', '_pipeline_child.py')], env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('PIPE ok')]
    assert len(line) == 1, out.stdout[-1000:]
    # with the early front in use, a foreign proof between wait(A) and wait(B) must have found B's front outstanding
    assert line[0].endswith('refused_a_foreign_proof=%s' % (sorts_first == '1')), line[0]


def test_trim_releases_scratch_and_proofs_continue(ctx):
    """fk_trim: the grow-only scratch of earlier proofs is released (keys and resident systems stay), the next proof re-allocates
    what it needs and gives the same bytes; refused while a proof is submitted."""
    import fawkes_crypto_amd as fk
    cs, prod, dr, key, z = _system(ctx, 777)
    r, s = fx.mont_fr(11), fx.mont_fr(12)
    want = ctx.prove_witness(key, dr, z, r, s).tobytes()
    ctx.trim()
    assert ctx.prove_witness(key, dr, z, r, s).tobytes() == want
    pin = ctx.host_alloc(z.shape)
    pin[:] = z
    t = ctx.prove_witness_submit(key, dr, pin, r, s)
    with pytest.raises(fk.FkError) as e:
        ctx.trim()
    assert e.value.code == 1 and 'outstanding' in str(e.value)
    assert ctx.prove_witness_wait(t).tobytes() == want
    ctx.trim()
    t = ctx.prove_witness_submit(key, dr, pin, r, s)
    assert ctx.prove_witness_wait(t).tobytes() == want
    ctx.host_free(pin)
    key.free(); dr.free()


def test_stats_union_of_intervals(ctx):
    cs, prod, dr, key, z = _system(ctx, 99, gates=20000, num_input=2, num_aux=15000)
    r, s = fx.mont_fr(5), fx.mont_fr(6)
    ctx.stats_reset()
    for _ in range(2):
        ctx.prove_witness(key, dr, z, r, s)
    st = ctx.stats()
    for name in ('acc_g1', 'acc_g2'):
        a = st[name]
        assert a['launches'] > 0 and a['adds'] > 0
        assert 0 < a['union_ms'] <= a['ms'] * 1.0001 + 1e-3        # a union never exceeds the sum of its intervals
    assert st['acc_g1']['launches'] == 2 * 4 and st['acc_g2']['launches'] == 2
