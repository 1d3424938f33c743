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
