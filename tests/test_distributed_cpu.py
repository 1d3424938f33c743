"""world_size-2 gloo test of the multi-GPU path (runs on CPU): each rank owns one shard of the key, the
partial MSM results travel through fawkes_crypto_amd.parallel.all_gather_parts (the one collective of the
path) and are folded by the product's fk_prove_assemble.  The per-shard MSMs themselves need a GPU, so the
oracle stands in for them HERE ONLY (as the checker's input); the result must equal the oracle's
single-process proof byte for byte."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import bn254_ref as ref
    import c_oracle as co
    import fixtures as fx
    from fawkes_crypto_amd import api, parallel
    from helpers import params_from_oracle_key, TOXIC
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        cs, z_in, z_aux = ref.random_r1cs(5150, 90, 3, 100)
        csr = fx.r1cs_to_csr(cs)
        key = co.setup(csr, **TOXIC)
        z = fx.witness_mont(z_in, z_aux)
        a, b, c, aa, bi, ba = co.synthesize(csr, z)
        r, s = fx.mont_fr(0xabc), fx.mont_fr(0xdef)
        want = co.prove(key, a, b, c, z, aa, bi, ba, r, s)
        # this rank's shard of every MSM (same slicing rule as fk_key_load: fk_shard_range)
        h = co.quotient_h(a, b, c)
        nin = cs.num_input
        sa = np.concatenate([z[:nin], z[nin:][aa != 0]])
        sb = np.concatenate([z[:nin][bi != 0], z[nin:][ba != 0]])
        part = np.zeros(api.FK_MSM_RESULT_BYTES, np.uint8)
        for off, bases, scalars, fn in ((0, key.h, h, co.msm_g1), (64, key.l, z[nin:], co.msm_g1), (128, key.a, sa, co.msm_g1),
                                        (192, key.b_g1, sb, co.msm_g1), (256, key.b_g2, sb, co.msm_g2)):
            lo, hi = api.shard_range(len(bases), rank, world)
            res = fn(np.array(bases[lo:hi]), scalars[lo:hi]) if hi > lo else np.zeros(128 if off == 256 else 64, np.uint8)
            part[off:off + len(res)] = res
        parts = parallel.all_gather_parts(part)
        assert parts.shape == (world, api.FK_MSM_RESULT_BYTES)
        assert np.array_equal(parts[rank], part)
        vk = api.HostVk(params_from_oracle_key(key))
        got = api.assemble(vk.handle, parts, r, s)
        q.put((rank, got.tobytes() == want.tobytes(), got.tobytes().hex()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2])
def test_sharded_prove_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results), results
    assert len({hx for _, _, hx in results}) == 1      # every rank assembled the same proof
