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
        ok = got.tobytes() == want.tobytes()

        # ---- the work-balanced schedule (parallel.prove_balanced) with CPU stand-ins for the GPU MSMs:
        # rank 0 computes h and ships slices point-to-point, witness arrays are split by plan_z_fractions
        import torch
        fracs = parallel.plan_z_fractions(world, key.m, cs.num_aux, len(key.a), len(key.b_g1))
        h_ranges = [api.shard_range(key.m - 1, g, world) for g in range(world)]

        def frac_range(n_, lo_, hi_):
            a_ = int(n_ * lo_ + 0.5); b_ = n_ if hi_ >= 1.0 else int(n_ * hi_ + 0.5)
            return a_, max(a_, b_)

        def z_fn():
            rec = np.zeros(api.FK_MSM_RESULT_BYTES, np.uint8)
            for off, bases, scalars, fn in ((64, key.l, z[nin:], co.msm_g1), (128, key.a, sa, co.msm_g1),
                                            (192, key.b_g1, sb, co.msm_g1), (256, key.b_g2, sb, co.msm_g2)):
                lo_, hi_ = frac_range(len(bases), *fracs[rank])
                if hi_ > lo_:
                    res = fn(np.array(bases[lo_:hi_]), scalars[lo_:hi_])
                    rec[off:off + len(res)] = res
            return rec

        def h_fn(t):
            lo_, hi_ = h_ranges[rank]
            sl = np.frombuffer(t.numpy().tobytes(), np.uint64).reshape(-1, 4)
            assert sl.shape[0] == hi_ - lo_
            return co.msm_g1(np.array(key.h[lo_:hi_]), sl) if hi_ > lo_ else np.zeros(64, np.uint8)

        def quotient_fn():
            full = np.zeros((key.m, 4), np.uint64); full[:key.m - 1] = h
            return torch.from_numpy(full.view(np.uint8).reshape(-1).copy())

        recv = torch.zeros(max(h_ranges[rank][1] - h_ranges[rank][0], 1) * 32, dtype=torch.uint8)
        got2 = parallel.prove_balanced(rank, world, quotient_fn, z_fn, h_fn, lambda parts_: api.assemble(vk.handle, parts_, r, s),
                                       h_ranges, recv)
        ok = ok and got2.tobytes() == want.tobytes()
        q.put((rank, ok, got.tobytes().hex()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
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
