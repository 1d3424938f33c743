"""Distributed quotient (fk_dq_* + parallel.quotient_distributed) against the single-GPU pipeline, bit for bit.
The W ranks run as W threads of this process, each with its own library context on the one GPU of the box; the
all-to-all is emulated through host memory with a barrier (the RCCL version of the same exchange is
parallel.torch_all_to_all; its gloo twin is covered by tests/test_distributed_cpu.py)."""
import threading

import numpy as np
import pytest

import bn254_ref as ref
import fixtures as fx
from helpers import rand_fr_mont, r1cs_product, TOXIC

pytestmark = pytest.mark.gpu


class DevBuf:
    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes, self.ptr = ctx, nbytes, ctx.dev_alloc(nbytes)

    def data_ptr(self):
        return self.ptr

    def free(self):
        self.ctx.dev_free(self.ptr)


class HostExchange:
    """all-to-all for W threads: chunk q of rank j's buffer lands in slot j of rank q's buffer"""

    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world

    def a2a_for(self, ctx, rank):
        def a2a(dst, src):
            ctx.sync()
            self.slots[rank] = [ctx.download(s.data_ptr(), s.nbytes, np.uint8).reshape(self.world, -1) for s in src]
            self.barrier.wait()
            for k, d in enumerate(dst):
                ctx.upload(d.data_ptr(), np.ascontiguousarray(np.concatenate([self.slots[j][k][rank] for j in range(self.world)])))
            self.barrier.wait()
        return a2a


def run_ranks(world, fn):
    import fawkes_crypto_amd as fk
    out, err = [None] * world, []

    def body(rank):
        try:
            c = fk.Context(0)
            try:
                out[rank] = fn(c, rank)
            finally:
                c.close()
        except BaseException as e:     # noqa: BLE001 -- re-raised in the test thread
            err.append(e)
            raise

    th = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    if err:
        raise err[0]
    return out


@pytest.mark.parametrize('log_m,world,n', [(10, 1, 1000), (10, 2, 1024), (12, 4, 3000), (6, 8, 64), (14, 8, 16384), (20, 8, (1 << 20) - 77),
                                           (19, 4, 1 << 19), (4, 2, 9)])
def test_distributed_quotient_matches_single_gpu(ctx, log_m, world, n):
    from fawkes_crypto_amd import parallel
    m = 1 << log_m
    rng = np.random.default_rng(100 + log_m + world)
    a, b, c = (rand_fr_mont(rng, n, 'witness' if k == 1 else 'uniform') for k in range(3))
    want = ctx.quotient_h(a, b, c)                 # (m-1, 4) single-GPU pipeline (itself checked against the oracle)
    L = m // world
    ex = HostExchange(world)

    def rank_fn(c_, rank):
        full = [DevBuf(c_, n * 32) for _ in range(3)]
        for f, v in zip(full, (a, b, c)):
            c_.upload(f.data_ptr(), v)
        send = [DevBuf(c_, L * 32) for _ in range(3)]
        recv = [DevBuf(c_, L * 32) for _ in range(3)]
        blk = parallel.quotient_distributed(c_, rank, world, [f.data_ptr() for f in full], n, log_m, send, recv, ex.a2a_for(c_, rank))
        c_.sync()
        got = c_.download(blk.data_ptr(), L * 32, np.uint64).reshape(-1, 4)
        for x in full + send + recv:
            x.free()
        return got

    got = run_ranks(world, rank_fn)
    # random a, b, c are not a satisfied system: coefficient m-1 is not zero, bellman drops it (h has m-1 entries)
    for rank in range(world):
        hi = min((rank + 1) * L, m - 1)
        assert np.array_equal(got[rank][:hi - rank * L], want[rank * L:hi]), 'rank %d' % rank


def test_dq_argument_checks(ctx):
    import fawkes_crypto_amd as fk
    d = ctx.dev_alloc(1 << 12)
    try:
        with pytest.raises(fk.FkError):
            ctx.dq_cross_dev(d, 4, 0, 3, 0)          # 2^4 points cannot be cut 8 x 8
        with pytest.raises(fk.FkError):
            ctx.dq_local_dev(d, 6, 2, 1, 0)          # rank 2 of 2
        with pytest.raises(fk.FkError):
            ctx.dq_local_dev(d, 6, 0, 4, 0)          # 16 ranks
        with pytest.raises(fk.FkError):
            ctx.dq_local_dev(d, 6, 0, 1, 2)          # stage 2 without b, c
    finally:
        ctx.dev_free(d)


@pytest.mark.parametrize('world', [2, 4])
def test_distributed_proof_bit_exact(ctx, oracle, world):
    """whole proof over W ranks: device SpMV on every rank, distributed quotient, 1/W of every MSM, all-gather of the
    384-byte partials (emulated), assemble -- equals the oracle's single-process proof"""
    from fawkes_crypto_amd import parallel
    cs, z_in, z_aux = ref.random_r1cs(4242 + world, 1500, 3, 1600)
    csr = fx.r1cs_to_csr(cs)
    okey = oracle.setup(csr, **TOXIC)
    z = fx.witness_mont(z_in, z_aux)
    r, s = fx.mont_fr(0x1111), fx.mont_fr(0x2222)
    aa = oracle.synthesize(csr, z)
    want = oracle.prove(okey, *aa[:3], z, *aa[3:], r, s)
    r1cs = r1cs_product(csr)
    n = 1500 + 3
    log_m = 11
    L = (1 << log_m) // world
    ex = HostExchange(world)
    parts = [None] * world
    bar = threading.Barrier(world)
    tox = {k: fx.mont_fr(v) for k, v in TOXIC.items()}

    def rank_fn(c_, rank):
        key, _ = c_.setup(r1cs, shard_index=rank, shard_count=world, **tox)
        assert key.shard_info()['h'] == (rank * L, min((rank + 1) * L, (1 << log_m) - 1))
        dr = c_.load_r1cs(r1cs)
        dens = dr.density_ptrs()
        d_z = DevBuf(c_, z.nbytes)
        c_.upload(d_z.data_ptr(), z)
        full = [DevBuf(c_, (1 << log_m) * 32) for _ in range(3)]
        send = [DevBuf(c_, L * 32) for _ in range(3)]
        recv = [DevBuf(c_, L * 32) for _ in range(3)]
        c_.r1cs_eval_dev(dr, d_z.data_ptr(), *[f.data_ptr() for f in full])
        c_.prove_msms_z_begin_dev(key, d_z.data_ptr(), *dens)          # witness MSMs run underneath the quotient
        blk = parallel.quotient_distributed(c_, rank, world, [f.data_ptr() for f in full], n, log_m, send, recv, ex.a2a_for(c_, rank))
        part = c_.prove_msms_finish_dev(key, blk.data_ptr())
        assert c_.prove_msms_hz_dev(key, blk.data_ptr(), d_z.data_ptr(), *dens).tobytes() == part.tobytes()
        assert c_.prove_msms_hz_r1cs_dev(key, dr, blk.data_ptr(), d_z.data_ptr()).tobytes() == part.tobytes()     # index-list gathers
        split = np.array(c_.prove_msms_z_dev(key, d_z.data_ptr(), *dens), dtype=np.uint8, copy=True)     # the two-call form agrees
        split[:64] = c_.prove_msm_h_dev(key, blk.data_ptr())
        assert split.tobytes() == part.tobytes()
        parts[rank] = part
        bar.wait()
        proof = c_.prove_assemble(key, np.stack(parts), r, s)
        for x in full + send + recv + [d_z]:
            x.free()
        dr.free(); key.free()
        return proof.tobytes()

    got = run_ranks(world, rank_fn)
    assert all(g == want.tobytes() for g in got)
