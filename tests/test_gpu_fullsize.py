"""BASELINE.json's two large configurations at their FULL sizes on one GPU, where no CPU oracle can follow: the checks are
the Groth16 pairing equation (oracle/bn254_ref.py verifier -- an independent big-int implementation), a negative control
(one witness element changed: the proof must be rejected), and byte equality between independent routes to the same
proof (single GPU vs 8 thread-ranks with sharded keys and the distributed quotient).

  configs[4]  2^27-row synthetic R1CS with the G2 MSM included -- the largest domain bellman accepts;
              proved once with FK_MSM_PRECOMP=0 (the W-bucket-set path for all five multiplications at the largest index
              arithmetic of every kernel) and once with the key the loader makes by default (fixed-base levels for the
              arrays that fit beside the 48 GiB key): same bytes.
  configs[3]  2^25 rollup-style R1CS, 1024-transaction shape: 1024 tiled rollup transactions (19.7 M gates, 9.6e8 matrix
              terms), the workload bench.py reports.
"""
import threading

import numpy as np
import pytest

import bn254_ref as ref
from test_gpu_dist_quotient import DevBuf, HostExchange, run_ranks

pytestmark = pytest.mark.gpu


def _verifies(bench, vk, z_inputs, proof):
    try:
        return bench.pairing_check(vk, z_inputs, proof.tobytes())
    except AssertionError:
        return False


def test_config4_2p27_rows_with_g2_pairing_checked(ctx):
    import bench
    import fawkes_crypto_amd as fk
    log2n = 27
    r1cs, z = bench.build_workload(ctx, fk, log2n)
    assert r1cs.n_rows == 1 << log2n
    tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
    # FK_MSM_PRECOMP=0 (read at every key load): no fixed-base levels, i.e. the W-bucket-set path for all five
    # multiplications at the largest index arithmetic there is.  (Left alone, the loader gives the arrays whose levels fit
    # beside the 48 GiB key the merged form; that mix is proved below as well and must give the same bytes.)
    import os
    os.environ['FK_MSM_PRECOMP'] = '0'
    try:
        key, vk = ctx.setup(r1cs, **tox)
    finally:
        del os.environ['FK_MSM_PRECOMP']
    assert key.counts()['m'] == 1 << log2n
    assert all(v == 0 for v in key.precomputed().values()), 'expected the W-bucket-set path (no fixed-base levels)'
    dr = ctx.load_r1cs(r1cs)
    info = dr.info()
    assert info['n_b'] > (1 << 25)             # the G2 MSM is part of it: > 3e7 G2 points
    d_z = ctx.dev_alloc(z.nbytes)
    r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
    try:
        ctx.upload(d_z, z)
        proof = ctx.prove_witness_dev(key, dr, d_z, r, s)
        assert _verifies(bench, vk, z[1:r1cs.num_input], proof)
        assert ctx.prove_witness_dev(key, dr, d_z, r, s).tobytes() == proof.tobytes()       # deterministic
        # negative control: one dense witness element changed -> the constraint system is violated -> must be rejected
        j = r1cs.num_input + r1cs.num_aux - 12345       # an output of a product gate
        bad = z[j].copy(); bad[0] ^= np.uint64(2)
        ctx.upload(d_z + j * 32, bad)
        forged = ctx.prove_witness_dev(key, dr, d_z, r, s)
        assert forged.tobytes() != proof.tobytes()
        assert not _verifies(bench, vk, z[1:r1cs.num_input], forged)
        # the key as the loader makes it by default (levels for the arrays that fit): same proof bytes
        key.free()
        key, _ = ctx.setup(r1cs, **tox)
        ctx.upload(d_z + j * 32, z[j])
        assert ctx.prove_witness_dev(key, dr, d_z, r, s).tobytes() == proof.tobytes()
    finally:
        ctx.dev_free(d_z)
        dr.free(); key.free()


def test_config3_2p25_rollup1024_single_gpu_and_8_thread_ranks(ctx):
    """the 1024-transaction system: single-GPU proof pairing-checked against its 2048 public roots, then reproduced byte for
    byte by 8 ranks (threads with their own library contexts on this GPU): key shards from fk_setup_tiled (each rank derives
    only its shard), every rank evaluates the resident constraint system, distributed quotient over log_w = 3 with the
    all-to-all emulated through host memory, 1/8 of each of the five MSMs, fold of the eight 384-byte records."""
    import bench
    import fawkes_crypto_amd as fk
    from fawkes_crypto_amd import parallel
    copies, world = 1024, 8
    inst, zs = bench.load_rollup_instance()
    z = bench.tile_witness(zs, inst.num_input, copies)
    num_input = 1 + copies * (inst.num_input - 1)
    n = copies * inst.num_gates + num_input
    log_m = 25
    assert (1 << (log_m - 1)) < n <= (1 << log_m)
    tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
    r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
    dr = ctx.load_r1cs(inst, copies=copies)
    assert dr.info()['rows'] == n and sum(dr.info()['nnz']) > 9e8
    d_z = ctx.dev_alloc(z.nbytes)
    ctx.upload(d_z, z)
    key, vk = ctx.setup(inst, copies=copies, **tox)
    assert key.precomputed()['h'] > 0          # the single-GPU key carries the fixed-base levels (merged bucket sets)
    want = ctx.prove_witness_dev(key, dr, d_z, r, s)
    assert _verifies(bench, vk, z[1:num_input], want)
    # the product's own verifier (fk_verify, host) agrees with the oracle's, also on a proof for other public inputs
    vkb = fk.api.vk_to_borsh(vk)
    assert fk.api.verify(vkb, z[1:num_input], want.tobytes()) is True
    swapped = z[1:num_input].copy(); swapped[[0, 1]] = swapped[[1, 0]]
    assert fk.api.verify(vkb, swapped, want.tobytes()) is False
    # the host-witness pipeline gives the same bytes
    zp = ctx.host_alloc(z.shape)
    zp[:] = z
    t0 = ctx.prove_witness_submit(key, dr, zp, r, s)
    t1 = ctx.prove_witness_submit(key, dr, zp, r, s)
    assert ctx.prove_witness_wait(t0).tobytes() == want.tobytes() and ctx.prove_witness_wait(t1).tobytes() == want.tobytes()
    ctx.host_free(zp)
    key.free()                                  # 140 GiB of key + levels make room for the eight shards' scratch

    L = (1 << log_m) // world
    ex = HostExchange(world)
    parts = [None] * world
    bar = threading.Barrier(world)
    one_at_a_time = threading.Lock()
    dens = dr.density_ptrs()

    def rank_fn(c_, rank):
        with one_at_a_time:                     # the shard derivations run one after the other (host memory, setup scratch)
            sk, _ = c_.setup(inst, copies=copies, shard_index=rank, shard_count=world, **tox)
        assert sk.shard_info()['h'] == (rank * L, min((rank + 1) * L, (1 << log_m) - 1))
        full = [DevBuf(c_, (1 << log_m) * 32) for _ in range(3)]
        send = [DevBuf(c_, L * 32) for _ in range(3)]
        recv = [DevBuf(c_, L * 32) for _ in range(3)]
        # the resident constraint system and the witness are read-only device memory shared by all contexts of the process
        c_.r1cs_eval_dev(dr, d_z, *[f.data_ptr() for f in full])
        blk = parallel.quotient_distributed(c_, rank, world, [f.data_ptr() for f in full], n, log_m, send, recv, ex.a2a_for(c_, rank))
        parts[rank] = c_.prove_msms_hz_r1cs_dev(sk, dr, blk.data_ptr(), d_z)
        bar.wait()
        proof = c_.prove_assemble(sk, np.stack(parts), r, s)
        for x in full + send + recv:
            x.free()
        sk.free()
        return proof.tobytes()

    try:
        got = run_ranks(world, rank_fn)
        assert all(g == want.tobytes() for g in got)
    finally:
        ctx.dev_free(d_z)
        dr.free()
