"""The multi-GPU prover behind the C ABI (fk_init_devices / fk_multi_*, csrc/multi.hip): ONE call proves on N ranks, every
exchange inside the library.  A one-GPU box passes the same device id N times (the ranks then share the GPU and the
all-to-all copies are device-local); the proof bytes must not depend on N and must equal the oracle's.
Also here: the cyclic row slices of the constraint-system evaluation (fk_r1cs_eval_slice_dev) against the full evaluation."""
import numpy as np
import pytest

import bn254_ref as ref
import fixtures as fx
from helpers import params_from_oracle_key, r1cs_product, TOXIC
from test_gpu_r1cs import _ragged_system

pytestmark = pytest.mark.gpu


def _eval_full_and_slices(ctx, dr, z, rows, log_m):
    m = 1 << log_m
    d = [ctx.dev_alloc(m * 32) for _ in range(3)]
    d_z = ctx.dev_alloc(z.nbytes)
    try:
        ctx.upload(d_z, z)
        for p in d:
            ctx.upload(p, np.zeros(m * 4, np.uint64))
        ctx.r1cs_eval_dev(dr, d_z, *d)
        full = [ctx.download(p, m * 32, np.uint64).reshape(-1, 4) for p in d]
        for f in full:
            assert not f[rows:].any()
        for log_w in (0, 1, 2, 3):
            W, L = 1 << log_w, m >> log_w
            for rank in range(W):
                for p in d:
                    ctx.upload(p, np.full(L * 4, 0xdeadbeefdeadbeef, np.uint64))
                ctx.r1cs_eval_slice_dev(dr, d_z, log_m, rank, log_w, *d)
                for k in range(3):
                    got = ctx.download(d[k], L * 32, np.uint64).reshape(-1, 4)
                    assert np.array_equal(got, full[k][rank::W]), (log_w, rank, k)
        return full
    finally:
        for p in d + [d_z]:
            ctx.dev_free(p)


@pytest.mark.parametrize('lens', [[0, 1, 1, 1, 2, 3, 4, 5, 31, 32, 33, 100, 257, 512], [1, 1, 2], [40, 41, 1500]])
@pytest.mark.parametrize('copies,gates', [(1, 700), (5, 700), (7, 333), (4, 252)])
def test_eval_slice_matches_full_evaluation(ctx, oracle, lens, copies, gates):
    """every cyclic slice (W = 1, 2, 4, 8, every rank) of a system with short rows only, with the Poseidon-like mix of row
    lengths (length-class kernel) and with very long rows; untiled and tiled, with gate counts of every residue mod 8 (which
    residue class a copy needs depends on copy * gates mod W)"""
    base = _ragged_system(len(lens) * 100 + copies, lens, gates, 3, 300)
    csr = fx.tile_r1cs(base, copies) if copies > 1 else base
    nv, rows = csr.num_input + csr.num_aux, csr.num_gates + csr.num_input
    rnd = np.random.default_rng(copies)
    z = fx.co.limbs_arr([int(x) % ref.R for x in rnd.integers(0, 2**63, nv).astype(object) * (2**190 + 12345)])
    want = oracle.synthesize(csr, z)
    log_m = max(rows - 1, 1).bit_length()
    loads = [lambda: ctx.load_r1cs(r1cs_product(csr))]
    if copies > 1:
        loads.append(lambda: ctx.load_r1cs(r1cs_product(base), copies=copies))
    for load in loads:
        dr = load()
        try:
            full = _eval_full_and_slices(ctx, dr, z, rows, log_m)
            for k in range(3):
                assert np.array_equal(full[k][:rows], want[k])
        finally:
            dr.free()


def test_eval_slice_argument_checks(ctx):
    import fawkes_crypto_amd as fk
    base = _ragged_system(5, [1, 2], 100, 2, 50)
    dr = ctx.load_r1cs(r1cs_product(base))
    d = ctx.dev_alloc(1 << 14)
    try:
        with pytest.raises(fk.FkError):
            ctx.r1cs_eval_slice_dev(dr, d, 7, 2, 1, d, d, d)        # rank 2 of 2
        with pytest.raises(fk.FkError):
            ctx.r1cs_eval_slice_dev(dr, d, 7, 0, 4, d, d, d)        # 16 ranks
        with pytest.raises(fk.FkError):
            ctx.r1cs_eval_slice_dev(dr, d, 6, 0, 1, d, d, d)        # 102 rows do not fit 2^6
    finally:
        ctx.dev_free(d)
        dr.free()


def _toy(oracle, seed, gates, nin, naux):
    cs, z_in, z_aux = ref.random_r1cs(seed, gates, nin, naux)
    csr = fx.r1cs_to_csr(cs)
    okey = oracle.setup(csr, **TOXIC)
    z = fx.witness_mont(z_in, z_aux)
    r, s = fx.mont_fr(0x1111 + seed), fx.mont_fr(0x2222 + seed)
    aa = oracle.synthesize(csr, z)
    want = oracle.prove(okey, *aa[:3], z, *aa[3:], r, s)
    return csr, okey, z, z_in, r, s, want


@pytest.mark.parametrize('world', [1, 2, 3, 4, 8])
def test_multi_prove_one_call_matches_oracle(ctx, oracle, world):
    """fk_multi_prove_r1cs on `world` ranks (sharing this GPU): key shards derived by fk_multi_setup, one constraint-system replica
    per rank, witness from host memory -- the oracle's bytes, whatever the rank count; also through fk_multi_key_load (host key
    arrays) and the two-slot submit / wait form"""
    import fawkes_crypto_amd as fk
    csr, okey, z, z_in, r, s, want = _toy(oracle, 5150 + world, 1500, 3, 1600)
    r1cs = r1cs_product(csr)
    tox = {k: fx.mont_fr(v) for k, v in TOXIC.items()}
    mc = fk.MultiContext([0] * world)
    try:
        assert mc.size == world
        key, vk = mc.setup(r1cs, **tox)
        dr = mc.load_r1cs(r1cs)
        # 4 and 8 ranks cut the transforms: h in blocks of the domain (what the distributed quotient leaves on a rank).  2 and 3 ranks run
        # the exchange-free schedule (round 4): rank 0 holds ALL of h (it computes the whole quotient), the others none
        n_h = (1 << 11) - 1
        q0 = world in (2, 3)
        for g in range(world):
            info = mc.key_shard(key, g).shard_info()
            assert info['h'] == (((0, n_h) if g == 0 else (n_h, n_h)) if q0 else fk.api.h_shard_range(n_h, g, world))
        # round 4: from two ranks on the witness arrays are dealt BY WORK (FK_Z_WORK_SPLIT): l | a | b_g1 | b_g2 laid end to end, a G2
        # point counting 2.8 G1 points, cut into `world` equal pieces -- the shards must tile every array exactly and carry equal work
        cnt = mc.key_shard(key, 0).counts()
        infos = [mc.key_shard(key, g).shard_info() for g in range(world)]
        for arr, n_arr in (('l', cnt['n_l']), ('a', cnt['n_a']), ('b', cnt['n_b']), ('b_g2', cnt['n_b'])):
            assert infos[0][arr][0] == 0 and infos[-1][arr][1] == n_arr, (arr, infos)
            assert all(infos[g][arr][1] == infos[g + 1][arr][0] for g in range(world - 1)), (arr, infos)
        if world > 1:
            work = [sum((i[a_][1] - i[a_][0]) * w for a_, w in (('l', 1.0), ('a', 1.0), ('b', 1.0), ('b_g2', 2.8))) for i in infos]
            if q0:          # rank 0's fixed work (evaluation, quotient, H: 2.2 units per domain point) counts towards its piece
                work[0] += 2.2 * (1 << 11)
                assert max(work[1:]) - min(work[1:]) <= 2 * 2.8 + 1e-6 and (abs(work[0] - work[1]) <= 2 * 2.8 + 1e-6 or infos[0]['l'][1] == 0), work
                assert [infos[g][a_] for g in range(world) for a_ in ('l', 'a', 'b', 'b_g2')] == \
                       [fk.api.work_shard_ranges(cnt['n_l'], cnt['n_a'], cnt['n_b'], g, world, q0_domain=1 << 11)[a_] for g in range(world) for a_ in ('l', 'a', 'b', 'b_g2')]
            else:
                assert max(work) - min(work) <= 2 * 2.8 + 1e-6, work
            assert q0 or any(i['b'] != i['b_g2'] for i in infos)          # b_g1 and b_g2 are sliced independently
        got = mc.prove_witness(key, dr, z, r, s)
        assert got.tobytes() == want.tobytes()
        assert ref.verify(fx.key_to_py(okey), z_in[1:], ref.proof_from_borsh(got.tobytes()))
        # round 5: the witness crosses PCIe ONCE -- every rank uploads its 1 / world piece, the ranks all-gather the rest among themselves
        tr = mc.witness_traffic()
        assert tr['pcie_bytes'] == z.nbytes and tr['gathered_bytes'] == (world - 1) * z.nbytes, tr
        if world in (3, 8):
            # ... FK_MULTI_WITNESS=whole (read by fk_init_devices): rounds 3-4, every rank uploads all of z -- same bytes
            import os
            os.environ['FK_MULTI_WITNESS'] = 'whole'
            try:
                mc_w = fk.MultiContext([0] * world)
            finally:
                del os.environ['FK_MULTI_WITNESS']
            try:
                key_w, _ = mc_w.setup(r1cs, **tox)
                dr_w = mc_w.load_r1cs(r1cs)
                assert mc_w.prove_witness(key_w, dr_w, z, r, s).tobytes() == want.tobytes()
                assert mc_w.witness_traffic() == dict(pcie_bytes=world * z.nbytes, gathered_bytes=0)
                key_w.free(); dr_w.free()
            finally:
                mc_w.close()
        if world in (2, 8):
            # the equal split of rounds 1-3 (FK_MULTI_SPLIT=equal, read at every key load): 1 / world of each array, same bytes
            import os
            os.environ['FK_MULTI_SPLIT'] = 'equal'
            try:
                key_e, _ = mc.setup(r1cs, **tox)
            finally:
                del os.environ['FK_MULTI_SPLIT']
            ie = [mc.key_shard(key_e, g).shard_info() for g in range(world)]
            assert all(i['b'] == i['b_g2'] and i['l'] == fk.api.shard_range(cnt['n_l'], g, world) and i['h'] == fk.api.h_shard_range(n_h, g, world)
                       for g, i in enumerate(ie))
            assert mc.prove_witness(key_e, dr, z, r, s).tobytes() == want.tobytes()        # (the cut-transform schedule: seven exchanges, also on 2 ranks)
            key_e.free()
            # ... and the work split WITH the cut transforms on 2 ranks (FK_MULTI_SPLIT=work)
            os.environ['FK_MULTI_SPLIT'] = 'work'
            try:
                key_w, _ = mc.setup(r1cs, **tox)
            finally:
                del os.environ['FK_MULTI_SPLIT']
            assert mc.key_shard(key_w, 1).shard_info()['h'] == fk.api.h_shard_range(n_h, 1, world)
            assert mc.prove_witness(key_w, dr, z, r, s).tobytes() == want.tobytes()
            key_w.free()
        # the verifying key of the sharded derivation is the oracle's
        assert np.array_equal(vk['ic'], np.array(okey.ic)) and vk['alpha_g1'].tobytes() == np.array(okey.alpha_g1).tobytes()
        # pipelined: two witnesses in flight, a different one second
        z2 = z.copy()
        t0 = mc.prove_witness_submit(key, dr, z, r, s)
        t1 = mc.prove_witness_submit(key, dr, z2, s, r)
        with pytest.raises(fk.FkError):
            mc.prove_witness_submit(key, dr, z, r, s)               # a third one does not fit
        assert mc.prove_witness_wait(t0).tobytes() == want.tobytes()
        p1 = mc.prove_witness_wait(t1)
        assert p1.tobytes() != want.tobytes() and ref.verify(fx.key_to_py(okey), z_in[1:], ref.proof_from_borsh(p1.tobytes()))
        key.free()
        # the same through host key arrays (fk_key_load on every rank, shard g of `world`)
        key2 = mc.load_key(params_from_oracle_key(okey, r1cs))
        assert mc.prove_witness(key2, dr, z, r, s).tobytes() == want.tobytes()
        key2.free(); dr.free()
    finally:
        mc.close()


def test_multi_prove_long_rows_tiled_and_errors(ctx, oracle):
    """a tiled system with the Poseidon-like mix of row lengths (the length-class kernel's slices), 4 ranks; a key that does not
    belong to the system is refused with the rank's message and the next proof still works"""
    import fawkes_crypto_amd as fk
    copies = 6
    base = _ragged_system(99, [0, 1, 1, 1, 2, 3, 4, 5, 31, 32, 33, 100, 257], 250, 3, 120)
    csr = fx.tile_r1cs(base, copies)
    nv = csr.num_input + csr.num_aux
    rnd = np.random.default_rng(3)
    z = fx.co.limbs_arr([int(x) % ref.R for x in rnd.integers(0, 2**63, nv).astype(object) * (2**190 + 12345)])
    z[0] = fx.mont_fr(1)
    tox = {k: fx.mont_fr(v) for k, v in TOXIC.items()}
    r, s = fx.mont_fr(77), fx.mont_fr(88)
    # single-GPU proof of the explicitly replicated system = the reference point (itself oracle-checked in test_gpu_tiled.py)
    k1, _ = ctx.setup(r1cs_product(base), copies=copies, **tox)
    d1 = ctx.load_r1cs(r1cs_product(base), copies=copies)
    want = ctx.prove_witness(k1, d1, z, r, s)
    okey = oracle.setup(csr, **TOXIC)
    aa = oracle.synthesize(csr, z)
    assert oracle.prove(okey, *aa[:3], z, *aa[3:], r, s).tobytes() == want.tobytes()
    k1.free(); d1.free()
    mc = fk.MultiContext([0, 0, 0, 0])
    try:
        key, _ = mc.setup(r1cs_product(base), copies=copies, **tox)
        dr = mc.load_r1cs(r1cs_product(base), copies=copies)
        assert mc.prove_witness(key, dr, z, r, s).tobytes() == want.tobytes()
        other = mc.load_r1cs(r1cs_product(_ragged_system(98, [1, 2], 250, 3, 121)), copies=copies)
        # a system that does not belong to the key: refused BEFORE anything is read from z (fk_multi_prove_r1cs compares the
        # variable counts first: ADVICE r3 -- it used to upload num_vars(other) * 32 bytes from the caller's shorter buffer) ...
        z_other = np.zeros((mc.r1cs_replica(other, 0).info()['num_vars'], 4), np.uint64)
        with pytest.raises(fk.FkError) as e:
            mc.prove_witness(key, other, z_other, r, s)
        assert e.value.code == 6 and 'variable counts' in str(e.value)
        # ... and the Python host refuses a witness of the wrong length for the system it is given
        with pytest.raises(fk.FkError) as e:
            mc.prove_witness(key, dr, z[:-1], r, s)
        assert e.value.code == 6 and 'variables' in str(e.value)
        assert mc.prove_witness(key, dr, z, r, s).tobytes() == want.tobytes()
        other.free(); key.free(); dr.free()
    finally:
        mc.close()


def test_multi_init_argument_checks():
    import fawkes_crypto_amd as fk
    with pytest.raises(fk.FkError):
        fk.MultiContext([])
    with pytest.raises(fk.FkError):
        fk.MultiContext([0, 99])


@pytest.mark.parametrize('host_events', [False, True])
def test_topology_and_preflight_on_one_device_named_three_times(monkeypatch, host_events):
    """fk_multi_topology / fk_multi_preflight where a one-GPU box can run them: three ranks on device 0 -- every pair is 'self',
    every ordered pair's 16 MiB pull behind the other rank's event is verified and timed, with in-stream and with host-side event waits; the
    argument checks of the preflight"""
    import fawkes_crypto_amd as fk
    if host_events:
        monkeypatch.setenv('FK_MULTI_HOST_EVENTS', '1')
    mc = fk.MultiContext([0, 0, 0])
    try:
        topo = mc.topology()
        assert topo == [['self'] * 3] * 3
        pf = mc.preflight(16 << 20)
        assert pf['ok'] is True and pf['host_events'] is host_events and pf['bytes'] == 16 << 20, pf
        for i in range(3):
            for j in range(3):
                assert pf['status'][i][j] == 0
                assert (pf['gbps'][i][j] > 1.0) == (i != j), pf['gbps']          # a device-to-device copy inside one MI355X: far above 1 GB/s
        bad = mc.preflight(12)                                                    # not a multiple of 8 / below 64 bytes
        assert bad['ok'] is False and bad['rc'] == 1
    finally:
        mc.close()


@pytest.mark.parametrize('transport', ['peer-dma', 'rccl'])
def test_one_rank_runs_the_exchanges_over_both_transports(ctx, oracle, monkeypatch, transport):
    """FK_MULTI_FORCE_EXCHANGE=1: a single rank runs the distributed schedule -- seven exchanges with itself -- once with the
    peer copies and once over a real RCCL communicator (FK_MULTI_TRANSPORT=rccl: librccl bound with dlopen, ncclCommInitAll,
    grouped ncclSend / ncclRecv on the exchange stream).  What a one-GPU box can execute of the RCCL transport; same bytes."""
    import fawkes_crypto_amd as fk
    csr, okey, z, z_in, r, s, want = _toy(oracle, 6200, 1500, 3, 1600)
    monkeypatch.setenv('FK_MULTI_FORCE_EXCHANGE', '1')
    if transport == 'rccl':
        monkeypatch.setenv('FK_MULTI_TRANSPORT', 'rccl')
    mc = fk.MultiContext([0])
    try:
        assert mc.transport == transport, mc.note()
        tox = {k: fx.mont_fr(v) for k, v in TOXIC.items()}
        key, _ = mc.setup(r1cs_product(csr), **tox)
        dr = mc.load_r1cs(r1cs_product(csr))
        for _ in range(2):
            assert mc.prove_witness(key, dr, z, r, s).tobytes() == want.tobytes()
        key.free(); dr.free()
    finally:
        mc.close()
    # two ranks on ONE device cannot share a communicator: the library says so and keeps the peer copies
    if transport == 'rccl':
        mc = fk.MultiContext([0, 0])
        try:
            assert mc.transport == 'peer-dma' and 'distinct devices' in mc.note()
        finally:
            mc.close()


def test_multi_context_grows_and_survives_changing_systems(ctx, oracle):
    """one MultiContext, three constraint systems of growing domain one after the other (the exchange buffers of every rank are
    re-allocated between proofs: nobody may still be pulling from them), then the small one again; oracle bytes every time"""
    import fawkes_crypto_amd as fk
    mc = fk.MultiContext([0, 0, 0, 0])
    tox = {k: fx.mont_fr(v) for k, v in TOXIC.items()}
    try:
        cases = [_toy(oracle, 7000 + i, g, 3, g + 40) for i, g in enumerate((300, 2500, 9000))]
        loaded = []
        for csr, okey, z, z_in, r, s, want in cases:
            key, _ = mc.setup(r1cs_product(csr), **tox)
            dr = mc.load_r1cs(r1cs_product(csr))
            loaded.append((key, dr))
            assert mc.prove_witness(key, dr, z, r, s).tobytes() == want.tobytes()
        for (key, dr), (csr, okey, z, z_in, r, s, want) in zip(loaded, cases):      # all keys resident at once, any order
            t = mc.prove_witness_submit(key, dr, z, r, s)
            assert mc.prove_witness_wait(t).tobytes() == want.tobytes()
        for key, dr in loaded:
            key.free(); dr.free()
    finally:
        mc.close()
