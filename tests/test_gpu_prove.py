"""GPU parity of the full prover path behind prover.rs:63-90: proof bytes bit-identical to the oracle
for fixed (r, s), pairing check, error behaviour, sharded == unsharded."""
import numpy as np
import pytest

import bn254_ref as ref
import fixtures as fx
from helpers import R, golden_instance, params_from_oracle_key, r1cs_product, TOXIC

pytestmark = pytest.mark.gpu


def _instance(oracle, seed, gates, nin, naux, toxic=TOXIC):
    cs, z_in, z_aux = ref.random_r1cs(seed, gates, nin, naux)
    csr = fx.r1cs_to_csr(cs)
    key = oracle.setup(csr, **toxic)
    return cs, csr, key, z_in, z_aux


def test_golden_proof_bit_exact(ctx, oracle):
    import fawkes_crypto_amd as fk
    g, cs, z_in, z_aux, tw, r, s = golden_instance()
    csr = fx.r1cs_to_csr(cs)
    key = oracle.setup(csr, **tw)
    params = params_from_oracle_key(key, r1cs_product(csr))
    dk = ctx.load_key(params)
    z = fx.witness_mont(z_in, z_aux)
    inputs, proof = fk.prove_with_rs(ctx, params, dk, z[:cs.num_input], z[cs.num_input:], fx.mont_fr(r), fx.mont_fr(s))
    assert proof.to_bytes().hex() == g['proof']
    assert np.array_equal(inputs, z[1:cs.num_input])       # prover.rs:84-87: inputs without the leading ONE
    # h coefficients too
    a, b, c, *_ = ctx.synthesize(params.r1cs, z)
    from helpers import mont_ints
    assert mont_ints(ctx.quotient_h(a, b, c)) == [int(x, 16) for x in g['h']]


@pytest.mark.parametrize('shape', [(5, 7, 1, 9), (6, 100, 4, 97), (7, 1000, 2, 1100)])
def test_prove_vs_oracle_small(ctx, oracle, shape):
    import fawkes_crypto_amd as fk
    seed, gates, nin, naux = shape
    cs, csr, key, z_in, z_aux = _instance(oracle, seed, gates, nin, naux)
    params = params_from_oracle_key(key, r1cs_product(csr))
    dk = ctx.load_key(params)
    z = fx.witness_mont(z_in, z_aux)
    r, s = fx.mont_fr(0x1234567 * seed), fx.mont_fr(0x7654321 + seed)
    a, b, c, aa, bi, ba = oracle.synthesize(csr, z)
    want, want_msm = oracle.prove(key, a, b, c, z, aa, bi, ba, r, s, want_msm=True)
    got_msm = ctx.prove_msms(dk, a, b, c, z, aa, bi, ba)
    assert got_msm.tobytes() == want_msm.tobytes()
    _, proof = fk.prove_with_rs(ctx, params, dk, z[:nin], z[nin:], r, s)
    assert proof.to_bytes() == want.tobytes()
    assert ref.verify(fx.key_to_py(key), z_in[1:], ref.proof_from_borsh(proof.to_bytes()))


def test_config1_shape_bit_exact_and_verifies(ctx, oracle):
    """BASELINE configs[0]: poseidon-merkle depth-32 shape (7362 gates + 2 input rows = 7364 rows, m = 2^13,
    2 inputs, 7394 aux), synthetic satisfiable R1CS.  HIP proof == oracle proof, and the pairing holds."""
    import fawkes_crypto_amd as fk
    cs, csr, key, z_in, z_aux = _instance(oracle, 11, 7362, 2, 7394)
    assert key.m == 1 << 13
    params = params_from_oracle_key(key, r1cs_product(csr))
    dk = ctx.load_key(params)
    z = fx.witness_mont(z_in, z_aux)
    r, s = fx.mont_fr(0xabcdef0123456789), fx.mont_fr(0x9876543210fedcba)
    a, b, c, aa, bi, ba = oracle.synthesize(csr, z)
    want = oracle.prove(key, a, b, c, z, aa, bi, ba, r, s)
    inputs, proof, tm = fk.prove_with_rs(ctx, params, dk, z[:2], z[2:], r, s, want_timings=True)
    assert proof.to_bytes() == want.tobytes()
    assert ref.verify(fx.key_to_py(key), z_in[1:], ref.proof_from_borsh(proof.to_bytes()))
    assert tm['total_ms'] > 0
    # random r, s through the OsRng-style sampler: different bytes, still a valid proof
    _, p2 = fk.prove(ctx, params, dk, z[:2], z[2:])
    assert p2.to_bytes() != proof.to_bytes()
    assert ref.verify(fx.key_to_py(key), z_in[1:], ref.proof_from_borsh(p2.to_bytes()))


def test_sharded_equals_unsharded(ctx, oracle):
    """MSM sharding by points: 3 shards run one after another on this GPU, partials folded by
    fk_prove_assemble, must reproduce the single-GPU proof bytes."""
    cs, csr, key, z_in, z_aux = _instance(oracle, 21, 500, 3, 520)
    params = params_from_oracle_key(key, r1cs_product(csr))
    z = fx.witness_mont(z_in, z_aux)
    r, s = fx.mont_fr(77), fx.mont_fr(88)
    a, b, c, aa, bi, ba = ctx.synthesize(params.r1cs, z)
    dk = ctx.load_key(params)
    want = ctx.prove_raw(dk, a, b, c, z, aa, bi, ba, r, s)
    parts = []
    for i in range(3):
        sk = ctx.load_key(params, shard_index=i, shard_count=3)
        parts.append(ctx.prove_msms(sk, a, b, c, z, aa, bi, ba))
        sk.free()
    got = ctx.prove_assemble(dk, np.stack(parts), r, s)
    assert got.tobytes() == want.tobytes()
    assert want.tobytes() == oracle.prove(key, a, b, c, z, aa, bi, ba, r, s).tobytes()


def test_balanced_schedule_pieces(ctx, oracle):
    """The work-balanced multi-GPU split, ranks simulated one after another on this GPU: weighted witness
    shards (fk_prove_msms_z_dev), h slices shipped to fk_prove_msm_h_dev, folded by fk_prove_assemble --
    must reproduce the single-GPU proof bytes."""
    from fawkes_crypto_amd import api, parallel
    cs, csr, key, z_in, z_aux = _instance(oracle, 23, 700, 2, 760)
    params = params_from_oracle_key(key, r1cs_product(csr))
    z = fx.witness_mont(z_in, z_aux)
    r, s = fx.mont_fr(1234), fx.mont_fr(5678)
    a, b, c, aa, bi, ba = ctx.synthesize(params.r1cs, z)
    dk = ctx.load_key(params)
    want = ctx.prove_raw(dk, a, b, c, z, aa, bi, ba, r, s)
    world = 3
    fracs = parallel.plan_z_fractions(world, params.m, params.num_aux, params.a.shape[0], params.b_g1.shape[0])
    assert fracs[0][0] == 0.0 and fracs[-1][1] == 1.0 and all(abs(fracs[i][1] - fracs[i + 1][0]) < 1e-12 for i in range(world - 1))
    m = params.m
    d = {k: ctx.dev_alloc(m * 32) for k in 'abch'}
    d_z = ctx.dev_alloc(z.nbytes)
    d_aa, d_bi, d_ba = ctx.dev_alloc(max(len(aa), 1)), ctx.dev_alloc(len(bi)), ctx.dev_alloc(max(len(ba), 1))
    try:
        for k, v in (('a', a), ('b', b), ('c', c)):
            ctx.upload(d[k], v)
        ctx.upload(d_z, z); ctx.upload(d_aa, aa); ctx.upload(d_bi, bi); ctx.upload(d_ba, ba)
        ctx.quotient_h_dev(d['a'], d['b'], d['c'], a.shape[0], d['h'])          # "rank 0"
        parts = []
        for g in range(world):
            sk = ctx.load_key(params, shard_index=g, shard_count=world, z_frac=fracs[g])
            info = sk.shard_info()
            assert info['h'] == api.h_shard_range(m - 1, g, world)
            part = ctx.prove_msms_z_dev(sk, d_z, d_aa, d_bi, d_ba)
            assert part[:64].tobytes() == bytes(64)
            part[:64] = ctx.prove_msm_h_dev(sk, d['h'] + info['h'][0] * 32)
            parts.append(part)
            sk.free()
        got = ctx.prove_assemble(dk, np.stack(parts), r, s)
        assert got.tobytes() == want.tobytes()
    finally:
        for p_ in list(d.values()) + [d_z, d_aa, d_bi, d_ba]:
            ctx.dev_free(p_)


@pytest.mark.parametrize('loader', ['key_load', 'setup'])
def test_balanced_schedule_rank0_holds_no_witness_points(ctx, oracle, loader):
    """World 4 with rank 0's witness share clamped to nothing -- what plan_z_fractions gives the bench workload from 4
    ranks on.  (0, 0) must be the EMPTY slice (it used to alias "equal split": rank 0 then re-counted the first quarter
    of L, A, B1, B2 and the proof was invalid).  Through fk_key_load and through fk_setup."""
    from fawkes_crypto_amd import api
    cs, csr, key, z_in, z_aux = _instance(oracle, 29, 600, 2, 650)
    r1cs = r1cs_product(csr)
    params = params_from_oracle_key(key, r1cs)
    z = fx.witness_mont(z_in, z_aux)
    r, s = fx.mont_fr(4321), fx.mont_fr(8765)
    a, b, c, aa, bi, ba = ctx.synthesize(params.r1cs, z)
    dk = ctx.load_key(params)
    want = ctx.prove_raw(dk, a, b, c, z, aa, bi, ba, r, s)
    assert want.tobytes() == oracle.prove(key, a, b, c, z, aa, bi, ba, r, s).tobytes()
    world = 4
    fracs = [(0.0, 0.0), (0.0, 1 / 3), (1 / 3, 2 / 3), (2 / 3, 1.0)]
    tox = {k: fx.mont_fr(v) for k, v in TOXIC.items()}
    m = params.m
    d = {k: ctx.dev_alloc(m * 32) for k in 'abch'}
    d_z = ctx.dev_alloc(z.nbytes)
    d_aa, d_bi, d_ba = ctx.dev_alloc(max(len(aa), 1)), ctx.dev_alloc(len(bi)), ctx.dev_alloc(max(len(ba), 1))
    try:
        for k, v in (('a', a), ('b', b), ('c', c)):
            ctx.upload(d[k], v)
        ctx.upload(d_z, z); ctx.upload(d_aa, aa); ctx.upload(d_bi, bi); ctx.upload(d_ba, ba)
        ctx.quotient_h_dev(d['a'], d['b'], d['c'], a.shape[0], d['h'])
        parts, covered = [], {'l': 0, 'a': 0, 'b': 0}
        for g in range(world):
            if loader == 'key_load':
                sk = ctx.load_key(params, shard_index=g, shard_count=world, z_frac=fracs[g])
            else:
                sk, _ = ctx.setup(r1cs, shard_index=g, shard_count=world, z_frac=fracs[g], **tox)
            info = sk.shard_info()
            assert info['h'] == api.h_shard_range(m - 1, g, world)
            for nm in 'lab':
                assert info[nm][0] == covered[nm], (g, nm, info)        # slices tile the arrays without overlap
                covered[nm] = info[nm][1]
            if g == 0:
                assert info['l'] == (0, 0) and info['a'] == (0, 0) and info['b'] == (0, 0)
            part = ctx.prove_msms_z_dev(sk, d_z, d_aa, d_bi, d_ba)
            part[:64] = ctx.prove_msm_h_dev(sk, d['h'] + info['h'][0] * 32)
            parts.append(part)
            sk.free()
        assert covered == {'l': params.num_aux, 'a': params.a.shape[0], 'b': params.b_g1.shape[0]}
        got = ctx.prove_assemble(dk, np.stack(parts), r, s)
        assert got.tobytes() == want.tobytes()
    finally:
        for p_ in list(d.values()) + [d_z, d_aa, d_bi, d_ba]:
            ctx.dev_free(p_)
    # a lone shard must hold everything: a zero-initialised fraction range is refused, not proved from
    import fawkes_crypto_amd as fk
    for bad in ((0.0, 0.0), (0.0, 0.5), (0.5, 0.25), (0.0, 1.5)):
        with pytest.raises(fk.FkError) as e:
            ctx.load_key(params, z_frac=bad)
        assert e.value.code == 1
    ctx.load_key(params, z_frac=(0.0, 1.0)).free()


def test_error_behaviour(ctx, oracle):
    """C ABI returns codes where bellman returns SynthesisError."""
    import fawkes_crypto_amd as fk
    cs, csr, key, z_in, z_aux = _instance(oracle, 31, 20, 2, 25)
    params = params_from_oracle_key(key, r1cs_product(csr))
    dk = ctx.load_key(params)
    z = fx.witness_mont(z_in, z_aux)
    a, b, c, aa, bi, ba = ctx.synthesize(params.r1cs, z)
    r, s = fx.mont_fr(1), fx.mont_fr(2)
    # density map that selects a different number of points than the key holds
    bad = ba.copy(); bad[:] = 1 - bad
    with pytest.raises(fk.FkError) as e:
        ctx.prove_raw(dk, a, b, c, z, aa, bi, bad, r, s)
    assert e.value.code == 6
    # wrong row count for the key's domain
    with pytest.raises(fk.FkError) as e:
        ctx.prove_raw(dk, a[:5], b[:5], c[:5], z, aa, bi, ba, r, s)
    assert e.value.code == 6
    # delta = identity -> UnexpectedIdentity
    arrays = dict(m=key.m, num_input=key.num_input, num_aux=key.num_aux, alpha_g1=key.alpha_g1, beta_g1=key.beta_g1,
                  beta_g2=key.beta_g2, delta_g1=np.zeros(64, np.uint8), delta_g2=key.delta_g2,
                  h=np.array(key.h), l=np.array(key.l), a=np.array(key.a), b_g1=np.array(key.b_g1), b_g2=np.array(key.b_g2))
    dk0 = ctx.load_key(fk.Parameters(arrays))
    with pytest.raises(fk.FkError) as e:
        ctx.prove_raw(dk0, a, b, c, z, aa, bi, ba, r, s)
    assert e.value.code == 3
    # malformed key shapes
    arrays['h'] = arrays['h'][:-1]
    with pytest.raises(fk.FkError) as e:
        ctx.load_key(fk.Parameters(arrays))
    assert e.value.code == 6


def test_synthetic_key_full_pipeline_2_16(ctx):
    """Device-resident pipeline with a synthetic key at 2^16: runs, is deterministic, and differs when
    the witness changes (smoke for the bench path)."""
    m = 1 << 16
    v_in, v_aux = 4, m - 4
    n = m
    rng = np.random.default_rng(3)
    dens_a = (rng.integers(0, 10, v_aux) < 6).astype(np.uint8)
    dens_bi = np.ones(v_in, np.uint8)
    dens_ba = (rng.integers(0, 10, v_aux) < 6).astype(np.uint8)
    n_a = v_in + int(dens_a.sum()); n_b = int(dens_bi.sum()) + int(dens_ba.sum())
    key = ctx.synthetic_key(m, v_in, v_aux, n_a, n_b, seed=5)
    bufs = {k: ctx.dev_alloc(m * 32) for k in 'abc'}
    d_z = ctx.dev_alloc((v_in + v_aux) * 32)
    d_da, d_dbi, d_dba = ctx.dev_alloc(v_aux), ctx.dev_alloc(v_in), ctx.dev_alloc(v_aux)
    try:
        ctx.upload(d_da, dens_a); ctx.upload(d_dbi, dens_bi); ctx.upload(d_dba, dens_ba)
        ctx.gen_scalars_dev(d_z, v_in + v_aux, 9, 1)
        outs = []
        for seed in (1, 1, 2):
            for i, k in enumerate('abc'):
                ctx.gen_scalars_dev(bufs[k], n, seed * 10 + i, 0)
            outs.append(ctx.prove_dev(key, bufs['a'], bufs['b'], bufs['c'], n, d_z, d_da, d_dbi, d_dba,
                                      fx.mont_fr(3), fx.mont_fr(4)).tobytes())
        assert outs[0] == outs[1] and outs[0] != outs[2] and outs[0] != bytes(256)
    finally:
        for p in list(bufs.values()) + [d_z, d_da, d_dbi, d_dba]:
            ctx.dev_free(p)
        key.free()


def test_repeated_proofs_are_deterministic(ctx, oracle):
    """The five MSMs of a proof run on two streams with overlapped tails and reused scratch: 40 back-to-back proofs of
    alternating systems (so every buffer is reused with different sizes) must each equal the oracle's bytes."""
    import fawkes_crypto_amd as fk
    cases = []
    for seed, gates, nin, naux in ((901, 700, 2, 650), (902, 3000, 3, 3300), (903, 90, 1, 120)):
        cs, z_in, z_aux = ref.random_r1cs(seed, gates, nin, naux)
        csr = fx.r1cs_to_csr(cs)
        key = oracle.setup(csr, **TOXIC)
        params = params_from_oracle_key(key, r1cs_product(csr))
        z = fx.witness_mont(z_in, z_aux)
        r, s = fx.mont_fr(seed), fx.mont_fr(seed * 7)
        a, b, c, aa, bi, ba = oracle.synthesize(csr, z)
        want = oracle.prove(key, a, b, c, z, aa, bi, ba, r, s).tobytes()
        cases.append((ctx.load_key(params), ctx.load_r1cs(params.r1cs), z, r, s, want))
    for it in range(40):
        dk, dr, z, r, s, want = cases[it % 3]
        assert ctx.prove_witness(dk, dr, z, r, s).tobytes() == want, it
    for dk, dr, *_ in cases:
        dr.free(); dk.free()


def test_two_contexts_prove_concurrently_from_one_key(ctx, oracle):
    """A proving service keeps the device busy with several contexts (own streams and scratch) in as many host threads; the
    key and the resident constraint system are plain read-only device memory and are shared.  Same bytes from every thread."""
    import threading
    import fawkes_crypto_amd as fk
    csr, z, _, _ = fx.fast_r1cs(91, 30000, 2, 31000)
    r1cs = r1cs_product(csr)
    dk, _ = ctx.setup(r1cs, **{k: fx.mont_fr(v) for k, v in TOXIC.items()})
    dr = ctx.load_r1cs(r1cs)
    r, s = fx.mont_fr(31), fx.mont_fr(32)
    want = ctx.prove_witness(dk, dr, z, r, s).tobytes()
    others = [fk.Context(0) for _ in range(2)]
    got = {}

    def work(i, c):
        got[i] = [c.prove_witness(dk, dr, z, r, s).tobytes() for _ in range(6)]

    th = [threading.Thread(target=work, args=(i, c)) for i, c in enumerate([ctx] + others)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert len(got) == 3 and all(p == want for v in got.values() for p in v)
    for c in others:
        c.close()
    dr.free(); dk.free()
