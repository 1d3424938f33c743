"""Fixed-base precomputation of the key (fk_key_precomputed; msm.hip: merged bucket set): with the window levels
2^(offset_w) * P resident, all Pippenger windows share ONE bucket set.  Same group elements, so every proof must stay
bit-identical to the oracle's.  By default only arrays of >= 2^21 points get levels (round 4; 2^24 before); the tests lower the threshold
(FK_MSM_PRE_MIN_LOG2) so that the merged path -- G1 and G2, oversized buckets included -- runs on small keys."""
import numpy as np
import pytest

import bn254_ref as ref
import fixtures as fx
from helpers import params_from_oracle_key, r1cs_product, TOXIC

pytestmark = pytest.mark.gpu


@pytest.fixture
def low_threshold(monkeypatch):
    monkeypatch.setenv('FK_MSM_PRE_MIN_LOG2', '8')


def _instance(oracle, seed, gates, nin, naux):
    csr, z, z_in, z_aux = fx.fast_r1cs(seed, gates, nin, naux)
    key = oracle.setup(csr, **TOXIC)
    return csr, key, z, z_in


@pytest.mark.parametrize('shape', [(31, 1500, 2, 1400), (32, 20000, 3, 21000), (33, 40000, 2, 36000)])
def test_merged_path_bit_exact(ctx, oracle, low_threshold, shape):
    """half of fast_r1cs's aux values are 0 / 1: the bucket of digit 1 is far over the cap, so the oversized-bucket path runs
    in merged form too"""
    csr, key, z, z_in = _instance(oracle, *shape)
    params = params_from_oracle_key(key, r1cs_product(csr))
    dk = ctx.load_key(params)
    lev = dk.precomputed()
    assert all(v > 1 for v in lev.values()), lev
    dr = ctx.load_r1cs(params.r1cs)
    r, s = fx.mont_fr(0x1234 + shape[0]), fx.mont_fr(0x4321)
    a, b, c, aa, bi, ba = oracle.synthesize(csr, z)
    want = oracle.prove(key, a, b, c, z, aa, bi, ba, r, s).tobytes()
    assert ctx.prove_witness(dk, dr, z, r, s).tobytes() == want
    assert ctx.prove_witness(dk, dr, z, r, s).tobytes() == want          # lanes reused
    assert ref.verify(fx.key_to_py(key), z_in[1:], ref.proof_from_borsh(want))
    # the GPU-generated key of the same system carries levels as well
    dk2, _ = ctx.setup(params.r1cs, **{k: fx.mont_fr(v) for k, v in TOXIC.items()})
    assert all(v > 1 for v in dk2.precomputed().values())
    assert ctx.prove_witness(dk2, dr, z, r, s).tobytes() == want
    dk2.free(); dr.free(); dk.free()


def test_merged_shards_fold_to_the_same_proof(ctx, oracle, low_threshold):
    csr, key, z, _ = _instance(oracle, 41, 6000, 3, 6500)
    params = params_from_oracle_key(key, r1cs_product(csr))
    r, s = fx.mont_fr(5), fx.mont_fr(6)
    a, b, c, aa, bi, ba = oracle.synthesize(csr, z)
    want = oracle.prove(key, a, b, c, z, aa, bi, ba, r, s).tobytes()
    dk = ctx.load_key(params)
    parts = []
    for i in range(3):
        sk = ctx.load_key(params, shard_index=i, shard_count=3)
        assert all(v > 1 for v in sk.precomputed().values())
        parts.append(ctx.prove_msms(sk, a, b, c, z, aa, bi, ba))
        sk.free()
    assert ctx.prove_assemble(dk, np.stack(parts), r, s).tobytes() == want
    dk.free()


def test_switches(ctx, oracle, monkeypatch):
    csr, key, z, _ = _instance(oracle, 51, 3000, 2, 3100)
    params = params_from_oracle_key(key, r1cs_product(csr))
    dk = ctx.load_key(params)                       # default threshold: a key this small keeps the ordinary path
    assert not any(dk.precomputed().values())
    r, s = fx.mont_fr(9), fx.mont_fr(10)
    a, b, c, aa, bi, ba = oracle.synthesize(csr, z)
    want = ctx.prove_raw(dk, a, b, c, z, aa, bi, ba, r, s).tobytes()
    assert want == oracle.prove(key, a, b, c, z, aa, bi, ba, r, s).tobytes()
    monkeypatch.setenv('FK_MSM_PRE_MIN_LOG2', '8')
    dk1 = ctx.load_key(params)
    assert all(dk1.precomputed().values())
    assert ctx.prove_raw(dk1, a, b, c, z, aa, bi, ba, r, s).tobytes() == want
    monkeypatch.setenv('FK_MSM_PRECOMP', '0')
    dk0 = ctx.load_key(params)
    assert not any(dk0.precomputed().values())
    assert ctx.prove_raw(dk0, a, b, c, z, aa, bi, ba, r, s).tobytes() == want
    for k in (dk, dk0, dk1):
        k.free()


def test_merged_sharded_setup_and_weighted_shards(ctx, oracle, low_threshold):
    """keys made by fk_setup for a shard (the bench's multi-GPU path), by points and by weighted witness fractions"""
    csr, key, z, _ = _instance(oracle, 61, 5000, 2, 5200)
    r1cs = r1cs_product(csr)
    tox = {k: fx.mont_fr(v) for k, v in TOXIC.items()}
    r, s = fx.mont_fr(21), fx.mont_fr(22)
    a, b, c, aa, bi, ba = oracle.synthesize(csr, z)
    want = oracle.prove(key, a, b, c, z, aa, bi, ba, r, s).tobytes()
    dk, _ = ctx.setup(r1cs, **tox)
    for fracs in (None, [(0.0, 0.3), (0.3, 0.55), (0.55, 1.0)]):
        parts = []
        for i in range(3):
            sk, _ = ctx.setup(r1cs, shard_index=i, shard_count=3, z_frac=fracs[i] if fracs else (-1.0, -1.0), **tox)
            assert sk.precomputed()['h'] > 1
            parts.append(ctx.prove_msms(sk, a, b, c, z, aa, bi, ba))
            sk.free()
        assert ctx.prove_assemble(dk, np.stack(parts), r, s).tobytes() == want
    dk.free()


@pytest.mark.parametrize('levels', [True, False])
def test_one_multiplication_over_a_resident_key_array(ctx, oracle, monkeypatch, levels):
    """fk_prove_msm_array_dev (round 4: the micro entry point behind bench.py's standalone figures): ONE multiplication over each of
    the key's resident arrays -- with its fixed-base levels and without -- against the oracle's multiexp over the downloaded array,
    uniform and witness-like scalars; also for the shards of a key split by work (b_g1 and b_g2 sliced independently)."""
    import c_oracle as co
    import fawkes_crypto_amd as fk
    monkeypatch.setenv('FK_MSM_PRE_MIN_LOG2', '8' if levels else '30')
    csr, key, z, _ = _instance(oracle, 77, 5000, 3, 5200)
    params = params_from_oracle_key(key, r1cs_product(csr))
    for shard, count, split in ((0, 1, fk.api.Z_EQUAL_SPLIT), (1, 3, fk.api.Z_WORK_SPLIT), (2, 3, fk.api.Z_WORK_SPLIT)):
        dk = ctx.load_key(params, shard_index=shard, shard_count=count, z_frac=split)
        assert any(v > 1 for v in dk.precomputed().values()) == levels
        info = dk.shard_info()
        if count > 1:
            assert info['b'] != info['b_g2']
        for arr, cnt_key in (('h', 'h'), ('l', 'l'), ('a', 'a'), ('b_g1', 'b'), ('b_g2', 'b_g2')):
            n = info[cnt_key][1] - info[cnt_key][0]
            bases = dk.download(arr)
            assert bases.shape == (n, 128 if arr == 'b_g2' else 64)
            if n == 0:
                continue
            d_s = ctx.dev_alloc(n * 32)
            for kind in (0, 1):
                ctx.gen_scalars_dev(d_s, n, 900 + kind, kind)
                sc = ctx.download(d_s, n * 32, np.uint64).reshape(-1, 4)
                got = ctx.prove_msm_array_dev(dk, arr, d_s)
                want = (co.msm_g2 if arr == 'b_g2' else co.msm_g1)(bases, sc)
                assert got.tobytes() == np.asarray(want).tobytes(), (arr, kind, shard, count)
            ctx.dev_free(d_s)
        dk.free()
