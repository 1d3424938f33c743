"""GPU parity: Fr batch multiply, NTT / iNTT / coset variants and the quotient against the oracle.
Bit-exact (integer arithmetic)."""
import numpy as np
import pytest

import bn254_ref as ref
import fixtures as fx
from helpers import R, golden, mont_ints, rand_fr_mont

pytestmark = pytest.mark.gpu


def test_fr_mul_batch(ctx, oracle):
    rng = np.random.default_rng(1)
    a, b = rand_fr_mont(rng, 3000), rand_fr_mont(rng, 3000)
    edge = oracle.limbs_arr([0, ref.to_mont(1, R), ref.to_mont(R - 1, R), R - 1])  # incl. the largest limb image
    a[:4], b[:4] = edge, edge[::-1]
    assert np.array_equal(ctx.fr_mul_batch(a, b), oracle.fe_mul_batch(oracle.FR, a, b))


def test_ntt_golden(ctx, oracle):
    for case in golden('ntt_golden.json')['cases']:
        vm = oracle.limbs_arr([ref.to_mont(int(x, 16), R) for x in case['input']])
        for key, kw in (('forward', {}), ('inverse', dict(inverse=True)), ('coset_forward', dict(coset=True)),
                        ('coset_inverse', dict(inverse=True, coset=True))):
            assert mont_ints(ctx.ntt(vm, **kw)) == [int(x, 16) for x in case[key]], (case['log_n'], key)


@pytest.mark.parametrize('log_n', [0, 1, 2, 3, 5, 8, 9, 10, 11, 13, 14, 16])
def test_ntt_vs_oracle(ctx, oracle, log_n):
    rng = np.random.default_rng(100 + log_n)
    v = rand_fr_mont(rng, 1 << log_n)
    for kw in ({}, dict(inverse=True), dict(coset=True), dict(inverse=True, coset=True)):
        assert np.array_equal(ctx.ntt(v, **kw), oracle.fr_ntt(v, **kw)), (log_n, kw)


def test_ntt_roundtrip_large(ctx):
    """BASELINE configs[1] size (2^20) and beyond: iNTT(NTT(x)) == x, coset variants too."""
    for log_n in (18, 20, 22):
        rng = np.random.default_rng(log_n)
        # cheap uniform-looking Montgomery images: random limbs with the top limb cleared (< r)
        v = rng.integers(0, 1 << 63, size=(1 << log_n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(1 << log_n, 4), dtype=np.uint64)
        v[:, 3] &= np.uint64((1 << 60) - 1)
        f = ctx.ntt(v)
        assert not np.array_equal(f, v)
        assert np.array_equal(ctx.ntt(f, inverse=True), v)
        assert np.array_equal(ctx.ntt(ctx.ntt(v, coset=True), inverse=True, coset=True), v)


def test_ntt_linearity_2_20(ctx, oracle):
    """NTT(a) + NTT(b) == NTT(a + b) at 2^20, spot-checked on 64 positions with oracle field adds."""
    n = 1 << 20
    rng = np.random.default_rng(7)
    a = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); a[:, 3] &= np.uint64((1 << 59) - 1)
    b = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); b[:, 3] &= np.uint64((1 << 59) - 1)
    # a + b without modular wrap: limbs < 2^62 so limb-wise addition has no carries and stays < r
    fa, fb, fab = ctx.ntt(a), ctx.ntt(b), ctx.ntt(a + b)
    idx = rng.integers(0, n, 64)
    for i in idx:
        assert np.array_equal(oracle.fe_add(oracle.FR, fa[i], fb[i]), fab[i])


@pytest.mark.parametrize('n', [1, 2, 3, 5, 33, 257, 1000, 4096, 7364])
def test_quotient_vs_oracle(ctx, oracle, n):
    """ragged n (not a power of two) incl. BASELINE configs[0]'s 7364 rows"""
    rng = np.random.default_rng(n)
    a, b, c = rand_fr_mont(rng, n, 'witness'), rand_fr_mont(rng, n), rand_fr_mont(rng, n)
    assert np.array_equal(ctx.quotient_h(a, b, c), oracle.quotient_h(a, b, c))


def test_domain_too_large(ctx):
    """bellman: PolynomialDegreeTooLarge once exp >= S = 28"""
    import fawkes_crypto_amd as fk
    with pytest.raises(fk.FkError) as e:
        ctx.lib.fk_ntt  # symbol exists
        ctx._ck(ctx.lib.fk_ntt_dev(ctx.handle, 1, 28, 0, 0))
    assert e.value.code == 2
