"""GPU parity: G1 / G2 Pippenger MSM against the oracle (bellman's multiexp restated).  Bit-exact."""
import numpy as np
import pytest

import bn254_ref as ref
import fixtures as fx
from helpers import R, golden, rand_fr_mont, g1_bases, g2_bases

pytestmark = pytest.mark.gpu


def test_msm_golden(ctx, oracle):
    g = golden('msm_golden.json')
    sm = oracle.limbs_arr([ref.to_mont(int(x, 16), R) for x in g['g1_scalars']])
    b1 = np.frombuffer(bytes.fromhex(''.join(g['g1_bases'])), np.uint8).reshape(-1, 64)
    assert ctx.msm_g1(b1, sm).tobytes().hex() == g['g1_result']
    b2 = np.frombuffer(bytes.fromhex(''.join(g['g2_bases'])), np.uint8).reshape(-1, 128)
    assert ctx.msm_g2(b2, sm[:len(b2)]).tobytes().hex() == g['g2_result']


@pytest.mark.parametrize('n', [0, 1, 2, 31, 32, 33, 1000, 4096])
def test_msm_g1_vs_oracle(ctx, oracle, n):
    rng = np.random.default_rng(n + 1)
    bases = g1_bases(n, seed=n) if n else np.zeros((0, 64), np.uint8)
    for kind in ('uniform', 'witness'):
        sc = rand_fr_mont(rng, n, kind)
        assert ctx.msm_g1(bases, sc).tobytes() == oracle.msm_g1(bases, sc).tobytes(), (n, kind)


@pytest.mark.parametrize('n', [0, 1, 3, 64, 700])
def test_msm_g2_vs_oracle(ctx, oracle, n):
    rng = np.random.default_rng(n + 11)
    bases = g2_bases(n, seed=n) if n else np.zeros((0, 128), np.uint8)
    for kind in ('uniform', 'witness'):
        sc = rand_fr_mont(rng, n, kind)
        assert ctx.msm_g2(bases, sc).tobytes() == oracle.msm_g2(bases, sc).tobytes(), (n, kind)


def test_msm_edge_cases(ctx, oracle):
    """all-zero scalars, all-one scalars, scalar r-1, infinity bases, identical bases (doubling branch),
    P and -P in one bucket (cancellation to infinity), result = infinity."""
    n = 200
    bases = g1_bases(n, seed=9)
    zero = np.zeros((n, 4), np.uint64)
    assert ctx.msm_g1(bases, zero).tobytes() == bytes(64)
    one = np.tile(fx.mont_fr(1), (n, 1))
    assert ctx.msm_g1(bases, one).tobytes() == oracle.msm_g1(bases, one).tobytes()
    rm1 = np.tile(fx.mont_fr(R - 1), (n, 1))
    assert ctx.msm_g1(bases, rm1).tobytes() == oracle.msm_g1(bases, rm1).tobytes()
    rng = np.random.default_rng(3)
    sc = rand_fr_mont(rng, n)
    b = bases.copy()
    b[5] = 0; b[17] = 0                      # infinity bases
    b[30:60] = b[30]                          # the same point 30 times
    sc[30:60] = sc[30]                        # ... with the same scalar: every bucket add is a doubling
    neg = ref.g1_from_raw_le(b[70].tobytes())
    b[71] = np.frombuffer(ref.g1_raw_le(ref.G1.neg(neg)), np.uint8)
    sc[71] = sc[70]                           # P and -P with equal scalars cancel
    assert ctx.msm_g1(b, sc).tobytes() == oracle.msm_g1(b, sc).tobytes()
    # whole MSM cancels to the identity
    b2 = np.stack([b[70], b[71]]); s2 = np.stack([sc[70], sc[70]])
    assert ctx.msm_g1(b2, s2).tobytes() == bytes(64)
    # G2 flavours of the same
    g2b = g2_bases(40, seed=2)
    g2b[3] = 0
    g2b[10:20] = g2b[10]
    s = rand_fr_mont(rng, 40); s[10:20] = s[10]
    assert ctx.msm_g2(g2b, s).tobytes() == oracle.msm_g2(g2b, s).tobytes()


def test_msm_oversized_bucket_path(ctx, oracle):
    """Skewed scalars: 6000 copies of the scalar 1 and 5000 of one random value force the
    oversized-bucket (segment + wave-shuffle fold) path, several segments per bucket."""
    n = 12000
    rng = np.random.default_rng(21)
    bases = g1_bases(n, seed=4)
    sc = rand_fr_mont(rng, n)
    sc[:6000] = fx.mont_fr(1)
    sc[6000:11000] = sc[6000]
    perm = rng.permutation(n)
    sc, bases = sc[perm], bases[perm]
    assert ctx.msm_g1(bases, sc).tobytes() == oracle.msm_g1(bases, sc).tobytes()
    g2b = g2_bases(5000, seed=5)
    s2 = rand_fr_mont(rng, 5000); s2[:4500] = fx.mont_fr(1)
    assert ctx.msm_g2(g2b, s2).tobytes() == oracle.msm_g2(g2b, s2).tobytes()


@pytest.mark.parametrize('c', [2, 5, 8, 13, 16, 17, 19, 21])
def test_msm_window_bits(ctx, oracle, c):
    n = 1500
    rng = np.random.default_rng(c)
    bases, sc = g1_bases(n, seed=6), rand_fr_mont(rng, n, 'witness')
    want = oracle.msm_g1(bases, sc).tobytes()
    ctx.set_window_bits(c)
    try:
        assert ctx.msm_g1(bases, sc).tobytes() == want
    finally:
        ctx.set_window_bits(0)


def test_msm_two_pass_sort_paths(ctx, oracle):
    """c > 16 goes through the two-pass radix sort (high bits, then low bits per segment): ragged sizes, skewed
    scalars that overflow one bucket, G1 and G2, several window sizes; FK_SORT2-style forcing is covered by c = 17."""
    rng = np.random.default_rng(1717)
    for n, c in ((1, 17), (77, 18), (5000, 17), (40000, 20), (40000, 22)):
        bases, sc = g1_bases(n, seed=n + c), rand_fr_mont(rng, n, 'witness')
        if n >= 5000:
            sc[: n // 3] = fx.mont_fr(1)            # one oversized bucket
            sc[n // 3: n // 2] = sc[n // 3]         # and another one in every window
        want = oracle.msm_g1(bases, sc).tobytes()
        ctx.set_window_bits(c)
        try:
            assert ctx.msm_g1(bases, sc).tobytes() == want, (n, c)
        finally:
            ctx.set_window_bits(0)
    g2b, s2 = g2_bases(3000, seed=8), rand_fr_mont(rng, 3000, 'witness')
    want = oracle.msm_g2(g2b, s2).tobytes()
    ctx.set_window_bits(18)
    try:
        assert ctx.msm_g2(g2b, s2).tobytes() == want
    finally:
        ctx.set_window_bits(0)


def test_msm_2_20_linearity_and_generator(ctx, oracle):
    """BASELINE configs[1] size: n = 2^20 with device-generated bases.  Size-independent properties:
    generated points are on the curve; MSM(s) + MSM(t) == MSM(s + t); MSM(k * e_i) == k * base_i."""
    n = 1 << 20
    d_b = ctx.dev_alloc(n * 64)
    d_s = ctx.dev_alloc(n * 32); d_t = ctx.dev_alloc(n * 32); d_u = ctx.dev_alloc(n * 32)
    try:
        ctx.gen_points_g1_dev(d_b, n, 77)
        pts = ctx.download(d_b, 64 * 4096).reshape(-1, 64)
        for row in pts[::257]:
            assert ref.G1.on_curve(ref.g1_from_raw_le(row.tobytes()))
        assert len({r.tobytes() for r in pts}) == len(pts)
        rng = np.random.default_rng(8)
        s = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); s[:, 3] &= np.uint64((1 << 59) - 1)
        t = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); t[:, 3] &= np.uint64((1 << 59) - 1)
        ctx.upload(d_s, s); ctx.upload(d_t, t); ctx.upload(d_u, s + t)   # limb-wise sums stay < r, no carries
        ps, pt, pu = ctx.msm_g1_dev(d_b, d_s, n), ctx.msm_g1_dev(d_b, d_t, n), ctx.msm_g1_dev(d_b, d_u, n)
        assert oracle.g1_add(ps, pt).tobytes() == pu.tobytes()
        e = np.zeros((n, 4), np.uint64); k = fx.mont_fr(0xdeadbeefcafe)
        e[123456] = k
        ctx.upload(d_s, e)
        base = ctx.download(d_b + 123456 * 64, 64)
        assert ctx.msm_g1_dev(d_b, d_s, n).tobytes() == oracle.g1_mul(base, k).tobytes()
    finally:
        for p in (d_b, d_s, d_t, d_u):
            ctx.dev_free(p)


def test_gen_points_g2_on_curve(ctx):
    n = 300
    d = ctx.dev_alloc(n * 128)
    try:
        ctx.gen_points_g2_dev(d, n, 5)
        pts = ctx.download(d, n * 128).reshape(-1, 128)
        for row in pts[::13]:
            assert ref.G2.on_curve(ref.g2_from_raw_le(row.tobytes()))
    finally:
        ctx.dev_free(d)


def test_msm_many_repeated_scalars(ctx, oracle):
    """6000 distinct scalars, 48 copies each: ~90 000 buckets exceed the statistical cap at once (a tiled batch witness does
    this).  The oversized-bucket list is sized for the worst case W * n / cap, so the path must neither fail nor truncate."""
    rng = np.random.default_rng(4848)
    base = rand_fr_mont(rng, 6000)
    sc = np.tile(base, (48, 1))
    n = sc.shape[0]
    bases = g1_bases(n, seed=9)
    assert ctx.msm_g1(bases, sc).tobytes() == oracle.msm_g1(bases, sc).tobytes()
