"""External known-answer vectors for the curve arithmetic and the pairing: the public EIP-196 / EIP-197 (alt_bn128 add, mul,
pairing check) test vectors (tests/golden/eip196_197_kats.json) through
  * oracle/bn254_ref.py   (big-int restatement: the checker of every GPU parity test),
  * oracle/groth16_oracle.c (the C restatement: add, scalar multiplication, multiexp),
  * the product's own pairing (csrc/pairing.hpp behind fk_verify -- host code of the library, no GPU needed), and
  * under -m gpu the HIP multi-scalar multiplication (fk_msm_g1).
The reference holds no vector for this path (tests/bellman_groth16.rs:45-46 asserts `verify == true` only); these are the
nearest public facts about the same curve that neither the oracle nor the product was written from."""
import numpy as np
import pytest

import bn254_ref as ref
import fixtures as fx
from helpers import golden

K = golden('eip196_197_kats.json')
Q, R = ref.Q, ref.R


def words(h):
    return [int(h[i:i + 64], 16) for i in range(0, len(h), 64)]


def g1(x, y):
    return None if (x, y) == (0, 0) else (x, y)


def g2(xi, xr, yi, yr):          # EIP-197 order: imaginary part first
    return None if (xi, xr, yi, yr) == (0, 0, 0, 0) else ((xr, xi), (yr, yi))


def pairs(h):
    w = words(h)
    assert len(w) % 6 == 0
    return [(g1(*w[o:o + 2]), g2(*w[o + 2:o + 6])) for o in range(0, len(w), 6)]


@pytest.mark.parametrize('case', K['ecadd'], ids=lambda c: c['name'])
def test_ecadd_python_and_c(oracle, case):
    w, o = words(case['in']), words(case['out'])
    P, S, want = g1(*w[0:2]), g1(*w[2:4]), g1(*o)
    assert ref.G1.on_curve(P) and ref.G1.on_curve(S) and ref.G1.on_curve(want)
    assert ref.G1.add(P, S) == want
    got = oracle.g1_add(np.frombuffer(ref.g1_raw_le(P), np.uint8), np.frombuffer(ref.g1_raw_le(S), np.uint8))
    assert ref.g1_from_raw_le(got.tobytes()) == want
    # ... and as a two-term multiexp with scalars 1, 1 (bellman's "scalar one" branch)
    bases = np.stack([np.frombuffer(ref.g1_raw_le(P), np.uint8), np.frombuffer(ref.g1_raw_le(S), np.uint8)])
    assert ref.g1_from_raw_le(oracle.msm_g1(bases, np.stack([fx.mont_fr(1), fx.mont_fr(1)])).tobytes()) == want


@pytest.mark.parametrize('case', K['ecmul'], ids=lambda c: c['name'])
def test_ecmul_python_and_c(oracle, case):
    w, o = words(case['in']), words(case['out'])
    P, k, want = g1(*w[0:2]), w[2], g1(*o)
    assert ref.G1.on_curve(P) and ref.G1.on_curve(want)
    assert ref.G1.mul(P, k % R) == want              # the precompile takes any 256-bit scalar; the group has order r
    praw = np.frombuffer(ref.g1_raw_le(P), np.uint8)
    assert ref.g1_from_raw_le(oracle.g1_mul(praw, fx.mont_fr(k % R)).tobytes()) == want
    assert ref.g1_from_raw_le(oracle.msm_g1(praw.reshape(1, 64), fx.mont_fr(k % R).reshape(1, 4)).tobytes()) == want


def _as_groth16_instance(ps):
    """prod e(a_i, b_i) == 1 for one or two pairs, phrased as a Groth16 verification: e(A, B) = e(alpha, beta) e(acc, gamma)
    e(C, delta) with A = a_1, B = b_1, alpha = -a_2, beta = b_2, acc = C = infinity.  Returns (vk dict, proof bytes)."""
    (a1, b1) = ps[0]
    (a2, b2) = ps[1] if len(ps) > 1 else (None, None)
    raw1 = lambda P: np.frombuffer(ref.g1_raw_le(P), np.uint8)
    raw2 = lambda P: np.frombuffer(ref.g2_raw_le(P), np.uint8)
    vk = dict(alpha_g1=raw1(ref.G1.neg(a2)), beta_g2=raw2(b2), gamma_g2=raw2(ref.G2_GEN), delta_g2=raw2(ref.G2_GEN),
              ic=raw1(None).reshape(1, 64))
    return vk, ref.proof_borsh(a1, b1, None)


@pytest.mark.parametrize('case', K['pairing'], ids=lambda c: c['name'])
def test_pairing_check_python_and_product_verifier(case):
    ps = pairs(case['in'])
    for a, b in ps:
        assert ref.G1.on_curve(a) and ref.G2.on_curve(b) and ref.G2.mul(b, R) is None
    f = ref.f12_one()
    for a, b in ps:
        f = ref.f12_mul(f, ref.miller_loop(b, a))
    assert (ref.final_exp(f) == ref.f12_one()) == bool(case['expect'])
    # the product's pairing (csrc/pairing.hpp, a different pairing variant and tower) decides the same
    import fawkes_crypto_amd as fk
    vk, proof = _as_groth16_instance(ps)
    assert fk.api.verify(fk.api.vk_to_borsh(vk), np.zeros((0, 4), np.uint64), proof) is bool(case['expect'])


@pytest.mark.gpu
def test_ecadd_ecmul_on_the_gpu(ctx):
    """the same vectors through the HIP multi-scalar multiplication (fk_msm_g1)"""
    for case in K['ecadd']:
        w, o = words(case['in']), words(case['out'])
        bases = np.stack([np.frombuffer(ref.g1_raw_le(g1(*w[0:2])), np.uint8), np.frombuffer(ref.g1_raw_le(g1(*w[2:4])), np.uint8)])
        got = ctx.msm_g1(bases, np.stack([fx.mont_fr(1), fx.mont_fr(1)]))
        assert ref.g1_from_raw_le(got.tobytes()) == g1(*o), case['name']
    for case in K['ecmul']:
        w, o = words(case['in']), words(case['out'])
        got = ctx.msm_g1(np.frombuffer(ref.g1_raw_le(g1(*w[0:2])), np.uint8).reshape(1, 64), fx.mont_fr(w[2] % R).reshape(1, 4))
        assert ref.g1_from_raw_le(got.tobytes()) == g1(*o), case['name']


@pytest.mark.gpu
def test_pairing_check_on_the_gpu(ctx):
    """... and the pairing vectors through the batch verifier kernel (fk_verify_batch_dev)"""
    import fawkes_crypto_amd as fk
    for case in K['pairing']:
        vk, proof = _as_groth16_instance(pairs(case['in']))
        acc = fk.api.verify_batch(ctx, fk.api.vk_to_borsh(vk), np.zeros((3, 0, 4), np.uint64), np.frombuffer(proof * 3, np.uint8))
        assert list(acc) == [bool(case['expect'])] * 3, case['name']
