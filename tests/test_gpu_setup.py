"""GPU key generation against the oracle's generate_parameters restatement, and the
payoff: VALID keys at BASELINE's sizes, so a GPU proof at 2^20 constraints is checked by the Groth16 pairing
equation (the reference's own acceptance test, fawkes-crypto/tests/bellman_groth16.rs:45-46)."""
import time

import numpy as np
import pytest

import bn254_ref as ref
import fixtures as fx
from helpers import r1cs_product, TOXIC

pytestmark = pytest.mark.gpu


def _toxic_mont():
    return {k: fx.mont_fr(v) for k, v in TOXIC.items()}


@pytest.mark.parametrize('shape', [(1, 3, 1, 5), (2, 60, 3, 70), (3, 900, 2, 1000)])
def test_setup_vs_oracle(ctx, oracle, shape):
    seed, gates, nin, naux = shape
    cs, z_in, z_aux = ref.random_r1cs(seed, gates, nin, naux)
    csr = fx.r1cs_to_csr(cs)
    want = oracle.setup(csr, **TOXIC)
    dk, vk = ctx.setup(r1cs_product(csr), **_toxic_mont())
    for name in ('h', 'l', 'a', 'b_g1', 'b_g2'):
        assert dk.download(name).tobytes() == np.array(getattr(want, name)).tobytes(), name
    for name in ('alpha_g1', 'beta_g1', 'beta_g2', 'gamma_g2', 'delta_g1', 'delta_g2'):
        assert vk[name].tobytes() == getattr(want, name).tobytes(), name
    assert vk['ic'].tobytes() == np.array(want.ic).tobytes()
    # and the key proves: same bytes as the oracle prover with the oracle key
    z = fx.witness_mont(z_in, z_aux)
    a, b, c, aa, bi, ba = oracle.synthesize(csr, z)
    r, s = fx.mont_fr(11), fx.mont_fr(22)
    assert ctx.prove_raw(dk, a, b, c, z, aa, bi, ba, r, s).tobytes() == oracle.prove(want, a, b, c, z, aa, bi, ba, r, s).tobytes()


def test_setup_heavy_column_vs_oracle(ctx, oracle):
    """The constant ONE occurs in thousands of B-side LCs (boolean constraints b*(b-1)=0): the transposed sparse
    product takes the segmented heavy-column path; result must still equal the oracle's key."""
    import c_oracle as co
    nin, nbits = 2, 9000
    one, minus_one = fx.mont_fr(1), fx.mont_fr(-1)
    seq = np.arange(nbits + 1, dtype=np.uint64)
    a = co.Csr(seq, (nin + np.arange(nbits)).astype(np.uint32), np.tile(one, (nbits, 1)))
    b_col = np.zeros(2 * nbits, np.uint32); b_col[0::2] = nin + np.arange(nbits); b_col[1::2] = 0
    b_val = np.tile(one, (2 * nbits, 1)); b_val[1::2] = minus_one
    b = co.Csr(2 * seq, b_col, b_val)
    c = co.Csr(np.zeros(nbits + 1, np.uint64), np.zeros(0, np.uint32), np.zeros((0, 4), np.uint64))
    csr = co.R1csC(nin, nbits, a, b, c)
    want = oracle.setup(csr, **TOXIC)
    dk, vk = ctx.setup(r1cs_product(csr), **_toxic_mont())
    for name in ('h', 'l', 'a', 'b_g1', 'b_g2'):
        assert dk.download(name).tobytes() == np.array(getattr(want, name)).tobytes(), name
    assert vk['ic'].tobytes() == np.array(want.ic).tobytes()


def _vk_to_py(vk, nin):
    g1 = lambda b: ref.g1_from_raw_le(bytes(b))
    g2 = lambda b: ref.g2_from_raw_le(bytes(b))
    return dict(alpha_g1=g1(vk['alpha_g1']), beta_g1=g1(vk['beta_g1']), beta_g2=g2(vk['beta_g2']), gamma_g2=g2(vk['gamma_g2']),
                delta_g1=g1(vk['delta_g1']), delta_g2=g2(vk['delta_g2']), ic=[g1(r.tobytes()) for r in vk['ic']])


@pytest.mark.parametrize('log2n', [16, 20])
def test_full_size_proof_verifies(ctx, oracle, log2n):
    """BASELINE configs[1] size: 2^20 rows.  GPU setup -> resident R1CS -> witness in, proof out -> the
    Groth16 pairing equation holds (python big-int verifier), and fails for a wrong public input."""
    nin = 3
    gates = (1 << log2n) - nin
    naux = (1 << log2n)
    t0 = time.time()
    cs, z, z_in, z_aux = fx.fast_r1cs(99 + log2n, gates, nin, naux)
    r1cs = r1cs_product(cs)
    dk, vk = ctx.setup(r1cs, **_toxic_mont())
    assert dk.shard_info()['h'] == (0, (1 << log2n) - 1)
    # vk points are plain scalar multiples of the generators: cross-check with the oracle
    gen1 = np.frombuffer(ref.g1_raw_le(ref.G1_GEN), np.uint8)
    gen2 = np.frombuffer(ref.g2_raw_le(ref.G2_GEN), np.uint8)
    assert vk['alpha_g1'].tobytes() == oracle.g1_mul(gen1, fx.mont_fr(TOXIC['alpha'])).tobytes()
    assert vk['delta_g2'].tobytes() == oracle.g2_mul(gen2, fx.mont_fr(TOXIC['delta'])).tobytes()
    assert vk['gamma_g2'].tobytes() == oracle.g2_mul(gen2, fx.mont_fr(TOXIC['gamma'])).tobytes()
    dr = ctx.load_r1cs(r1cs)
    r, s = fx.mont_fr(0x314159), fx.mont_fr(0x271828)
    proof = ctx.prove_witness(dk, dr, z, r, s)
    pk = _vk_to_py(vk, nin)
    P = ref.proof_from_borsh(proof.tobytes())
    assert ref.verify(pk, z_in[1:], P)
    assert not ref.verify(pk, [(z_in[1] + 1) % ref.R] + z_in[2:], P)
    # deterministic for fixed (r, s); different (r, s) -> different but still valid proof
    assert ctx.prove_witness(dk, dr, z, r, s).tobytes() == proof.tobytes()
    p2 = ctx.prove_witness(dk, dr, z, fx.mont_fr(5), fx.mont_fr(6))
    assert p2.tobytes() != proof.tobytes() and ref.verify(pk, z_in[1:], ref.proof_from_borsh(p2.tobytes()))
    dr.free(); dk.free()
    print('2^%d end-to-end in %.1f s' % (log2n, time.time() - t0))
