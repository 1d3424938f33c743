"""The verifier half (fk_verify / fk_verify_batch_dev, csrc/pairing.hpp: `verifier::verify`, verifier.rs:75-81) against the
oracle's independent big-int pairing verifier: same accept / reject on valid proofs, tampered proofs, wrong inputs and a
wrong key.  fk_verify is host code of the product library (no GPU needed); the batch kernel is a gpu test."""
import numpy as np
import pytest

import bn254_ref as ref
import fixtures as fx
from helpers import golden, golden_instance, TOXIC


def _vk_of(key):
    return dict(alpha_g1=key.alpha_g1, beta_g2=key.beta_g2, gamma_g2=key.gamma_g2, delta_g2=key.delta_g2, ic=np.array(key.ic))


def _instance(oracle, seed, gates, nin, naux):
    cs, z_in, z_aux = ref.random_r1cs(seed, gates, nin, naux)
    csr = fx.r1cs_to_csr(cs)
    key = oracle.setup(csr, **TOXIC)
    z = fx.witness_mont(z_in, z_aux)
    a, b, c, aa, bi, ba = oracle.synthesize(csr, z)
    proof = oracle.prove(key, a, b, c, z, aa, bi, ba, fx.mont_fr(seed * 7 + 1), fx.mont_fr(seed * 11 + 2))
    return key, z, z_in, proof


def test_host_verifier_matches_oracle_verifier(oracle):
    import fawkes_crypto_amd as fk
    from fawkes_crypto_amd import api
    g, cs, z_in, z_aux, tw, r, s = golden_instance()
    key = oracle.setup(fx.r1cs_to_csr(cs), **tw)
    vkb = api.vk_to_borsh(_vk_of(key))
    assert len(vkb) == 64 + 3 * 128 + 4 + 64 * cs.num_input
    proof = bytes.fromhex(g['proof'])
    inputs = fx.witness_mont(z_in, [])[1:]
    assert ref.verify(fx.key_to_py(key), z_in[1:], ref.proof_from_borsh(proof))
    assert api.verify(vkb, inputs, proof) is True
    # tampered proof coordinates (still canonical field elements): rejected, like the oracle's verifier
    for off in (0, 40, 64 + 5, 192 + 33):
        bad = bytearray(proof); bad[off] ^= 1
        try:
            want = ref.verify(fx.key_to_py(key), z_in[1:], ref.proof_from_borsh(bytes(bad)))
        except Exception:
            want = False
        assert want is False and api.verify(vkb, inputs, bytes(bad)) is False
    # a wrong public input
    wrong = inputs.copy(); wrong[0] = fx.mont_fr(12345)
    assert api.verify(vkb, wrong, proof) is False
    # input count != #ic - 1: bellman's MalformedVerifyingKey
    with pytest.raises(fk.FkError) as e:
        api.verify(vkb, inputs[:-1] if len(inputs) > 1 else np.zeros((len(inputs) + 1, 4), np.uint64), proof)
    assert e.value.code == 6
    # a coordinate that is not a field element: Num<Fq>::deserialize fails
    bad = bytearray(proof); bad[0:32] = ref.Q.to_bytes(32, 'little')
    with pytest.raises(fk.FkError) as e:
        api.verify(vkb, inputs, bytes(bad))
    assert e.value.code == 7
    # the identity as a proof point (all-zero bytes, group.rs:55) is a legal encoding and is rejected by the equation
    assert api.verify(vkb, inputs, bytes(64) + proof[64:]) is False


@pytest.mark.parametrize('shape', [(3, 30, 1, 35), (4, 200, 5, 210)])
def test_host_verifier_random_instances(oracle, shape):
    from fawkes_crypto_amd import api
    seed, gates, nin, naux = shape
    key, z, z_in, proof = _instance(oracle, seed, gates, nin, naux)
    vkb = api.vk_to_borsh(_vk_of(key))
    assert api.verify(vkb, z[1:nin], proof.tobytes()) is True
    # the proof of another instance under this key's vk
    key2, z2, _, proof2 = _instance(oracle, seed + 100, gates, nin, naux)
    assert api.verify(vkb, z[1:nin], proof2.tobytes()) is False
    assert api.verify(api.vk_to_borsh(_vk_of(key2)), z2[1:nin], proof2.tobytes()) is True


def test_host_verifier_on_the_committed_circuit_proofs(oracle):
    """poseidon merkle proof (BASELINE configs[0]) and the rollup-style transaction: the committed golden proofs verify"""
    import fawkes_circuit as fc
    import random
    from fawkes_crypto_amd import api
    g = golden('rollup_tx_golden.json')
    rnd = random.Random(g['seed'])
    sibling, path = [rnd.randrange(ref.R) for _ in range(32)], [rnd.randrange(2) for _ in range(32)]
    cs = fc.rollup_tx_circuit(int(g['sk'], 16), g['bal_old'], g['bal_new'], sibling, path, int(g['rho'], 16))
    key = oracle.setup(fx.r1cs_to_csr(cs.r1cs()), **TOXIC)
    vkb = api.vk_to_borsh(_vk_of(key))
    inputs = fx.witness_mont(cs.z_in, [])[1:]
    assert api.verify(vkb, inputs, bytes.fromhex(g['proof'])) is True
    assert api.verify(vkb, inputs[::-1].copy(), bytes.fromhex(g['proof'])) is False      # old and new root swapped


@pytest.mark.gpu
def test_batch_verifier_on_the_gpu(ctx, oracle):
    """one lane per proof: 70 proofs of one key (more than a wave), every third one tampered with"""
    from fawkes_crypto_amd import api
    key, z, z_in, proof = _instance(oracle, 9, 60, 3, 70)
    vkb = api.vk_to_borsh(_vk_of(key))
    n = 70
    proofs = np.tile(proof, (n, 1))
    inputs = np.tile(z[1:3], (n, 1, 1))
    want = np.ones(n, bool)
    for i in range(0, n, 3):
        if i % 2:
            proofs[i, 200] ^= 4
        else:
            inputs[i, 1] = fx.mont_fr(i + 5)
        want[i] = False
    got = api.verify_batch(ctx, vkb, inputs, proofs)
    assert np.array_equal(got, want)
    assert all(api.verify(vkb, inputs[i], proofs[i].tobytes()) == bool(want[i]) for i in (0, 1, 3, 4))
    # a proof that does not decode -- a coordinate >= q, found by tools/fuzz_parity.py -- is rejected as one proof of the batch;
    # the verdicts on the others stand (the single-proof entry point refuses it as malformed, like upstream's Borsh reader)
    proofs[7, 0:32] = 0xff
    want[7] = False
    assert np.array_equal(api.verify_batch(ctx, vkb, inputs, proofs), want)
    with pytest.raises(api.FkError) as e:
        api.verify(vkb, inputs[7], proofs[7].tobytes(), ctx)
    assert 'FORMAT' in str(e.value)


def test_host_verifier_rejects_points_outside_the_groups(oracle):
    """proof points that are canonical field elements but not group elements -- A or C off the curve, B on the twist but outside
    the order-r subgroup -- are rejected outright (the pairing loop assumes prime-order points); honest proofs are unaffected"""
    from fawkes_crypto_amd import api
    key, z, z_in, proof = _instance(oracle, 9, 40, 2, 45)
    vkb = api.vk_to_borsh(_vk_of(key))
    inputs = z[1:2]
    assert api.verify(vkb, inputs, proof.tobytes()) is True
    A, B, C = ref.proof_from_borsh(proof.tobytes())
    assert api.verify(vkb, inputs, ref.proof_borsh((A[0], (A[1] + 1) % ref.Q), B, C)) is False        # A off the curve
    assert api.verify(vkb, inputs, ref.proof_borsh(A, B, (C[0], (C[1] + 1) % ref.Q))) is False        # C off the curve
    # a point ON the twist that is not in the order-r subgroup (the twist's cofactor is large: almost every curve point)
    F2, b2 = ref.F2, ref.G2.b

    def fq_sqrt(v):
        y = pow(v, (ref.Q + 1) // 4, ref.Q)
        return y if y * y % ref.Q == v % ref.Q else None

    def fq2_sqrt(a0, a1):
        alpha = fq_sqrt((a0 * a0 + a1 * a1) % ref.Q)
        if alpha is None:
            return None
        for d in ((a0 + alpha) * pow(2, -1, ref.Q) % ref.Q, (a0 - alpha) * pow(2, -1, ref.Q) % ref.Q):
            x0 = fq_sqrt(d)
            if x0:
                return x0, a1 * pow(2 * x0, -1, ref.Q) % ref.Q
        return None
    x = (11, 3)
    while True:
        rhs = F2.add(F2.mul(F2.sqr(x), x), b2)
        y = fq2_sqrt(*rhs)
        if y is not None and F2.sqr(y) == rhs:
            break
        x = (x[0] + 1, x[1])
    assert ref.G2.on_curve((x, y)) and ref.G2.mul((x, y), ref.R) is not None
    assert api.verify(vkb, inputs, ref.proof_borsh(A, (x, y), C)) is False
    assert api.verify(vkb, inputs, ref.proof_borsh(A, (B[0], (B[1][0], (B[1][1] + 1) % ref.Q)), C)) is False   # B off the twist
