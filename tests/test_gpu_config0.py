"""BASELINE configs[0] on the GPU: the reference's own Groth16 test (tests/bellman_groth16.rs:19-48) --
setup(circuit) -> Parameters file -> prove(params, root, (leaf, merkle proof), circuit) -> verify -- with the
circuit produced by the restated DSL (oracle/fawkes_circuit.py) and every heavy step on the HIP path:
fk_setup, the Parameters key file loader, SpMV synthesis, quotient, the five MSMs, assembly."""
import random

import numpy as np
import pytest

import bn254_ref as ref
import fawkes_circuit as fc
import fixtures as fx
from helpers import golden, r1cs_product, TOXIC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def instance():
    g = golden('poseidon_merkle_golden.json')
    rnd = random.Random(g['seed'])
    leaf = rnd.randrange(ref.R)
    sib = [rnd.randrange(ref.R) for _ in range(32)]
    path = [rnd.randrange(2) for _ in range(32)]
    cs, root = fc.poseidon_merkle_circuit(leaf, sib, path)
    return g, cs, root, fx.r1cs_to_csr(cs.r1cs())


def test_config0_bit_exact_and_verifies(ctx, oracle, instance):
    import fawkes_crypto_amd as fk
    from fawkes_crypto_amd import params_io as pio
    g, cs, root, csr = instance
    r1cs = r1cs_product(csr)
    # setup(circuit) on the GPU == the oracle's generate_parameters restatement
    want_key = oracle.setup(csr, **TOXIC)
    dk, vk = ctx.setup(r1cs, **{k: fx.mont_fr(v) for k, v in TOXIC.items()})
    assert dk.counts()['m'] == 1 << 13
    arrays = {}
    for name in ('h', 'l', 'a', 'b_g1', 'b_g2'):
        arrays[name] = dk.download(name)
        assert arrays[name].tobytes() == np.array(getattr(want_key, name)).tobytes(), name
    assert vk['ic'].tobytes() == np.array(want_key.ic).tobytes()
    # Parameters::write -> Parameters::read (mod.rs:144-175): gate stream + const tracker + bellman key file
    for name in ('alpha_g1', 'beta_g1', 'beta_g2', 'gamma_g2', 'delta_g1', 'delta_g2'):
        arrays[name] = vk[name]
    arrays.update(ic=vk['ic'], m=1 << 13, num_input=2, num_aux=cs.num_aux)
    from helpers import brotli_compress
    # the reference's blob format (brotli quality 9 / lgwin 22, setup.rs:26) when the system has the encoder, else the raw stream
    data = pio.store_parameters(arrays, r1cs, const_tracker_bits=cs.const_tracker, compress=brotli_compress if brotli_compress(b'x') else None)
    dk2, dr, hdr = pio.load_parameters(ctx, data, checked=True, disallow_points_at_infinity=False, want_host_r1cs=True)
    r1cs2 = hdr['r1cs']
    assert hdr['num_gates'] == 7362 and hdr['const_tracker'] == cs.const_tracker and hdr['gates_info']['nnz'] == tuple(len(c) for _, c, _ in r1cs.mats)
    params = fk.Parameters(dict(arrays), r1cs2)
    z_in, z_aux = fx.witness_mont(cs.z_in, []), fx.witness_mont([], cs.z_aux)
    r, s = fx.mont_fr(int(g['r'], 16)), fx.mont_fr(int(g['s'], 16))
    # prove: resident constraint system, and the host-synthesis entry, both against the committed golden proof
    inputs, proof = fk.prove_with_rs(ctx, params, dk2, z_in, z_aux, r, s, device_r1cs=dr)
    assert proof.to_bytes().hex() == g['proof']
    inputs_b, proof_b = fk.prove_with_rs(ctx, params, dk2, z_in, z_aux, r, s)
    assert proof_b.to_bytes() == proof.to_bytes()
    assert ref.from_mont(int(fx.co.ints(inputs)[0]), ref.R) == root
    # verify(vk, proof, inputs): tests/bellman_groth16.rs:45-46
    pk = fx.key_to_py(want_key)
    assert ref.verify(pk, [root], ref.proof_from_borsh(proof.to_bytes()))
    # random (r, s) like create_random_proof: different bytes, still accepted
    _, p3 = fk.prove(ctx, params, dk2, z_in, z_aux, device_r1cs=dr)
    assert p3.to_bytes() != proof.to_bytes() and ref.verify(pk, [root], ref.proof_from_borsh(p3.to_bytes()))
    dr.free(); dk.free(); dk2.free()


def test_config0_wrong_witness_is_rejected_by_verifier(ctx, oracle, instance):
    """a witness that violates the circuit (flipped path bit) still yields 256 bytes, but not a valid proof"""
    g, cs, root, csr = instance
    r1cs = r1cs_product(csr)
    dk, vk = ctx.setup(r1cs, **{k: fx.mont_fr(v) for k, v in TOXIC.items()})
    dr = ctx.load_r1cs(r1cs)
    z_aux = list(cs.z_aux)
    z_aux[40] ^= 1
    z = fx.witness_mont(cs.z_in, z_aux)
    proof = ctx.prove_witness(dk, dr, z, fx.mont_fr(3), fx.mont_fr(4))
    pk = fx.key_to_py(oracle.setup(csr, **TOXIC))
    try:
        ok = ref.verify(pk, [root], ref.proof_from_borsh(proof.tobytes()))
    except Exception:
        ok = False
    assert not ok
    dr.free(); dk.free()
