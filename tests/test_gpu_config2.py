"""BASELINE configs[2] on the GPU: the eddsa-poseidon signature check (circuit/eddsaposeidon.rs:17-47) produced by the
restated DSL (oracle/fawkes_circuit.py) -- one signature, and a batch of 16 signatures replicated into ONE constraint
system (the shape of "batch of 4096 signatures as one R1CS"; the full 4096 go through the tiled entry points,
tests/test_gpu_tiled.py).  LCs here are long (133 terms per gate on average, up to 512): this is the
workload that exercises the device SpMV, unlike the synthetic rollup shape."""
import random

import numpy as np
import pytest

import bn254_ref as ref
import fawkes_circuit as fc
import fixtures as fx
from helpers import r1cs_product, TOXIC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def signatures():
    rnd = random.Random(20261003)
    pp, jj = fc.PoseidonParams(4, 8, 54), fc.JubJubBN256()
    return [fc.eddsa_circuit(rnd.randrange(fc.FS), rnd.randrange(ref.R), rnd.randrange(fc.FS), pp, jj)[0] for _ in range(4)]


def _prove_and_check(ctx, oracle, csr, z, public):
    r1cs = r1cs_product(csr)
    dk, vk = ctx.setup(r1cs, **{k: fx.mont_fr(v) for k, v in TOXIC.items()})
    dr = ctx.load_r1cs(r1cs)
    r, s = fx.mont_fr(0xedd5a), fx.mont_fr(0x5eed)
    got = ctx.prove_witness(dk, dr, z, r, s)
    okey = oracle.setup(csr, **TOXIC)
    a, b, c, aa, bi, ba = oracle.synthesize(csr, z)
    want = oracle.prove(okey, a, b, c, z, aa, bi, ba, r, s)
    assert got.tobytes() == want.tobytes()
    pk = fx.key_to_py(okey)
    P = ref.proof_from_borsh(got.tobytes())
    assert ref.verify(pk, public, P)
    assert not ref.verify(pk, [(public[0] + 1) % ref.R] + public[1:], P)
    dr.free(); dk.free()


def test_one_signature(ctx, oracle, signatures):
    cs = signatures[0]
    csr = fx.r1cs_to_csr(cs.r1cs())
    assert csr.num_gates == 4123
    _prove_and_check(ctx, oracle, csr, fx.witness_mont(cs.z_in, cs.z_aux), cs.z_in[1:])


def test_golden_signature_proof(ctx):
    """the committed golden vector (tests/golden/eddsa_golden.json): GPU setup + proof bytes for a fixed signature and fixed (r, s)"""
    from helpers import golden
    g = golden('eddsa_golden.json')
    cs, (s_, r_x, a_x) = fc.eddsa_circuit(int(g['sk'], 16), int(g['m'], 16), int(g['rho'], 16))
    assert '%064x' % r_x == g['signature']['r_x'] and len(cs.gates) == g['num_gates']
    r1cs = r1cs_product(fx.r1cs_to_csr(cs.r1cs()))
    dk, _ = ctx.setup(r1cs, **{k: fx.mont_fr(v) for k, v in TOXIC.items()})
    dr = ctx.load_r1cs(r1cs)
    got = ctx.prove_witness(dk, dr, fx.witness_mont(cs.z_in, cs.z_aux), fx.mont_fr(int(g['r'], 16)), fx.mont_fr(int(g['s'], 16)))
    assert got.tobytes().hex() == g['proof']
    dr.free(); dk.free()


def test_batch_of_16_signatures_as_one_system(ctx, oracle, signatures):
    one = fx.r1cs_to_csr(signatures[0].r1cs())
    copies = 16
    batch = fx.tile_r1cs(one, copies)
    pick = [signatures[k % 4] for k in range(copies)]
    z = fx.tile_witness([c.z_in for c in pick], [c.z_aux for c in pick])
    assert batch.num_gates == copies * 4123 and batch.num_input == 1 + copies
    _prove_and_check(ctx, oracle, batch, z, [c.z_in[1] for c in pick])
