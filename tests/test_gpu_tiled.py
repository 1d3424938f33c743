"""Batch circuits as tiled constraint systems (fk_r1cs_load_tiled / fk_setup_tiled): ONE instance of a gadget resident in
HBM stands for `copies` of it -- how BASELINE configs[2] ("eddsa batch of 4096 signatures as one R1CS": 2.2e9 matrix
terms if written out) fits the boundary.  The bar: indistinguishable, bit for bit, from the explicitly replicated system
(oracle/fixtures.py: tile_r1cs) going through the ordinary entry points -- matrix evaluation, every array of the key,
the proof -- and at the full 4096 signatures, where nothing on the CPU can follow, the Groth16 pairing equation."""
import random

import numpy as np
import pytest

import bn254_ref as ref
import fawkes_circuit as fc
import fixtures as fx
from helpers import r1cs_product, TOXIC

pytestmark = pytest.mark.gpu
TOX = {k: fx.mont_fr(v) for k, v in TOXIC.items()}


def _tile_z(zs, num_input, picks):
    """batch witness from per-instance Montgomery witnesses ((nv, 4) uint64 each): ONE, every copy's inputs, every copy's aux"""
    one = zs[0][:1]
    return np.ascontiguousarray(np.concatenate([one] + [zs[p][1:num_input] for p in picks] + [zs[p][num_input:] for p in picks]))


def _same_key(dk1, dk2):
    assert dk1.counts() == dk2.counts()
    for name in ('h', 'l', 'a', 'b_g1', 'b_g2'):
        assert np.array_equal(dk1.download(name), dk2.download(name)), name


@pytest.mark.parametrize('copies,num_input,num_aux,gates', [(5, 3, 40, 50), (7, 1, 33, 61), (64, 2, 100, 130), (3, 4, 9000, 9001)])
def test_tiled_equals_replicated(ctx, copies, num_input, num_aux, gates):
    # (3, 4, 9000, 9001): the ONE column of C holds 9001 entries per copy -- the replicated system sends it through the
    # heavy-column path of the setup, the tiled one does not
    cs, z1, _, _ = fx.fast_r1cs(1000 + copies, gates, num_input, num_aux)
    batch = fx.tile_r1cs(cs, copies)
    base_p, batch_p = r1cs_product(cs), r1cs_product(batch)
    dr_t, dr_e = ctx.load_r1cs(base_p, copies=copies), ctx.load_r1cs(batch_p)
    assert dr_t.info() == dr_e.info()
    assert all(np.array_equal(ctx.download(p, n), ctx.download(q, n)) for p, q, n in
               zip(dr_t.density_ptrs(), dr_e.density_ptrs(), (batch.num_aux, batch.num_input, batch.num_aux)))
    # A z, B z, C z on a random vector
    rnd = random.Random(copies)
    nv, rows = batch.num_input + batch.num_aux, batch.num_gates + batch.num_input
    zr = fx.co.limbs_arr([rnd.randrange(ref.R) for _ in range(nv)])
    m = 1 << max(rows - 1, 1).bit_length()
    d_z = ctx.dev_alloc(zr.nbytes)
    outs = [[ctx.dev_alloc(m * 32) for _ in range(3)] for _ in range(2)]
    ctx.upload(d_z, zr)
    ctx.r1cs_eval_dev(dr_t, d_z, *outs[0])
    ctx.r1cs_eval_dev(dr_e, d_z, *outs[1])
    ctx.sync()
    for x, y in zip(*outs):
        assert np.array_equal(ctx.download(x, rows * 32), ctx.download(y, rows * 32))
    for p_ in [d_z] + outs[0] + outs[1]:
        ctx.dev_free(p_)
    # the key and the proof
    dk_t, vk_t = ctx.setup(base_p, copies=copies, **TOX)
    dk_e, vk_e = ctx.setup(batch_p, **TOX)
    _same_key(dk_t, dk_e)
    assert all(np.array_equal(vk_t[k], vk_e[k]) for k in vk_e)
    z = _tile_z([z1], num_input, [0] * copies)      # the proof needs a satisfying witness: the instance's, in every copy
    r, s = fx.mont_fr(7 + copies), fx.mont_fr(99)
    proofs = [ctx.prove_witness(k, d, z, r, s).tobytes() for k in (dk_t, dk_e) for d in (dr_t, dr_e)]
    assert len(set(proofs)) == 1
    for o in (dr_t, dr_e, dk_t, dk_e):
        o.free()


def test_tiled_rejects_bad_arguments(ctx):
    import fawkes_crypto_amd as fk
    cs, _, _, _ = fx.fast_r1cs(3, 20, 2, 10)
    p = r1cs_product(cs)
    with pytest.raises(fk.FkError):
        ctx.load_r1cs(p, copies=0)
    with pytest.raises(fk.FkError):
        ctx.setup(p, copies=0, **TOX)
    with pytest.raises(fk.FkError):
        ctx.load_r1cs(p, copies=1 << 30)      # 2^30 copies x 11 variables: past 32-bit variable indices
    # a key for 4 copies does not accept a system of 5
    dk, _ = ctx.setup(p, copies=4, **TOX)
    dr = ctx.load_r1cs(p, copies=5)
    with pytest.raises(fk.FkError):
        ctx.prove_witness(dk, dr, np.zeros((1 + 5 * 1 + 5 * 10, 4), np.uint64), fx.mont_fr(1), fx.mont_fr(2))
    dr.free(); dk.free()


@pytest.fixture(scope='module')
def signatures():
    rnd = random.Random(4096)
    pp, jj = fc.PoseidonParams(4, 8, 54), fc.JubJubBN256()
    return [fc.eddsa_circuit(rnd.randrange(fc.FS), rnd.randrange(ref.R), rnd.randrange(fc.FS), pp, jj)[0] for _ in range(3)]


def test_eddsa_batch_tiled_equals_oracle(ctx, oracle, signatures):
    """24 signatures: the tiled system's proof == the oracle's proof of the replicated system"""
    one = fx.r1cs_to_csr(signatures[0].r1cs())
    copies, picks = 24, [k % 3 for k in range(24)]
    zs = [fx.witness_mont(c.z_in, c.z_aux) for c in signatures]
    z = _tile_z(zs, one.num_input, picks)
    base_p = r1cs_product(one)
    dk, vk = ctx.setup(base_p, copies=copies, **TOX)
    dr = ctx.load_r1cs(base_p, copies=copies)
    r, s = fx.mont_fr(0x7113d), fx.mont_fr(0xba7c4)
    got = ctx.prove_witness(dk, dr, z, r, s)
    batch = fx.tile_r1cs(one, copies)
    okey = oracle.setup(batch, **TOXIC)
    a, b, c, aa, bi, ba = oracle.synthesize(batch, z)
    assert got.tobytes() == oracle.prove(okey, a, b, c, z, aa, bi, ba, r, s).tobytes()
    assert np.array_equal(vk['ic'], np.asarray(okey.ic).view(np.uint8).reshape(-1, 64))
    dr.free(); dk.free()


def test_config2_full_batch_4096_signatures(ctx, signatures):
    """BASELINE configs[2] at its full size: 4096 eddsa-poseidon verifiers as one R1CS (16.9 M gates, domain 2^25).
    The proof equals, byte for byte, the ORACLE's proof of the same batch (tests/golden/fullsize_digests.json, made by
    tests/golden/make_fullsize_digests.py: oracle/groth16_oracle.c on the host cores), satisfies the pairing
    equation for the 4096 public keys, fails it when one of them is altered, and is the same proof on a second run."""
    import hashlib
    from helpers import eddsa_batch_inputs, fullsize_digest
    copies = 4096
    sigs, one, picks, z, r, s = eddsa_batch_inputs(copies)
    assert [c.z_in for c in sigs] == [c.z_in for c in signatures]          # the shared builder is the module fixture's
    base_p = r1cs_product(one)
    dk, vk = ctx.setup(base_p, copies=copies, **TOX)
    dr = ctx.load_r1cs(base_p, copies=copies)
    assert dk.counts()['m'] == 1 << 25 and dr.info()['rows'] == copies * 4123 + 1 + copies
    proof = ctx.prove_witness(dk, dr, z, r, s)
    assert ctx.prove_witness(dk, dr, z, r, s).tobytes() == proof.tobytes()
    dg = fullsize_digest('eddsa%d' % copies)
    assert dg is not None and dg['witness_sha256'] == hashlib.sha256(z.tobytes()).hexdigest(), 'tests/golden/fullsize_digests.json: no entry for this batch'
    assert proof.tobytes().hex() == dg['proofs'][0], 'the 4096-signature proof differs from the oracle\'s bytes'
    g1 = lambda b: ref.g1_from_raw_le(bytes(b)); g2 = lambda b: ref.g2_from_raw_le(bytes(b))
    pk = dict(alpha_g1=g1(vk['alpha_g1']), beta_g2=g2(vk['beta_g2']), gamma_g2=g2(vk['gamma_g2']), delta_g2=g2(vk['delta_g2']),
              ic=[g1(x.tobytes()) for x in vk['ic']])
    public = [signatures[p].z_in[1] for p in picks]
    P = ref.proof_from_borsh(proof.tobytes())
    assert ref.verify(pk, public, P)
    public[1234] = signatures[(picks[1234] + 1) % 3].z_in[1]
    assert not ref.verify(pk, public, P)
    dr.free(); dk.free()


def test_rollup_style_transactions_tiled_equals_oracle(ctx, oracle):
    """BASELINE configs[3]'s "1024-tx shape" with real gadgets (oracle/fawkes_circuit.py: rollup_tx_circuit, 19270 gates and
    942 k matrix terms per transaction): 3 distinct transactions as one tiled system, proof == the oracle's proof of the
    replicated system, and it verifies against the 6 public roots"""
    rnd = random.Random(2025)
    txs = [fc.rollup_tx_circuit(rnd.randrange(fc.FS), 500 + k, 400 + k, [rnd.randrange(ref.R) for _ in range(32)], [rnd.randrange(2) for _ in range(32)],
                                rnd.randrange(fc.FS)) for k in range(3)]
    one = fx.r1cs_to_csr(txs[0].r1cs())
    zs = [fx.witness_mont(c.z_in, c.z_aux) for c in txs]
    z = _tile_z(zs, one.num_input, [0, 1, 2])
    base_p = r1cs_product(one)
    dk, vk = ctx.setup(base_p, copies=3, **TOX)
    dr = ctx.load_r1cs(base_p, copies=3)
    r, s = fx.mont_fr(0x70110), fx.mont_fr(0x7a)
    got = ctx.prove_witness(dk, dr, z, r, s)
    batch = fx.tile_r1cs(one, 3)
    okey = oracle.setup(batch, **TOXIC)
    a, b, c, aa, bi, ba = oracle.synthesize(batch, z)
    assert got.tobytes() == oracle.prove(okey, a, b, c, z, aa, bi, ba, r, s).tobytes()
    assert ref.verify(fx.key_to_py(okey), [v for c_ in txs for v in c_.z_in[1:]], ref.proof_from_borsh(got.tobytes()))
    dr.free(); dk.free()


def test_rollup_tx_golden_proof(ctx):
    """the committed golden vector (tests/golden/rollup_tx_golden.json, made by the C oracle): GPU setup + proof bytes of one
    rollup-style transaction for fixed inputs and fixed (r, s) -- 19270 gates, 942 k matrix terms through the length-class SpMV"""
    from helpers import golden
    g = golden('rollup_tx_golden.json')
    rnd = random.Random(g['seed'])
    sibling, path = [rnd.randrange(ref.R) for _ in range(32)], [rnd.randrange(2) for _ in range(32)]
    cs = fc.rollup_tx_circuit(int(g['sk'], 16), g['bal_old'], g['bal_new'], sibling, path, int(g['rho'], 16))
    assert '%064x' % cs.z_in[1] == g['old_root'] and len(cs.gates) == g['num_gates']
    r1cs = r1cs_product(fx.r1cs_to_csr(cs.r1cs()))
    dk, _ = ctx.setup(r1cs, **TOX)
    dr = ctx.load_r1cs(r1cs)
    got = ctx.prove_witness(dk, dr, fx.witness_mont(cs.z_in, cs.z_aux), fx.mont_fr(int(g['r'], 16)), fx.mont_fr(int(g['s'], 16)))
    assert got.tobytes().hex() == g['proof']
    dr.free(); dk.free()
