"""CPU tests of the oracle itself: pinned against the reference's own known-answer data, the curve's
public constants, the Groth16 pairing equation and python<->C agreement on the golden vectors."""
import numpy as np
import pytest

import bn254_ref as ref
import fixtures as fx
from helpers import R, Q, golden, golden_instance, mont_ints, TOXIC


def test_ff_uint_kats(oracle):
    """ff-uint/tests/ff-uint_tests.rs:35-156 against the generic 4-limb Montgomery code of the C oracle."""
    k = golden('ff_uint_kats.json')
    p = int(k['modulus'])
    oracle.field_custom(p)
    f = lambda x: oracle.fe_from_canon(oracle.FX, int(x))
    g = lambda a: oracle.fe_to_canon(oracle.FX, a)
    for a, b, want in k['add']:
        assert g(oracle.fe_add(oracle.FX, f(a), f(b))) == int(want)
    for a, b, want in k['sub']:
        assert g(oracle.fe_sub(oracle.FX, f(a), f(b))) == int(want)
    for a, b, want in k['mul']:
        assert g(oracle.fe_mul(oracle.FX, f(a), f(b))) == int(want)
    for a, b, want in k['div']:
        assert g(oracle.fe_mul(oracle.FX, f(a), oracle.fe_inv(oracle.FX, f(b)))) == int(want)
    for a, e, want in k['pow']:
        assert g(oracle.fe_pow(oracle.FX, f(a), int(e))) == int(want)
    for a, want in k['neg']:
        assert g(oracle.fe_neg(oracle.FX, f(a))) == int(want)
    # legendre symbol = a^((p-1)/2) and the sqrt KATs (ff-uint_tests.rs:102-141) through the same pow / mul code
    for a, want in k['legendre']:
        l = g(oracle.fe_pow(oracle.FX, f(a), (p - 1) // 2))
        assert (0 if l == 0 else (1 if l == 1 else -1)) == want and l in (0, 1, p - 1)
    for a, root in k['sqrt']:
        if root is None:
            assert g(oracle.fe_pow(oracle.FX, f(a), (p - 1) // 2)) == p - 1          # non-residue: no root
        else:
            assert g(oracle.fe_mul(oracle.FX, f(root), f(root))) == int(a)


def test_bn254_constants(oracle):
    # SURVEY.md Appendix B.1 (values derived from engines/bn256/mod.rs:13,23,24)
    assert oracle.to_int(oracle.fe_from_canon(oracle.FR, 1)) == 0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb
    assert oracle.to_int(oracle.fe_from_canon(oracle.FQ, 1)) == 0x0e0a77c19a07df2f666ea36f7879462c0a78eb28f5c70b3dd35d438dc58f0d9d
    assert ref.FR_ROOT_OF_UNITY == 0x03ddb9f5166d18b798865ea93dd31f743215cf6dd39329c8d34f1ed960c37c9c
    assert pow(ref.FR_ROOT_OF_UNITY, 1 << 28, R) == 1 and pow(ref.FR_ROOT_OF_UNITY, 1 << 27, R) != 1
    assert ref.G1.on_curve(ref.G1_GEN) and ref.G2.on_curve(ref.G2_GEN)
    # the well-known 2*G1 of alt_bn128
    assert ref.G1.mul(ref.G1_GEN, 2) == (
        1368015179489954701390400359078579693043519447331113978918064868415326638035,
        9918110051302171585080402603319702774565515993150576347155970296011118125764)
    assert ref.G1.mul(ref.G1_GEN, R) is None and ref.G2.mul(ref.G2_GEN, R) is None


def test_field_python_vs_c(oracle):
    rng = ref.Lcg(99)
    for fid, p in ((oracle.FQ, Q), (oracle.FR, R)):
        for _ in range(50):
            a, b = rng.below(p), rng.below(p)
            fa, fb = oracle.fe_from_canon(fid, a), oracle.fe_from_canon(fid, b)
            assert oracle.to_int(fa) == ref.to_mont(a, p)
            assert oracle.fe_to_canon(fid, oracle.fe_mul(fid, fa, fb)) == a * b % p
            assert oracle.fe_to_canon(fid, oracle.fe_add(fid, fa, fb)) == (a + b) % p
            assert oracle.fe_to_canon(fid, oracle.fe_sub(fid, fa, fb)) == (a - b) % p
        a = rng.below(p)
        assert oracle.fe_to_canon(fid, oracle.fe_inv(fid, oracle.fe_from_canon(fid, a))) == pow(a, -1, p)
    for edge in (0, 1, R - 1):
        assert oracle.fe_to_canon(oracle.FR, oracle.fe_mul(oracle.FR, oracle.fe_from_canon(oracle.FR, edge), oracle.fe_from_canon(oracle.FR, R - 1))) == edge * (R - 1) % R


def test_pairing_bilinear():
    e1 = ref.pairing(ref.G2_GEN, ref.G1_GEN)
    assert e1 != ref.f12_one()
    assert ref.pairing(ref.G2.mul(ref.G2_GEN, 5), ref.G1.mul(ref.G1_GEN, 7)) == ref.f12_pow(e1, 35)


def test_ntt_golden_python_and_c(oracle):
    for case in golden('ntt_golden.json')['cases']:
        v = [int(x, 16) for x in case['input']]
        n = len(v)
        w = ref.omega_for(n)
        assert [int(x, 16) for x in case['forward']] == ref.ntt(v, w)
        vm = oracle.limbs_arr([ref.to_mont(x, R) for x in v])
        assert mont_ints(oracle.fr_ntt(vm)) == [int(x, 16) for x in case['forward']]
        assert mont_ints(oracle.fr_ntt(vm, inverse=True)) == [int(x, 16) for x in case['inverse']]
        assert mont_ints(oracle.fr_ntt(vm, coset=True)) == [int(x, 16) for x in case['coset_forward']]
        assert mont_ints(oracle.fr_ntt(vm, inverse=True, coset=True)) == [int(x, 16) for x in case['coset_inverse']]


def test_msm_golden_python_and_c(oracle):
    g = golden('msm_golden.json')
    sc = [int(x, 16) for x in g['g1_scalars']]
    bases = np.frombuffer(bytes.fromhex(''.join(g['g1_bases'])), np.uint8).reshape(-1, 64)
    sm = oracle.limbs_arr([ref.to_mont(x, R) for x in sc])
    assert oracle.msm_g1(bases, sm).tobytes().hex() == g['g1_result']
    pts = [ref.g1_from_raw_le(bytes(b)) for b in bases]
    assert ref.g1_raw_le(ref.G1.msm(pts, sc)).hex() == g['g1_result']
    b2 = np.frombuffer(bytes.fromhex(''.join(g['g2_bases'])), np.uint8).reshape(-1, 128)
    assert oracle.msm_g2(b2, sm[:len(b2)]).tobytes().hex() == g['g2_result']


def test_msm_density_filter(oracle):
    """bellman's DensityTracker path: bases are compacted to the selected scalars (App. A.3)."""
    from helpers import g1_bases
    rng = np.random.default_rng(5)
    n = 50
    dens = (rng.integers(0, 2, n)).astype(np.uint8)
    from helpers import rand_fr_mont
    sc = rand_fr_mont(rng, n, 'witness')
    bases = g1_bases(int(dens.sum()), 3)
    got = oracle.msm_g1(bases, sc, dens)
    want = oracle.msm_g1(bases, sc[dens != 0])
    assert got.tobytes() == want.tobytes()


def test_proof_golden_python_and_c(oracle):
    g, cs, z_in, z_aux, tw, r, s = golden_instance()
    pk = ref.setup(cs, **tw)
    proof = ref.prove(pk, cs, z_in, z_aux, r, s)
    assert ref.proof_borsh(*proof).hex() == g['proof']
    assert ref.verify(pk, z_in[1:], proof)
    assert not ref.verify(pk, [(z_in[1] + 1) % R] + z_in[2:], proof)
    csr = fx.r1cs_to_csr(cs)
    key = oracle.setup(csr, **tw)
    assert key.m == g['m'] and key.a.shape[0] == g['n_a'] and key.b_g1.shape[0] == g['n_b']
    assert key.h.tobytes() == b''.join(ref.g1_raw_le(p) for p in pk['h'])
    assert key.b_g2.tobytes() == b''.join(ref.g2_raw_le(p) for p in pk['b_g2'])
    z = fx.witness_mont(z_in, z_aux)
    a, b, c, aa, bi, ba = oracle.synthesize(csr, z)
    h = oracle.quotient_h(a, b, c)
    assert mont_ints(h) == [int(x, 16) for x in g['h']]
    out = oracle.prove(key, a, b, c, z, aa, bi, ba, fx.mont_fr(r), fx.mont_fr(s))
    assert out.tobytes().hex() == g['proof']


def test_quotient_identity(oracle):
    """A(x)B(x) - C(x) = h(x) (x^m - 1) at a random point (size-independent pin of the quotient)."""
    cs, z_in, z_aux = ref.random_r1cs(77, 300, 2, 310)
    csr = fx.r1cs_to_csr(cs)
    a, b, c, *_ = oracle.synthesize(csr, fx.witness_mont(z_in, z_aux))
    h = mont_ints(oracle.quotient_h(a, b, c))
    m = 512
    w = ref.omega_for(m)
    pad = lambda v: mont_ints(v) + [0] * (m - len(v))
    ca, cb, cc = (ref.intt(pad(v), w) for v in (a, b, c))
    x = 0x123456789abcdef
    ev = lambda co_: sum(cf * pow(x, i, R) for i, cf in enumerate(co_)) % R
    assert (ev(ca) * ev(cb) - ev(cc)) % R == ev(h) * (pow(x, m, R) - 1) % R


def test_config1_shape_prove_verify(oracle):
    """BASELINE configs[0] shape (poseidon-merkle depth 32: 7362 gates, 2 inputs, 7394 aux => m = 2^13),
    synthetic satisfiable R1CS; C oracle proof must satisfy the pairing equation."""
    cs, z_in, z_aux = ref.random_r1cs(11, 7362, 2, 7394)
    csr = fx.r1cs_to_csr(cs)
    key = oracle.setup(csr, **TOXIC)
    assert key.m == 1 << 13
    z = fx.witness_mont(z_in, z_aux)
    a, b, c, aa, bi, ba = oracle.synthesize(csr, z)
    proof = oracle.prove(key, a, b, c, z, aa, bi, ba, fx.mont_fr(0x1111), fx.mont_fr(0x2222))
    assert ref.verify(fx.key_to_py(key), z_in[1:], ref.proof_from_borsh(proof.tobytes()))


def test_threaded_prover_same_bytes(oracle):
    """bellman's multicore split restated (parallel_fft, one task per multiexp region; the all-cores CPU baseline of
    bench.py) gives the serial prover's bytes for every thread count, including ones that are not a power of two."""
    cs, z, _, _ = fx.fast_r1cs(5, 3000, 3, 3100)
    key = oracle.setup(cs, **TOXIC)
    a, b, c, aa, bi, ba = oracle.synthesize(cs, z)
    r, s = fx.mont_fr(5), fx.mont_fr(6)
    want = oracle.prove(key, a, b, c, z, aa, bi, ba, r, s)
    for t in (2, 3, 8, 64):
        assert oracle.prove(key, a, b, c, z, aa, bi, ba, r, s, threads=t).tobytes() == want.tobytes(), t


def test_synthesize_tiled_equals_replicated_system(oracle):
    """orc_synthesize_tiled (what lets the CPU baseline run at the benchmark's full size) = orc_synthesize of the explicitly
    replicated system: a, b, c and the three density maps, for gate counts of different residues"""
    import numpy as np
    import fixtures as fx
    from test_gpu_r1cs import _ragged_system
    for copies, gates in ((1, 50), (3, 41), (6, 64)):
        base = _ragged_system(7 + copies, [0, 1, 1, 2, 5, 33], gates, 3, 40)
        big = fx.tile_r1cs(base, copies)
        nv = big.num_input + big.num_aux
        rnd = np.random.default_rng(copies)
        z = fx.co.limbs_arr([int(x) % R for x in rnd.integers(0, 2**63, nv).astype(object) * (2**190 + 12345)])
        want = oracle.synthesize(big, z)
        got = oracle.synthesize_tiled(base, copies, z)
        for w, g in zip(want, got):
            assert np.array_equal(w, g)
