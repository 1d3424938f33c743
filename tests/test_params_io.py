"""`Parameters` file round trip: host-side wrapper + gate stream on CPU, and the bellman key
part through the on-GPU big-endian -> Montgomery loader (gpu)."""
import numpy as np
import pytest

import bn254_ref as ref
import fixtures as fx
from helpers import r1cs_product, TOXIC


def _key_arrays(key):
    return dict(alpha_g1=key.alpha_g1, beta_g1=key.beta_g1, beta_g2=key.beta_g2, gamma_g2=key.gamma_g2, delta_g1=key.delta_g1,
                delta_g2=key.delta_g2, ic=np.array(key.ic), h=np.array(key.h), l=np.array(key.l), a=np.array(key.a),
                b_g1=np.array(key.b_g1), b_g2=np.array(key.b_g2))


def test_wrapper_and_gate_stream_roundtrip(oracle):
    from fawkes_crypto_amd import params_io as pio
    cs, z_in, z_aux = ref.random_r1cs(31, 50, 3, 60)
    csr = fx.r1cs_to_csr(cs)
    r1cs = r1cs_product(csr)
    stream = pio.encode_gate_stream(r1cs)
    # first item of the first gate: 4-byte count, then 32-byte canonical coefficient, tag, index (cs.rs:193-213)
    cnt = int.from_bytes(stream[:4], 'little')
    assert cnt == len(cs.rows[0][0])
    coeff, (kind, idx) = cs.rows[0][0][0]
    assert int.from_bytes(stream[4:36], 'little') == coeff % ref.R
    assert stream[36] == (0 if kind == 'i' else 1) and int.from_bytes(stream[37:41], 'little') == idx
    back = pio.decode_gate_stream(stream, r1cs.num_gates, r1cs.num_input, r1cs.num_aux)
    for (p0, c0, v0), (p1, c1, v1) in zip(r1cs.mats, back.mats):
        assert np.array_equal(p0, p1) and np.array_equal(c0, c1) and np.array_equal(v0, v1)
    bits = [True, False, False, True, True, False, True, False, True, True]
    assert pio.bits_to_bytes(bits) == bytes([0b10011010, 0b11000000])      # MSB-first packing (bit-vec)
    blob = pio.write_parameters(r1cs.num_gates, pio.RAW_MAGIC + stream, bits, b'BELLMAN')
    hdr = pio.read_parameters(blob)
    assert hdr['num_gates'] == r1cs.num_gates and hdr['const_tracker'] == bits and hdr['bellman'] == b'BELLMAN'
    with pytest.raises(ValueError):
        pio.read_parameters(blob[:20])
    with pytest.raises(ValueError):
        pio.decode_gate_stream(stream[:-3], r1cs.num_gates, r1cs.num_input, r1cs.num_aux)
    # point encodings: big-endian canonical, c1 before c0, 0x40 = infinity
    g1 = ref.g1_raw_le(ref.G1_GEN)
    enc = pio.g1_uncompressed(g1)
    assert int.from_bytes(enc[:32], 'big') == 1 and int.from_bytes(enc[32:], 'big') == 2
    enc2 = pio.g2_uncompressed(ref.g2_raw_le(ref.G2_GEN))
    assert int.from_bytes(enc2[:32], 'big') == ref.G2_GEN[0][1] and int.from_bytes(enc2[32:64], 'big') == ref.G2_GEN[0][0]
    assert pio.g1_uncompressed(bytes(64))[0] == 0x40


@pytest.mark.gpu
def test_key_file_roundtrip_and_prove(ctx, oracle):
    """oracle key -> file bytes -> fk_key_load_bellman (GPU conversion) -> identical device arrays -> identical proof"""
    from fawkes_crypto_amd import params_io as pio
    import fawkes_crypto_amd as fk
    cs, z_in, z_aux = ref.random_r1cs(41, 300, 3, 330)
    csr = fx.r1cs_to_csr(cs)
    key = oracle.setup(csr, **TOXIC)
    r1cs = r1cs_product(csr)
    arrays = _key_arrays(key)
    arrays['l'] = arrays['l'].copy(); arrays['l'][5] = 0          # an identity point inside an array (0x40 flag path)
    data = pio.store_parameters(arrays, r1cs, const_tracker_bits=[True, False, True])
    dk, r1cs2, hdr = pio.load_parameters(ctx, data)
    assert hdr['const_tracker'] == [True, False, True] and hdr['num_gates'] == r1cs.num_gates
    assert dk.counts()['m'] == key.m and dk.counts()['num_input'] == 3
    for name in ('h', 'l', 'a', 'b_g1', 'b_g2'):
        assert dk.download(name).tobytes() == arrays[name].tobytes(), name
    vk = dk.vk()
    for name in ('alpha_g1', 'beta_g1', 'delta_g1', 'beta_g2', 'delta_g2'):
        assert vk[name].tobytes() == arrays[name].tobytes(), name
    assert hdr['gamma_g2'].tobytes() == arrays['gamma_g2'].tobytes() and hdr['ic'].tobytes() == arrays['ic'].tobytes()
    for (p0, c0, v0), (p1, c1, v1) in zip(r1cs.mats, r1cs2.mats):
        assert np.array_equal(p0, p1) and np.array_equal(c0, c1) and np.array_equal(v0, v1)
    # sharded load keeps the right slices
    sk, _, _ = pio.load_parameters(ctx, data, shard_index=1, shard_count=3)
    lo, hi = sk.shard_info()['h']
    assert sk.download('h').tobytes() == arrays['h'][lo:hi].tobytes()
    # malformed files are rejected with a status code
    with pytest.raises(fk.FkError):
        ctx.load_key_bellman(hdr['bellman'][:1000])
    bad = bytearray(hdr['bellman']); bad[576 + 4 + 3 * 64 + 4] |= 0x80     # compression flag on h[0]
    with pytest.raises(fk.FkError):
        ctx.load_key_bellman(bytes(bad))
