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


def _same_system(x, y):
    assert (x.num_input, x.num_aux, x.num_gates) == (y.num_input, y.num_aux, y.num_gates)
    for (p0, c0, v0), (p1, c1, v1) in zip(x.mats, y.mats):
        assert np.array_equal(p0, p1) and np.array_equal(c0, c1) and np.array_equal(v0, v1)


def test_native_gate_decoder_raw_brotli_and_malformed(oracle):
    """csrc/gatestream.hip (host code, no GPU): the reference's blob format -- brotli(quality 9, lgwin 22) over the Borsh gate
    stream, written here with the system's brotli ENCODER -- decodes to the system that was written, also across the
    decoder's 4 MiB output chunks; malformed streams are FK_ERR_FORMAT the way the reference's reader fails."""
    import fawkes_crypto_amd as fk
    from fawkes_crypto_amd import api, params_io as pio
    from helpers import brotli_compress
    cs, _, _, _ = fx.fast_r1cs(77, 40000, 3, 41000)        # 160 k terms = 5.9 MB of gate stream: more than one output chunk
    r1cs = r1cs_product(cs)
    stream = pio.encode_gate_stream(r1cs)
    assert len(stream) > (5 << 20)
    g = api.Gates(stream, api.FK_GATES_RAW, r1cs.num_gates, r1cs.num_input, r1cs.num_aux)
    info = g.info()
    assert info['nnz'] == tuple(len(c) for _, c, _ in r1cs.mats) and info['decoded_bytes'] == len(stream)
    _same_system(g.to_r1cs(), r1cs)
    blob = brotli_compress(stream)
    if blob is None:
        pytest.skip('libbrotlienc.so.1 not present: cannot write a reference-format blob')
    assert len(blob) < len(stream)
    gb = api.Gates(blob, api.FK_GATES_BROTLI, r1cs.num_gates, r1cs.num_input, r1cs.num_aux)
    assert gb.info() == info
    _same_system(gb.to_r1cs(), r1cs)
    # the slow per-term restatement of GateStreamedIterator agrees on a small system
    cs2, _, _ = ref.random_r1cs(5, 40, 3, 50)
    small = r1cs_product(fx.r1cs_to_csr(cs2))
    st2 = pio.encode_gate_stream(small)
    _same_system(api.Gates(brotli_compress(st2), api.FK_GATES_BROTLI, small.num_gates, 3, 50).to_r1cs(), pio.decode_gate_stream(st2, small.num_gates, 3, 50))

    def rejects(data, fmt=api.FK_GATES_RAW, gates=small.num_gates, code=7):
        with pytest.raises(fk.FkError) as e:
            api.Gates(data, fmt, gates, 3, 50)
        assert e.value.code == code
    rejects(st2[:-3])                                       # truncated
    rejects(st2 + b'\x00')                                  # trailing bytes
    rejects(st2, gates=small.num_gates + 1)                 # fewer gates than announced
    bad = bytearray(st2); bad[4 + 32] = 2; rejects(bytes(bad))                                 # "enum elements overflow" (cs.rs:209)
    bad = bytearray(st2); bad[4 + 33:4 + 37] = (1 << 20).to_bytes(4, 'little'); rejects(bytes(bad))     # variable index out of range
    bad = bytearray(st2); bad[4:4 + 32] = ref.R.to_bytes(32, 'little'); rejects(bytes(bad))   # coefficient not below r (from_uint fails)
    rejects(brotli_compress(st2)[:-5], fmt=api.FK_GATES_BROTLI)                               # brotli stream cut short
    rejects(b'\xff' * 64, fmt=api.FK_GATES_BROTLI)                                            # not brotli at all


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
    dk, dr2, hdr = pio.load_parameters(ctx, data, want_host_r1cs=True)
    r1cs2 = hdr['r1cs']
    assert dr2.info()['nnz'] == tuple(len(c) for _, c, _ in r1cs.mats)
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
    # the proof from the loaded file (reference-format brotli blob when the encoder is present) equals the oracle's
    from helpers import brotli_compress
    if brotli_compress(b'x') is not None:
        data_b = pio.store_parameters(arrays, r1cs, const_tracker_bits=[True], compress=brotli_compress)
        assert len(data_b) < len(data)
        dk_b, dr_b, _ = pio.load_parameters(ctx, data_b)
        z = fx.witness_mont(z_in, z_aux)
        r, s = fx.mont_fr(3), fx.mont_fr(4)
        want = ctx.prove_witness(dk, dr2, z, r, s)             # same key (l[5] replaced by the identity in both files)
        assert ctx.prove_witness(dk_b, dr_b, z, r, s).tobytes() == want.tobytes()
        dk_b.free(); dr_b.free()
    # malformed files are rejected with a status code: Parameters::read(reader, disallow_points_at_infinity, checked) (mod.rs:159)
    bell = hdr['bellman']
    H0 = 576 + 4 + 3 * 64 + 4                  # offset of h[0]: vk (576 B), ic count + 3 points, h count
    with pytest.raises(fk.FkError) as e:
        ctx.load_key_bellman(bell[:1000])
    assert e.value.code == 7

    def rejected(mut, flags=fk.api.FK_KEY_CHECKED, code=7):
        bad = bytearray(bell); mut(bad)
        with pytest.raises(fk.FkError) as e:
            ctx.load_key_bellman(bytes(bad), flags=flags)
        assert e.value.code == code, e.value
    def setb(off, val):
        return lambda b: b.__setitem__(slice(off, off + len(val)), val)
    rejected(lambda b: b.__setitem__(H0, b[H0] | 0x80), flags=0)                         # compression flag on h[0]: always an error
    rejected(setb(H0, ref.Q.to_bytes(32, 'big')), flags=0)                              # x = q: not a field element, checked or not
    L5 = H0 + len(arrays['h']) * 64 + 4 + 5 * 64
    assert bell[L5] == 0x40
    rejected(setb(L5 + 40, b'\x01'), flags=0)                                            # infinity flag with a non-zero rest
    rejected(lambda b: None, flags=fk.api.FK_KEY_NO_INFINITY)                            # l[5] IS the identity: disallow_points_at_infinity
    ctx.load_key_bellman(bell, flags=0)[0].free()                                       # ... and is fine otherwise
    ctx.load_key_bellman(bell, flags=fk.api.FK_KEY_CHECKED)[0].free()
    flip = lambda b: b.__setitem__(H0 + 63, b[H0 + 63] ^ 1)                              # y of h[0] changed: off the curve
    rejected(flip)
    bad = bytearray(bell); flip(bad)
    ctx.load_key_bellman(bytes(bad), flags=0)[0].free()                                 # `checked = false` does not look (bellman: into_affine_unchecked)
    # a point ON the twist but outside the order-r subgroup in b_g2[0]: only the checked read rejects it
    def fq_sqrt(v):
        y = pow(v, (ref.Q + 1) // 4, ref.Q)
        return y if y * y % ref.Q == v % ref.Q else None
    def fq2_sqrt(a0, a1):
        alpha = fq_sqrt((a0 * a0 + a1 * a1) % ref.Q)
        if alpha is None:
            return None
        for dlt in ((a0 + alpha) * pow(2, -1, ref.Q) % ref.Q, (a0 - alpha) * pow(2, -1, ref.Q) % ref.Q):
            x0 = fq_sqrt(dlt)
            if x0:
                return x0, a1 * pow(2 * x0, -1, ref.Q) % ref.Q
        return None
    F2, b2 = ref.F2, ref.G2.b
    x = (5, 7)
    while True:
        rhs = F2.add(F2.mul(F2.sqr(x), x), b2)
        y = fq2_sqrt(*rhs)
        if y is not None and F2.sqr(y) == rhs:
            break
        x = (x[0] + 1, x[1])
    assert ref.G2.on_curve((x, y)) and ref.G2.mul((x, y), ref.R) is not None       # on the curve, r * P != identity
    B2 = L5 - 5 * 64 + len(arrays['l']) * 64 + 4 + len(arrays['a']) * 64 + 4 + len(arrays['b_g1']) * 64 + 4
    enc = b''.join(v.to_bytes(32, 'big') for v in (x[1], x[0], y[1], y[0]))
    rejected(setb(B2, enc))
    bad = bytearray(bell); bad[B2:B2 + 128] = enc
    ctx.load_key_bellman(bytes(bad), flags=0)[0].free()


def test_bitvec_packing_is_bit_vec_0_6(oracle):
    """`BitVec::to_bytes` / `from_bytes` (mod.rs:155,169) pack the FIRST bit into the HIGH-order bit of byte 0.  The cases are the
    documentation examples of the bit-vec crate (0.6), i.e. facts that do not come from this repository: a self round trip
    cannot tell MSB-first from LSB-first, these can -- including lengths that are not a multiple of 8."""
    from fawkes_crypto_amd import params_io as pio
    # to_bytes: BitVec::from_elem(3, true) with bit 1 cleared -> [0b10100000]
    assert pio.bits_to_bytes([True, False, True]) == bytes([0b10100000])
    # to_bytes: BitVec::from_elem(9, false) with bits 2 and 8 set -> [0b00100000, 0b10000000]
    bits9 = [False] * 9
    bits9[2] = bits9[8] = True
    assert pio.bits_to_bytes(bits9) == bytes([0b00100000, 0b10000000])
    # from_bytes(&[0b10100000, 0b00010010]) == [t f t f f f f f  f f f t f f t f]
    want = [True, False, True, False, False, False, False, False, False, False, False, True, False, False, True, False]
    assert pio.bytes_to_bits(bytes([0b10100000, 0b00010010]), 16) == want
    assert pio.bytes_to_bits(bytes([0b10100000, 0b00010010]), 12) == want[:12]        # mod.rs:170: truncate to the stored bit length
    with pytest.raises(ValueError):
        pio.bytes_to_bits(bytes([0xff]), 9)                                            # mod.rs:165-167: inconsistent length
    # the fawkes wrapper around it, byte for byte (mod.rs:150-157): u32 LE gate count | Borsh Vec<u8> | u32 LE bit length | Borsh Vec<u8>
    data = pio.write_parameters(7, b'\xaa\xbb', bits9, b'TAIL')
    assert data == bytes.fromhex('07000000' '02000000' 'aabb' '09000000' '02000000' '2080') + b'TAIL'
    hdr = pio.read_parameters(data)
    assert hdr['num_gates'] == 7 and hdr['gates_blob'] == b'\xaa\xbb' and hdr['const_tracker'] == bits9 and hdr['bellman'] == b'TAIL'


@pytest.mark.gpu
def test_handmade_bellman_key_bytes(ctx):
    """A `Parameters::write` image assembled BY HAND from the upstream format description -- not by
    params_io.encode_bellman_parameters -- and loaded through fk_key_load_bellman: big-endian coordinates, G2 as
    x.c1 | x.c0 | y.c1 | y.c0 (for the generator that is literally the EIP-197 hex of it), `u32` BIG-endian counts, the 0x40
    infinity byte, the 0x80 compression bit.  Expected device bytes come from the big-int reference (Montgomery LE)."""
    import fawkes_crypto_amd as fk
    be = lambda v: v.to_bytes(32, 'big')
    G1x, G1y = 1, 2
    D = (1368015179489954701390400359078579693043519447331113978918064868415326638035,
         9918110051302171585080402603319702774565515993150576347155970296011118125764)          # 2 * G1, public constant
    g1 = lambda P: be(P[0]) + be(P[1])
    G2HEX = ('198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2' '1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed'
             '090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b' '12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa')
    g2gen = bytes.fromhex(G2HEX)
    assert ref.G2_GEN == ((int(G2HEX[64:128], 16), int(G2HEX[0:64], 16)), (int(G2HEX[192:256], 16), int(G2HEX[128:192], 16)))
    P3, P5 = ref.G1.mul(ref.G1_GEN, 3), ref.G1.mul(ref.G1_GEN, 5)
    Q2 = ref.G2.mul(ref.G2_GEN, 2)
    g2 = lambda P: be(P[0][1]) + be(P[0][0]) + be(P[1][1]) + be(P[1][0])
    INF1, INF2 = b'\x40' + bytes(63), b'\x40' + bytes(127)
    u32be = lambda v: v.to_bytes(4, 'big')
    vk = g1((G1x, G1y)) + g1(D) + g2gen + g2(Q2) + g1(P3) + g2gen           # alpha_g1 beta_g1 beta_g2 gamma_g2 delta_g1 delta_g2
    ic = [(G1x, G1y), D]
    h = [D, P3, P5]                                                        # m - 1 = 3 points: domain 4
    l = [P5, None]                                                         # num_aux = 2, the second one the identity
    a = [(G1x, G1y), D, P3]
    b1 = [P3, None]
    b2 = [Q2, None]
    enc1 = lambda pts: u32be(len(pts)) + b''.join(INF1 if P is None else g1(P) for P in pts)
    enc2 = lambda pts: u32be(len(pts)) + b''.join(INF2 if P is None else g2(P) for P in pts)
    data = vk + enc1(ic) + enc1(h) + enc1(l) + enc1(a) + enc1(b1) + enc2(b2)
    assert len(data) == 576 + 4 + 128 + 4 + 192 + 4 + 128 + 4 + 192 + 4 + 128 + 4 + 256
    raw1 = lambda pts: b''.join(ref.g1_raw_le(P) for P in pts)
    raw2 = lambda pts: b''.join(ref.g2_raw_le(P) for P in pts)
    for flags in (0, fk.api.FK_KEY_CHECKED):
        dk, gamma, icv = ctx.load_key_bellman(data, flags=flags)
        c = dk.counts()
        assert (c['m'], c['num_input'], c['num_aux'], c['n_a'], c['n_b']) == (4, 2, 2, 3, 2)
        assert dk.download('h').tobytes() == raw1(h) and dk.download('l').tobytes() == raw1(l) and dk.download('a').tobytes() == raw1(a)
        assert dk.download('b_g1').tobytes() == raw1(b1) and dk.download('b_g2').tobytes() == raw2(b2)
        v = dk.vk()
        assert v['alpha_g1'].tobytes() == ref.g1_raw_le((G1x, G1y)) and v['beta_g1'].tobytes() == ref.g1_raw_le(D) and v['delta_g1'].tobytes() == ref.g1_raw_le(P3)
        assert v['beta_g2'].tobytes() == ref.g2_raw_le(ref.G2_GEN) and v['delta_g2'].tobytes() == ref.g2_raw_le(ref.G2_GEN)
        assert gamma.tobytes() == ref.g2_raw_le(Q2) and icv.tobytes() == raw1(ic)
        dk.free()

    def rejected(mutated, flags=0):
        with pytest.raises(fk.FkError) as e:
            ctx.load_key_bellman(bytes(mutated), flags=flags)
        assert e.value.code == 7, e.value

    rejected(data, flags=fk.api.FK_KEY_NO_INFINITY)                          # l[1], b_g1[1], b_g2[1] are the identity
    H0 = 576 + 4 + 128 + 4
    bad = bytearray(data); bad[H0] |= 0x80; rejected(bad)                    # compression bit on an uncompressed point
    bad = bytearray(data); bad[H0 + 64 + 128 + 4 + 64 + 9] = 1; rejected(bad)                # l[1]: infinity byte with a non-zero rest
    bad = bytearray(data); bad[576:580] = (2).to_bytes(4, 'little'); rejected(bad)           # a little-endian count reads as 2^25 points: truncated
    # the identity among the ic points is refused whatever the flags; inside alpha .. delta it is an encoding like any other
    bad = bytearray(data); bad[580:644] = INF1; rejected(bad)
    bad = bytearray(data); bad[64:128] = INF1                                # beta_g1
    ctx.load_key_bellman(bytes(bad), flags=0)[0].free()
    # the verifying key is always decoded with the curve check (VerifyingKey::read uses into_affine), `checked` or not
    bad = bytearray(data); bad[63] ^= 1; rejected(bad, flags=0)


def test_gate_decoder_refuses_a_forged_gate_count():
    """`num_gates` comes from the file header: a count the stream cannot hold is InvalidData at once (no 3 x 32 GiB reservation,
    no exception leaving the C ABI)"""
    import fawkes_crypto_amd as fk
    from fawkes_crypto_amd import api
    stream = b'\x00' * 12 * 5                                  # five gates of three empty linear combinations
    api.Gates(stream, api.FK_GATES_RAW, 5, 1, 0).free()
    with pytest.raises(fk.FkError) as e:
        api.Gates(stream, api.FK_GATES_RAW, 0xffffffff, 1, 0)
    assert e.value.code == 7


@pytest.mark.gpu
def test_native_bellman_writer_matches_the_restatement_and_round_trips(ctx, oracle):
    """fk_key_write_bellman (round 4): `Parameters::write`'s bellman part (mod.rs:156) from the device layout, converted on the GPU --
    byte-equal to the per-point Python restatement (params_io.encode_bellman_parameters) incl. an identity point inside an array
    (0x40 flag byte), and fk_key_load_bellman(checked) of those bytes gives back identical device arrays, vk, gamma_g2 and ic;
    a shard is refused; the whole `Parameters` file written through store_parameters_dev loads and proves like the original."""
    from fawkes_crypto_amd import params_io as pio
    import fawkes_crypto_amd as fk
    cs, z_in, z_aux = ref.random_r1cs(43, 500, 3, 560)
    csr = fx.r1cs_to_csr(cs)
    key = oracle.setup(csr, **TOXIC)
    r1cs = r1cs_product(csr)
    arrays = _key_arrays(key)
    arrays['a'] = arrays['a'].copy(); arrays['a'][7] = 0          # an identity point inside an array
    shape = dict(m=key.m, num_input=key.num_input, num_aux=key.num_aux)
    params = fk.Parameters(dict(arrays, **shape), r1cs)
    dk = ctx.load_key(params)
    vk = dict(gamma_g2=arrays['gamma_g2'], ic=arrays['ic'])
    got = ctx.write_key_bellman(dk, vk)
    want = pio.encode_bellman_parameters(arrays)
    assert got.tobytes() == want
    dk2, gamma, ic = ctx.load_key_bellman(got, flags=fk.api.FK_KEY_CHECKED)
    for name in ('h', 'l', 'a', 'b_g1', 'b_g2'):
        assert dk2.download(name).tobytes() == arrays[name].tobytes(), name
    assert gamma.tobytes() == arrays['gamma_g2'].tobytes() and ic.tobytes() == np.asarray(arrays['ic'], np.uint8).tobytes()
    assert all(dk2.vk()[n_].tobytes() == arrays[n_].tobytes() for n_ in ('alpha_g1', 'beta_g1', 'delta_g1', 'beta_g2', 'delta_g2'))
    sk = ctx.load_key(params, shard_index=1, shard_count=2)
    with pytest.raises(fk.FkError) as e:
        ctx.write_key_bellman(sk, vk)
    assert e.value.code == 1 and 'shard' in str(e.value)
    sk.free()
    # the whole file: fawkes header (host) + bellman part (GPU) -> load -> prove = the oracle's bytes
    arrays_ok = _key_arrays(key)
    dk3 = ctx.load_key(fk.Parameters(dict(arrays_ok, **shape), r1cs))
    data = pio.store_parameters_dev(ctx, dk3, dict(gamma_g2=arrays_ok['gamma_g2'], ic=arrays_ok['ic']), r1cs, const_tracker_bits=[True, False],
                                    gates_blob=pio.RAW_MAGIC + pio.encode_gate_stream(r1cs))
    assert data.tobytes() == pio.store_parameters(arrays_ok, r1cs, const_tracker_bits=[True, False])
    # ... and with the gate blob written natively (fk_gates_encode, the reference's brotli setting): another blob, the same system and key
    data_n = pio.store_parameters_dev(ctx, dk3, dict(gamma_g2=arrays_ok['gamma_g2'], ic=arrays_ok['ic']), r1cs, const_tracker_bits=[True, False])
    h_n, h_r = pio.read_parameters(data_n), pio.read_parameters(data)
    assert bytes(h_n['bellman']) == bytes(h_r['bellman']) and h_n['num_gates'] == h_r['num_gates'] and len(h_n['gates_blob']) < len(h_r['gates_blob'])
    from helpers import brotli_decompress
    assert brotli_decompress(bytes(h_n['gates_blob'])) == pio.encode_gate_stream(r1cs)
    dk4, dr4, hdr = pio.load_parameters(ctx, data)
    z = fx.witness_mont(z_in, z_aux)
    r, s = fx.mont_fr(3), fx.mont_fr(4)
    a, b, c, aa, bi, ba = oracle.synthesize(csr, z)
    assert ctx.prove_witness(dk4, dr4, z, r, s).tobytes() == oracle.prove(key, a, b, c, z, aa, bi, ba, r, s).tobytes()
    for k_ in (dk, dk2, dk3, dk4):
        k_.free()
    dr4.free()


def _replicate(r1cs, copies):
    """the explicit system fk_r1cs_load_tiled stands for: ONE shared, copy j's inputs at 1 + j*(ni-1), its aux at j*na (include/fawkes_hip.h)"""
    from fawkes_crypto_amd import api
    ni, na = r1cs.num_input, r1cs.num_aux
    n_in = 1 + copies * (ni - 1)
    mats = []
    for ptr, col, val in r1cs.mats:
        col = col.astype(np.int64)
        cols, ptrs = [], [np.zeros(1, np.uint64)]
        for j in range(copies):
            c = col.copy()
            is_in = (c > 0) & (c < ni)
            is_aux = c >= ni
            c[is_in] += j * (ni - 1)
            c[is_aux] += n_in + j * na - ni
            cols.append(c)
            ptrs.append(ptr[1:] + np.uint64(j * len(col)))
        mats.append((np.concatenate(ptrs), np.concatenate(cols).astype(np.uint32), None if val is None else np.tile(val, (copies, 1))))
    return api.R1cs(n_in, copies * na, *mats)


def test_native_gate_encoder_matches_the_restatement(oracle, monkeypatch):
    """fk_gates_encode (csrc/gatestream.hip, host code): Gate::serialize of every gate (cs.rs:184-191) -- byte-equal to the per-term
    Python restatement for a system and for `copies` copies of it (= the explicitly replicated system), as a raw stream and through
    the system's brotli encoder at the reference's setting (quality 9, lgwin 22) and at the fast one the benchmark uses; the native
    decoder returns the system that went in, with the SAME dictionary order whatever the thread count."""
    import fawkes_crypto_amd as fk
    from fawkes_crypto_amd import api, params_io as pio
    from helpers import brotli_decompress
    cs, _, _ = ref.random_r1cs(91, 120, 4, 150)
    r1cs = r1cs_product(fx.r1cs_to_csr(cs))
    want = pio.encode_gate_stream(r1cs)
    raw = api.GateBlob(r1cs, None, fmt=api.FK_GATES_RAW)
    assert raw.data.tobytes() == want and raw.num_gates == r1cs.num_gates
    rep = _replicate(r1cs, 5)
    want5 = pio.encode_gate_stream(rep)
    raw5 = api.GateBlob(r1cs, 5, fmt=api.FK_GATES_RAW)
    assert raw5.data.tobytes() == want5 and raw5.num_gates == 5 * r1cs.num_gates
    if brotli_decompress(b'\x06') is None:            # (the empty stream) -- no libbrotli in this environment
        pytest.skip('libbrotlidec.so.1 not present')
    for q, lg in ((9, 22), (1, 22), (0, 18)):
        b = api.GateBlob(r1cs, 5, fmt=api.FK_GATES_BROTLI, quality=q, lgwin=lg)
        assert b.data.size < len(want5) and brotli_decompress(b.data.tobytes()) == want5
        prof = b.profile()
        assert prof['stream_bytes'] == len(want5) and prof['blob_bytes'] == b.data.size
        g = api.Gates(b.data, api.FK_GATES_BROTLI, b.num_gates, rep.num_input, rep.num_aux)       # (a numpy view: no copy)
        _same_system(g.to_r1cs(), rep)
        b.free(); g.free()
    with pytest.raises(fk.FkError) as e:
        api.GateBlob(r1cs, 5, fmt=api.FK_GATES_BROTLI, quality=12)
    assert e.value.code == 1
    # a system of several blocks: the tables of a 1-thread and of a many-thread decode are the same arrays
    cs2, _, _, _ = fx.fast_r1cs(78, 30000, 3, 31000)
    big = r1cs_product(cs2)
    blob = api.GateBlob(big, 9, fmt=api.FK_GATES_BROTLI, quality=1)
    n_in, n_aux = 1 + 9 * (big.num_input - 1), 9 * big.num_aux
    outs = []
    for threads in ('1', '3', '8'):
        monkeypatch.setenv('FK_HOST_THREADS', threads)
        g = api.Gates(blob.data, api.FK_GATES_BROTLI, blob.num_gates, n_in, n_aux)
        assert g.profile()['parse_threads'] == max(1, int(threads) - (int(threads) > 2)) and g.profile()['blocks'] >= 4
        raw_tab = []
        for k in range(3):
            i = g.info()
            ptr = np.zeros(i['num_gates'] + 1, np.uint64); col = np.zeros(i['nnz'][k], np.uint32); val = np.zeros((i['nnz'][k], 4), np.uint64)
            assert g.lib.fk_gates_export(g.handle, k, ptr.ctypes.data_as(api.C.c_void_p), col.ctypes.data_as(api.C.c_void_p), val.ctypes.data_as(api.C.c_void_p)) == 0
            raw_tab.append((ptr, col, val))
        outs.append((g.info(), raw_tab))
        g.free()
    for info, tabs in outs[1:]:
        assert info == outs[0][0]
        for (p0, c0, v0), (p1, c1, v1) in zip(tabs, outs[0][1]):
            assert np.array_equal(p0, p1) and np.array_equal(c0, c1) and np.array_equal(v0, v1)
    _same_system(api.R1cs(n_in, n_aux, *outs[0][1]), _replicate(big, 9))


def test_gate_decoder_blocks_giant_gate_and_earliest_error(monkeypatch):
    """the threaded decoder's block cutting: a linear combination larger than a block (10 MiB of terms in ONE gate) passes through both
    the in-memory and the decompressing path; with errors in two distant blocks the EARLIEST one is reported (what the reference's
    serial reader would have hit)."""
    import fawkes_crypto_amd as fk
    from fawkes_crypto_amd import api
    from helpers import brotli_compress
    n_big = 300000                                   # x 37 B = 11.1 MB > the 8 MiB block target
    one = (1).to_bytes(32, 'little')
    item = lambda tag, idx, c=one: c + bytes([tag]) + idx.to_bytes(4, 'little')
    small_gate = (1).to_bytes(4, 'little') + item(1, 3) + (1).to_bytes(4, 'little') + item(0, 0) + (0).to_bytes(4, 'little')
    giant = (n_big).to_bytes(4, 'little') + b''.join(item(1, i % 7) for i in range(n_big)) + (0).to_bytes(4, 'little') + (1).to_bytes(4, 'little') + item(1, 6, (5).to_bytes(32, 'little'))
    stream = small_gate * 3 + giant + small_gate * 2
    monkeypatch.setenv('FK_HOST_THREADS', '4')
    for data, fmt in ((stream, api.FK_GATES_RAW), (brotli_compress(stream), api.FK_GATES_BROTLI)):
        if data is None:
            continue
        g = api.Gates(data, fmt, 6, 1, 7)
        i = g.info()
        assert i['nnz'] == (5 + n_big, 5, 1) and i['decoded_bytes'] == len(stream) and i['distinct_coefficients'] == 2
        r = g.to_r1cs()
        assert int(r.mats[0][0][4] - r.mats[0][0][3]) == n_big and np.array_equal(r.mats[0][1][3:3 + 14] - 1, np.arange(14) % 7)
        g.free()
    # two malformed items, ~40 MB apart: the first one's message wins whichever thread meets its block first
    many = small_gate * 600000                       # 51 MB: several blocks
    bad = bytearray(many)
    first, second = 1000 * len(small_gate) + 4 + 32, 590000 * len(small_gate) + 4 + 33
    bad[first] = 9                                   # tag 9: "enum elements overflow"
    bad[second:second + 4] = (1 << 30).to_bytes(4, 'little')     # aux index out of range
    for _ in range(3):
        with pytest.raises(fk.FkError) as e:
            api.Gates(bytes(bad), api.FK_GATES_RAW, 600000, 1, 7)
        assert e.value.code == 7 and 'enum elements overflow' in str(e.value)


def test_load_parameters_plans_levels_early_and_replans(oracle, monkeypatch):
    """host logic of params_io.load_parameters (no GPU: a stand-in context records the calls; the gate blob is decoded by the real native
    decoder): the key is read with FK_KEY_NO_LEVELS and its levels derived while the decoder runs; once the system is resident the
    headroom is checked and the levels are planned again when it is negative; an out-of-memory system upload drops the levels, retries,
    and derives them last; early_levels=False keeps the old order.  (warm=False: the throw-away proof depends on whether the decoder is still busy --
    its guard is tested below.)"""
    from fawkes_crypto_amd import api, params_io as pio
    cs, _, _ = ref.random_r1cs(43, 200, 3, 230)
    csr = fx.r1cs_to_csr(cs)
    key = oracle.setup(csr, **TOXIC)
    r1cs = r1cs_product(csr)
    data = pio.store_parameters(_key_arrays(key), r1cs, const_tracker_bits=[True])

    class Key:
        def __init__(self, log, headroom):
            self.log, self.headroom = log, headroom
        def derive_levels(self): self.log.append('derive')
        def drop_levels(self): self.log.append('drop')
        def levels_headroom(self): self.log.append('headroom'); return self.headroom.pop(0)
        def load_profile(self): return dict(arrays_s=0.0, levels_s=0.0)
        def free(self): self.log.append('key.free')

    class Ctx:
        def __init__(self, headroom):
            self.log, self.headroom = [], headroom
        def load_key_bellman(self, blob, si, sc, zf, flags=0):
            self.log.append('key(flags=%d)' % flags)
            return Key(self.log, self.headroom), b'g' * 128, np.zeros((4, 8), np.uint64)

    class Dr:
        def free(self): pass

    fails = []

    def fake_load(self, ctx):
        ctx.log.append('system')
        if fails:
            raise api.FkError(fails.pop(0), 'r1cs: device allocation failed')
        return Dr()
    monkeypatch.setattr(api.Gates, 'load', fake_load)
    nl = api.FK_KEY_CHECKED | api.FK_KEY_NO_LEVELS
    # 1. the ordinary case: levels underneath the decoding, room confirmed afterwards
    c = Ctx([5 << 30]); tm = {}
    pio.load_parameters(c, data, timings=tm, warm=False)
    assert c.log == ['key(flags=%d)' % nl, 'derive', 'system', 'headroom'] and tm['key_levels_early'] and tm['key_levels_headroom_GiB'] == 5.0 and 'key_levels_replanned_s' not in tm
    # 2. the system took the room the proofs need: planned again
    c = Ctx([-(3 << 30)]); tm = {}
    pio.load_parameters(c, data, timings=tm, warm=False)
    assert c.log == ['key(flags=%d)' % nl, 'derive', 'system', 'headroom', 'derive'] and tm['key_levels_headroom_GiB'] == -3.0 and 'key_levels_replanned_s' in tm
    # 3. the upload itself ran out of memory: levels dropped, upload retried, levels last
    c = Ctx([]); tm = {}; fails.append(5)
    pio.load_parameters(c, data, timings=tm, warm=False)
    assert c.log == ['key(flags=%d)' % nl, 'derive', 'system', 'drop', 'system', 'derive'] and tm['key_levels_early'] is False
    # ... any other failure, or a second out-of-memory, is the caller's: nothing stays behind
    c = Ctx([]); fails.extend([5, 5])
    with pytest.raises(api.FkError):
        pio.load_parameters(c, data, warm=False)
    assert c.log[-1] == 'key.free'
    c = Ctx([]); fails.append(3)
    with pytest.raises(api.FkError):
        pio.load_parameters(c, data, warm=False)
    assert 'drop' not in c.log and c.log[-1] == 'key.free'
    # 4. early_levels=False / overlap=False: the order of the first version
    for kw in (dict(early_levels=False), dict(overlap=False)):
        c = Ctx([])
        pio.load_parameters(c, data, warm=False, **kw)
        assert c.log == ['key(flags=%d)' % nl, 'system', 'derive'], (kw, c.log)
    # 5. the warm-up's guard: skipped when the HBM left beside the levels would not cover the system still to come; otherwise run, and the
    # statistics it touched are reset when they were empty before; nothing it does can raise
    hdr = pio.read_parameters(data)
    cnt = pio.bellman_counts(hdr['bellman'])

    class WKey(Key):
        def counts(self): return dict(m=256, n_a=10, n_b=10)

    class WCtx(Ctx):
        def __init__(self, headroom, launches):
            super().__init__(headroom); self.launches = launches
        def stats(self): self.log.append('stats'); return dict(acc_g1=dict(launches=self.launches), ntt=dict(launches=0))
        def stats_reset(self): self.log.append('stats_reset')
        def dev_alloc(self, n): raise api.FkError(5, 'no device here')
        def dev_free(self, p_): pass
    w = WCtx([1 << 10], 0); tm = {}
    pio._maybe_warm_up(w, WKey(w.log, w.headroom), hdr, cnt, tm)
    assert 'warm_up_skipped' in tm and 'stats' not in w.log and 'warm_up_s' not in tm
    w = WCtx([1 << 40], 0); tm = {}
    pio._maybe_warm_up(w, WKey(w.log, w.headroom), hdr, cnt, tm)
    assert w.log == ['headroom', 'stats', 'stats_reset'] and 'FkError' in tm['warm_up_error'] and tm['warm_up_s'] >= 0
    w = WCtx([1 << 40], 7); tm = {}            # the caller's own counters are running: left alone
    pio._maybe_warm_up(w, WKey(w.log, w.headroom), hdr, cnt, tm)
    assert w.log == ['headroom', 'stats']
    w = WCtx([], 0); tm = {}                    # a key that cannot answer: an error string, never an exception
    pio._maybe_warm_up(w, WKey(w.log, w.headroom), hdr, cnt, tm)
    assert 'IndexError' in tm['warm_up_error']
