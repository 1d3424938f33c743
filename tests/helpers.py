"""Shared test helpers.  The oracle (oracle/) is used here only as the checker."""
import json
import os

import numpy as np

import bn254_ref as ref
import c_oracle as co
import fixtures as fx

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
R, Q = ref.R, ref.Q


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def rand_fr_mont(rng, n, kind='uniform'):
    """n Montgomery Fr elements as (n,4) uint64 from a numpy Generator.  kind: uniform | witness."""
    raw = rng.integers(0, 1 << 63, size=(n, 5), dtype=np.uint64)
    vals = []
    for row in raw:
        v = 0
        for x in row:
            v = (v << 63) | int(x)
        vals.append(v % R)
    if kind == 'witness':
        sel = rng.integers(0, 4, size=n)
        for i in range(n):
            if sel[i] == 0:
                vals[i] = 0
            elif sel[i] == 1:
                vals[i] = 1
    return co.limbs_arr([ref.to_mont(v, R) for v in vals]) if n else np.zeros((0, 4), np.uint64)


def mont_ints(arr, p=R):
    """(n,4) Montgomery limbs -> canonical python ints"""
    return [ref.from_mont(x, p) for x in co.ints(arr)]


def g1_bases(n, seed=1):
    """n distinct G1 points k*G as (n,64) raw LE (C oracle, fast)."""
    g = np.frombuffer(ref.g1_raw_le(ref.G1_GEN), np.uint8)
    return co.g1_series(g, fx.mont_fr(0x1234567 + seed * 7919), n)


def g2_bases(n, seed=1):
    g = np.frombuffer(ref.g2_raw_le(ref.G2_GEN), np.uint8)
    return co.g2_series(g, fx.mont_fr(0x7654321 + seed * 104729), n)


def params_from_oracle_key(key, r1cs=None):
    """c_oracle.Key -> fawkes_crypto_amd.Parameters"""
    import fawkes_crypto_amd as fk
    arrays = dict(m=key.m, num_input=key.num_input, num_aux=key.num_aux,
                  alpha_g1=key.alpha_g1, beta_g1=key.beta_g1, beta_g2=key.beta_g2,
                  delta_g1=key.delta_g1, delta_g2=key.delta_g2,
                  h=np.array(key.h), l=np.array(key.l), a=np.array(key.a), b_g1=np.array(key.b_g1), b_g2=np.array(key.b_g2))
    return fk.Parameters(arrays, r1cs)


def r1cs_product(csr):
    """c_oracle.R1csC -> fawkes_crypto_amd.R1cs (same CSR arrays)"""
    import fawkes_crypto_amd as fk
    return fk.R1cs(csr.num_input, csr.num_aux, (csr.A.ptr, csr.A.col, csr.A.val), (csr.B.ptr, csr.B.col, csr.B.val),
                   (csr.C.ptr, csr.C.col, csr.C.val))


def golden_instance():
    """The toy Groth16 instance of tests/golden/proof_golden.json as oracle objects."""
    g = golden('proof_golden.json')
    rows = [tuple([(int(cf, 16), (kind, idx)) for cf, kind, idx in lc] for lc in row) for row in g['rows']]
    cs = ref.R1CS(g['num_input'], g['num_aux'], rows)
    z_in = [int(x, 16) for x in g['z_in']]
    z_aux = [int(x, 16) for x in g['z_aux']]
    tw = {k: int(v, 16) for k, v in g['toxic'].items()}
    return g, cs, z_in, z_aux, tw, int(g['r'], 16), int(g['s'], 16)


TOXIC = dict(tau=0x1f2e3d4c5b6a79880123456789abcdef0fedcba987654321, alpha=0xa11ce, beta=0xb0b, gamma=0xc0ffee, delta=0xdec0de)


def brotli_compress(data, quality=9, lgwin=22):
    """what fawkes' setup writes (setup.rs:26: CompressorWriter::new(_, 4096, 9, 22)), through the system's libbrotlienc.so.1
    (test-side only: the product binds the DECODER, csrc/gatestream.hip).  Returns None when the library is absent."""
    import ctypes as C
    import ctypes.util
    try:
        enc = C.CDLL(ctypes.util.find_library('brotlienc') or 'libbrotlienc.so.1')
    except OSError:
        return None
    enc.BrotliEncoderMaxCompressedSize.restype = C.c_size_t
    enc.BrotliEncoderMaxCompressedSize.argtypes = [C.c_size_t]
    cap = enc.BrotliEncoderMaxCompressedSize(len(data)) or (len(data) + 1024)
    out = C.create_string_buffer(cap)
    n = C.c_size_t(cap)
    enc.BrotliEncoderCompress.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.c_char_p]
    ok = enc.BrotliEncoderCompress(quality, lgwin, 0, len(data), bytes(data), C.byref(n), out)
    assert ok == 1, 'BrotliEncoderCompress failed'
    return out.raw[:n.value]


def brotli_decompress(data, cap=1 << 28):
    """the system's libbrotlidec.so.1, one-shot (test-side check of what fk_gates_encode wrote).  None when the library is absent."""
    import ctypes as C
    import ctypes.util
    try:
        dec = C.CDLL(ctypes.util.find_library('brotlidec') or 'libbrotlidec.so.1')
    except OSError:
        return None
    dec.BrotliDecoderDecompress.argtypes = [C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.c_char_p]
    out = C.create_string_buffer(cap)
    n = C.c_size_t(cap)
    ok = dec.BrotliDecoderDecompress(len(data), bytes(data), C.byref(n), out)
    assert ok == 1, 'BrotliDecoderDecompress failed'
    return out.raw[:n.value]


FULLSIZE_DIGESTS = os.path.join(GOLDEN, 'fullsize_digests.json')


def fullsize_digest(name):
    """entry `name` of tests/golden/fullsize_digests.json (made by tests/golden/make_fullsize_digests.py: the C oracle's 256 proof
    bytes of a BASELINE configuration at its FULL size), or None when the file holds none"""
    if not os.path.exists(FULLSIZE_DIGESTS):
        return None
    return golden('fullsize_digests.json')['entries'].get(name)


def eddsa_batch_inputs(copies=4096):
    """BASELINE configs[2] at full size, as tests/test_gpu_tiled.py::test_config2_full_batch_4096_signatures and
    tests/golden/make_fullsize_digests.py both build it: three eddsa-poseidon signature circuits (oracle/fawkes_circuit.py), dealt
    to `copies` instances by a seeded choice; r and s drawn from the same generator.
    Returns (signatures, one (R1csC of ONE verifier), picks, z (tiled Montgomery witness), r, s)."""
    import random
    import fawkes_circuit as fc
    rnd = random.Random(4096)
    pp, jj = fc.PoseidonParams(4, 8, 54), fc.JubJubBN256()
    sigs = [fc.eddsa_circuit(rnd.randrange(fc.FS), rnd.randrange(ref.R), rnd.randrange(fc.FS), pp, jj)[0] for _ in range(3)]
    one = fx.r1cs_to_csr(sigs[0].r1cs())
    rnd = random.Random(7)
    picks = [rnd.randrange(3) for _ in range(copies)]
    zs = [fx.witness_mont(c.z_in, c.z_aux) for c in sigs]
    ni = one.num_input
    z = np.ascontiguousarray(np.concatenate([zs[0][:1]] + [zs[p][1:ni] for p in picks] + [zs[p][ni:] for p in picks]))
    r, s = fx.mont_fr(rnd.randrange(ref.R)), fx.mont_fr(rnd.randrange(ref.R))
    return sigs, one, picks, z, r, s
