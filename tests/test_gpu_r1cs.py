"""GPU parity of the device-resident constraint system: SpMV a = Az, b = Bz, c = Cz and
the witness-in / proof-out entry point against the oracle's ProvingAssignment restatement.  Bit-exact."""
import numpy as np
import pytest

import bn254_ref as ref
import fixtures as fx
from helpers import params_from_oracle_key, r1cs_product, TOXIC

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape', [(1, 1, 1, 3), (2, 40, 3, 44), (3, 500, 2, 480), (4, 3000, 5, 3100)])
def test_spmv_vs_oracle(ctx, oracle, shape):
    seed, gates, nin, naux = shape
    cs, z_in, z_aux = ref.random_r1cs(seed, gates, nin, naux)
    csr = fx.r1cs_to_csr(cs)
    z = fx.witness_mont(z_in, z_aux)
    want = oracle.synthesize(csr, z)
    dr = ctx.load_r1cs(r1cs_product(csr))
    info = dr.info()
    rows = gates + nin
    assert info['rows'] == rows
    assert info['n_a'] == nin + int(want[3].sum()) and info['n_b'] == int(want[4].sum()) + int(want[5].sum())
    m = 1
    while m < rows:
        m *= 2
    d = [ctx.dev_alloc(m * 32) for _ in range(3)]
    d_z = ctx.dev_alloc(z.nbytes)
    try:
        ctx.upload(d_z, z)
        ctx.r1cs_eval_dev(dr, d_z, *d)
        for k in range(3):
            got = ctx.download(d[k], rows * 32, np.uint64).reshape(-1, 4)
            assert np.array_equal(got, want[k]), 'matrix %d' % k
    finally:
        for p in d + [d_z]:
            ctx.dev_free(p)
        dr.free()


def test_prove_witness_bit_exact(ctx, oracle):
    """witness vector in -> 256-byte proof out, everything else resident in HBM"""
    import fawkes_crypto_amd as fk
    cs, z_in, z_aux = ref.random_r1cs(77, 900, 3, 950)
    csr = fx.r1cs_to_csr(cs)
    key = oracle.setup(csr, **TOXIC)
    params = params_from_oracle_key(key, r1cs_product(csr))
    dk = ctx.load_key(params)
    dr = ctx.load_r1cs(params.r1cs)
    z = fx.witness_mont(z_in, z_aux)
    r, s = fx.mont_fr(0x13579), fx.mont_fr(0x2468a)
    a, b, c, aa, bi, ba = oracle.synthesize(csr, z)
    want = oracle.prove(key, a, b, c, z, aa, bi, ba, r, s)
    got = ctx.prove_witness(dk, dr, z, r, s)
    assert got.tobytes() == want.tobytes()
    # the host mirror of prove() can use it too
    _, proof = fk.prove_with_rs(ctx, params, dk, z[:3], z[3:], r, s, device_r1cs=dr)
    assert proof.to_bytes() == want.tobytes()
    assert ref.verify(fx.key_to_py(key), z_in[1:], ref.proof_from_borsh(proof.to_bytes()))
    # a constraint system that does not belong to the key is rejected
    cs2, _, _ = ref.random_r1cs(78, 900, 3, 951)
    dr2 = ctx.load_r1cs(r1cs_product(fx.r1cs_to_csr(cs2)))
    with pytest.raises(fk.FkError) as e:
        ctx.prove_witness(dk, dr2, np.zeros((954, 4), np.uint64), r, s)
    assert e.value.code == 6


def test_r1cs_load_rejects_bad_input(ctx):
    import fawkes_crypto_amd as fk
    one = fx.mont_fr(1)
    ptr = np.array([0, 1], np.uint64)
    good = (ptr, np.array([1], np.uint32), one.reshape(1, 4))
    bad = (ptr, np.array([9], np.uint32), one.reshape(1, 4))
    with pytest.raises(fk.FkError):
        ctx.load_r1cs(fk.R1cs(1, 1, bad, good, good))


def _ragged_system(seed, lens_choices, gates, nin, naux):
    """random matrices with the given row lengths (zero-length rows and ONE coefficients included); not satisfiable, the
    product does not need it"""
    rng = np.random.default_rng(seed)
    nv = nin + naux
    coeffs = fx.co.limbs_arr([1, ref.R - 1, 2] + [int(x) for x in rng.integers(3, 2**62, 40)])
    coeffs = np.stack([fx.mont_fr(int.from_bytes(c.tobytes(), 'little')) for c in coeffs])

    def mat():
        lens = rng.choice(lens_choices, size=gates)
        ptr = np.zeros(gates + 1, np.uint64)
        ptr[1:] = np.cumsum(lens)
        nnz = int(ptr[-1])
        return fx.co.Csr(ptr, rng.integers(0, nv, nnz).astype(np.uint32), np.ascontiguousarray(coeffs[rng.integers(0, len(coeffs), nnz)]))

    return fx.co.R1csC(nin, naux, mat(), mat(), mat())


@pytest.mark.parametrize('lens', [[0, 1, 1, 1, 2, 3, 4, 5, 31, 32, 33, 100, 257, 512], [1, 1, 1, 7, 8, 9], [40, 41, 1500]])
@pytest.mark.parametrize('copies', [1, 5])
def test_spmv_long_rows_vs_oracle(ctx, oracle, monkeypatch, lens, copies):
    """rows of very different lengths (the Poseidon shape) go through the length-class kernel: same result as the oracle,
    as the single-class kernel (FK_SPMV_BIN_MIN=0), and as a tiled system"""
    base = _ragged_system(len(lens) * 10 + copies, lens, 700, 3, 300)
    csr = fx.tile_r1cs(base, copies) if copies > 1 else base
    nv, rows = csr.num_input + csr.num_aux, csr.num_gates + csr.num_input
    rnd = np.random.default_rng(copies)
    z = fx.co.limbs_arr([int(x) % ref.R for x in rnd.integers(0, 2**63, nv).astype(object) * (2**190 + 12345)])
    want = oracle.synthesize(csr, z)
    m = 1 << max(rows - 1, 1).bit_length()
    d = [ctx.dev_alloc(m * 32) for _ in range(3)]
    d_z = ctx.dev_alloc(z.nbytes)
    ctx.upload(d_z, z)
    loads = [lambda: ctx.load_r1cs(r1cs_product(csr))]
    if copies > 1:
        loads.append(lambda: ctx.load_r1cs(r1cs_product(base), copies=copies))
    try:
        for env in ('', '0'):
            if env:
                monkeypatch.setenv('FK_SPMV_BIN_MIN', env)
            for load in loads:
                dr = load()
                for p in d:
                    ctx.upload(p, np.full(rows * 4, 0xdeadbeefdeadbeef, np.uint64))
                ctx.r1cs_eval_dev(dr, d_z, *d)
                for k in range(3):
                    assert np.array_equal(ctx.download(d[k], rows * 32, np.uint64).reshape(-1, 4), want[k]), (env, k)
                dr.free()
    finally:
        for p in d + [d_z]:
            ctx.dev_free(p)


@pytest.mark.parametrize('copies,gates', [(64, 90), (65, 211), (130, 64), (257, 33), (1024, 17), (1000, 9), (1088, 5)])
def test_spmv_wave_form_of_batch_circuits(ctx, oracle, copies, gates):
    """batch circuits of 64 copies and more: rows of 4 .. 63 terms are evaluated one wave per (row, 64 copies)
    (spmv_tiled_wave_kernel), the others by the length-class kernel -- every boundary length, copy counts that do not fill the
    last wave, against the oracle's evaluation of the explicitly replicated system; the slices a multi-GPU rank evaluates
    (which keep the length-class form) agree with it too"""
    lens = [0, 1, 2, 3, 4, 5, 7, 8, 9, 31, 32, 33, 62, 63, 64, 65, 127, 128, 300]
    base = _ragged_system(copies * 7 + gates, lens, gates, 3, 150)
    csr = fx.tile_r1cs(base, copies)
    nv, rows = csr.num_input + csr.num_aux, csr.num_gates + csr.num_input
    rnd = np.random.default_rng(copies)
    z = fx.co.limbs_arr([int(x) % ref.R for x in rnd.integers(0, 2**63, nv).astype(object) * (2**190 + 12345)])
    want = oracle.synthesize_tiled(base, copies, z)
    ref_full = oracle.synthesize(csr, z)
    for k in range(3):
        assert np.array_equal(want[k], ref_full[k])
    log_m = max(rows - 1, 1).bit_length()
    m = 1 << log_m
    d = [ctx.dev_alloc(m * 32) for _ in range(3)]
    d_z = ctx.dev_alloc(z.nbytes)
    ctx.upload(d_z, z)
    dr = ctx.load_r1cs(r1cs_product(base), copies=copies)
    try:
        for p in d:
            ctx.upload(p, np.full(m * 4, 0xdeadbeefdeadbeef, np.uint64))
        ctx.r1cs_eval_dev(dr, d_z, *d)
        for k in range(3):
            assert np.array_equal(ctx.download(d[k], rows * 32, np.uint64).reshape(-1, 4), want[k]), k
        for log_w, rank in ((1, 1), (2, 3)):
            W, L = 1 << log_w, m >> log_w
            ctx.r1cs_eval_slice_dev(dr, d_z, log_m, rank, log_w, *d)
            for k in range(3):
                got = ctx.download(d[k], L * 32, np.uint64).reshape(-1, 4)
                full = np.zeros((m, 4), np.uint64); full[:rows] = want[k]
                assert np.array_equal(got, full[rank::W]), (log_w, rank, k)
    finally:
        dr.free()
        for p in d + [d_z]:
            ctx.dev_free(p)


def test_spmv_large_flat_system_block_ordering(ctx, oracle):
    """a flat system of more than 16 x 4096 rows with long rows: its length classes are ordered by length inside blocks of
    consecutive rows (locality), not over the whole system -- same a, b, c as the oracle, whole and as multi-GPU slices"""
    lens = [0, 1, 1, 1, 2, 3, 4, 9, 31, 32, 33, 70, 200]
    base = _ragged_system(4242, lens, 70001, 3, 50000)
    nv, rows = base.num_input + base.num_aux, base.num_gates + base.num_input
    rnd = np.random.default_rng(11)
    z = fx.co.limbs_arr([int(x) % ref.R for x in rnd.integers(0, 2**63, nv).astype(object) * (2**190 + 12345)])
    want = oracle.synthesize(base, z)
    log_m = max(rows - 1, 1).bit_length()
    m = 1 << log_m
    d = [ctx.dev_alloc(m * 32) for _ in range(3)]
    d_z = ctx.dev_alloc(z.nbytes)
    ctx.upload(d_z, z)
    dr = ctx.load_r1cs(r1cs_product(base))
    try:
        ctx.r1cs_eval_dev(dr, d_z, *d)
        for k in range(3):
            assert np.array_equal(ctx.download(d[k], rows * 32, np.uint64).reshape(-1, 4), want[k]), k
        for log_w, rank in ((1, 0), (3, 5)):
            W, L = 1 << log_w, m >> log_w
            ctx.r1cs_eval_slice_dev(dr, d_z, log_m, rank, log_w, *d)
            for k in range(3):
                got = ctx.download(d[k], L * 32, np.uint64).reshape(-1, 4)
                full = np.zeros((m, 4), np.uint64); full[:rows] = want[k]
                assert np.array_equal(got, full[rank::W]), (log_w, rank, k)
    finally:
        dr.free()
        for p in d + [d_z]:
            ctx.dev_free(p)


def _local_system(seed, lens_choices, gates, nin, naux):
    """as _ragged_system, but a row reads variables allocated around its own position and before, the way a circuit's gates do"""
    rng = np.random.default_rng(seed)
    nv = nin + naux
    coeffs = fx.co.limbs_arr([1, ref.R - 1, 2] + [int(x) for x in rng.integers(3, 2**62, 40)])
    coeffs = np.stack([fx.mont_fr(int.from_bytes(c.tobytes(), 'little')) for c in coeffs])

    def mat():
        lens = rng.choice(lens_choices, size=gates)
        ptr = np.zeros(gates + 1, np.uint64)
        ptr[1:] = np.cumsum(lens)
        nnz = int(ptr[-1])
        row = np.repeat(np.arange(gates, dtype=np.int64), lens)
        newest = nin + (row + 1) * naux // gates                      # the variables that exist when the row is written
        col = (newest - 1 - rng.integers(0, 3000, nnz)).clip(0, nv - 1)
        col[rng.integers(0, nnz, nnz // 50)] = 0                        # the constant ONE everywhere
        return fx.co.Csr(ptr, col.astype(np.uint32), np.ascontiguousarray(coeffs[rng.integers(0, len(coeffs), nnz)]))

    return fx.co.R1csC(nin, naux, mat(), mat(), mat())


@pytest.mark.parametrize('local', [True, False])
def test_prove_chunked_hand_over_same_bytes(ctx, local):
    """fk_prove_r1cs hands a host witness over in pieces and evaluates a window of rows behind each piece (fk_r1cs_windows): the proof is
    the one of the whole-witness path (fk_prove_r1cs_dev, checked against the oracle above and in test_gpu_params_image).  local = False:
    rows that read variables from anywhere -- the first window needs all of z, the chunking degenerates and the bytes are still the same."""
    lens = [0, 1, 1, 1, 2, 3, 4, 9, 31, 32, 33, 70, 200]
    gates, nin, naux = 140001, 3, 120000
    base = (_local_system if local else _ragged_system)(99, lens, gates, nin, naux)
    nv = nin + naux
    dr = ctx.load_r1cs(r1cs_product(base))
    w = dr.windows()
    assert w is not None and len(w['need']) == 8 and w['rows'][0] == 0 and w['rows'][-1] == gates and w['need'][-1] == nv
    assert all(x % 4096 == 0 for x in w['rows'][:-1]) and sorted(w['rows']) == w['rows'] and sorted(w['need']) == w['need']
    if local:
        assert w['need'][0] < nv // 4 and w['need'][3] < 3 * nv // 4       # the pieces really are pieces
        col = np.concatenate([m.col[:int(m.ptr[w['rows'][1]])] for m in (base.A, base.B, base.C)])
        assert w['need'][0] == max(int(col.max()) + 1, nin)
    else:
        assert w['need'][0] >= nv - 5
    key, _ = ctx.setup(r1cs_product(base), **{k: fx.mont_fr(v) for k, v in TOXIC.items()})
    rnd = np.random.default_rng(5)
    zs = []
    for t in range(2):
        z = fx.co.limbs_arr([1] + [int(x) % ref.R for x in rnd.integers(1, 2**63, nv - 1).astype(object) * (2**190 + 12345 + t)])
        z[0] = fx.mont_fr(1)
        zs.append(z)
    r, s = fx.mont_fr(0x1111), fx.mont_fr(0x2222)
    d_z = ctx.dev_alloc(zs[0].nbytes)
    try:
        ctx.upload(d_z, zs[1])
        other = ctx.prove_witness_dev(key, dr, d_z, r, s)        # leaves ANOTHER witness's a, b, c in the staging buffers
        got = ctx.prove_witness(key, dr, zs[0], r, s)             # chunked
        ctx.upload(d_z, zs[0])
        want = ctx.prove_witness_dev(key, dr, d_z, r, s)         # whole
        assert got.tobytes() == want.tobytes() and got.tobytes() != other.tobytes()
        got2 = ctx.prove_witness(key, dr, zs[1], r, s)
        assert got2.tobytes() == other.tobytes()
    finally:
        ctx.dev_free(d_z)
        dr.free()
        key.free()
