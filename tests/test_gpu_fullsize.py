"""BASELINE.json's two large configurations at their FULL sizes on one GPU.  Since round 6 the ORACLE's proof bytes at these sizes are committed
data (tests/golden/fullsize_digests.json: minutes of 16 host threads per proof on the GPU box, tests/golden/make_fullsize_digests.py) and every
route below is compared with them; beside that the checks of rounds 1-5 remain:
the Groth16 pairing equation (oracle/bn254_ref.py verifier -- an independent big-int implementation), a negative control
(one witness element changed: the proof must be rejected), and byte equality between independent routes to the same
proof (single GPU vs ONE fk_multi_prove_r1cs call on 8 ranks with sharded keys and the distributed quotient).

  configs[4]  2^27-row synthetic R1CS with the G2 MSM included -- the largest domain bellman accepts;
              proved once with FK_MSM_PRECOMP=0 (the W-bucket-set path for all five multiplications at the largest index
              arithmetic of every kernel) and once with the key the loader makes by default (fixed-base levels for the
              arrays that fit beside the 48 GiB key): same bytes.
  configs[3]  2^25 rollup-style R1CS, "1024-tx shape", FILLING the domain as BASELINE.md config 4 defines it (rows = 2^25):
              1741 tiled rollup transactions = 33 552 553 rows (99.99 % of 2^25), 1.64e9 matrix terms -- the workload bench.py
              reports since round 4 (rounds 1-3: 1024 transactions, 59 % of the domain).
"""
import numpy as np
import pytest

import bn254_ref as ref

pytestmark = pytest.mark.gpu


def _digest(name, zs=None):
    """the oracle's proof bytes of a full-size configuration (helpers.fullsize_digest), None when the entry was made for another witness set"""
    import hashlib
    from helpers import fullsize_digest
    e = fullsize_digest(name)
    if e is None:
        return None
    if zs is not None and e.get('witness_set_sha256') != hashlib.sha256(np.ascontiguousarray(zs).tobytes()).hexdigest():
        return None
    assert all(e['gpu_equal']) and all(e['pairing'])         # what the generator itself saw on the box that made the file
    return e


def _verifies(bench, vk, z_inputs, proof):
    try:
        return bench.pairing_check(vk, z_inputs, proof.tobytes())
    except AssertionError:
        return False


def test_config4_2p27_rows_with_g2_pairing_checked(ctx):
    import bench
    import fawkes_crypto_amd as fk
    log2n = 27
    r1cs, z = bench.build_workload(ctx, fk, log2n)
    assert r1cs.n_rows == 1 << log2n
    tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
    # FK_MSM_PRECOMP=0 (read at every key load): no fixed-base levels, i.e. the W-bucket-set path for all five
    # multiplications at the largest index arithmetic there is.  (Left alone, the loader gives the arrays whose levels fit
    # beside the 48 GiB key the merged form; that mix is proved below as well and must give the same bytes.)
    import os
    os.environ['FK_MSM_PRECOMP'] = '0'
    try:
        key, vk = ctx.setup(r1cs, **tox)
    finally:
        del os.environ['FK_MSM_PRECOMP']
    assert key.counts()['m'] == 1 << log2n
    assert all(v == 0 for v in key.precomputed().values()), 'expected the W-bucket-set path (no fixed-base levels)'
    dr = ctx.load_r1cs(r1cs)
    info = dr.info()
    assert info['n_b'] > (1 << 25)             # the G2 MSM is part of it: > 3e7 G2 points
    d_z = ctx.dev_alloc(z.nbytes)
    r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
    try:
        ctx.upload(d_z, z)
        proof = ctx.prove_witness_dev(key, dr, d_z, r, s)
        assert _verifies(bench, vk, z[1:r1cs.num_input], proof)
        assert ctx.prove_witness_dev(key, dr, d_z, r, s).tobytes() == proof.tobytes()       # deterministic
        # ... and equal, byte for byte, to the ORACLE's proof of this very system (tests/golden/fullsize_digests.json, entry synthetic2p27:
        # oracle/groth16_oracle.c, 16 host threads, 326 s; made by tests/golden/make_fullsize_digests.py --only synthetic2p27)
        dg = _digest('synthetic2p%d' % log2n)
        assert dg is not None and dg['rows'] == 1 << log2n and dg['num_aux'] == r1cs.num_aux
        import hashlib
        assert hashlib.sha256(np.asarray(vk['ic'], np.uint8).tobytes()).hexdigest() == dg['vk_ic_sha256']
        assert proof.tobytes().hex() == dg['proofs'][0], 'the 2^27 proof differs from the oracle\'s bytes'
        # negative control: one dense witness element changed -> the constraint system is violated -> must be rejected
        j = r1cs.num_input + r1cs.num_aux - 12345       # an output of a product gate
        bad = z[j].copy(); bad[0] ^= np.uint64(2)
        ctx.upload(d_z + j * 32, bad)
        forged = ctx.prove_witness_dev(key, dr, d_z, r, s)
        assert forged.tobytes() != proof.tobytes()
        assert not _verifies(bench, vk, z[1:r1cs.num_input], forged)
        # the key as the loader makes it by default (levels for the arrays that fit): same proof bytes
        key.free()
        key, _ = ctx.setup(r1cs, **tox)
        ctx.upload(d_z + j * 32, z[j])
        assert ctx.prove_witness_dev(key, dr, d_z, r, s).tobytes() == proof.tobytes()
    finally:
        ctx.dev_free(d_z)
        dr.free(); key.free()


def test_config3_2p25_filled_domain_single_gpu_and_one_call_on_8_ranks(ctx):
    """the 1741-transaction system (33.55 M rows on the 2^25 domain): single-GPU proof pairing-checked against its 3482 public
    roots, then reproduced byte for byte by ONE fk_multi_prove_r1cs call on 8 ranks (this box's GPU named eight times: a library
    context and a worker thread per rank, all-to-all and fold inside the library)."""
    import bench
    import fawkes_crypto_amd as fk
    copies, world = 1741, 8
    inst, zs = bench.load_rollup_instance()
    z = bench.tile_witness(zs, inst.num_input, copies)
    num_input = 1 + copies * (inst.num_input - 1)
    n = copies * inst.num_gates + num_input
    log_m = 25
    assert 0.99 * (1 << log_m) <= n <= (1 << log_m)            # the domain is FILLED: what "2^25 constraints" means
    tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
    r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
    # the session's context may still hold the scratch of the 2^27 proofs above (~170 GB of grow-only lane and transform buffers): a
    # service that moves on to another key releases it first, so that this key's fixed-base levels (~120 GB) fit
    ctx.trim()
    dr = ctx.load_r1cs(inst, copies=copies)
    assert dr.info()['rows'] == n and sum(dr.info()['nnz']) > 1.6e9
    d_z = ctx.dev_alloc(z.nbytes)
    ctx.upload(d_z, z)
    key, vk = ctx.setup(inst, copies=copies, **tox)
    assert all(v > 0 for v in key.precomputed().values()), key.precomputed()   # every array of the single-GPU key carries its fixed-base levels (merged bucket sets)
    want = ctx.prove_witness_dev(key, dr, d_z, r, s)
    assert _verifies(bench, vk, z[1:num_input], want)
    # full-size parity against the ORACLE: tests/golden/fullsize_digests.json holds the 256 bytes oracle/groth16_oracle.c produced
    # for this very system, key, witness pair, r and s (tests/golden/make_fullsize_digests.py).  Every route below is compared with `want`, hence with
    # the oracle: pipelined, separate multiplications, the reloaded bellman key, ONE call on 8 ranks and on 2 ranks.
    dg = _digest('rollup%d' % copies, zs)
    assert dg is not None, 'tests/golden/fullsize_digests.json has no applicable entry for the %d-transaction system' % copies
    assert want.tobytes().hex() == dg['proofs'][0], 'the single-GPU proof differs from the oracle\'s bytes at full size'
    # the product's own verifier (fk_verify, host) agrees with the oracle's, also on a proof for other public inputs
    vkb = fk.api.vk_to_borsh(vk)
    assert fk.api.verify(vkb, z[1:num_input], want.tobytes()) is True
    swapped = z[1:num_input].copy(); swapped[[0, 1]] = swapped[[1, 0]]
    assert fk.api.verify(vkb, swapped, want.tobytes()) is False
    # the host-witness pipeline gives the same bytes -- with a DIFFERENT witness in the other slot (the transactions dealt to the
    # copies in reverse order): at this size the early front is on, so proof k + 1's evaluation and sorts are queued out of the
    # other slot while proof k runs (with the same witness in both slots a mix-up would go unnoticed)
    z2 = bench.tile_witness(zs[::-1], inst.num_input, copies)
    d_z2 = ctx.dev_alloc(z2.nbytes)
    ctx.upload(d_z2, z2)
    want2 = ctx.prove_witness_dev(key, dr, d_z2, r, s)
    ctx.dev_free(d_z2)
    assert want2.tobytes() != want.tobytes() and _verifies(bench, vk, z2[1:num_input], want2)
    assert want2.tobytes().hex() == dg['proofs'][1], 'the second witness\' proof differs from the oracle\'s bytes at full size'
    zp = [ctx.host_alloc(z.shape), ctx.host_alloc(z.shape)]
    zp[0][:] = z; zp[1][:] = z2
    tk = ctx.prove_witness_submit(key, dr, zp[0], r, s)
    for i in range(4):
        nxt = ctx.prove_witness_submit(key, dr, zp[(i + 1) & 1], r, s)
        assert ctx.prove_witness_wait(tk).tobytes() == (want2 if i & 1 else want).tobytes(), 'pipelined proof %d' % i
        tk = nxt
    assert ctx.prove_witness_wait(tk).tobytes() == want.tobytes()
    for p_ in zp:
        ctx.host_free(p_)
    del z2
    # the multiplications as separate calls: quotient -> fk_prove_msm_h_dev (H on its own over the resident levels), the four witness
    # multiplications by fk_prove_msms_z_dev, folded on the host = same bytes
    m = 1 << log_m
    d_abc = [ctx.dev_alloc(m * 32) for _ in range(3)]
    d_h = ctx.dev_alloc(m * 32)
    try:
        ctx.r1cs_eval_dev(dr, d_z, *d_abc)
        ctx.quotient_h_dev(d_abc[0], d_abc[1], d_abc[2], n, d_h)
        part = ctx.prove_msms_z_dev(key, d_z, *dr.density_ptrs())
        assert part[:64].tobytes() == bytes(64)
        part[:64] = ctx.prove_msm_h_dev(key, d_h)
        assert ctx.prove_assemble(key, np.stack([part]), r, s).tobytes() == want.tobytes()
    finally:
        for p_ in d_abc + [d_h]:
            ctx.dev_free(p_)
    # `Parameters::write` / `Parameters::read(.., checked = true)` (mod.rs:150-175) at this size: the bellman part of the key -- 146.5 M points,
    # 11 GB -- written from HBM by the GPU (fk_key_write_bellman) and read back with every point checked on the GPU (curve equation,
    # G2 subgroup): the reloaded key proves the same bytes
    import time
    t0 = time.time()
    blob = ctx.write_key_bellman(key, vk)
    t_write = time.time() - t0
    cnt = key.counts()
    assert blob.nbytes == 3 * 64 + 3 * 128 + 4 + num_input * 64 + 5 * 4 + (cnt['n_h'] + cnt['n_l'] + cnt['n_a'] + cnt['n_b']) * 64 + cnt['n_b'] * 128
    key.free()                                  # the key and its fixed-base levels (≈ 140 GB) make room
    t0 = time.time()
    key, gamma_g2, ic = ctx.load_key_bellman(blob, flags=fk.api.FK_KEY_CHECKED)
    t_read = time.time() - t0
    print('Parameters::write of the 2^25 key: %.1f s for %.1f GB; Parameters::read(checked) + fixed-base levels: %.1f s' % (t_write, blob.nbytes / 1e9, t_read))
    assert gamma_g2.tobytes() == np.asarray(vk['gamma_g2'], np.uint8).tobytes() and ic.tobytes() == np.asarray(vk['ic'], np.uint8).tobytes()
    assert ctx.prove_witness_dev(key, dr, d_z, r, s).tobytes() == want.tobytes()
    del blob
    key.free()

    ctx.dev_free(d_z); dr.free()
    ctx.trim()                                  # ... and so does the single-GPU prover's scratch (~70 GB)
    # ---- the same proof from ONE call on 8 ranks: fk_init_devices with this box's GPU named eight times, shard keys from
    # fk_multi_setup_tiled (each rank derives only its shard), one constraint-system replica per rank, every rank evaluates only
    # its rows t = g (mod 8), distributed quotient with the all-to-all inside the library, 1/8 of each multiplication
    mc = fk.MultiContext([0] * world)
    try:
        mkey, mvk = mc.setup(inst, copies=copies, **tox)
        L = (1 << log_m) // world
        for g in range(world):
            assert mc.key_shard(mkey, g).shard_info()['h'] == (g * L, min((g + 1) * L, (1 << log_m) - 1))
        assert all(np.array_equal(mvk[k], vk[k]) for k in vk)
        mdr = mc.load_r1cs(inst, copies=copies)
        got, tm = mc.prove_witness(mkey, mdr, z, r, s, want_timings=True)
        assert got.tobytes() == want.tobytes()
        # ... and pipelined from pinned host memory, twice
        zp = mc.ctx(0).host_alloc(z.shape)
        zp[:] = z
        t0 = mc.prove_witness_submit(mkey, mdr, zp, r, s)
        t1 = mc.prove_witness_submit(mkey, mdr, zp, r, s)
        assert mc.prove_witness_wait(t0).tobytes() == want.tobytes() and mc.prove_witness_wait(t1).tobytes() == want.tobytes()
        mc.ctx(0).host_free(zp)
        mkey.free(); mdr.free()
    finally:
        mc.close()
    # ... and on 2 ranks: the exchange-free schedule (round 4) -- rank 0 holds ALL of h (it evaluates a, b, c, computes the whole quotient
    # and H) and, its fixed work counted, l and the head of a; rank 1 holds no h, the rest of a, b_g1 and all of b_g2 -- every piece
    # with its fixed-base levels, nothing exchanged but the 384-byte partial sums
    mc = fk.MultiContext([0, 0])
    try:
        mkey, _ = mc.setup(inst, copies=copies, **tox)
        i0, i1 = mc.key_shard(mkey, 0).shard_info(), mc.key_shard(mkey, 1).shard_info()
        assert i0['h'] == (0, (1 << log_m) - 1) and i1['h'][0] == i1['h'][1]
        assert i0['l'][1] - i0['l'][0] == copies * inst.num_aux and i1['l'] == (i0['l'][1], i0['l'][1]) and 0 < i0['a'][1] == i1['a'][0]
        assert i1['b_g2'][1] - i1['b_g2'][0] > 2e7 and i0['b_g2'][1] == i0['b_g2'][0] and i0['b'][1] == 0
        pre0, pre1 = mc.key_shard(mkey, 0).precomputed(), mc.key_shard(mkey, 1).precomputed()
        assert pre0['h'] > 0 and pre0['l'] > 0 and pre0['a'] > 0 and pre1['h'] == 0 and pre1['b_g2'] > 0 and pre1['b_g1'] > 0 and pre1['l'] == 0
        mdr = mc.load_r1cs(inst, copies=copies)
        assert mc.prove_witness(mkey, mdr, z, r, s).tobytes() == want.tobytes()
        mkey.free(); mdr.free()
    finally:
        mc.close()


def test_full_size_properties_transform_round_trips_and_two_routes_to_one_multiplication(ctx):
    """Size-independent properties at BASELINE's full sizes, where no CPU oracle follows (SURVEY 8(c) substitute pins):
    * iNTT(NTT(x)) = x and icoset(coset(x)) = x on 2^25 and 2^26 random elements, every byte compared;
    * ONE 2^25-point G1 multiplication by two independent routes: over the key's resident h array with its fixed-base levels (one
      merged bucket set, c = 22, 12 digits per scalar) and over the SAME points as plain device memory in two halves (the W-bucket-set
      path, c = 20, 13 digits; bare-pointer entry point), the halves added by the oracle's group law -- same affine point."""
    import bench
    import c_oracle as co
    for log_n in (25, 26):
        n = 1 << log_n
        d = ctx.dev_alloc(n * 32)
        ctx.gen_scalars_dev(d, n, 4242 + log_n, 0)
        want = ctx.download(d, n * 32, np.uint64)
        for coset in (False, True):
            ctx.ntt_dev(d, log_n, inverse=False, coset=coset)
            mid = ctx.download(d, 1 << 20, np.uint64)
            assert not np.array_equal(mid, want[:mid.size])
            ctx.ntt_dev(d, log_n, inverse=True, coset=coset)
            got = ctx.download(d, n * 32, np.uint64)
            assert np.array_equal(got, want), (log_n, coset)
        ctx.dev_free(d)
    ctx.trim()
    # a synthetic key of the benchmark's shape: 2^25 - 1 valid h bases with their levels
    m = 1 << 25
    key = ctx.synthetic_key(m, 2, 1 << 20, 1 << 20, 1 << 19, seed=5)
    assert key.precomputed()['h'] > 0
    n_h = m - 1
    d_s = ctx.dev_alloc(n_h * 32)
    ctx.gen_scalars_dev(d_s, n_h, 99, 0)
    merged = ctx.prove_msm_array_dev(key, 'h', d_s)
    bases = key.download('h')
    key.free()
    d_b = ctx.dev_alloc(bases.nbytes)
    ctx.upload(d_b, bases)
    half = n_h // 2
    p0 = ctx.msm_g1_dev(d_b, d_s, half)
    p1 = ctx.msm_g1_dev(d_b + half * 64, d_s + half * 32, n_h - half)
    ctx.dev_free(d_b); ctx.dev_free(d_s)
    assert merged.tobytes() != bytes(64) and merged.tobytes() == np.asarray(co.g1_add(p0, p1)).tobytes()


def test_config3_parameters_image_at_full_size_equals_the_oracle_bytes(ctx):
    """The benchmark's own path at its full size inside the test suite: the 1741-transaction key and circuit WRITTEN as a `Parameters` image
    (Parameters::write, mod.rs:150-157), everything dropped, the prover set up from the image alone (load_parameters: fk_gates_decode ->
    fk_r1cs_load_gates, 1.64e9 explicit terms; fk_key_load_bellman(checked)) -- and both benchmark witnesses proved through the two-slot
    host-witness pipeline: the 256 bytes equal the ORACLE's (tests/golden/fullsize_digests.json), which is what bench.py's `proof_sha256` names."""
    import hashlib
    import bench
    from fawkes_crypto_amd import params_io as pio
    copies = 1741
    inst, zs = bench.load_rollup_instance()
    dg = _digest('rollup%d' % copies, zs)
    assert dg is not None
    tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
    r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
    ctx.trim()
    key, vk = ctx.setup(inst, copies=copies, **tox)
    image = pio.store_parameters_dev(ctx, key, vk, inst, copies=copies, quality=1, lgwin=22)      # (quality 1: the fastest writer; any setting decodes to the same stream)
    key.free()
    key, dr, hdr = pio.load_parameters(ctx, image, checked=True)
    del image
    assert sum(dr.info()['nnz']) > 1.6e9 and dr.info()['rows'] == dg['rows']
    assert hashlib.sha256(np.asarray(hdr['ic'], np.uint8).tobytes()).hexdigest() == dg['vk_ic_sha256']
    nv = dg['num_input'] + dg['num_aux']
    zp = [ctx.host_alloc((nv, 4)), ctx.host_alloc((nv, 4))]
    bench.tile_witness(zs, inst.num_input, copies, out=zp[0])
    bench.tile_witness(zs[::-1], inst.num_input, copies, out=zp[1])
    try:
        tk = ctx.prove_witness_submit(key, dr, zp[0], r, s)
        for i in range(3):
            nxt = ctx.prove_witness_submit(key, dr, zp[(i + 1) & 1], r, s)
            assert ctx.prove_witness_wait(tk).tobytes().hex() == dg['proofs'][i & 1], 'pipelined proof %d from the Parameters image differs from the oracle\'s bytes' % i
            tk = nxt
        ctx.prove_witness_wait(tk)
        # one proof at a time (the chunked hand-over: fk_prove_r1cs)
        assert ctx.prove_witness(key, dr, zp[1], r, s).tobytes().hex() == dg['proofs'][1]
    finally:
        for p_ in zp:
            ctx.host_free(p_)
        dr.free(); key.free()
        ctx.trim()
