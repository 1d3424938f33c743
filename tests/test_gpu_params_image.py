"""The prover set up from a `Parameters` IMAGE alone -- the reference's own input form (backend/bellman_groth16/mod.rs:139-175: bellman key +
num_gates + brotli(Borsh gates) + const tracker; setup.rs:25-32 writes it, prover.rs:63-90 consumes it) -- at sizes where the threaded gate
decoder, the explicit resident system and the checked key reader all engage:

  * 40 rollup-style transactions (770 881 rows, domain 2^20, 37.7 M matrix terms, 1.4 GB of gate stream): image written at the reference's
    brotli setting is too slow for a test (quality 9: 54 MB/s), so quality 5 here and quality 9 on 6 transactions; the proof out of the
    image is compared BYTE FOR BYTE with the C oracle's proof of the same system (bellman's algorithm restated, oracle/groth16_oracle.c);
  * 218 transactions (4 200 941 rows >= 2^22, domain 2^23, 2.05e8 terms, 7.6 GB of stream): image -> load_parameters(checked) -> proof equal
    to the proof over the tiled resident form (fk_r1cs_load_tiled) of the same circuit and accepted by both verifiers.
bench.py's headline runs the same path at 1741 transactions (33.55 M rows, 61 GB of stream)."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _image_roundtrip(ctx, inst, zs, copies, quality, bench):
    """tiled setup -> Parameters image -> everything dropped -> load_parameters(checked) -> (key, explicit system, vk, tiled proof, timings)"""
    from fawkes_crypto_amd import params_io as pio
    tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
    r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
    z = bench.tile_witness(zs, inst.num_input, copies)
    key0, vk = ctx.setup(inst, copies=copies, **tox)
    dr0 = ctx.load_r1cs(inst, copies=copies)
    tiled_proof = ctx.prove_witness(key0, dr0, z, r, s).tobytes()
    tiled_info = dr0.info()
    tm_w, tm_r = {}, {}
    image = pio.store_parameters_dev(ctx, key0, vk, inst, const_tracker_bits=[True, False, True], copies=copies, quality=quality, lgwin=22, timings=tm_w)
    key0.free(); dr0.free()
    assert isinstance(image, np.ndarray) and image.dtype == np.uint8
    hdr0 = pio.read_parameters(image)
    assert hdr0['num_gates'] == copies * inst.num_gates and hdr0['const_tracker'] == [True, False, True] and len(hdr0['gates_blob']) == tm_w['blob_bytes']
    key, dr, hdr = pio.load_parameters(ctx, image, checked=True, disallow_points_at_infinity=False, timings=tm_r)
    info = dr.info()
    assert info['rows'] == tiled_info['rows'] and tuple(info['nnz']) == tuple(tiled_info['nnz']) and (info['n_a'], info['n_b']) == (tiled_info['n_a'], tiled_info['n_b'])
    assert hdr['gates_info']['decoded_bytes'] == tm_w['gates_encode_profile']['stream_bytes']
    assert np.array_equal(np.asarray(hdr['ic']), np.asarray(vk['ic'])) and bytes(hdr['gamma_g2']) == bytes(vk['gamma_g2'])
    return key, dr, vk, z, r, s, tiled_proof, dict(write=tm_w, read=tm_r)


def test_parameters_image_2p20_rows_proof_equals_the_oracle(ctx, oracle):
    import bench
    import c_oracle as co
    inst, zs = bench.load_rollup_instance()
    # the reference's own blob setting (quality 9, lgwin 22) on a small system first: same proof as the tiled form
    key, dr, vk, z, r, s, tiled_proof, _ = _image_roundtrip(ctx, inst, zs, 6, 9, bench)
    assert ctx.prove_witness(key, dr, z, r, s).tobytes() == tiled_proof
    key.free(); dr.free()
    # FK_KEY_NO_LEVELS + fk_key_derive_levels (how load_parameters reads the key while the blob is being decoded): the key proves the same bytes
    # without levels, and after the levels are derived it carries exactly the plan of a key loaded in one go
    import os
    from fawkes_crypto_amd import api, params_io as pio
    os.environ['FK_MSM_PRE_MIN_LOG2'] = '8'                  # (read at every key load: levels on this small key too)
    try:
        tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
        key0, vk0 = ctx.setup(inst, copies=6, **tox)
        dr0 = ctx.load_r1cs(inst, copies=6)
        bell = ctx.write_key_bellman(key0, vk0)
        want_plan = key0.levels_plan()
        assert any(v['levels'] for v in want_plan.values())
        key0.free()
        k1, _, _ = ctx.load_key_bellman(bell, flags=api.FK_KEY_CHECKED | api.FK_KEY_NO_LEVELS)
        assert not any(k1.precomputed().values())
        assert ctx.prove_witness(k1, dr0, z, r, s).tobytes() == tiled_proof
        k1.derive_levels()
        assert k1.levels_plan() == want_plan and k1.load_profile()['levels_s'] > 0
        assert ctx.prove_witness(k1, dr0, z, r, s).tobytes() == tiled_proof
        k1.free(); dr0.free()
    finally:
        del os.environ['FK_MSM_PRE_MIN_LOG2']
    copies = 40
    key, dr, vk, z, r, s, tiled_proof, tm = _image_roundtrip(ctx, inst, zs, copies, 5, bench)
    try:
        cnt = key.counts()
        assert cnt['m'] == 1 << 20
        got = ctx.prove_witness(key, dr, z, r, s).tobytes()
        assert got == tiled_proof
        one = co.R1csC(inst.num_input, inst.num_aux, *[co.Csr(p_, c_, v_) for p_, c_, v_ in inst.mats])
        a, b, c, aa, bi, ba = co.synthesize_tiled(one, copies, z)
        okey = bench.oracle_key(key, vk, cnt['m'], cnt['num_input'], cnt['num_aux'])
        want = co.prove(okey, a, b, c, z, aa, bi, ba, r, s, threads=min(16, bench.usable_cores()))
        assert got == want.tobytes(), 'proof out of the Parameters image differs from the oracle proof'
        prof = tm['read']['gates_decode_profile']
        print('2^20: gate stream %.2f GB decoded in %.2f s (decompressor %.2f s, %d parsing threads), key read checked %.2f s + levels %.2f s'
              % (prof['blob_bytes'] and tm['write']['gates_encode_profile']['stream_bytes'] / 1e9, tm['read']['gates_decode_s'], prof['decompressor_s'], prof['parse_threads'],
                 tm['read']['key_read_profile']['arrays_s'], tm['read']['key_read_profile']['levels_s']))
    finally:
        key.free(); dr.free()


def test_parameters_image_2p22_rows_explicit_system_equals_tiled(ctx):
    import bench
    import fawkes_crypto_amd as fk
    inst, zs = bench.load_rollup_instance()
    copies = 218
    n = copies * inst.num_gates + 1 + copies * (inst.num_input - 1)
    assert n >= 1 << 22
    ctx.trim()
    t0 = time.time()
    key, dr, vk, z, r, s, tiled_proof, tm = _image_roundtrip(ctx, inst, zs, copies, 1, bench)
    try:
        assert key.counts()['m'] == 1 << 23 and sum(dr.info()['nnz']) > 2.0e8
        assert all(v > 0 for v in key.precomputed().values()), key.precomputed()          # the checked reader's key carries its fixed-base levels
        # ... derived WHILE the blob was still being decoded, and found to leave room once the system was resident
        assert tm['read'].get('key_levels_early') is True and tm['read']['key_levels_headroom_GiB'] >= 0 and 'key_levels_replanned_s' not in tm['read']
        assert key.levels_headroom() > 0
        assert tm['read'].get('warm_up_error') is None           # (the throw-away proof of the warm-up, when the decoder left time for it, ran)
        got = ctx.prove_witness(key, dr, z, r, s).tobytes()
        assert got == tiled_proof
        # the levels can be dropped and derived again (what load_parameters does when a system leaves too little room): same bytes each way
        plan = key.levels_plan()
        head0 = key.levels_headroom()
        key.drop_levels()
        assert not any(key.precomputed().values()) and key.levels_headroom() > head0
        assert ctx.prove_witness(key, dr, z, r, s).tobytes() == tiled_proof
        key.derive_levels()
        assert key.levels_plan() == plan
        assert ctx.prove_witness(key, dr, z, r, s).tobytes() == tiled_proof
        num_input = 1 + copies * (inst.num_input - 1)
        assert bench.pairing_check(vk, z[1:num_input].copy(), got)
        assert fk.api.verify(fk.api.vk_to_borsh(vk), z[1:num_input], got) is True
        # pipelined from pinned memory with two different witnesses: same bytes per slot
        z2 = bench.tile_witness(zs[::-1], inst.num_input, copies)
        zp = [ctx.host_alloc(z.shape), ctx.host_alloc(z.shape)]
        zp[0][:] = z; zp[1][:] = z2
        want2 = ctx.prove_witness(key, dr, z2, r, s).tobytes()
        assert want2 != got
        tk = ctx.prove_witness_submit(key, dr, zp[0], r, s)
        for i in range(4):
            nxt = ctx.prove_witness_submit(key, dr, zp[(i + 1) & 1], r, s)
            assert ctx.prove_witness_wait(tk).tobytes() == (want2 if i & 1 else got), 'pipelined proof %d' % i
            tk = nxt
        ctx.prove_witness_wait(tk)
        for p_ in zp:
            ctx.host_free(p_)
        prof = tm['read']['gates_decode_profile']
        print('218 transactions: %.2f GB of gate stream behind a %.0f MB blob, written in %.1f s, decoded in %.2f s (decompressor %.2f s, parsing %.1f CPU-s on %d threads), '
              'resident system loaded in %.2f s, key read checked %.2f s + levels %.2f s; whole test %.0f s'
              % (tm['write']['gates_encode_profile']['stream_bytes'] / 1e9, tm['write']['blob_bytes'] / 1e6, tm['write']['gates_encode_s'], tm['read']['gates_decode_s'],
                 prof['decompressor_s'], prof['parse_cpu_s'], prof['parse_threads'], tm['read']['r1cs_load_s'], tm['read']['key_read_profile']['arrays_s'],
                 tm['read']['key_read_profile']['levels_s'], time.time() - t0))
    finally:
        key.free(); dr.free()
