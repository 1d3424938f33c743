"""A/B of load_parameters(early_levels=...) at the benchmark size on one box: seconds from the Parameters image to the first proof, with the
fixed-base levels derived after the constraint system is resident (rounds 5a) or underneath the gate decoding (5b).  The context is trimmed
between the runs so that every first proof allocates its scratch again.  usage: python tools/load_probe.py [copies]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
import fawkes_crypto_amd as fk  # noqa: E402
from fawkes_crypto_amd import params_io as pio  # noqa: E402

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 1741
ctx = fk.Context(0)
inst, zs = bench.load_rollup_instance()
tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
z = bench.tile_witness(zs, inst.num_input, copies)
zp = ctx.host_alloc(z.shape)
zp[:] = z
key, vk = ctx.setup(inst, copies=copies, **tox)
image = pio.store_parameters_dev(ctx, key, vk, inst, copies=copies, quality=int(os.environ.get("BLOB_QUALITY", "2")), lgwin=22)
key.free()
want = None
for early in ((False, True, False, True) if os.environ.get("PROBE_AB", "1") == "1" else (True, True)):
    ctx.trim()
    tm = {}
    t0 = time.perf_counter()
    key, dr, hdr = pio.load_parameters(ctx, image, checked=True, timings=tm, early_levels=early)
    t1 = time.perf_counter()
    p1, tm1 = ctx.prove_witness(key, dr, zp, r, s, want_timings=True)
    p1 = p1.tobytes()
    t2 = time.perf_counter()
    p2 = ctx.prove_witness(key, dr, zp, r, s).tobytes()
    t3 = time.perf_counter()
    assert p1 == p2 and (want is None or p1 == want)
    want = p1
    print(json.dumps(dict(early_levels=early, load_s=round(t1 - t0, 2), first_proof_s=round(t2 - t1, 2), second_proof_s=round(t3 - t2, 3), to_first_proof_s=round(t2 - t0, 2),
                          decode_s=round(tm['gates_decode_s'], 2), key_read_s=round(tm['key_read_s'], 2), levels_s=round(tm['key_levels_s'], 2),
                          r1cs_load_s=round(tm['r1cs_load_s'], 2), header_s=round(tm.get('header_s', 0), 3), gates_free_s=round(tm.get('gates_free_s', 0), 3), headroom_GiB=tm.get('key_levels_headroom_GiB'), warm_up_s=tm.get('warm_up_s'), warm_up_error=tm.get('warm_up_error'), first_proof_stages_ms={k: round(v, 1) for k, v in tm1.items()}, levels=key.precomputed())), flush=True)
    key.free(); dr.free()
