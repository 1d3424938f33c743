"""How the brotli quality of the WRITER decides what the gate decoder sees (host only).  The reference writes `Parameters.2` at quality 9, lgwin 22
(setup.rs:26); the benchmark cannot afford that encoder for 61 GB (one thread, ~20 minutes) -- which fast setting gives a blob that decodes like
the reference's?  For each quality: blob size, encode seconds, decode wall / decompressor seconds (best of 3) of the native threaded decoder.
usage: python tools/blob_quality_probe.py [copies] [copies for quality 9]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fawkes_crypto_amd import api  # noqa: E402

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 400
copies9 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
inst, _ = bench.load_rollup_instance()


def leg(q, n):
    n_in, n_aux = 1 + n * (inst.num_input - 1), n * inst.num_aux
    t0 = time.time()
    b = api.GateBlob(inst, n, fmt=api.FK_GATES_BROTLI, quality=q, lgwin=22)
    te = time.time() - t0
    best = None
    for _ in range(3):
        g = api.Gates(b.data, api.FK_GATES_BROTLI, b.num_gates, n_in, n_aux)
        p = g.profile()
        g.free()
        if best is None or p['wall_s'] < best['wall_s']:
            best = p
    sb = b.profile()['stream_bytes']
    print('quality %d, %4d transactions: stream %6.2f GB, blob %7.1f MB (1 : %4.1f), encode %6.1f s, decode %5.2f s wall = %4.2f GB/s (decompressor %5.2f s, parsers waited for %4.2f s)'
          % (q, n, sb / 1e9, b.data.size / 1e6, sb / b.data.size, te, best['wall_s'], sb / best['wall_s'] / 1e9, best['decompressor_s'], best['waited_for_parsers_s']), flush=True)
    b.free()


for q in (1, 2, 3, 5):
    leg(q, copies)
for q in (1, 2, 9):
    leg(q, copies9)
