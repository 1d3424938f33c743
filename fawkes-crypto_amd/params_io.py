"""
Reading / writing the reference's on-disk `Parameters`.

File layout (fawkes header: /root/reference/fawkes-crypto/src/backend/bellman_groth16/mod.rs:150-175):
    u32 LE   num_gates                      (Borsh u32, mod.rs:152)
    u32 LE   blob length, blob bytes        (Borsh Vec<u8>, mod.rs:153): brotli(concat of Borsh gates), setup.rs:26-32
    u32 LE   const_tracker bit length       (mod.rs:151,154)
    u32 LE   byte length, bytes             (Borsh Vec<u8> of BitVec::to_bytes(), mod.rs:155; bit i = byte i/8, MSB first)
    ...      bellman Parameters::write      (mod.rs:156)  -> fk_key_load_bellman (csrc/keyfile.hip), SURVEY Appendix B.2
Gate stream inside the blob (circuit/r1cs/cs.rs:193-223): per gate three parts, each
    u32 LE count, then count x ( 32 B canonical LE Fr | u8 tag 0 = Input / 1 = Aux (lc.rs:144-149) | u32 LE index ).

The gate blob is decoded natively (csrc/gatestream.hip: streaming brotli through the system's libbrotlidec.so.1 -> CSR with
dictionary-coded coefficients -> resident constraint system); `decode_gate_stream` below is the slow per-term restatement
the tests compare it with.  Writing a reference-format blob needs a brotli ENCODER: pass `compress=` (the tests bind
libbrotlienc.so.1); without it `store_parameters` stores the raw gate stream behind RAW_MAGIC.  The bellman part and the
bit-vector packing are restated from the un-vendored crates and could not be checked against a file written by the reference.
"""
import struct

import numpy as np

from . import api

FQ = api.FQ_MODULUS
FR = api.FR_MODULUS
_R = 1 << 256
RAW_MAGIC = b'FKRAWGATES\x00'     # marks a gate blob stored uncompressed by this module (never produced by fawkes)


# ------------------------------------------------------------------------------------------ bit vector
def bits_to_bytes(bits):
    """bit-vec 0.6 `BitVec::to_bytes`: first bit = high-order bit of byte 0"""
    out = bytearray((len(bits) + 7) // 8)
    for i, b in enumerate(bits):
        if b:
            out[i >> 3] |= 0x80 >> (i & 7)
    return bytes(out)


def bytes_to_bits(data, nbits):
    if nbits > len(data) * 8:
        raise ValueError('inconsistent bitvec length')          # mod.rs:165-167
    return [bool(data[i >> 3] & (0x80 >> (i & 7))) for i in range(nbits)]


# ------------------------------------------------------------------------------------------ gate stream
def encode_gate_stream(r1cs):
    """api.R1cs -> bytes in the reference's Borsh gate format (cs.rs:193-213), uncompressed."""
    out = bytearray()
    rinv = pow(_R, -1, FR)
    for g in range(r1cs.num_gates):
        for ptr, col, val in r1cs.mats:
            lo, hi = int(ptr[g]), int(ptr[g + 1])
            out += struct.pack('<I', hi - lo)
            for k in range(lo, hi):
                coeff = 1 if val is None else api.limbs_to_int(val[k]) * rinv % FR
                v = int(col[k])
                out += coeff.to_bytes(32, 'little')
                if v < r1cs.num_input:
                    out += b'\x00' + struct.pack('<I', v)
                else:
                    out += b'\x01' + struct.pack('<I', v - r1cs.num_input)
    return bytes(out)


def decode_gate_stream(data, num_gates, num_input, num_aux):
    """bytes (decompressed blob) -> api.R1cs.  Mirrors GateStreamedIterator (cs.rs:215-223)."""
    pos = 0
    mats = [([0], [], []) for _ in range(3)]
    for _ in range(num_gates):
        for ptr, col, val in mats:
            if pos + 4 > len(data):
                raise ValueError('gate stream truncated')
            (cnt,) = struct.unpack_from('<I', data, pos)
            pos += 4
            if pos + cnt * 37 > len(data):
                raise ValueError('gate stream truncated')
            for _k in range(cnt):
                coeff = int.from_bytes(data[pos:pos + 32], 'little')
                tag = data[pos + 32]
                (idx,) = struct.unpack_from('<I', data, pos + 33)
                pos += 37
                if tag == 0:
                    if idx >= num_input:
                        raise ValueError('input index out of range')
                    v = idx
                elif tag == 1:
                    if idx >= num_aux:
                        raise ValueError('aux index out of range')
                    v = num_input + idx
                else:
                    raise ValueError('enum elements overflow')        # cs.rs:209
                if coeff >= FR:
                    raise ValueError('non-canonical field element')
                col.append(v)
                val.append(coeff * _R % FR)
            ptr.append(len(col))
    conv = []
    for ptr, col, val in mats:
        v = np.frombuffer(b''.join(x.to_bytes(32, 'little') for x in val), np.uint64).reshape(-1, 4).copy() if val else np.zeros((0, 4), np.uint64)
        conv.append((np.array(ptr, np.uint64), np.array(col, np.uint32), v))
    return api.R1cs(num_input, num_aux, *conv)


# ------------------------------------------------------------------------------------------ bellman part (small keys, tests)
def _fq_be(x_mont_le):
    """32 raw Montgomery LE bytes -> 32 canonical big-endian bytes"""
    return (int.from_bytes(bytes(x_mont_le), 'little') * pow(_R, -1, FQ) % FQ).to_bytes(32, 'big')


def g1_uncompressed(raw):
    raw = bytes(raw)
    if raw == bytes(64):
        return b'\x40' + bytes(63)
    return _fq_be(raw[:32]) + _fq_be(raw[32:])


def g2_uncompressed(raw):
    raw = bytes(raw)
    if raw == bytes(128):
        return b'\x40' + bytes(127)
    x0, x1, y0, y1 = (raw[i * 32:(i + 1) * 32] for i in range(4))
    return _fq_be(x1) + _fq_be(x0) + _fq_be(y1) + _fq_be(y0)       # c1 before c0


def encode_bellman_parameters(k):
    """k: dict of raw Montgomery-LE arrays (alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1, delta_g2, ic, h, l, a, b_g1,
    b_g2) -> bellman `Parameters::write` bytes.  Python big-int conversion: for tests / small keys."""
    out = bytearray()
    out += g1_uncompressed(k['alpha_g1']) + g1_uncompressed(k['beta_g1']) + g2_uncompressed(k['beta_g2'])
    out += g2_uncompressed(k['gamma_g2']) + g1_uncompressed(k['delta_g1']) + g2_uncompressed(k['delta_g2'])
    for name, enc in (('ic', g1_uncompressed), ('h', g1_uncompressed), ('l', g1_uncompressed), ('a', g1_uncompressed),
                      ('b_g1', g1_uncompressed), ('b_g2', g2_uncompressed)):
        arr = np.asarray(k[name], np.uint8)
        arr = arr.reshape(-1, 128 if name == 'b_g2' else 64)
        out += struct.pack('>I', arr.shape[0])
        for row in arr:
            out += enc(row)
    return bytes(out)


# ------------------------------------------------------------------------------------------ the fawkes wrapper
def write_parameters(num_gates, gates_blob, const_tracker_bits, bellman_bytes):
    """mod.rs:150-157"""
    bv = bits_to_bytes(const_tracker_bits)
    return (struct.pack('<I', num_gates) + struct.pack('<I', len(gates_blob)) + bytes(gates_blob) +
            struct.pack('<I', len(const_tracker_bits)) + struct.pack('<I', len(bv)) + bv + bytes(bellman_bytes))


def read_parameters(data):
    """mod.rs:159-175.  Returns dict(num_gates, gates_blob, const_tracker (list of bool), bellman).  `gates_blob` and `bellman` are
    memoryviews INTO `data` (bytes, bytearray, memoryview or a uint8 ndarray): a benchmark-size image is 14 GB and is not copied."""
    data = memoryview(data).cast('B') if not isinstance(data, memoryview) else data.cast('B')
    pos = 0

    def u32():
        nonlocal pos
        if pos + 4 > len(data):
            raise ValueError('Parameters file truncated')
        (v,) = struct.unpack_from('<I', data, pos)
        pos += 4
        return v

    def blob():
        nonlocal pos
        n = u32()
        if pos + n > len(data):
            raise ValueError('Parameters file truncated')
        b = data[pos:pos + n]
        pos += n
        return b
    num_gates = u32()
    gates_blob = blob()
    nbits = u32()
    bv = blob()
    return dict(num_gates=num_gates, gates_blob=gates_blob, const_tracker=bytes_to_bits(bytes(bv), nbits), bellman=data[pos:])


def bellman_counts(bellman):
    """(num_input, num_aux, h, a, b) point counts of a bellman `Parameters::write` image without touching the points: the verifying key is
    576 bytes, every array a u32 BE count followed by its uncompressed points (ic, h, l, a, b_g1: 64 B; b_g2: 128 B)."""
    view = memoryview(bellman).cast('B') if not isinstance(bellman, memoryview) else bellman.cast('B')
    pos, out = 64 + 64 + 128 + 128 + 64 + 128, []
    for width in (64, 64, 64, 64, 64):
        if pos + 4 > len(view):
            raise ValueError('Parameters file truncated (bellman part)')
        (n,) = struct.unpack_from('>I', view, pos)
        out.append(n)
        pos += 4 + n * width
    ic, h, l, a, b = out
    return dict(num_input=ic, num_aux=l, h=h, a=a, b=b)


def _warm_up(ctx, key, counts):
    """One proof of the key's size from device-generated vectors (result discarded): fk_prove_dev over a, b, c, z filled by fk_gen_scalars_dev and
    all-ones density maps -- everything a first proof allocates or derives once per context and domain then exists.  Never an error: a failure
    here only means the first real proof does the work itself; returns None or what went wrong."""
    import numpy as np
    bufs = []
    try:
        m = key.counts()['m']
        n_in, n_aux = counts['num_input'], counts['num_aux']
        d = [ctx.dev_alloc(m * 32) for _ in range(3)]
        bufs += d
        d_z = ctx.dev_alloc((n_in + n_aux) * 32)
        bufs.append(d_z)
        for i, p_ in enumerate(d):
            ctx.gen_scalars_dev(p_, m, 101 + i, 0)
        ctx.gen_scalars_dev(d_z, n_in + n_aux, 104, 2)
        # density maps with as many entries set as the key's a and b queries hold points (WHICH ones does not matter here)
        kc = key.counts()
        a_ones = kc['n_a'] - n_in
        b_in_ones = min(n_in, kc['n_b'])
        b_aux_ones = kc['n_b'] - b_in_ones
        if not (0 <= a_ones <= n_aux and 0 <= b_aux_ones <= n_aux):
            return 'the key\'s query sizes do not fit density maps'
        dens = []
        for n, ones in ((n_aux, a_ones), (n_in, b_in_ones), (n_aux, b_aux_ones)):
            h = np.zeros(max(n, 1), np.uint8)
            h[:ones] = 1
            p_ = ctx.dev_alloc(h.size)
            bufs.append(p_)
            ctx.upload(p_, h)
            dens.append(p_)
        one = np.zeros((1, 4), np.uint64)
        ctx.prove_dev(key, d[0], d[1], d[2], m, d_z, dens[0], dens[1], dens[2], one, one)
        return None
    except Exception as e:       # noqa: BLE001
        return repr(e)
    finally:
        for p_ in bufs:
            try:
                ctx.dev_free(p_)
            except Exception:       # noqa: BLE001
                pass


def _maybe_warm_up(ctx, key, hdr, counts, tm):
    """the loader's throw-away proof (see load_parameters), guarded: one proof of the key's size over generated vectors while the decoder is still
    busy, so that what a context sets up on its first proof (the transform tables of the domain: 1.2 s at 2^25; the multiplications' lane scratch:
    0.3 s) is in place when the caller's first proof comes (tools/load_probe.py: 1.6 -> 0.3 s in a fresh process).  Only when the HBM left after the
    levels covers the resident system still to come (estimated from the header: 8 bytes per matrix term, ~4 terms per row and matrix at fawkes'
    densities) and the warm-up's own a, b, c, z -- on a smaller GPU its scratch must not be what pushes the system's upload into the
    drop-levels-and-retry path.  Never an error: anything that goes wrong here only means the first real proof does the work itself."""
    import time
    try:
        m = 1 << max(int(counts['h']), 1).bit_length()          # h holds m - 1 points
        est = 8 * 3 * 4 * int(hdr['num_gates']) + 32 * (3 * m + counts['num_input'] + counts['num_aux'])
        headroom = key.levels_headroom()
        if headroom < est:
            tm['warm_up_skipped'] = 'HBM headroom %.1f GiB < the %.1f GiB the system and a throw-away proof would need' % (headroom / 2**30, est / 2**30)
            return
        t1 = time.perf_counter()
        before = ctx.stats()
        tm['warm_up_error'] = _warm_up(ctx, key, counts)
        # the throw-away proof is not the caller's: the kernel statistics (fk_stats_get) do not show it when they were empty before
        if all(v.get('launches', 0) == 0 for v in before.values()):
            ctx.stats_reset()
        tm['warm_up_s'] = time.perf_counter() - t1
    except Exception as e:       # noqa: BLE001
        tm['warm_up_error'] = repr(e)


def load_parameters(ctx, data, shard_index=0, shard_count=1, z_frac=(-1.0, -1.0), checked=True, disallow_points_at_infinity=False,
                    want_host_r1cs=False, timings=None, overlap=True, early_levels=True, background_free=False, warm=True):
    """`Parameters::read(reader, disallow_points_at_infinity, checked)` (mod.rs:159-175) for the GPU prover: file bytes ->
    (DeviceKey resident in HBM, DeviceR1cs resident in HBM, header dict incl. gamma_g2 / ic / const_tracker for a verifier
    and for the witness generator).  The key part is converted and checked on the GPU (fk_key_load_bellman), the gate blob
    is decoded natively (api.Gates: one decompressing thread, the parsing on all host threads).  want_host_r1cs: also return the
    decoded system as an api.R1cs in hdr['r1cs'].  timings: a dict that receives the seconds of every stage.
    warm (default on, single-GPU loads only): while the decoder is still busy, one throw-away proof of the key's size over generated vectors
    (timings['warm_up_s']; skipped -- timings['warm_up_skipped'] -- when the HBM left beside the levels would not cover the system still to
    come).  Side effects: the context's grow-only proof scratch exists afterwards; fk_stats counters are reset if they were empty before."""
    import time
    tm = timings if timings is not None else {}
    t0 = time.perf_counter()
    hdr = read_parameters(data)
    tm['header_s'] = time.perf_counter() - t0
    flags = (api.FK_KEY_CHECKED if checked else 0) | (api.FK_KEY_NO_INFINITY if disallow_points_at_infinity else 0)
    # The circuit first, then the key: the loader sizes the key's fixed-base levels against the HBM that is free at that moment, and a
    # resident system of 1.6e9 terms is 14 GB the levels must not take.  num_input / num_aux are the lengths of ic and l.
    try:
        c = bellman_counts(hdr['bellman'])
    except ValueError:
        ctx.load_key_bellman(hdr['bellman'], shard_index, shard_count, z_frac, flags=flags)[0].free()      # raises the loader's own FK_ERR_FORMAT
        raise
    blob = hdr['gates_blob']
    raw = bytes(blob[:len(RAW_MAGIC)]) == RAW_MAGIC
    gates = dr = key = None
    import threading
    box = {}

    def decode():
        t1 = time.perf_counter()
        try:
            box['gates'] = api.Gates(blob[len(RAW_MAGIC):] if raw else blob, api.FK_GATES_RAW if raw else api.FK_GATES_BROTLI, hdr['num_gates'],
                                     c['num_input'], c['num_aux'], ctx=None)        # (context-free: the decoder is host code and runs beside the key reader)
        except BaseException as e:          # noqa: BLE001 -- re-raised on the calling thread
            box['error'] = e
        box['decode_s'] = time.perf_counter() - t1
    try:
        # The gate blob is decoded on host threads (ctypes releases the GIL) WHILE the key arrays are transferred, converted and checked on the
        # GPU, and (early_levels) the fixed-base levels are derived; early_levels=False: the levels wait until the constraint system is resident
        # (33 s to the first proof at the benchmark size instead of 28); overlap=False: one after the other (37.9 s).
        th = threading.Thread(target=decode)
        th.start()
        if not overlap:
            th.join()
        try:
            t1 = time.perf_counter()
            key, gamma_g2, ic = ctx.load_key_bellman(hdr['bellman'], shard_index, shard_count, z_frac, flags=flags | api.FK_KEY_NO_LEVELS)
            tm['key_read_s'] = time.perf_counter() - t1
            if overlap and early_levels:
                # the decoder is still busy (at the benchmark size for another 20 s): derive the levels NOW, against the HBM free at this moment
                # less the planner's allowance for the caller; whether they still leave room is checked once the system is resident
                t1 = time.perf_counter()
                key.derive_levels()
                tm['key_levels_s'] = time.perf_counter() - t1
                tm['key_levels_early'] = True
                if warm and shard_count == 1 and th.is_alive():
                    _maybe_warm_up(ctx, key, hdr, c, tm)
        finally:
            th.join()
        if 'error' in box:
            raise box['error']
        gates = box['gates']
        tm['gates_decode_s'] = box['decode_s']
        tm['gates_decode_profile'] = gates.profile()
        t1 = time.perf_counter()
        try:
            dr = gates.load(ctx)
        except api.FkError as e:
            if e.code != 5 or not tm.get('key_levels_early'):     # FK_ERR_OOM with the levels already in place: they took the system's room
                raise
            key.drop_levels()
            tm['key_levels_early'] = False
            dr = gates.load(ctx)
        tm['r1cs_load_s'] = time.perf_counter() - t1
        hdr['gates_info'] = gates.info()
        if want_host_r1cs:
            hdr['r1cs'] = gates.to_r1cs()
        # the decoder's arrays (8 bytes per term: 14 GB at the benchmark size) go back to the system: 0.7 - 0.9 s of unmapping.  background_free
        # does it on a thread of its own -- measured: the first proof then takes exactly that much longer (the unmapping holds the address
        # space's lock against the runtime's own mappings), so it is off by default
        t1 = time.perf_counter()
        if background_free:
            threading.Thread(target=gates.free, name='fk-gates-free').start()
        else:
            gates.free()
        gates = None
        tm['gates_free_s'] = time.perf_counter() - t1
        if tm.get('key_levels_early'):
            tm['key_levels_headroom_GiB'] = round(key.levels_headroom() / 2**30, 1)
            if tm['key_levels_headroom_GiB'] < 0:                  # the system turned out larger than the planner's allowance: plan again, now that it is in place
                t1 = time.perf_counter()
                key.derive_levels()
                tm['key_levels_replanned_s'] = time.perf_counter() - t1
        else:
            t1 = time.perf_counter()
            key.derive_levels()                                    # sized against the HBM that is free NOW: the system is in place
            tm['key_levels_s'] = time.perf_counter() - t1
        tm['key_read_profile'] = key.load_profile()
        hdr.update(gamma_g2=gamma_g2, ic=ic)
    except Exception:
        # nothing stays behind in HBM when either half of the file is bad (a 2^25 key is several GB)
        if dr is not None:
            dr.free()
        if key is not None:
            key.free()
        raise
    finally:
        if gates is not None:
            gates.free()
        elif box.get('gates') is not None and dr is None:
            box['gates'].free()
    tm['total_s'] = time.perf_counter() - t0
    return key, dr, hdr


def store_parameters(key_arrays, r1cs, const_tracker_bits=(), compress=None):
    """Inverse of load_parameters for small keys.  compress=brotli.compress reproduces the reference's blob format
    (quality 9, lgwin 22: setup.rs:26); without it the gate stream is stored raw behind RAW_MAGIC."""
    stream = encode_gate_stream(r1cs)
    blob = compress(stream) if compress is not None else RAW_MAGIC + stream
    return write_parameters(r1cs.num_gates, blob, list(const_tracker_bits), encode_bellman_parameters(key_arrays))


def store_parameters_dev(ctx, key, vk, r1cs, const_tracker_bits=(), compress=None, gates_blob=None, copies=None, quality=9, lgwin=22,
                         timings=None, alloc=None):
    """`Parameters::write` (mod.rs:150-157) for a key RESIDENT in HBM, at any size, as ONE uint8 array: fawkes' header from the host,
    the gate blob written natively (fk_gates_encode: Gate::serialize of every gate through libbrotlienc; setup.rs:25-32 uses quality 9 /
    lgwin 22 -- any setting decodes to the same stream; quality 2 is 16 x faster to write at 61 GB of stream and its blob decodes like a quality-9 one), the bellman part converted on the GPU
    straight into the image (fk_key_write_bellman: Montgomery limbs -> big-endian canonical points, ~seconds for a 2^25 key).
    vk: the dict fk_setup* / load_key_bellman returned (gamma_g2, ic).  The gate blob is `gates_blob` as given (e.g. the blob the key
    was loaded with), or `compress(encode_gate_stream(r1cs))` when a `compress` callable is given (the slow per-term restatement: small
    systems, tests), or the native encoding of `copies` copies of `r1cs` (fk_r1cs_load_tiled's variable order; None = the system itself).
    A blob of 4 GiB or more cannot be written: Borsh's Vec<u8> length is a u32 (the reference's writer fails the same way).
    alloc(nbytes) -> writable uint8 array of that length, e.g. a numpy.memmap over a file several processes will read (bench.py --gpus N: every
    rank sets its prover up from the same image); default: process memory."""
    import time
    tm = timings if timings is not None else {}
    t0 = time.perf_counter()
    owned = None
    if gates_blob is None:
        if compress is not None:
            gates_blob = compress(encode_gate_stream(r1cs))
        else:
            owned = api.GateBlob(r1cs, copies, fmt=api.FK_GATES_BROTLI, quality=quality, lgwin=lgwin, ctx=ctx)
            gates_blob = owned.data
            tm['gates_encode_profile'] = owned.profile()
    tm['gates_encode_s'] = time.perf_counter() - t0
    try:
        blob = api._bytes_view(gates_blob)
        if blob.size >= 1 << 32:
            raise ValueError('gate blob of %d bytes: Borsh Vec<u8> holds less than 4 GiB (mod.rs:153)' % blob.size)
        bits = list(const_tracker_bits)
        bv = bits_to_bytes(bits)
        head = struct.pack('<I', r1cs.num_gates * int(copies or 1)) + struct.pack('<I', blob.size)
        mid = struct.pack('<I', len(bits)) + struct.pack('<I', len(bv)) + bv
        t1 = time.perf_counter()
        need = ctx.write_key_bellman(key, vk, size_only=True)
        total = len(head) + blob.size + len(mid) + need
        image = np.empty(total, np.uint8) if alloc is None else alloc(total)
        if image.dtype != np.uint8 or image.ndim != 1 or image.size != total:
            raise ValueError('alloc(%d) must return a one-dimensional uint8 array of that length' % total)
        o = 0
        image[o:o + len(head)] = np.frombuffer(head, np.uint8); o += len(head)
        image[o:o + blob.size] = blob; o += blob.size
        image[o:o + len(mid)] = np.frombuffer(mid, np.uint8); o += len(mid)
        ctx.write_key_bellman(key, vk, out=image[o:])
        tm['key_write_s'] = time.perf_counter() - t1
        tm['blob_bytes'] = int(blob.size); tm['bellman_bytes'] = int(need); tm['total_s'] = time.perf_counter() - t0
        return image
    finally:
        if owned is not None:
            owned.free()
