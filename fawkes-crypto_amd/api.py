"""
ctypes binding of libfawkes_hip.so + the host-side mirror of the reference's prover interface.

Reference interface mirrored here (file:line under /root/reference/fawkes-crypto/src/backend/bellman_groth16/):
  Parameters            mod.rs:139-175     (bellman Parameters + circuit replay data)
  Proof / Borsh         prover.rs:13-60
  prove                 prover.rs:63-90    (returns (public inputs without ONE, proof))
  G1Point / G2Point     group.rs:13-122    ((0,0) <=> infinity)
  OsRng                 osrng.rs:12-18     (r, s source; `prove_with_rs` bypasses it)
No oracle code is imported here, and nothing falls back to the CPU.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
FK_PROOF_BYTES = 256
Z_EQUAL_SPLIT = (-1.0, -1.0)     # FK_Z_EQUAL_SPLIT: l / a / b sliced like h; any (lo, hi) with lo >= 0 is a fraction range, (0, 0) = empty
Z_WORK_SPLIT_Q0 = (-3.0, -3.0)   # FK_Z_WORK_SPLIT_Q0: all of h on shard 0 (which computes the whole quotient), its fixed work counted into the work line
Z_WORK_SPLIT = (-2.0, -2.0)      # FK_Z_WORK_SPLIT: l | a | b_g1 | b_g2 cut by work (G2 = 2.8 G1): one or two large pieces per rank
FK_MSM_RESULT_BYTES = 4 * 64 + 128
FR_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617
FQ_MODULUS = 21888242871839275222246405745257275088696311157297823662689037894645226208583

# every symbol include/fawkes_hip.h declares (tests check the .so exports all of them)
EXPORTED_SYMBOLS = [
    'fk_init', 'fk_free', 'fk_trim', 'fk_last_error', 'fk_set_window_bits',
    'fk_dev_alloc', 'fk_dev_free', 'fk_upload', 'fk_download', 'fk_dev_copy', 'fk_sync', 'fk_stream',
    'fk_host_alloc', 'fk_host_free', 'fk_witness_upload_async', 'fk_witness_ptr', 'fk_witness_slot', 'fk_witness_upload_part_async', 'fk_witness_mark_ready', 'fk_prove_r1cs_submit', 'fk_prove_r1cs_wait',
    'fk_key_load', 'fk_key_synthetic', 'fk_key_shard_info', 'fk_key_shard_info2', 'fk_key_host_vk', 'fk_key_free',
    'fk_prove', 'fk_prove_dev', 'fk_prove_msms', 'fk_prove_msms_dev', 'fk_prove_msms_z_dev', 'fk_prove_msm_h_dev', 'fk_prove_msm_array_dev', 'fk_prove_msms_hz_dev',
    'fk_prove_msms_z_begin_dev', 'fk_prove_msms_finish_dev', 'fk_prove_msms_hz_r1cs_dev', 'fk_prove_msms_z_begin_r1cs_dev',
    'fk_prove_assemble',
    'fk_fr_mul_batch', 'fk_ntt', 'fk_ntt_dev', 'fk_quotient_h', 'fk_quotient_h_dev',
    'fk_msm_g1', 'fk_msm_g2', 'fk_msm_g1_dev', 'fk_msm_g2_dev',
    'fk_gen_points_g1_dev', 'fk_gen_points_g2_dev', 'fk_gen_scalars_dev',
    'fk_synthesize', 'fk_roctx_active', 'fk_stats_reset', 'fk_stats_get', 'fk_calibrate', 'fk_verify', 'fk_verify_batch_dev', 'fk_shard_range', 'fk_h_shard_range', 'fk_work_shard_ranges', 'fk_work_shard_ranges_q0',
    'fk_dq_gather_dev', 'fk_dq_local_dev', 'fk_dq_cross_dev', 'fk_dq_cross_sub_dev',
    'fk_setup', 'fk_setup_tiled', 'fk_r1cs_load_tiled', 'fk_key_download', 'fk_key_load_bellman', 'fk_key_write_bellman', 'fk_key_vk', 'fk_key_counts', 'fk_key_precomputed', 'fk_key_load_profile', 'fk_key_levels_plan', 'fk_key_derive_levels', 'fk_key_levels_headroom', 'fk_key_drop_levels',
    'fk_gates_decode', 'fk_gates_free', 'fk_gates_info', 'fk_gates_export', 'fk_r1cs_load_gates', 'fk_gates_profile',
    'fk_gates_encode', 'fk_blob_data', 'fk_blob_profile', 'fk_blob_free',
    'fk_r1cs_load', 'fk_r1cs_load_coded', 'fk_r1cs_free', 'fk_r1cs_info', 'fk_r1cs_windows', 'fk_r1cs_density_ptrs', 'fk_r1cs_eval_dev', 'fk_r1cs_eval_slice_dev', 'fk_prove_r1cs', 'fk_prove_r1cs_dev',
    'fk_init_devices', 'fk_multi_free', 'fk_multi_last_error', 'fk_multi_size', 'fk_multi_transport', 'fk_multi_topology', 'fk_multi_preflight', 'fk_multi_ctx', 'fk_multi_sync', 'fk_multi_witness_traffic',
    'fk_multi_key_load', 'fk_multi_key_load_bellman', 'fk_multi_setup', 'fk_multi_setup_tiled', 'fk_multi_key_free', 'fk_multi_key_shard',
    'fk_multi_r1cs_load', 'fk_multi_r1cs_load_tiled', 'fk_multi_r1cs_load_gates', 'fk_multi_r1cs_free', 'fk_multi_r1cs_replica',
    'fk_multi_prove_r1cs', 'fk_multi_prove_r1cs_submit', 'fk_multi_prove_r1cs_wait',
]

_ERR = {1: 'FK_ERR_BAD_ARG', 2: 'FK_ERR_DOMAIN_TOO_LARGE (bellman: PolynomialDegreeTooLarge)',
        3: 'FK_ERR_UNEXPECTED_IDENTITY (bellman: UnexpectedIdentity)', 4: 'FK_ERR_HIP', 5: 'FK_ERR_OOM',
        6: 'FK_ERR_KEY_MISMATCH', 7: 'FK_ERR_FORMAT (InvalidData / GroupDecodingError)', 8: 'FK_ERR_UNSUPPORTED'}


class FkError(RuntimeError):
    def __init__(self, code, msg=''):
        self.code = code
        super().__init__('%s: %s' % (_ERR.get(code, 'error %d' % code), msg))


def lib_path():
    # FK_LIB_VARIANT=<suffix>: an experimental build of the same sources (tools/ab_probe.sh compares builds on one GPU box)
    v = os.environ.get('FK_LIB_VARIANT', '')
    return os.path.join(_HERE, 'libfawkes_hip%s.so' % (('_' + v) if v else ''))


def build_library(jobs=3):
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    subprocess.check_call(['make', '-C', os.path.join(_HERE, 'csrc'), '-j%d' % jobs, '-s'])


_LIB = None


class KeyDesc(C.Structure):
    _fields_ = [('m', C.c_uint64), ('num_input', C.c_uint32), ('num_aux', C.c_uint32),
                ('alpha_g1', C.c_void_p), ('beta_g1', C.c_void_p), ('delta_g1', C.c_void_p),
                ('beta_g2', C.c_void_p), ('delta_g2', C.c_void_p),
                ('h', C.c_void_p), ('n_h', C.c_uint64), ('l', C.c_void_p), ('n_l', C.c_uint64),
                ('a', C.c_void_p), ('n_a', C.c_uint64),
                ('b_g1', C.c_void_p), ('b_g2', C.c_void_p), ('n_b', C.c_uint64),
                ('shard_index', C.c_uint32), ('shard_count', C.c_uint32),
                ('z_frac_lo', C.c_double), ('z_frac_hi', C.c_double)]


class Timings(C.Structure):
    _fields_ = [(n, C.c_double) for n in ('upload_ms', 'ntt_ms', 'msm_h_ms', 'msm_l_ms', 'msm_a_ms', 'msm_b1_ms',
                                          'msm_b2_ms', 'assemble_ms', 'total_ms')]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class R1csStruct(C.Structure):
    _fields_ = [('num_input', C.c_uint32), ('num_aux', C.c_uint32), ('num_gates', C.c_uint64),
                ('a_ptr', C.c_void_p), ('a_col', C.c_void_p), ('a_val', C.c_void_p),
                ('b_ptr', C.c_void_p), ('b_col', C.c_void_p), ('b_val', C.c_void_p),
                ('c_ptr', C.c_void_p), ('c_col', C.c_void_p), ('c_val', C.c_void_p)]


def load_library():
    """Loads libfawkes_hip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise FileNotFoundError('%s missing: run __graft_entry__.build() (hipcc, gfx950)' % p)
        lib = C.CDLL(p)
        lib.fk_last_error.restype = C.c_char_p
        lib.fk_last_error.argtypes = [C.c_void_p]
        lib.fk_free.argtypes = [C.c_void_p]
        lib.fk_free.restype = None
        lib.fk_key_free.argtypes = [C.c_void_p, C.c_void_p]
        lib.fk_key_free.restype = None
        lib.fk_gates_free.argtypes = [C.c_void_p]
        lib.fk_gates_free.restype = None
        lib.fk_blob_free.argtypes = [C.c_void_p]
        lib.fk_blob_free.restype = None
        lib.fk_multi_last_error.restype = C.c_char_p
        lib.fk_multi_last_error.argtypes = [C.c_void_p]
        lib.fk_multi_free.argtypes = [C.c_void_p]
        lib.fk_multi_free.restype = None
        lib.fk_multi_key_free.argtypes = [C.c_void_p, C.c_void_p]
        lib.fk_multi_key_free.restype = None
        lib.fk_multi_r1cs_free.argtypes = [C.c_void_p, C.c_void_p]
        lib.fk_multi_r1cs_free.restype = None
        lib.fk_multi_transport.restype = C.c_char_p
        lib.fk_multi_transport.argtypes = [C.c_void_p]
        lib.fk_multi_ctx.restype = C.c_void_p
        lib.fk_multi_ctx.argtypes = [C.c_void_p, C.c_int]
        lib.fk_multi_key_shard.restype = C.c_void_p
        lib.fk_multi_key_shard.argtypes = [C.c_void_p, C.c_int]
        lib.fk_multi_r1cs_replica.restype = C.c_void_p
        lib.fk_multi_r1cs_replica.argtypes = [C.c_void_p, C.c_int]
        _LIB = lib
    return _LIB


def _vp(arr):
    if arr is None:
        return None
    return C.c_void_p(arr.ctypes.data)


def _fr(arr, rows=None):
    a = np.ascontiguousarray(arr, dtype=np.uint64)
    if a.ndim == 1:
        a = a.reshape(-1, 4)
    assert a.ndim == 2 and a.shape[1] == 4
    if rows is not None:
        assert a.shape[0] == rows, (a.shape, rows)
    return a


def _u8(arr, n=None):
    a = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1)
    if n is not None:
        assert a.shape[0] == n, (a.shape, n)
    return a


def int_to_limbs(x):
    return np.frombuffer(int(x).to_bytes(32, 'little'), dtype=np.uint64).copy()


def limbs_to_int(l):
    return int.from_bytes(np.ascontiguousarray(l, dtype=np.uint64).tobytes(), 'little')


class R1cs:
    """CSR constraint matrices (what `WitnessCS` streams out of the brotli gate blob, cs.rs:184-223):
    variable index = i for Input(i), num_input + j for Aux(j); coefficients Montgomery Fr."""

    def __init__(self, num_input, num_aux, a, b, c):
        self.num_input, self.num_aux = int(num_input), int(num_aux)
        self.mats = []
        for (ptr, col, val) in (a, b, c):       # val None: every coefficient of that matrix is ONE
            self.mats.append((np.ascontiguousarray(ptr, np.uint64), np.ascontiguousarray(col, np.uint32),
                              None if val is None else np.ascontiguousarray(val, np.uint64).reshape(-1, 4)))
        self.num_gates = len(self.mats[0][0]) - 1
        s = R1csStruct()
        s.num_input, s.num_aux, s.num_gates = self.num_input, self.num_aux, self.num_gates
        for nm, (ptr, col, val) in zip('abc', self.mats):
            setattr(s, nm + '_ptr', ptr.ctypes.data)
            setattr(s, nm + '_col', col.ctypes.data)
            setattr(s, nm + '_val', val.ctypes.data if val is not None else None)
        self.struct = s

    @property
    def n_rows(self):
        return self.num_gates + self.num_input


def num_to_json(v):
    """serde form of `Num<Fp>` (ff-uint/src/num/mod.rs:445-451): the canonical value as a DECIMAL string"""
    return str(int(v))


def num_from_json(sv, modulus=None):
    """`Num<Fp>`'s Deserialize (ff-uint/src/num/mod.rs:454-459 -> NumRepr::from_str, ff-uint/src/uint/mod.rs:367-388): a string of ASCII digits
    only ("Invalid character" otherwise; the empty string is 0, as upstream's loop gives), at most 256 bits ("Invalid length"), below the
    modulus ("Field overflow").  The messages are upstream's."""
    if not isinstance(sv, str):
        raise ValueError('Wrong number format')          # NumRepr's Deserialize maps every parse error to this (mod.rs:95-97)
    if not all(48 <= ord(ch) <= 57 for ch in sv):
        raise ValueError('Wrong number format: Invalid character')
    v = 0
    for ch in sv:
        v = v * 10 + (ord(ch) - 48)
        if v >> 256:
            raise ValueError('Wrong number format: Invalid length')
    if modulus is not None and v >= modulus:
        raise ValueError('Field overflow')
    return v


class G1Point:
    """group.rs:13 -- affine (x, y) as canonical ints; (0, 0) is the point at infinity (group.rs:55)."""

    def __init__(self, x, y):
        self.x, self.y = int(x), int(y)

    def to_json(self):
        """serde form (group.rs:12-13: a derived tuple struct of two `Num<Fq>`): ["x", "y"], decimal strings"""
        return [num_to_json(self.x), num_to_json(self.y)]

    @classmethod
    def from_json(cls, j):
        if not isinstance(j, (list, tuple)) or len(j) != 2:
            raise ValueError('G1Point: expected a sequence of 2 elements')
        return cls(num_from_json(j[0], FQ_MODULUS), num_from_json(j[1], FQ_MODULUS))

    def is_zero(self):
        return self.x == 0 and self.y == 0

    def to_bytes(self):  # Borsh (group.rs:16-21)
        return self.x.to_bytes(32, 'little') + self.y.to_bytes(32, 'little')

    @classmethod
    def from_bytes(cls, b):
        return cls(int.from_bytes(b[:32], 'little'), int.from_bytes(b[32:64], 'little'))

    def __eq__(self, o):
        return (self.x, self.y) == (o.x, o.y)


class G2Point:
    """group.rs:85 -- ((x_re, x_im), (y_re, y_im))."""

    def __init__(self, x, y):
        self.x, self.y = (int(x[0]), int(x[1])), (int(y[0]), int(y[1]))

    def to_json(self):
        """serde form (group.rs:83-85: a tuple struct of two pairs, "X+IY" little-endian): [["x_re", "x_im"], ["y_re", "y_im"]]"""
        return [[num_to_json(self.x[0]), num_to_json(self.x[1])], [num_to_json(self.y[0]), num_to_json(self.y[1])]]

    @classmethod
    def from_json(cls, j):
        if not isinstance(j, (list, tuple)) or len(j) != 2 or any(not isinstance(c, (list, tuple)) or len(c) != 2 for c in j):
            raise ValueError('G2Point: expected a sequence of 2 pairs')
        return cls(tuple(num_from_json(v, FQ_MODULUS) for v in j[0]), tuple(num_from_json(v, FQ_MODULUS) for v in j[1]))

    def is_zero(self):
        return self.x == (0, 0) and self.y == (0, 0)

    def to_bytes(self):  # Borsh (group.rs:33-39)
        return b''.join(v.to_bytes(32, 'little') for v in (self.x[0], self.x[1], self.y[0], self.y[1]))

    @classmethod
    def from_bytes(cls, b):
        v = [int.from_bytes(b[i * 32:(i + 1) * 32], 'little') for i in range(4)]
        return cls((v[0], v[1]), (v[2], v[3]))

    def __eq__(self, o):
        return (self.x, self.y) == (o.x, o.y)


class Proof:
    """prover.rs:13-17; Borsh = a || b || c = 256 bytes (prover.rs:39-45)."""

    def __init__(self, a, b, c):
        self.a, self.b, self.c = a, b, c

    def to_bytes(self):
        return self.a.to_bytes() + self.b.to_bytes() + self.c.to_bytes()

    @classmethod
    def from_bytes(cls, b):
        b = bytes(b)
        assert len(b) == FK_PROOF_BYTES
        return cls(G1Point.from_bytes(b[:64]), G2Point.from_bytes(b[64:192]), G1Point.from_bytes(b[192:]))

    def __eq__(self, o):
        return self.to_bytes() == o.to_bytes()

    def to_json(self):
        """serde form (prover.rs:11-17: derived, named fields): {"a": G1, "b": G2, "c": G1} -- what serde_json::to_value gives"""
        return {'a': self.a.to_json(), 'b': self.b.to_json(), 'c': self.c.to_json()}

    @classmethod
    def from_json(cls, j):
        if isinstance(j, (str, bytes)):
            import json as _json
            j = _json.loads(j)
        missing = [k for k in 'abc' if k not in j]
        if missing:
            raise ValueError('missing field `%s`' % missing[0])          # serde's message
        return cls(G1Point.from_json(j['a']), G2Point.from_json(j['b']), G1Point.from_json(j['c']))

    def to_json_str(self):
        import json as _json
        return _json.dumps(self.to_json(), separators=(',', ':'))           # serde_json::to_string's compact form, field order a, b, c


class VK:
    """verifier.rs:10-18 -- alpha (G1), beta, gamma, delta (G2), ic (Vec<G1>) as canonical coordinates.  Borsh = the five fields in order,
    `ic` as a u32 LE count + points (verifier.rs:46-54); serde = {"alpha", "beta", "gamma", "delta", "ic"} (derived, verifier.rs:10-11)."""

    def __init__(self, alpha, beta, gamma, delta, ic):
        self.alpha, self.beta, self.gamma, self.delta, self.ic = alpha, beta, gamma, delta, list(ic)

    def to_bytes(self):
        return (self.alpha.to_bytes() + self.beta.to_bytes() + self.gamma.to_bytes() + self.delta.to_bytes() +
                len(self.ic).to_bytes(4, 'little') + b''.join(p.to_bytes() for p in self.ic))

    @classmethod
    def from_bytes(cls, b):
        b = bytes(b)
        if len(b) < 64 + 3 * 128 + 4:
            raise ValueError('Unexpected length of input')            # Borsh's message
        n = int.from_bytes(b[448:452], 'little')
        if len(b) != 452 + 64 * n:
            raise ValueError('Unexpected length of input')
        return cls(G1Point.from_bytes(b[:64]), G2Point.from_bytes(b[64:192]), G2Point.from_bytes(b[192:320]), G2Point.from_bytes(b[320:448]),
                   [G1Point.from_bytes(b[452 + 64 * i:516 + 64 * i]) for i in range(n)])

    @classmethod
    def from_raw(cls, vk):
        """from the dict fk_setup* / load_key_bellman return (raw Montgomery LE arrays: alpha_g1, beta_g2, gamma_g2, delta_g2, ic)"""
        return cls.from_bytes(vk_to_borsh(vk))

    def to_json(self):
        return {'alpha': self.alpha.to_json(), 'beta': self.beta.to_json(), 'gamma': self.gamma.to_json(), 'delta': self.delta.to_json(),
                'ic': [p.to_json() for p in self.ic]}

    @classmethod
    def from_json(cls, j):
        if isinstance(j, (str, bytes)):
            import json as _json
            j = _json.loads(j)
        missing = [k for k in ('alpha', 'beta', 'gamma', 'delta', 'ic') if k not in j]
        if missing:
            raise ValueError('missing field `%s`' % missing[0])
        return cls(G1Point.from_json(j['alpha']), G2Point.from_json(j['beta']), G2Point.from_json(j['gamma']), G2Point.from_json(j['delta']),
                   [G1Point.from_json(p) for p in j['ic']])

    def to_json_str(self):
        import json as _json
        return _json.dumps(self.to_json(), separators=(',', ':'))

    def __eq__(self, o):
        return self.to_bytes() == o.to_bytes()


class Parameters:
    """Mirror of `Parameters<E>` (mod.rs:139): the bellman proving key arrays (raw Montgomery-LE points,
    the layout group.rs:57-66 exchanges) plus the constraint system that `WitnessCS` replays.
    `key_arrays`: dict with m, num_input, num_aux, alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2 (uint8
    arrays) and h, l, a, b_g1 (n x 64 uint8), b_g2 (n x 128)."""

    def __init__(self, key_arrays, r1cs=None):
        k = key_arrays
        self.m, self.num_input, self.num_aux = int(k['m']), int(k['num_input']), int(k['num_aux'])
        self.vk = {n: _u8(k[n], w) for n, w in (('alpha_g1', 64), ('beta_g1', 64), ('delta_g1', 64),
                                                 ('beta_g2', 128), ('delta_g2', 128))}
        self.h = np.ascontiguousarray(k['h'], np.uint8).reshape(-1, 64)
        self.l = np.ascontiguousarray(k['l'], np.uint8).reshape(-1, 64)
        self.a = np.ascontiguousarray(k['a'], np.uint8).reshape(-1, 64)
        self.b_g1 = np.ascontiguousarray(k['b_g1'], np.uint8).reshape(-1, 64)
        self.b_g2 = np.ascontiguousarray(k['b_g2'], np.uint8).reshape(-1, 128)
        self.r1cs = r1cs
        self._handles = {}

    def desc(self, shard_index=0, shard_count=1, z_frac=Z_EQUAL_SPLIT):
        d = KeyDesc()
        d.z_frac_lo, d.z_frac_hi = float(z_frac[0]), float(z_frac[1])
        d.m, d.num_input, d.num_aux = self.m, self.num_input, self.num_aux
        for n in self.vk:
            setattr(d, n, self.vk[n].ctypes.data)
        d.h, d.n_h = self.h.ctypes.data, self.h.shape[0]
        d.l, d.n_l = self.l.ctypes.data, self.l.shape[0]
        d.a, d.n_a = self.a.ctypes.data, self.a.shape[0]
        d.b_g1, d.b_g2, d.n_b = self.b_g1.ctypes.data, self.b_g2.ctypes.data, self.b_g1.shape[0]
        d.shard_index, d.shard_count = shard_index, shard_count
        return d


class DeviceKey:
    def __init__(self, ctx, handle, shard_index=0, shard_count=1):
        self.ctx, self.handle = ctx, handle
        self.shard_index, self.shard_count = shard_index, shard_count

    def free(self):
        if self.handle:
            self.ctx.lib.fk_key_free(self.ctx.handle, self.handle)
            self.handle = None

    def download(self, name):
        """this key's slice of one array as numpy: 'h' | 'l' | 'a' | 'b_g1' (n,64) or 'b_g2' (n,128)"""
        which = ['h', 'l', 'a', 'b_g1', 'b_g2'].index(name)
        info = self.shard_info()
        lo, hi = info[{'h': 'h', 'l': 'l', 'a': 'a', 'b_g1': 'b', 'b_g2': 'b_g2'}[name]]
        w = 128 if name == 'b_g2' else 64
        out = np.zeros((max(hi - lo, 0), w), np.uint8)
        buf = out if out.size else np.zeros((1, w), np.uint8)
        self.ctx._ck(self.ctx.lib.fk_key_download(self.ctx.handle, self.handle, C.c_int(which), C.c_void_p(buf.ctypes.data), C.c_size_t(buf.nbytes)))
        return out

    def counts(self):
        out = (C.c_uint64 * 8)()
        rc = self.ctx.lib.fk_key_counts(self.handle, out)
        if rc != 0:
            raise FkError(rc, 'fk_key_counts')
        v = list(out)
        return dict(m=v[0], num_input=v[1], num_aux=v[2], n_h=v[3], n_l=v[4], n_a=v[5], n_b=v[6], shard_count=v[7])

    def precomputed(self):
        """fk_key_precomputed: window levels held per array (0 = the ordinary W-bucket-set path)"""
        out = (C.c_uint32 * 5)()
        rc = self.ctx.lib.fk_key_precomputed(self.handle, out)
        if rc != 0:
            raise FkError(rc, 'fk_key_precomputed')
        return dict(zip(('h', 'l', 'a', 'b_g1', 'b_g2'), list(out)))

    def levels_headroom(self):
        """fk_key_levels_headroom: bytes of HBM free beyond what proofs with this key still allocate; negative -> derive_levels() again"""
        out = C.c_int64(0)
        self.ctx._ck(self.ctx.lib.fk_key_levels_headroom(self.ctx.handle, self.handle, C.byref(out)))
        return out.value

    def drop_levels(self):
        """fk_key_drop_levels: release the fixed-base levels (derive_levels() brings them back)"""
        self.ctx._ck(self.ctx.lib.fk_key_drop_levels(self.ctx.handle, self.handle))

    def derive_levels(self):
        """fk_key_derive_levels: the fixed-base levels against the HBM that is free now (a key loaded with FK_KEY_NO_LEVELS)"""
        self.ctx._ck(self.ctx.lib.fk_key_derive_levels(self.ctx.handle, self.handle))

    def levels_plan(self):
        """fk_key_levels_plan: {array: {levels, GiB (negative: what the array left out would have needed), est_ms_saved}}"""
        out = (C.c_double * 15)()
        self.ctx._ck(self.ctx.lib.fk_key_levels_plan(self.handle, out))
        return {nm: dict(levels=int(out[3 * i]), GiB=round(out[3 * i + 1], 2), est_ms_saved=round(out[3 * i + 2], 2)) for i, nm in enumerate(('h', 'l', 'a', 'b_g1', 'b_g2'))}

    def load_profile(self):
        """fk_key_load_profile: seconds the loader spent on the arrays (transfer, conversion, checks) and on the fixed-base levels"""
        out = (C.c_double * 2)()
        self.ctx._ck(self.ctx.lib.fk_key_load_profile(self.handle, out))
        return dict(arrays_s=out[0], levels_s=out[1])

    def vk(self):
        """prover-side vk points as raw Montgomery LE uint8 arrays"""
        buf = np.zeros(3 * 64 + 2 * 128, np.uint8)
        rc = self.ctx.lib.fk_key_vk(self.handle, C.c_void_p(buf.ctypes.data))
        if rc != 0:
            raise FkError(rc, 'fk_key_vk')
        return dict(alpha_g1=buf[0:64].copy(), beta_g1=buf[64:128].copy(), delta_g1=buf[128:192].copy(),
                    beta_g2=buf[192:320].copy(), delta_g2=buf[320:448].copy())

    def shard_info(self):
        """dict of the [lo, hi) slices this key holds: h, l, a, b"""
        out = (C.c_uint64 * 8)()
        rc = self.ctx.lib.fk_key_shard_info(self.handle, out)
        if rc != 0:
            raise FkError(rc, 'fk_key_shard_info')
        v = list(out)
        out2 = (C.c_uint64 * 10)()
        rc = self.ctx.lib.fk_key_shard_info2(self.handle, out2)
        if rc != 0:
            raise FkError(rc, 'fk_key_shard_info2')
        w = list(out2)
        return dict(h=(v[0], v[1]), l=(v[2], v[3]), a=(v[4], v[5]), b=(v[6], v[7]), b_g2=(w[8], w[9]))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceR1cs:
    """Constraint system resident in HBM (fk_r1cs_load): CSR + coefficient dictionary + density maps."""

    def __init__(self, ctx, handle):
        self.ctx, self.handle = ctx, handle

    def info(self):
        out = (C.c_uint64 * 8)()
        rc = self.ctx.lib.fk_r1cs_info(self.handle, out)
        if rc != 0:
            raise FkError(rc, 'fk_r1cs_info')
        v = list(out)
        return dict(rows=v[0], nnz=(v[1], v[2], v[3]), distinct_coefficients=v[4], n_a=v[5], n_b=v[6], num_vars=v[7])

    def windows(self):
        """row windows of the chunked witness hand-over (fk_r1cs_windows): dict(rows=[K + 1 gate bounds], need=[K witness prefixes]) or None"""
        k = C.c_uint32(0)
        rows, need = (C.c_uint64 * 17)(), (C.c_uint64 * 16)()
        rc = self.ctx.lib.fk_r1cs_windows(self.handle, C.byref(k), rows, need)
        if rc != 0:
            raise FkError(rc, 'fk_r1cs_windows')
        if not k.value:
            return None
        return dict(rows=list(rows)[:k.value + 1], need=list(need)[:k.value])

    def check_witness(self, z):
        """the prove calls read (num_input + num_aux) * 32 bytes from the caller's witness: refuse a vector of another length HERE
        (the C ABI cannot know the size of the buffer behind a pointer)"""
        nv = self.info()['num_vars']
        if z.size != nv * 4:
            raise FkError(6, 'witness holds %d field elements, the constraint system has %d variables' % (z.size // 4, nv))

    def density_ptrs(self):
        """device pointers (a_aux, b_input, b_aux) of the structural density maps"""
        out = (C.c_void_p * 3)()
        rc = self.ctx.lib.fk_r1cs_density_ptrs(self.handle, out)
        if rc != 0:
            raise FkError(rc, 'fk_r1cs_density_ptrs')
        return tuple(int(x or 0) for x in out)

    def free(self):
        if self.handle:
            self.ctx.lib.fk_r1cs_free(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


FK_GATES_RAW, FK_GATES_BROTLI = 0, 1
FK_KEY_CHECKED, FK_KEY_NO_INFINITY, FK_KEY_NO_LEVELS = 1, 2, 4


def _bytes_view(data):
    """uint8 numpy view of bytes / bytearray / memoryview / ndarray without copying"""
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    return np.frombuffer(data, np.uint8)


class GateBlob:
    """fk_blob: the gate blob of a `Parameters` object as fk_gates_encode wrote it (host memory owned by the library).
    `.data` is a uint8 view valid until free()."""

    def __init__(self, r1cs, copies=None, fmt=1, quality=9, lgwin=22, ctx=None):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.fk_gates_encode(ctx.handle if ctx else None, C.byref(r1cs.struct), C.c_uint32(int(copies or 1)), C.c_int(fmt), C.c_int(quality),
                                      C.c_int(lgwin), C.byref(h))
        if rc != 0:
            msg = self.lib.fk_last_error(ctx.handle if ctx else None)
            raise FkError(rc, msg.decode() if msg else 'fk_gates_encode')
        self.handle = h
        p, n = C.c_void_p(), C.c_size_t()
        self.lib.fk_blob_data(h, C.byref(p), C.byref(n))
        self.data = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)) if n.value else np.zeros(0, np.uint8)
        self.num_gates = r1cs.num_gates * int(copies or 1)

    def profile(self):
        out = (C.c_double * 4)()
        self.lib.fk_blob_profile(self.handle, out)
        v = list(out)
        return dict(wall_s=v[0], compressor_s=v[1], stream_bytes=int(v[2]), blob_bytes=int(v[3]))

    def free(self):
        if getattr(self, 'handle', None):
            self.data = None
            self.lib.fk_blob_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Gates:
    """fk_gates: the constraint system decoded from the gate blob of a `Parameters` file (host memory, native decoder).
    fmt: FK_GATES_BROTLI (what the reference writes, setup.rs:25-32) or FK_GATES_RAW (the bare Borsh gate stream)."""

    def __init__(self, blob, fmt, num_gates, num_input, num_aux, ctx=None):
        self.lib = load_library()
        buf = _bytes_view(blob)              # no copy: a benchmark-size blob is GBs
        h = C.c_void_p()
        rc = self.lib.fk_gates_decode(ctx.handle if ctx else None, _vp(buf), C.c_size_t(buf.size), C.c_int(fmt), C.c_uint32(num_gates),
                                      C.c_uint32(num_input), C.c_uint32(num_aux), C.byref(h))
        if rc != 0:
            msg = self.lib.fk_last_error(ctx.handle if ctx else None)
            raise FkError(rc, msg.decode() if msg else 'fk_gates_decode')
        self.handle = h

    def profile(self):
        """fk_gates_profile: how the decoding went (seconds; the decompressor is the serial floor, the parsing runs beside it)"""
        out = (C.c_double * 8)()
        rc = self.lib.fk_gates_profile(self.handle, out)
        if rc != 0:
            raise FkError(rc, 'fk_gates_profile')
        v = list(out)
        return dict(wall_s=v[0], decompressor_s=v[1], waited_for_parsers_s=v[2], parse_cpu_s=v[3], renumber_s=v[4], parse_threads=int(v[5]),
                    blocks=int(v[6]), blob_bytes=int(v[7]))

    def info(self):
        out = (C.c_uint64 * 8)()
        rc = self.lib.fk_gates_info(self.handle, out)
        if rc != 0:
            raise FkError(rc, 'fk_gates_info')
        v = list(out)
        return dict(num_gates=v[0], nnz=(v[1], v[2], v[3]), distinct_coefficients=v[4], decoded_bytes=v[5], num_input=v[6], num_aux=v[7])

    def to_r1cs(self):
        """the decoded system as an api.R1cs (explicit 32-byte coefficients: for tests and for fk_setup)"""
        i = self.info()
        mats = []
        for k in range(3):
            ptr = np.zeros(i['num_gates'] + 1, np.uint64)
            col = np.zeros(max(i['nnz'][k], 1), np.uint32)[:i['nnz'][k]]
            val = np.zeros((max(i['nnz'][k], 1), 4), np.uint64)[:i['nnz'][k]]
            rc = self.lib.fk_gates_export(self.handle, C.c_int(k), _vp(ptr), C.c_void_p(col.ctypes.data), C.c_void_p(val.ctypes.data))
            if rc != 0:
                raise FkError(rc, 'fk_gates_export')
            mats.append((ptr, col, val))
        return R1cs(i['num_input'], i['num_aux'], *mats)

    def load(self, ctx):
        """fk_r1cs_load_gates: resident constraint system straight from the decoded stream"""
        h = C.c_void_p()
        ctx._ck(ctx.lib.fk_r1cs_load_gates(ctx.handle, self.handle, C.byref(h)))
        return DeviceR1cs(ctx, h)

    def free(self):
        h, self.handle = getattr(self, 'handle', None), None          # (taken first: load_parameters frees on a background thread)
        if h:
            self.lib.fk_gates_free(h)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class HostVk:
    """Host-only key (vk points only) for fk_prove_assemble -- needs no GPU."""

    def __init__(self, params):
        self.lib = load_library()
        h = C.c_void_p()
        v = params.vk
        rc = self.lib.fk_key_host_vk(_vp(v['alpha_g1']), _vp(v['beta_g1']), _vp(v['delta_g1']), _vp(v['beta_g2']),
                                     _vp(v['delta_g2']), C.byref(h))
        if rc != 0:
            raise FkError(rc, 'fk_key_host_vk')
        self.handle = h

    def __del__(self):
        try:
            if self.handle:
                self.lib.fk_key_free(None, self.handle)
                self.handle = None
        except Exception:
            pass


class Context:
    """One GPU (one process per GPU for multi-GPU runs).  Raises FkError if no GPU is usable."""

    def __init__(self, device_id=0, _borrowed=None):
        self.lib = load_library()
        self.device_id = device_id
        self._owned = _borrowed is None
        if _borrowed is not None:           # a rank's context of a MultiContext: owned by the fk_multi
            self.handle = C.c_void_p(_borrowed)
            return
        h = C.c_void_p()
        rc = self.lib.fk_init(C.c_int(device_id), C.byref(h))
        if rc != 0:
            raise FkError(rc, 'fk_init(device %d) failed -- no usable MI355X/HIP device; there is no CPU fallback' % device_id)
        self.handle = h

    def close(self):
        if getattr(self, 'handle', None):
            if self._owned:
                self.lib.fk_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            msg = self.lib.fk_last_error(self.handle)
            raise FkError(rc, msg.decode() if msg else '')

    # ---- device memory
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._ck(self.lib.fk_dev_alloc(self.handle, C.c_size_t(nbytes), C.byref(p)))
        return p.value

    def dev_free(self, dptr):
        self._ck(self.lib.fk_dev_free(self.handle, C.c_void_p(dptr)))

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._ck(self.lib.fk_upload(self.handle, C.c_void_p(dptr), _vp(arr), C.c_size_t(arr.nbytes)))

    def download(self, dptr, nbytes, dtype=np.uint8):
        out = np.zeros(nbytes // np.dtype(dtype).itemsize, dtype)
        self._ck(self.lib.fk_download(self.handle, _vp(out), C.c_void_p(dptr), C.c_size_t(nbytes)))
        return out

    def sync(self):
        self._ck(self.lib.fk_sync(self.handle))

    def trim(self):
        """fk_trim: release the scratch grown for earlier proofs (keys and resident constraint systems stay)"""
        self._ck(self.lib.fk_trim(self.handle))

    def stream_handle(self):
        """fk_stream: the library's main HIP stream as an integer (for torch.cuda.ExternalStream)"""
        p = C.c_void_p()
        self._ck(self.lib.fk_stream(self.handle, C.byref(p)))
        return p.value or 0

    # ---- witness hand-over from host memory (pinned buffers, two device slots filled on a copy stream)
    def host_alloc(self, shape, dtype=np.uint64):
        """fk_host_alloc: a numpy array over pinned host memory (free with host_free(arr))."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._ck(self.lib.fk_host_alloc(self.handle, C.c_size_t(n), C.byref(p)))
        arr = np.frombuffer((C.c_uint8 * max(n, 1)).from_address(p.value), dtype=np.uint8, count=n).view(dtype).reshape(shape)
        self._pinned = getattr(self, '_pinned', {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        p = getattr(self, '_pinned', {}).pop(arr.ctypes.data, None)
        if p is not None:
            self._ck(self.lib.fk_host_free(self.handle, C.c_void_p(p)))

    def witness_upload_async(self, slot, z):
        """fk_witness_upload_async: z must stay alive (and should be pinned) until the proof that reads the slot returns"""
        assert z.flags['C_CONTIGUOUS']
        self._ck(self.lib.fk_witness_upload_async(self.handle, C.c_int(slot), _vp(z), C.c_size_t(z.nbytes)))

    def witness_ptr(self, slot):
        p = C.c_void_p()
        self._ck(self.lib.fk_witness_ptr(self.handle, C.c_int(slot), C.byref(p)))
        return p.value

    # ---- the sharded hand-over: a rank uploads its own piece, the ranks all-gather the rest (parallel.witness_all_gather)
    def witness_slot(self, slot, total_bytes):
        """fk_witness_slot: (device pointer of the slot with room for total_bytes, hipStream_t of the copy stream as an integer)"""
        p, st = C.c_void_p(), C.c_void_p()
        self._ck(self.lib.fk_witness_slot(self.handle, C.c_int(slot), C.c_size_t(total_bytes), C.byref(p), C.byref(st)))
        return p.value or 0, st.value or 0

    def witness_upload_part_async(self, slot, part, offset):
        """fk_witness_upload_part_async: `part` (contiguous array; pinned memory overlaps) -> slot bytes [offset, offset + part.nbytes)"""
        assert part.flags['C_CONTIGUOUS']
        self._ck(self.lib.fk_witness_upload_part_async(self.handle, C.c_int(slot), _vp(part) if part.nbytes else None, C.c_size_t(offset), C.c_size_t(part.nbytes)))

    def witness_mark_ready(self, slot):
        self._ck(self.lib.fk_witness_mark_ready(self.handle, C.c_int(slot)))

    def prove_witness_submit(self, key, dr, z, r, s):
        """fk_prove_r1cs_submit -> ticket.  z, r, s are kept referenced until prove_witness_wait(ticket)."""
        assert z.dtype == np.uint64 and z.flags['C_CONTIGUOUS']
        dr.check_witness(z)
        r, s = _fr(r, 1), _fr(s, 1)
        t = C.c_int(-1)
        self._ck(self.lib.fk_prove_r1cs_submit(self.handle, key.handle, dr.handle, _vp(z), _vp(r), _vp(s), C.byref(t)))
        self._tickets = getattr(self, '_tickets', {})
        self._tickets[t.value] = (z, r, s, key, dr)
        return t.value

    def prove_witness_wait(self, ticket, want_timings=False):
        out = np.zeros(FK_PROOF_BYTES, np.uint8)
        tm = Timings()
        try:
            self._ck(self.lib.fk_prove_r1cs_wait(self.handle, C.c_int(ticket), _vp(out), C.byref(tm)))
        finally:
            getattr(self, '_tickets', {}).pop(ticket, None)
        return (out, tm.as_dict()) if want_timings else out

    def set_window_bits(self, c):
        self._ck(self.lib.fk_set_window_bits(self.handle, C.c_uint(c)))

    # ---- keys
    def load_key(self, params, shard_index=0, shard_count=1, z_frac=Z_EQUAL_SPLIT):
        d = params.desc(shard_index, shard_count, z_frac)
        h = C.c_void_p()
        self._ck(self.lib.fk_key_load(self.handle, C.byref(d), C.byref(h)))
        return DeviceKey(self, h, shard_index, shard_count)

    def synthetic_key(self, m, num_input, num_aux, n_a, n_b, seed=1, shard_index=0, shard_count=1, z_frac=Z_EQUAL_SPLIT):
        h = C.c_void_p()
        self._ck(self.lib.fk_key_synthetic(self.handle, C.c_uint64(m), C.c_uint32(num_input), C.c_uint32(num_aux),
                                           C.c_uint64(n_a), C.c_uint64(n_b), C.c_uint64(seed), C.c_uint32(shard_index),
                                           C.c_uint32(shard_count), C.c_double(z_frac[0]), C.c_double(z_frac[1]), C.byref(h)))
        return DeviceKey(self, h, shard_index, shard_count)

    # ---- building blocks (host arrays)
    def fr_mul_batch(self, a, b):
        a, b = _fr(a), _fr(b, None)
        out = np.zeros_like(a)
        self._ck(self.lib.fk_fr_mul_batch(self.handle, _vp(a), _vp(b), _vp(out), C.c_size_t(a.shape[0])))
        return out

    def ntt(self, data, inverse=False, coset=False):
        d = _fr(data).copy()
        log_n = int(d.shape[0]).bit_length() - 1
        if (1 << log_n) != d.shape[0]:
            raise FkError(1, 'ntt size must be a power of two')
        self._ck(self.lib.fk_ntt(self.handle, _vp(d), C.c_uint32(log_n), C.c_int(int(inverse)), C.c_int(int(coset))))
        return d

    def ntt_dev(self, dptr, log_n, inverse=False, coset=False):
        self._ck(self.lib.fk_ntt_dev(self.handle, C.c_void_p(dptr), C.c_uint32(log_n), C.c_int(int(inverse)), C.c_int(int(coset))))

    def quotient_h(self, a, b, c):
        a, b, c = _fr(a), _fr(b), _fr(c)
        n = a.shape[0]
        m = 1
        while m < n:
            m *= 2
        h = np.zeros((max(m - 1, 0), 4), np.uint64)
        hbuf = h if m > 1 else np.zeros((1, 4), np.uint64)
        self._ck(self.lib.fk_quotient_h(self.handle, _vp(a), _vp(b), _vp(c), C.c_uint64(n), _vp(hbuf)))
        return h

    def quotient_h_dev(self, d_a, d_b, d_c, n, d_h):
        self._ck(self.lib.fk_quotient_h_dev(self.handle, C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_c), C.c_uint64(n), C.c_void_p(d_h)))

    # distributed quotient building blocks (include/fawkes_hip.h: fk_dq_*); world = 2^log_w ranks
    def dq_gather_dev(self, d_full, n, log_m, rank, log_w, d_local):
        self._ck(self.lib.fk_dq_gather_dev(self.handle, C.c_void_p(d_full), C.c_uint64(n), C.c_uint32(log_m), C.c_uint32(rank),
                                           C.c_uint32(log_w), C.c_void_p(d_local)))

    def dq_local_dev(self, d_x, log_m, rank, log_w, stage, d_xb=0, d_xc=0):
        self._ck(self.lib.fk_dq_local_dev(self.handle, C.c_void_p(d_x), C.c_void_p(d_xb), C.c_void_p(d_xc), C.c_uint32(log_m),
                                          C.c_uint32(rank), C.c_uint32(log_w), C.c_int(stage)))

    def dq_cross_dev(self, d_buf, log_m, rank, log_w, mode):
        self._ck(self.lib.fk_dq_cross_dev(self.handle, C.c_void_p(d_buf), C.c_uint32(log_m), C.c_uint32(rank), C.c_uint32(log_w),
                                          C.c_int(mode)))

    def dq_cross_sub_dev(self, d_buf, d_sub, log_m, rank, log_w):
        self._ck(self.lib.fk_dq_cross_sub_dev(self.handle, C.c_void_p(d_buf), C.c_void_p(d_sub), C.c_uint32(log_m), C.c_uint32(rank), C.c_uint32(log_w)))

    def msm_g1(self, bases, scalars):
        bases = np.ascontiguousarray(bases, np.uint8).reshape(-1, 64)
        scalars = _fr(scalars, bases.shape[0])
        out = np.zeros(64, np.uint8)
        self._ck(self.lib.fk_msm_g1(self.handle, _vp(bases), _vp(scalars), C.c_size_t(bases.shape[0]), _vp(out)))
        return out

    def msm_g2(self, bases, scalars):
        bases = np.ascontiguousarray(bases, np.uint8).reshape(-1, 128)
        scalars = _fr(scalars, bases.shape[0])
        out = np.zeros(128, np.uint8)
        self._ck(self.lib.fk_msm_g2(self.handle, _vp(bases), _vp(scalars), C.c_size_t(bases.shape[0]), _vp(out)))
        return out

    def msm_g1_dev(self, d_bases, d_scalars, n):
        out = np.zeros(64, np.uint8)
        self._ck(self.lib.fk_msm_g1_dev(self.handle, C.c_void_p(d_bases), C.c_void_p(d_scalars), C.c_size_t(n), _vp(out)))
        return out

    def msm_g2_dev(self, d_bases, d_scalars, n):
        out = np.zeros(128, np.uint8)
        self._ck(self.lib.fk_msm_g2_dev(self.handle, C.c_void_p(d_bases), C.c_void_p(d_scalars), C.c_size_t(n), _vp(out)))
        return out

    def gen_points_g1_dev(self, dptr, n, seed):
        self._ck(self.lib.fk_gen_points_g1_dev(self.handle, C.c_void_p(dptr), C.c_size_t(n), C.c_uint64(seed)))

    def gen_points_g2_dev(self, dptr, n, seed):
        self._ck(self.lib.fk_gen_points_g2_dev(self.handle, C.c_void_p(dptr), C.c_size_t(n), C.c_uint64(seed)))

    def gen_scalars_dev(self, dptr, n, seed, kind=0):
        self._ck(self.lib.fk_gen_scalars_dev(self.handle, C.c_void_p(dptr), C.c_size_t(n), C.c_uint64(seed), C.c_int(kind)))

    # ---- synthesis + proving
    def synthesize(self, r1cs, z):
        return synthesize(r1cs, z, ctx=self)

    def prove_raw(self, key, a, b, c, z, a_aux, b_in, b_aux, r, s, want_timings=False):
        """fk_prove: returns the 256-byte proof (numpy uint8)."""
        # every converted array is bound to a local for the duration of the call: np.ascontiguousarray may copy, and
        # the address of a temporary would dangle
        a, b, c, z = _fr(a), _fr(b), _fr(c), _fr(z)
        a_aux, b_in, b_aux, r, s = _u8(a_aux), _u8(b_in), _u8(b_aux), _fr(r, 1), _fr(s, 1)
        out = np.zeros(FK_PROOF_BYTES, np.uint8)
        tm = Timings()
        self._ck(self.lib.fk_prove(self.handle, key.handle, _vp(a), _vp(b), _vp(c), C.c_uint64(a.shape[0]), _vp(z),
                                   C.c_void_p(a_aux.ctypes.data), _vp(b_in), C.c_void_p(b_aux.ctypes.data),
                                   _vp(r), _vp(s), _vp(out), C.byref(tm)))
        return (out, tm.as_dict()) if want_timings else out

    def prove_msms(self, key, a, b, c, z, a_aux, b_in, b_aux):
        a, b, c, z = _fr(a), _fr(b), _fr(c), _fr(z)
        a_aux, b_in, b_aux = _u8(a_aux), _u8(b_in), _u8(b_aux)
        out = np.zeros(FK_MSM_RESULT_BYTES, np.uint8)
        self._ck(self.lib.fk_prove_msms(self.handle, key.handle, _vp(a), _vp(b), _vp(c), C.c_uint64(a.shape[0]), _vp(z),
                                        C.c_void_p(a_aux.ctypes.data), _vp(b_in), C.c_void_p(b_aux.ctypes.data),
                                        _vp(out), None))
        return out

    def prove_msms_dev(self, key, d_a, d_b, d_c, n, d_z, d_a_aux, d_b_in, d_b_aux, want_timings=False):
        out = np.zeros(FK_MSM_RESULT_BYTES, np.uint8)
        tm = Timings()
        self._ck(self.lib.fk_prove_msms_dev(self.handle, key.handle, C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_c),
                                            C.c_uint64(n), C.c_void_p(d_z), C.c_void_p(d_a_aux), C.c_void_p(d_b_in),
                                            C.c_void_p(d_b_aux), _vp(out), C.byref(tm)))
        return (out, tm.as_dict()) if want_timings else out

    def prove_msms_z_dev(self, key, d_z, d_a_aux, d_b_in, d_b_aux):
        """witness MSMs L, A, B1, B2 of this key's slices -> 384-byte record with H = identity"""
        out = np.zeros(FK_MSM_RESULT_BYTES, np.uint8)
        self._ck(self.lib.fk_prove_msms_z_dev(self.handle, key.handle, C.c_void_p(d_z), C.c_void_p(d_a_aux), C.c_void_p(d_b_in),
                                              C.c_void_p(d_b_aux), _vp(out), None))
        return out

    def prove_msms_hz_dev(self, key, d_h_slice, d_z, d_a_aux, d_b_in, d_b_aux):
        """all five MSMs of this key's slices, H over the given block of quotient coefficients -> 384-byte record"""
        out = np.zeros(FK_MSM_RESULT_BYTES, np.uint8)
        self._ck(self.lib.fk_prove_msms_hz_dev(self.handle, key.handle, C.c_void_p(d_h_slice), C.c_void_p(d_z), C.c_void_p(d_a_aux),
                                               C.c_void_p(d_b_in), C.c_void_p(d_b_aux), _vp(out), None))
        return out

    def prove_msms_hz_r1cs_dev(self, key, device_r1cs, d_h_slice, d_z):
        """all five MSMs of this key's slices for a resident constraint system -> 384-byte record"""
        out = np.zeros(FK_MSM_RESULT_BYTES, np.uint8)
        self._ck(self.lib.fk_prove_msms_hz_r1cs_dev(self.handle, key.handle, device_r1cs.handle, C.c_void_p(d_h_slice), C.c_void_p(d_z), _vp(out)))
        return out

    def prove_msms_z_begin_dev(self, key, d_z, d_a_aux, d_b_in, d_b_aux):
        """queue L, A, B1, B2 on the MSM streams and return; pair with prove_msms_finish_dev"""
        self._ck(self.lib.fk_prove_msms_z_begin_dev(self.handle, key.handle, C.c_void_p(d_z), C.c_void_p(d_a_aux), C.c_void_p(d_b_in),
                                                    C.c_void_p(d_b_aux)))

    def prove_msms_z_begin_r1cs_dev(self, key, device_r1cs, d_z):
        """the same for a resident constraint system (index-list gathers instead of the density compaction)"""
        self._ck(self.lib.fk_prove_msms_z_begin_r1cs_dev(self.handle, key.handle, device_r1cs.handle, C.c_void_p(d_z)))

    def prove_msms_finish_dev(self, key, d_h_slice):
        out = np.zeros(FK_MSM_RESULT_BYTES, np.uint8)
        self._ck(self.lib.fk_prove_msms_finish_dev(self.handle, key.handle, C.c_void_p(d_h_slice), _vp(out)))
        return out

    def prove_msm_h_dev(self, key, d_h_slice):
        out = np.zeros(64, np.uint8)
        self._ck(self.lib.fk_prove_msm_h_dev(self.handle, key.handle, C.c_void_p(d_h_slice), _vp(out)))
        return out

    ARRAYS = {'h': 0, 'l': 1, 'a': 2, 'b_g1': 3, 'b_g2': 4}

    def prove_msm_array_dev(self, key, which, d_scalars):
        """fk_prove_msm_array_dev: ONE multiplication over the key's resident array `which` ('h', 'l', 'a', 'b_g1', 'b_g2'; with its
        fixed-base levels when held), one scalar per point of the key's slice; raw affine result (64 B, 128 B for 'b_g2')"""
        out = np.zeros(128 if which == 'b_g2' else 64, np.uint8)
        self._ck(self.lib.fk_prove_msm_array_dev(self.handle, key.handle, C.c_int(self.ARRAYS[which]), C.c_void_p(d_scalars), _vp(out)))
        return out

    def prove_dev(self, key, d_a, d_b, d_c, n, d_z, d_a_aux, d_b_in, d_b_aux, r, s, want_timings=False):
        out = np.zeros(FK_PROOF_BYTES, np.uint8)
        tm = Timings()
        r, s = _fr(r, 1), _fr(s, 1)
        self._ck(self.lib.fk_prove_dev(self.handle, key.handle, C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_c),
                                       C.c_uint64(n), C.c_void_p(d_z), C.c_void_p(d_a_aux), C.c_void_p(d_b_in),
                                       C.c_void_p(d_b_aux), _vp(r), _vp(s), _vp(out), C.byref(tm)))
        return (out, tm.as_dict()) if want_timings else out

    def load_key_bellman(self, data, shard_index=0, shard_count=1, z_frac=Z_EQUAL_SPLIT, flags=FK_KEY_CHECKED):
        """fk_key_load_bellman: `data` = bytes of bellman's Parameters::write; flags = FK_KEY_CHECKED | FK_KEY_NO_INFINITY (the
        `checked` / `disallow_points_at_infinity` arguments of Parameters::read, mod.rs:159).  Returns (DeviceKey, gamma_g2, ic)."""
        buf = _bytes_view(data)      # (no copy of a multi-GB array)
        h = C.c_void_p()
        gamma = np.zeros(128, np.uint8)
        n_ic = C.c_uint32()
        cap = 1 << 16
        ic = np.zeros((cap, 64), np.uint8)
        self._ck(self.lib.fk_key_load_bellman(self.handle, _vp(buf), C.c_size_t(buf.size), C.c_uint32(flags), C.c_uint32(shard_index), C.c_uint32(shard_count),
                                              C.c_double(z_frac[0]), C.c_double(z_frac[1]), C.byref(h), _vp(gamma), _vp(ic), C.c_uint32(cap),
                                              C.byref(n_ic)))
        return DeviceKey(self, h, shard_index, shard_count), gamma, ic[:min(n_ic.value, cap)].copy()

    def write_key_bellman(self, key, vk, out=None, size_only=False):
        """fk_key_write_bellman: bellman `Parameters::write` bytes of a whole resident key (GPU conversion); vk: the dict fk_setup* /
        load_key_bellman returned (gamma_g2 and ic are not part of a proving key).  out: a contiguous uint8 array of at least the
        needed size to write into (e.g. the tail of a `Parameters` image); size_only: return the byte count."""
        gamma = np.ascontiguousarray(vk['gamma_g2'], np.uint8).reshape(-1)
        ic = np.ascontiguousarray(vk['ic'], np.uint8).reshape(-1, 64)
        need = C.c_size_t()
        self._ck(self.lib.fk_key_write_bellman(self.handle, key.handle, _vp(gamma), _vp(ic), C.c_uint32(ic.shape[0]), None, C.c_size_t(0), C.byref(need)))
        if size_only:
            return need.value
        if out is None:
            out = np.empty(need.value, np.uint8)
        assert out.dtype == np.uint8 and out.flags['C_CONTIGUOUS'] and out.nbytes >= need.value
        self._ck(self.lib.fk_key_write_bellman(self.handle, key.handle, _vp(gamma), _vp(ic), C.c_uint32(ic.shape[0]), _vp(out), C.c_size_t(out.nbytes), C.byref(need)))
        return out[:need.value]

    def setup(self, r1cs, tau, alpha, beta, gamma, delta, shard_index=0, shard_count=1, z_frac=Z_EQUAL_SPLIT, copies=None):
        """fk_setup: GPU key generation with explicit toxic waste (Montgomery limbs).  Returns (DeviceKey, vk dict)
        with vk = alpha_g1, beta_g1, beta_g2, gamma_g2, delta_g1, delta_g2 (raw LE uint8 arrays) and ic (num_input, 64).
        copies: fk_setup_tiled -- r1cs is one instance of a batch circuit, the key is for `copies` of it."""
        h = C.c_void_p()
        vk = np.zeros(6 * 128, np.uint8)
        tau, alpha, beta, gamma, delta = _fr(tau, 1), _fr(alpha, 1), _fr(beta, 1), _fr(gamma, 1), _fr(delta, 1)
        tail = (_vp(tau), _vp(alpha), _vp(beta), _vp(gamma), _vp(delta), C.c_uint32(shard_index),
                C.c_uint32(shard_count), C.c_double(z_frac[0]), C.c_double(z_frac[1]), C.byref(h), _vp(vk))
        if copies is None:
            ic = np.zeros((r1cs.num_input, 64), np.uint8)
            self._ck(self.lib.fk_setup(self.handle, C.byref(r1cs.struct), *tail, _vp(ic)))
        else:
            ic = np.zeros((1 + int(copies) * (r1cs.num_input - 1), 64), np.uint8)
            self._ck(self.lib.fk_setup_tiled(self.handle, C.byref(r1cs.struct), C.c_uint32(int(copies)), *tail, _vp(ic)))
        names = (('alpha_g1', 64), ('beta_g1', 64), ('beta_g2', 128), ('gamma_g2', 128), ('delta_g1', 64), ('delta_g2', 128))
        out = {n: vk[i * 128:i * 128 + w].copy() for i, (n, w) in enumerate(names)}
        out['ic'] = ic
        return DeviceKey(self, h, shard_index, shard_count), out

    # ---- device-resident constraint system: only the witness crosses the boundary
    def load_r1cs(self, r1cs, copies=None):
        """copies: fk_r1cs_load_tiled -- r1cs is one instance, the resident system stands for `copies` of it"""
        h = C.c_void_p()
        if copies is None:
            self._ck(self.lib.fk_r1cs_load(self.handle, C.byref(r1cs.struct), C.byref(h)))
        else:
            self._ck(self.lib.fk_r1cs_load_tiled(self.handle, C.byref(r1cs.struct), C.c_uint32(int(copies)), C.byref(h)))
        return DeviceR1cs(self, h)

    def load_r1cs_coded(self, num_input, num_aux, mats, table):
        """fk_r1cs_load_coded: mats = three (ptr u64, col u32, cidx u32) triples, cidx indexing `table` ((n, 4) uint64 Montgomery,
        table[0] = ONE).  8 bytes per term: how a system with 10^9 explicit terms is loaded."""
        keep = []
        st = R1csStruct()
        st.num_input, st.num_aux, st.num_gates = num_input, num_aux, len(mats[0][0]) - 1
        cidx = []
        for nm, (ptr, col, ci) in zip('abc', mats):
            ptr = np.ascontiguousarray(ptr, np.uint64); col = np.ascontiguousarray(col, np.uint32); ci = np.ascontiguousarray(ci, np.uint32)
            assert len(col) == len(ci) == int(ptr[-1])
            keep += [ptr, col, ci]
            setattr(st, nm + '_ptr', ptr.ctypes.data); setattr(st, nm + '_col', col.ctypes.data if len(col) else None); setattr(st, nm + '_val', None)
            cidx.append(ci)
        table = _fr(table)
        h = C.c_void_p()
        self._ck(self.lib.fk_r1cs_load_coded(self.handle, C.byref(st), _vp(cidx[0]), _vp(cidx[1]), _vp(cidx[2]), _vp(table), C.c_uint64(len(table)), C.byref(h)))
        return DeviceR1cs(self, h)

    def r1cs_eval_dev(self, dr, d_z, d_a, d_b, d_c):
        self._ck(self.lib.fk_r1cs_eval_dev(self.handle, dr.handle, C.c_void_p(d_z), C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_c)))

    def r1cs_eval_slice_dev(self, dr, d_z, log_m, rank, log_w, d_a, d_b, d_c):
        """fk_r1cs_eval_slice_dev: the cyclic row slice of rank `rank` of 2^log_w (2^(log_m - log_w) elements per array)"""
        self._ck(self.lib.fk_r1cs_eval_slice_dev(self.handle, dr.handle, C.c_void_p(d_z), C.c_uint32(log_m), C.c_uint32(rank), C.c_uint32(log_w),
                                                 C.c_void_p(d_a), C.c_void_p(d_b), C.c_void_p(d_c)))

    def prove_witness(self, key, dr, z, r, s, want_timings=False):
        """fk_prove_r1cs: z (host, (num_input+num_aux, 4) uint64 Montgomery) -> 256-byte proof."""
        z, r, s = _fr(z), _fr(r, 1), _fr(s, 1)
        dr.check_witness(z)
        out = np.zeros(FK_PROOF_BYTES, np.uint8)
        tm = Timings()
        self._ck(self.lib.fk_prove_r1cs(self.handle, key.handle, dr.handle, _vp(z), _vp(r), _vp(s), _vp(out), C.byref(tm)))
        return (out, tm.as_dict()) if want_timings else out

    def prove_witness_dev(self, key, dr, d_z, r, s, want_timings=False):
        out = np.zeros(FK_PROOF_BYTES, np.uint8)
        tm = Timings()
        r, s = _fr(r, 1), _fr(s, 1)
        self._ck(self.lib.fk_prove_r1cs_dev(self.handle, key.handle, dr.handle, C.c_void_p(d_z), _vp(r), _vp(s), _vp(out), C.byref(tm)))
        return (out, tm.as_dict()) if want_timings else out

    def prove_assemble(self, key, parts, r, s):
        return assemble(key.handle, parts, r, s, ctx=self)

    def stats_reset(self):
        self._ck(self.lib.fk_stats_reset(self.handle))

    def stats(self):
        """HIP-event kernel times since stats_reset(): G1/G2 bucket accumulation and NTT passes."""
        out = {}
        for which, name in ((0, 'acc_g1'), (1, 'acc_g2'), (2, 'ntt')):
            ms, n, u = C.c_double(), C.c_uint64(), C.c_uint64()
            self._ck(self.lib.fk_stats_get(self.handle, C.c_int(which), C.byref(ms), C.byref(n), C.byref(u)))
            out[name] = dict(ms=ms.value, launches=n.value, units=u.value)
        for which, name in ((3, 'acc_g1'), (4, 'acc_g2')):     # the same kernels, counted in mixed point additions
            u = C.c_uint64()
            self._ck(self.lib.fk_stats_get(self.handle, C.c_int(which), None, None, C.byref(u)))
            out[name]['adds'] = u.value
        for which, name in ((5, 'acc_g1'), (6, 'acc_g2')):     # union of the launches' intervals (launches side by side)
            ms = C.c_double()
            self._ck(self.lib.fk_stats_get(self.handle, C.c_int(which), C.byref(ms), None, None))
            out[name]['union_ms'] = ms.value
        return out

    def calibrate(self):
        """fk_calibrate: dict(mad_lane_ops_per_s, modmul_per_s) measured live on this device"""
        out = (C.c_double * 2)()
        self._ck(self.lib.fk_calibrate(self.handle, out))
        return dict(mad_lane_ops_per_s=out[0], modmul_per_s=out[1])

    def dev_copy(self, dst, src, nbytes):
        self._ck(self.lib.fk_dev_copy(self.handle, C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes)))


class _MultiHandle:
    """a key / constraint system loaded through a MultiContext (one shard / replica per GPU)"""

    def __init__(self, multi, handle, free_fn):
        self.multi, self.handle, self._free = multi, handle, free_fn

    def free(self):
        if self.handle is not None:
            self._free(self.multi.handle or None, self.handle)      # a closed MultiContext: the device memory is released all the same
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MultiContext:
    """N GPUs of one node behind ONE call (fk_init_devices / fk_multi_*, csrc/multi.hip): one process, a library context and a
    worker thread per GPU, exchanges inside the library.  `device_ids` may repeat (ranks sharing a GPU: one-GPU test boxes).
    The reference's `prove` (prover.rs:63-90) is one call; so is `prove_witness` here, whatever N is."""

    def __init__(self, device_ids):
        self.lib = load_library()
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        h = C.c_void_p()
        rc = self.lib.fk_init_devices(C.c_int(len(device_ids)), ids, C.byref(h))
        if rc != 0:
            raise FkError(rc, 'fk_init_devices(%s) failed -- no usable MI355X/HIP device; there is no CPU fallback' % list(device_ids))
        self.handle = h
        self.device_ids = list(device_ids)
        self._tickets = {}

    @property
    def size(self):
        return int(self.lib.fk_multi_size(self.handle))

    @property
    def transport(self):
        """'peer-dma' (hipMemcpyPeerAsync pulls) or 'rccl' (FK_MULTI_TRANSPORT=rccl)"""
        return self.lib.fk_multi_transport(self.handle).decode()

    def note(self):
        """text left by fk_init_devices / the last call (e.g. why RCCL was not used)"""
        msg = self.lib.fk_multi_last_error(self.handle)
        return msg.decode() if msg else ''

    PEER_STATES = {0: 'self', 1: 'direct', 2: 'staged (not reachable)', 3: 'staged (enable refused)'}

    def topology(self):
        """fk_multi_topology: N x N list of strings -- entry [i][j] says how copies INTO rank i's device FROM rank j's travel:
        'self' (same device), 'direct' (peer access granted: DMA over xGMI), 'staged (...)' (the runtime stages them)"""
        n = self.size
        out = (C.c_int32 * (n * n))()
        self._ck(self.lib.fk_multi_topology(self.handle, out))
        return [[self.PEER_STATES.get(int(out[i * n + j]), '?') for j in range(n)] for i in range(n)]

    def preflight(self, nbytes=64 << 20):
        """fk_multi_preflight: one verified, timed `nbytes` pull per ordered pair of ranks behind a cross-device event wait.  Returns
        dict(ok, gbps (N x N), status (N x N), host_events (the library switched itself to host-side event waits), note).  Never raises for
        a failing pair: the caller decides what to do with a node whose links do not work."""
        n = self.size
        gb = (C.c_double * (n * n))()
        st = (C.c_int32 * (n * n))()
        he = C.c_int(0)
        rc = self.lib.fk_multi_preflight(self.handle, C.c_size_t(int(nbytes)), gb, st, C.byref(he))
        return dict(ok=rc == 0, rc=int(rc), bytes=int(nbytes), gbps=[[round(float(gb[i * n + j]), 2) for j in range(n)] for i in range(n)],
                    status=[[int(st[i * n + j]) for j in range(n)] for i in range(n)], host_events=bool(he.value), note=self.note())

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.fk_multi_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            msg = self.lib.fk_multi_last_error(self.handle)
            raise FkError(rc, msg.decode() if msg else '')

    def ctx(self, rank):
        """rank's single-GPU Context (borrowed: statistics, calibration, pinned host buffers)"""
        p = self.lib.fk_multi_ctx(self.handle, C.c_int(rank))
        if not p:
            raise IndexError(rank)
        return Context(self.device_ids[rank], _borrowed=p)

    def sync(self):
        self._ck(self.lib.fk_multi_sync(self.handle))

    def witness_traffic(self):
        """fk_multi_witness_traffic: bytes the latest witness hand-over moved, summed over the ranks (host -> device, device -> device)"""
        out = (C.c_uint64 * 2)()
        self._ck(self.lib.fk_multi_witness_traffic(self.handle, out))
        return dict(pcie_bytes=int(out[0]), gathered_bytes=int(out[1]))

    def load_key(self, params):
        d = params.desc(0, 1, Z_EQUAL_SPLIT)
        h = C.c_void_p()
        self._ck(self.lib.fk_multi_key_load(self.handle, C.byref(d), C.byref(h)))
        return _MultiHandle(self, h, self.lib.fk_multi_key_free)

    def load_key_bellman(self, data, flags=FK_KEY_CHECKED):
        buf = _bytes_view(data)      # (no copy of a multi-GB array)
        h = C.c_void_p()
        gamma = np.zeros(128, np.uint8)
        n_ic = C.c_uint32()
        cap = 1 << 16
        ic = np.zeros((cap, 64), np.uint8)
        self._ck(self.lib.fk_multi_key_load_bellman(self.handle, _vp(buf), C.c_size_t(buf.size), C.c_uint32(flags), C.byref(h), _vp(gamma), _vp(ic),
                                                    C.c_uint32(cap), C.byref(n_ic)))
        return _MultiHandle(self, h, self.lib.fk_multi_key_free), gamma, ic[:min(n_ic.value, cap)].copy()

    def setup(self, r1cs, tau, alpha, beta, gamma, delta, copies=None):
        """fk_multi_setup / fk_multi_setup_tiled: rank g derives only shard g.  Returns (key, vk dict) like Context.setup."""
        h = C.c_void_p()
        vk = np.zeros(6 * 128, np.uint8)
        tau, alpha, beta, gamma, delta = _fr(tau, 1), _fr(alpha, 1), _fr(beta, 1), _fr(gamma, 1), _fr(delta, 1)
        tox = (_vp(tau), _vp(alpha), _vp(beta), _vp(gamma), _vp(delta))
        if copies is None:
            ic = np.zeros((r1cs.num_input, 64), np.uint8)
            self._ck(self.lib.fk_multi_setup(self.handle, C.byref(r1cs.struct), *tox, C.byref(h), _vp(vk), _vp(ic)))
        else:
            ic = np.zeros((1 + int(copies) * (r1cs.num_input - 1), 64), np.uint8)
            self._ck(self.lib.fk_multi_setup_tiled(self.handle, C.byref(r1cs.struct), C.c_uint32(int(copies)), *tox, C.byref(h), _vp(vk), _vp(ic)))
        names = (('alpha_g1', 64), ('beta_g1', 64), ('beta_g2', 128), ('gamma_g2', 128), ('delta_g1', 64), ('delta_g2', 128))
        out = {n: vk[i * 128:i * 128 + w].copy() for i, (n, w) in enumerate(names)}
        out['ic'] = ic
        return _MultiHandle(self, h, self.lib.fk_multi_key_free), out

    def key_shard(self, key, rank):
        """rank's shard as a (borrowed) DeviceKey: counts(), shard_info(), precomputed()"""
        dk = DeviceKey(self.ctx(rank), C.c_void_p(self.lib.fk_multi_key_shard(key.handle, C.c_int(rank))), rank, self.size)
        dk.free = lambda: None
        return dk

    def load_r1cs(self, r1cs, copies=None):
        h = C.c_void_p()
        if copies is None:
            self._ck(self.lib.fk_multi_r1cs_load(self.handle, C.byref(r1cs.struct), C.byref(h)))
        else:
            self._ck(self.lib.fk_multi_r1cs_load_tiled(self.handle, C.byref(r1cs.struct), C.c_uint32(int(copies)), C.byref(h)))
        return _MultiHandle(self, h, self.lib.fk_multi_r1cs_free)

    def load_gates(self, gates):
        h = C.c_void_p()
        self._ck(self.lib.fk_multi_r1cs_load_gates(self.handle, gates.handle, C.byref(h)))
        return _MultiHandle(self, h, self.lib.fk_multi_r1cs_free)

    def r1cs_replica(self, dr, rank):
        d = DeviceR1cs(self.ctx(rank), C.c_void_p(self.lib.fk_multi_r1cs_replica(dr.handle, C.c_int(rank))))
        d.free = lambda: None
        return d

    def prove_witness(self, key, dr, z, r, s, want_timings=False):
        """fk_multi_prove_r1cs: z (host, (num_input + num_aux, 4) uint64 Montgomery) -> 256-byte proof, on all GPUs"""
        z, r, s = _fr(z), _fr(r, 1), _fr(s, 1)
        self.r1cs_replica(dr, 0).check_witness(z)
        out = np.zeros(FK_PROOF_BYTES, np.uint8)
        tm = Timings()
        self._ck(self.lib.fk_multi_prove_r1cs(self.handle, key.handle, dr.handle, _vp(z), _vp(r), _vp(s), _vp(out), C.byref(tm)))
        return (out, tm.as_dict()) if want_timings else out

    def prove_witness_submit(self, key, dr, z, r, s):
        assert z.dtype == np.uint64 and z.flags['C_CONTIGUOUS']
        self.r1cs_replica(dr, 0).check_witness(z)
        r, s = _fr(r, 1), _fr(s, 1)
        t = C.c_int(-1)
        self._ck(self.lib.fk_multi_prove_r1cs_submit(self.handle, key.handle, dr.handle, _vp(z), _vp(r), _vp(s), C.byref(t)))
        self._tickets[t.value] = (z, r, s, key, dr)
        return t.value

    def prove_witness_wait(self, ticket, want_timings=False):
        out = np.zeros(FK_PROOF_BYTES, np.uint8)
        tm = Timings()
        try:
            self._ck(self.lib.fk_multi_prove_r1cs_wait(self.handle, C.c_int(ticket), _vp(out), C.byref(tm)))
        finally:
            self._tickets.pop(ticket, None)
        return (out, tm.as_dict()) if want_timings else out


def synthesize(r1cs, z, ctx=None):
    """ProvingAssignment::enforce/eval: a, b, c and the density maps (host code of the
    product library; needs no GPU)."""
    lib = load_library()
    z = _fr(z, r1cs.num_input + r1cs.num_aux)
    n = r1cs.n_rows
    a = np.zeros((n, 4), np.uint64); b = np.zeros((n, 4), np.uint64); c = np.zeros((n, 4), np.uint64)
    a_aux = np.zeros(max(r1cs.num_aux, 1), np.uint8)[:r1cs.num_aux]
    b_in = np.zeros(r1cs.num_input, np.uint8)
    b_aux = np.zeros(max(r1cs.num_aux, 1), np.uint8)[:r1cs.num_aux]
    rc = lib.fk_synthesize(ctx.handle if ctx else None, C.byref(r1cs.struct), _vp(z), _vp(a), _vp(b), _vp(c),
                           C.c_void_p(a_aux.ctypes.data), _vp(b_in), C.c_void_p(b_aux.ctypes.data))
    if rc != 0:
        msg = lib.fk_last_error(ctx.handle) if ctx else b''
        raise FkError(rc, msg.decode() if msg else '')
    return a, b, c, a_aux, b_in, b_aux


def assemble(key_handle, parts, r, s, ctx=None):
    """fk_prove_assemble: fold n_parts x 384 B partial MSM results into the 256-byte proof (host only)."""
    lib = load_library()
    parts = np.ascontiguousarray(parts, np.uint8).reshape(-1, FK_MSM_RESULT_BYTES)
    out = np.zeros(FK_PROOF_BYTES, np.uint8)
    r, s = _fr(r, 1), _fr(s, 1)
    rc = lib.fk_prove_assemble(ctx.handle if ctx else None, key_handle, _vp(parts), C.c_uint32(parts.shape[0]),
                               _vp(r), _vp(s), _vp(out))
    if rc != 0:
        msg = lib.fk_last_error(ctx.handle) if ctx else b''
        raise FkError(rc, msg.decode() if msg else '')
    return out


def shard_range(n, index, count):
    lib = load_library()
    lo, hi = C.c_uint64(), C.c_uint64()
    lib.fk_shard_range(C.c_uint64(n), C.c_uint32(index), C.c_uint32(count), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def work_shard_ranges(n_l, n_a, n_b, index, count, q0_domain=None):
    """fk_work_shard_ranges: dict of the [lo, hi) slices of l, a, b (= b_g1), b_g2 of shard `index` of `count` under Z_WORK_SPLIT
    (q0_domain = m: under Z_WORK_SPLIT_Q0, fk_work_shard_ranges_q0)"""
    lib = load_library()
    out = (C.c_uint64 * 8)()
    if q0_domain is None:
        lib.fk_work_shard_ranges(C.c_uint64(n_l), C.c_uint64(n_a), C.c_uint64(n_b), C.c_uint32(index), C.c_uint32(count), out)
    else:
        lib.fk_work_shard_ranges_q0(C.c_uint64(n_l), C.c_uint64(n_a), C.c_uint64(n_b), C.c_uint64(q0_domain), C.c_uint32(index), C.c_uint32(count), out)
    v = list(out)
    return dict(l=(v[0], v[1]), a=(v[2], v[3]), b=(v[4], v[5]), b_g2=(v[6], v[7]))


def h_shard_range(n_h, index, count):
    """[lo, hi) of the h bases shard `index` holds: blocks of the evaluation domain (fk_h_shard_range)"""
    lib = load_library()
    lo, hi = C.c_uint64(), C.c_uint64()
    lib.fk_h_shard_range(C.c_uint64(n_h), C.c_uint32(index), C.c_uint32(count), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


# ------------------------------------------------------------------------------------------ verifier
_FQ_RINV = pow(1 << 256, -1, FQ_MODULUS)


def vk_to_borsh(vk):
    """fawkes' Borsh `VK` (verifier.rs:46-54) from raw Montgomery-LE points (what fk_setup / fk_key_load_bellman hand out):
    vk = dict(alpha_g1 (64 B), beta_g2, gamma_g2, delta_g2 (128 B), ic (n, 64)).  Coordinates become canonical LE integers."""
    def canon(raw):
        raw = bytes(raw)
        return b''.join((int.from_bytes(raw[i:i + 32], 'little') * _FQ_RINV % FQ_MODULUS).to_bytes(32, 'little') for i in range(0, len(raw), 32))
    ic = np.ascontiguousarray(vk['ic'], np.uint8).reshape(-1, 64)
    return (canon(vk['alpha_g1']) + canon(vk['beta_g2']) + canon(vk['gamma_g2']) + canon(vk['delta_g2']) +
            len(ic).to_bytes(4, 'little') + b''.join(canon(r) for r in ic))


def verify(vk_borsh, inputs, proof, ctx=None):
    """`verifier::verify(vk, proof, inputs)` (verifier.rs:75-81) on the host (fk_verify; no GPU needed).  inputs: Montgomery Fr
    (n, 4) uint64 without the leading ONE; proof: 256 bytes or a Proof.  Returns True / False."""
    lib = load_library()
    vkb = np.frombuffer(bytes(vk_borsh), np.uint8)
    pr = np.frombuffer(proof.to_bytes() if isinstance(proof, Proof) else bytes(proof), np.uint8)
    assert pr.size == FK_PROOF_BYTES
    inp = np.ascontiguousarray(inputs, np.uint64).reshape(-1, 4)
    ok = C.c_int(0)
    rc = lib.fk_verify(ctx.handle if ctx else None, _vp(vkb), C.c_size_t(vkb.size), _vp(inp), C.c_uint32(inp.shape[0]), _vp(pr), C.byref(ok))
    if rc != 0:
        msg = lib.fk_last_error(ctx.handle) if ctx else b''
        raise FkError(rc, msg.decode() if msg else 'fk_verify')
    return bool(ok.value)


def verify_batch(ctx, vk_borsh, inputs, proofs):
    """fk_verify_batch_dev: `count` proofs of one key on the GPU, one lane per proof.  inputs (count, n_inputs, 4) uint64,
    proofs (count, 256) uint8.  Returns a bool array; a proof that does not even decode (coordinate >= q) is False like any other
    bad proof, it does not fail the batch (the single-proof `verify` raises FK_ERR_FORMAT for it, as upstream's Borsh reader would)."""
    vkb = np.frombuffer(bytes(vk_borsh), np.uint8)
    pr = np.ascontiguousarray(proofs, np.uint8).reshape(-1, FK_PROOF_BYTES)
    inp = np.ascontiguousarray(inputs, np.uint64)
    inp = inp.reshape(pr.shape[0], -1, 4) if inp.size else np.zeros((pr.shape[0], 0, 4), np.uint64)     # a key without public inputs
    out = np.zeros(pr.shape[0], np.uint8)
    ctx._ck(ctx.lib.fk_verify_batch_dev(ctx.handle, _vp(vkb), C.c_size_t(vkb.size), _vp(inp), C.c_uint32(inp.shape[1]), _vp(pr),
                                        C.c_uint32(pr.shape[0]), _vp(out)))
    return out.astype(bool)


# ------------------------------------------------------------------------------------------ r, s sampling
def _osrng_next_u64(rand=os.urandom):
    # fawkes OsRng::next_u32 = 4 getrandom bytes big-endian (osrng.rs:13-17); rand-0.4 composes
    # next_u64 from two next_u32, high word first
    hi = int.from_bytes(rand(4), 'big')
    lo = int.from_bytes(rand(4), 'big')
    return (hi << 32) | lo


def sample_fr(rand=os.urandom):
    """bellman `Fr::rand`: 4 x next_u64 limbs, top 2 bits cleared, rejected unless < r; the accepted
    limbs are the MONTGOMERY representation.  Returns uint64[4]."""
    while True:
        limbs = [_osrng_next_u64(rand) for _ in range(4)]
        limbs[3] &= (1 << 62) - 1
        v = sum(l << (64 * i) for i, l in enumerate(limbs))
        if v < FR_MODULUS:
            return np.array(limbs, dtype=np.uint64)


# ------------------------------------------------------------------------------------------ prove
def prove_with_rs(ctx, params, key, z_input, z_aux, r, s, want_timings=False, device_r1cs=None):
    """Deterministic twin of `prove` (bellman's create_proof(circuit, params, r, s)).
    z_input / z_aux: the `WitnessCS` value vectors (cs.rs:100-101), Montgomery limbs (n,4) uint64;
    z_input[0] must be ONE (cs.rs:111).  Returns (public inputs without ONE, Proof)."""
    if params.r1cs is None:
        raise FkError(1, 'Parameters carries no constraint system')
    z_input, z_aux = _fr(z_input, params.num_input), _fr(z_aux, params.num_aux) if params.num_aux else np.zeros((0, 4), np.uint64)
    z = np.concatenate([z_input, z_aux], axis=0)
    if device_r1cs is not None:     # constraint system resident in HBM: only z crosses the boundary
        res = ctx.prove_witness(key, device_r1cs, z, r, s, want_timings=want_timings)
    else:                           # host synthesis (bellman's ProvingAssignment restated in the library)
        a, b, c, a_aux, b_in, b_aux = ctx.synthesize(params.r1cs, z)
        res = ctx.prove_raw(key, a, b, c, z, a_aux, b_in, b_aux, r, s, want_timings=want_timings)
    proof_bytes, tm = (res if want_timings else (res, None))
    inputs = z_input[1:].copy()   # prover.rs:84-87
    proof = Proof.from_bytes(proof_bytes.tobytes())
    return (inputs, proof, tm) if want_timings else (inputs, proof)


def prove(ctx, params, key, z_input, z_aux, device_r1cs=None):
    """Mirror of prover.rs:63-90 below the DSL: r, s are drawn from the OS entropy source exactly like
    `create_random_proof` does through fawkes' OsRng (prover.rs:78-80)."""
    return prove_with_rs(ctx, params, key, z_input, z_aux, sample_fr(), sample_fr(), device_r1cs=device_r1cs)
