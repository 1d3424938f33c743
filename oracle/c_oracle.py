"""
ORACLE (test infrastructure, NOT product code): ctypes binding of oracle/liboracle.so
(oracle/groth16_oracle.c, the plain-C restatement of the bellman CPU path; see its header for the
parity status -- "parity unpinned").  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

U64P = C.POINTER(C.c_uint64)
U32P = C.POINTER(C.c_uint32)
U8P = C.POINTER(C.c_uint8)


def build():
    subprocess.check_call(['make', '-C', _HERE, '-s'], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'liboracle.so')
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_init()
        _LIB.orc_quotient_h.restype = C.c_uint64
        _LIB.orc_setup.restype = C.POINTER(OrcKey)
    return _LIB


class OrcR1cs(C.Structure):
    _fields_ = [('num_input', C.c_uint32), ('num_aux', C.c_uint32), ('num_gates', C.c_uint64),
                ('a_ptr', C.c_void_p), ('a_col', C.c_void_p), ('a_val', C.c_void_p),
                ('b_ptr', C.c_void_p), ('b_col', C.c_void_p), ('b_val', C.c_void_p),
                ('c_ptr', C.c_void_p), ('c_col', C.c_void_p), ('c_val', C.c_void_p)]


class OrcKey(C.Structure):
    _fields_ = [('m', C.c_uint64), ('num_input', C.c_uint32), ('num_aux', C.c_uint32),
                ('n_h', C.c_uint64), ('n_l', C.c_uint64), ('n_a', C.c_uint64), ('n_b', C.c_uint64),
                ('alpha_g1', C.c_uint8 * 64), ('beta_g1', C.c_uint8 * 64), ('beta_g2', C.c_uint8 * 128),
                ('gamma_g2', C.c_uint8 * 128), ('delta_g1', C.c_uint8 * 64), ('delta_g2', C.c_uint8 * 128),
                ('ic', C.c_void_p), ('h', C.c_void_p), ('l', C.c_void_p), ('a', C.c_void_p),
                ('b_g1', C.c_void_p), ('b_g2', C.c_void_p)]


def _p(arr):
    return arr.ctypes.data_as(C.c_void_p)


def limbs(x):
    """python int -> np.uint64[4] little-endian limbs"""
    return np.frombuffer(int(x).to_bytes(32, 'little'), dtype=np.uint64).copy()


def limbs_arr(xs):
    return np.frombuffer(b''.join(int(x).to_bytes(32, 'little') for x in xs), dtype=np.uint64).reshape(-1, 4).copy()


def to_int(l):
    return int.from_bytes(np.ascontiguousarray(l, dtype=np.uint64).tobytes(), 'little')


def ints(arr):
    b = np.ascontiguousarray(arr, dtype=np.uint64).tobytes()
    return [int.from_bytes(b[i:i + 32], 'little') for i in range(0, len(b), 32)]


FQ, FR, FX = 0, 1, 2


def field_custom(modulus):
    lib().orc_field_custom(_p(limbs(modulus)))


def fe_from_canon(f, x):
    o = np.zeros(4, np.uint64); lib().orc_fe_from_canon(f, _p(limbs(x)), _p(o)); return o


def fe_to_canon(f, a):
    o = np.zeros(4, np.uint64); lib().orc_fe_to_canon(f, _p(np.ascontiguousarray(a)), _p(o)); return to_int(o)


def _bin(name, f, a, b):
    o = np.zeros(4, np.uint64); getattr(lib(), name)(f, _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(o)); return o


def fe_add(f, a, b): return _bin('orc_fe_add', f, a, b)
def fe_sub(f, a, b): return _bin('orc_fe_sub', f, a, b)
def fe_mul(f, a, b): return _bin('orc_fe_mul', f, a, b)
def fe_pow(f, a, e): return _bin('orc_fe_pow', f, a, limbs(e))


def fe_neg(f, a):
    o = np.zeros(4, np.uint64); lib().orc_fe_neg(f, _p(np.ascontiguousarray(a)), _p(o)); return o


def fe_inv(f, a):
    o = np.zeros(4, np.uint64); lib().orc_fe_inv(f, _p(np.ascontiguousarray(a)), _p(o)); return o


def fe_mul_batch(f, a, b):
    a = np.ascontiguousarray(a, np.uint64); b = np.ascontiguousarray(b, np.uint64)
    o = np.zeros_like(a); lib().orc_fe_mul_batch(f, _p(a), _p(b), _p(o), C.c_size_t(a.shape[0])); return o


class _threads:
    """bellman's Worker size for the calls inside (1 = serial; more = its multicore split restated: same results)"""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        lib().orc_set_threads(C.c_int(self.n))

    def __exit__(self, *exc):
        lib().orc_set_threads(C.c_int(1))


def fr_ntt(a, inverse=False, coset=False, threads=1):
    """a: (2^k, 4) uint64 Montgomery Fr; returns transformed copy (natural order)."""
    a = np.ascontiguousarray(a, np.uint64).copy()
    logn = int(a.shape[0]).bit_length() - 1
    assert 1 << logn == a.shape[0]
    fn = lib().orc_fr_coset_ntt if coset else lib().orc_fr_ntt
    with _threads(threads):
        rc = fn(_p(a), C.c_uint32(logn), C.c_int(1 if inverse else 0))
    assert rc == 0
    return a


def quotient_h(a, b, c):
    a = np.ascontiguousarray(a, np.uint64); b = np.ascontiguousarray(b, np.uint64); c = np.ascontiguousarray(c, np.uint64)
    n = a.shape[0]
    m = 1
    while m < n:
        m *= 2
    h = np.zeros((m, 4), np.uint64)
    got = lib().orc_quotient_h(_p(a), _p(b), _p(c), C.c_uint64(n), _p(h))
    assert got == m
    return h[:m - 1]


def msm_g1(bases, scalars, density=None, threads=1):
    """bases: (nb, 64) uint8 raw LE; scalars (n, 4) uint64 Montgomery; density: optional uint8[n]; threads: one task per multiexp region."""
    bases = np.ascontiguousarray(bases, np.uint8); scalars = np.ascontiguousarray(scalars, np.uint64)
    out = np.zeros(64, np.uint8)
    dp = _p(np.ascontiguousarray(density, np.uint8)) if density is not None else None
    with _threads(threads):
        lib().orc_msm_g1(_p(bases), C.c_size_t(bases.shape[0]), _p(scalars), dp, C.c_size_t(scalars.shape[0]), _p(out))
    return out


def msm_g2(bases, scalars, density=None, threads=1):
    bases = np.ascontiguousarray(bases, np.uint8); scalars = np.ascontiguousarray(scalars, np.uint64)
    out = np.zeros(128, np.uint8)
    dp = _p(np.ascontiguousarray(density, np.uint8)) if density is not None else None
    with _threads(threads):
        lib().orc_msm_g2(_p(bases), C.c_size_t(bases.shape[0]), _p(scalars), dp, C.c_size_t(scalars.shape[0]), _p(out))
    return out


def g1_mul(p, k_mont):
    o = np.zeros(64, np.uint8); lib().orc_g1_mul(_p(np.ascontiguousarray(p, np.uint8)), _p(np.ascontiguousarray(k_mont, np.uint64)), _p(o)); return o


def g2_mul(p, k_mont):
    o = np.zeros(128, np.uint8); lib().orc_g2_mul(_p(np.ascontiguousarray(p, np.uint8)), _p(np.ascontiguousarray(k_mont, np.uint64)), _p(o)); return o


def g1_add(p, q):
    o = np.zeros(64, np.uint8); lib().orc_g1_add(_p(np.ascontiguousarray(p, np.uint8)), _p(np.ascontiguousarray(q, np.uint8)), _p(o)); return o


def g2_add(p, q):
    o = np.zeros(128, np.uint8); lib().orc_g2_add(_p(np.ascontiguousarray(p, np.uint8)), _p(np.ascontiguousarray(q, np.uint8)), _p(o)); return o


def g1_series(p, start_mont, n):
    o = np.zeros((n, 64), np.uint8); lib().orc_g1_series(_p(np.ascontiguousarray(p, np.uint8)), _p(np.ascontiguousarray(start_mont, np.uint64)), C.c_size_t(n), _p(o)); return o


def g2_series(p, start_mont, n):
    o = np.zeros((n, 128), np.uint8); lib().orc_g2_series(_p(np.ascontiguousarray(p, np.uint8)), _p(np.ascontiguousarray(start_mont, np.uint64)), C.c_size_t(n), _p(o)); return o


class Csr:
    """One R1CS matrix in CSR form; coefficients Montgomery Fr."""

    def __init__(self, ptr, col, val):
        self.ptr = np.ascontiguousarray(ptr, np.uint64)
        self.col = np.ascontiguousarray(col, np.uint32)
        self.val = np.ascontiguousarray(val, np.uint64).reshape(-1, 4)


class R1csC:
    def __init__(self, num_input, num_aux, A, B, Cm):
        self.num_input, self.num_aux = num_input, num_aux
        self.A, self.B, self.C = A, B, Cm
        self.num_gates = len(A.ptr) - 1
        s = OrcR1cs()
        s.num_input, s.num_aux, s.num_gates = num_input, num_aux, self.num_gates
        for nm, mtx in (('a', A), ('b', B), ('c', Cm)):
            setattr(s, nm + '_ptr', mtx.ptr.ctypes.data)
            setattr(s, nm + '_col', mtx.col.ctypes.data)
            setattr(s, nm + '_val', mtx.val.ctypes.data)
        self.struct = s

    @property
    def n_rows(self):
        return self.num_gates + self.num_input


def synthesize(cs, z):
    """z: (num_input+num_aux, 4) Montgomery.  Returns a, b, c (n,4) and the three density byte maps."""
    z = np.ascontiguousarray(z, np.uint64)
    n = cs.n_rows
    a = np.zeros((n, 4), np.uint64); b = np.zeros((n, 4), np.uint64); c = np.zeros((n, 4), np.uint64)
    a_aux = np.zeros(cs.num_aux, np.uint8); b_in = np.zeros(cs.num_input, np.uint8); b_aux = np.zeros(cs.num_aux, np.uint8)
    lib().orc_synthesize(C.byref(cs.struct), _p(z), _p(a), _p(b), _p(c), _p(a_aux), _p(b_in), _p(b_aux))
    return a, b, c, a_aux, b_in, b_aux


def synthesize_tiled(inst, copies, z):
    """synthesize() of fixtures.tile_r1cs(inst, copies) without materialising it (orc_synthesize_tiled)."""
    z = np.ascontiguousarray(z, np.uint64)
    nin, naux = 1 + copies * (inst.num_input - 1), copies * inst.num_aux
    n = copies * inst.num_gates + nin
    assert z.shape == (nin + naux, 4)
    a = np.zeros((n, 4), np.uint64); b = np.zeros((n, 4), np.uint64); c = np.zeros((n, 4), np.uint64)
    a_aux = np.zeros(naux, np.uint8); b_in = np.zeros(nin, np.uint8); b_aux = np.zeros(naux, np.uint8)
    lib().orc_synthesize_tiled(C.byref(inst.struct), C.c_uint32(copies), _p(z), _p(a), _p(b), _p(c), _p(a_aux), _p(b_in), _p(b_aux))
    return a, b, c, a_aux, b_in, b_aux


class Key:
    """Owns an orc_key; exposes numpy views of the key arrays."""

    def __init__(self, ptr):
        self.ptr = ptr
        k = ptr.contents
        self.m, self.num_input, self.num_aux = k.m, k.num_input, k.num_aux

        def view(addr, n, w):
            return np.ctypeslib.as_array(C.cast(addr, U8P), shape=(int(n) * w,)).reshape(int(n), w)
        self.h = view(k.h, k.n_h, 64); self.l = view(k.l, k.n_l, 64); self.a = view(k.a, k.n_a, 64)
        self.b_g1 = view(k.b_g1, k.n_b, 64); self.b_g2 = view(k.b_g2, k.n_b, 128)
        self.ic = view(k.ic, k.num_input, 64)
        for nm in ('alpha_g1', 'beta_g1', 'beta_g2', 'gamma_g2', 'delta_g1', 'delta_g2'):
            setattr(self, nm, np.frombuffer(bytes(getattr(k, nm)), dtype=np.uint8).copy())

    def __del__(self):
        try:
            lib().orc_key_free(self.ptr)
        except Exception:
            pass


class ArrayKey:
    """An orc_key whose arrays are numpy-owned (e.g. a synthetic key downloaded from the GPU)."""

    def __init__(self, m, num_input, num_aux, vk, h, l, a, b_g1, b_g2, ic=None):
        self._keep = []

        def arr(x, w):
            x = np.ascontiguousarray(x, np.uint8).reshape(-1, w)
            self._keep.append(x)
            return x
        k = OrcKey()
        k.m, k.num_input, k.num_aux = m, num_input, num_aux
        self.h, self.l, self.a = arr(h, 64), arr(l, 64), arr(a, 64)
        self.b_g1, self.b_g2 = arr(b_g1, 64), arr(b_g2, 128)
        self.ic = arr(ic if ic is not None else np.zeros((num_input, 64), np.uint8), 64)
        k.n_h, k.n_l, k.n_a, k.n_b = self.h.shape[0], self.l.shape[0], self.a.shape[0], self.b_g1.shape[0]
        for nm, w in (('alpha_g1', 64), ('beta_g1', 64), ('beta_g2', 128), ('gamma_g2', 128), ('delta_g1', 64), ('delta_g2', 128)):
            v = np.ascontiguousarray(vk.get(nm, np.zeros(w, np.uint8)), np.uint8).reshape(-1)
            C.memmove(getattr(k, nm), v.ctypes.data, w)
            setattr(self, nm, v.copy())
        for nm in ('ic', 'h', 'l', 'a', 'b_g1', 'b_g2'):
            setattr(k, nm, getattr(self, nm).ctypes.data)
        self.struct = k
        self.ptr = C.pointer(k)
        self.m, self.num_input, self.num_aux = m, num_input, num_aux


def setup(cs, tau, alpha, beta, gamma, delta, g1=None, g2=None):
    """Toxic waste as python ints (canonical)."""
    import bn254_ref as ref
    g1b = np.frombuffer(ref.g1_raw_le(ref.G1_GEN if g1 is None else g1), np.uint8).copy()
    g2b = np.frombuffer(ref.g2_raw_le(ref.G2_GEN if g2 is None else g2), np.uint8).copy()
    tw = [limbs(ref.to_mont(x % ref.R, ref.R)) for x in (tau, alpha, beta, gamma, delta)]
    ptr = lib().orc_setup(C.byref(cs.struct), *[_p(t) for t in tw], _p(g1b), _p(g2b))
    assert bool(ptr), 'orc_setup failed (domain too large?)'
    return Key(ptr)


def prove(key, a, b, c, z, a_aux, b_in, b_aux, r_mont, s_mont, want_msm=False, threads=1):
    """threads: bellman's Worker size.  1 = the serial prover (fawkes-crypto's configuration); more = bellman's multicore
    split restated (parallel_fft, one task per multiexp region) -- same proof bytes."""
    lib().orc_set_threads(C.c_int(int(threads)))
    a = np.ascontiguousarray(a, np.uint64); b = np.ascontiguousarray(b, np.uint64); c = np.ascontiguousarray(c, np.uint64)
    z = np.ascontiguousarray(z, np.uint64)
    out = np.zeros(256, np.uint8)
    msm = np.zeros(4 * 64 + 128, np.uint8)
    rc = lib().orc_prove(key.ptr, _p(a), _p(b), _p(c), C.c_uint64(a.shape[0]), _p(z),
                         _p(np.ascontiguousarray(a_aux, np.uint8)), _p(np.ascontiguousarray(b_in, np.uint8)),
                         _p(np.ascontiguousarray(b_aux, np.uint8)),
                         _p(np.ascontiguousarray(r_mont, np.uint64)), _p(np.ascontiguousarray(s_mont, np.uint64)),
                         _p(out), _p(msm))
    lib().orc_set_threads(C.c_int(1))
    if rc != 0:
        raise RuntimeError('orc_prove rc=%d' % rc)
    return (out, msm) if want_msm else out


def num_threads():
    return lib().orc_num_threads()
