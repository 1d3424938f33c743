"""world_size-2 gloo test of the multi-GPU path (runs on CPU): each rank owns one shard of the key, the
partial MSM results travel through fawkes_crypto_amd.parallel.all_gather_parts (the one collective of the
path) and are folded by the product's fk_prove_assemble.  The per-shard MSMs themselves need a GPU, so the
oracle stands in for them HERE ONLY (as the checker's input); the result must equal the oracle's
single-process proof byte for byte."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import bn254_ref as ref
    import c_oracle as co
    import fixtures as fx
    from fawkes_crypto_amd import api, parallel
    from helpers import params_from_oracle_key, TOXIC
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        cs, z_in, z_aux = ref.random_r1cs(5150, 90, 3, 100)
        csr = fx.r1cs_to_csr(cs)
        key = co.setup(csr, **TOXIC)
        z = fx.witness_mont(z_in, z_aux)
        a, b, c, aa, bi, ba = co.synthesize(csr, z)
        r, s = fx.mont_fr(0xabc), fx.mont_fr(0xdef)
        want = co.prove(key, a, b, c, z, aa, bi, ba, r, s)
        # this rank's shard of every MSM (same slicing rule as fk_key_load: fk_shard_range)
        h = co.quotient_h(a, b, c)
        nin = cs.num_input
        sa = np.concatenate([z[:nin], z[nin:][aa != 0]])
        sb = np.concatenate([z[:nin][bi != 0], z[nin:][ba != 0]])
        part = np.zeros(api.FK_MSM_RESULT_BYTES, np.uint8)
        for off, bases, scalars, fn in ((0, key.h, h, co.msm_g1), (64, key.l, z[nin:], co.msm_g1), (128, key.a, sa, co.msm_g1),
                                        (192, key.b_g1, sb, co.msm_g1), (256, key.b_g2, sb, co.msm_g2)):
            lo, hi = (api.h_shard_range if off == 0 else api.shard_range)(len(bases), rank, world)
            res = fn(np.array(bases[lo:hi]), scalars[lo:hi]) if hi > lo else np.zeros(128 if off == 256 else 64, np.uint8)
            part[off:off + len(res)] = res
        parts = parallel.all_gather_parts(part)
        assert parts.shape == (world, api.FK_MSM_RESULT_BYTES)
        assert np.array_equal(parts[rank], part)
        vk = api.HostVk(params_from_oracle_key(key))
        got = api.assemble(vk.handle, parts, r, s)
        ok = got.tobytes() == want.tobytes()

        # ---- the work-balanced schedule (parallel.prove_balanced) with CPU stand-ins for the GPU MSMs:
        # rank 0 computes h and ships slices point-to-point, witness arrays are split by plan_z_fractions
        import torch
        fracs = parallel.plan_z_fractions(world, key.m, cs.num_aux, len(key.a), len(key.b_g1))
        h_ranges = [api.h_shard_range(key.m - 1, g, world) for g in range(world)]

        def frac_range(n_, lo_, hi_):
            a_ = int(n_ * lo_ + 0.5); b_ = n_ if hi_ >= 1.0 else int(n_ * hi_ + 0.5)
            return a_, max(a_, b_)

        def z_fn():
            rec = np.zeros(api.FK_MSM_RESULT_BYTES, np.uint8)
            for off, bases, scalars, fn in ((64, key.l, z[nin:], co.msm_g1), (128, key.a, sa, co.msm_g1),
                                            (192, key.b_g1, sb, co.msm_g1), (256, key.b_g2, sb, co.msm_g2)):
                lo_, hi_ = frac_range(len(bases), *fracs[rank])
                if hi_ > lo_:
                    res = fn(np.array(bases[lo_:hi_]), scalars[lo_:hi_])
                    rec[off:off + len(res)] = res
            return rec

        def h_fn(t):
            lo_, hi_ = h_ranges[rank]
            sl = np.frombuffer(t.numpy().tobytes(), np.uint64).reshape(-1, 4)
            assert sl.shape[0] == hi_ - lo_
            return co.msm_g1(np.array(key.h[lo_:hi_]), sl) if hi_ > lo_ else np.zeros(64, np.uint8)

        def quotient_fn():
            full = np.zeros((key.m, 4), np.uint64); full[:key.m - 1] = h
            return torch.from_numpy(full.view(np.uint8).reshape(-1).copy())

        recv = torch.zeros(max(h_ranges[rank][1] - h_ranges[rank][0], 1) * 32, dtype=torch.uint8)
        got2 = parallel.prove_balanced(rank, world, quotient_fn, z_fn, h_fn, lambda parts_: api.assemble(vk.handle, parts_, r, s),
                                       h_ranges, recv)
        ok = ok and got2.tobytes() == want.tobytes()
        q.put((rank, ok, got.tobytes().hex()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_prove_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results), results
    assert len({hx for _, _, hx in results}) == 1      # every rank assembled the same proof


# ------------------------------------------------------------------------------------------ distributed quotient
class _CpuDq:
    """CPU stand-in (python big ints, direct O(L^2) transforms) for the three rank-local device calls of the distributed
    quotient, so that parallel.quotient_distributed + parallel.torch_all_to_all run over real gloo ranks here.
    Buffers are uint8 torch tensors holding Montgomery limbs, looked up by data_ptr()."""

    def __init__(self, ref, co, bufs):
        self.ref, self.co = ref, co
        self.bufs = {t.data_ptr(): t for t in bufs}

    def sync(self):
        pass

    def _get(self, ptr):
        R = self.ref.R
        arr = np.frombuffer(self.bufs[ptr].numpy().tobytes(), np.uint64).reshape(-1, 4)
        return [self.ref.from_mont(x, R) for x in self.co.ints(arr)]

    def _put(self, ptr, vals):
        import torch
        R = self.ref.R
        raw = self.co.limbs_arr([self.ref.to_mont(v % R, R) for v in vals]).tobytes()
        self.bufs[ptr].copy_(torch.frombuffer(bytearray(raw), dtype=torch.uint8))

    def dq_gather_dev(self, d_full, n, log_m, rank, log_w, d_local):
        full = self._get(d_full)
        L = 1 << (log_m - log_w)
        self._put(d_local, [full[rank + (j << log_w)] if rank + (j << log_w) < n else 0 for j in range(L)])

    def dq_local_dev(self, d_x, log_m, rank, log_w, stage, d_xb=0, d_xc=0):
        R = self.ref.R
        x = self._get(d_x)
        L = len(x)
        if stage == 2:
            xb, xc = self._get(d_xb), self._get(d_xc)
            x = [(u * v - w) % R for u, v, w in zip(x, xb, xc)]
        if stage == 3:                                   # six-transform form: a * b only
            x = [u * v % R for u, v in zip(x, self._get(d_xb))]
        wm = self.ref.omega_for(1 << log_m)
        wl = pow(wm, 1 << log_w, R)
        if stage != 1:
            wl, wm = pow(wl, -1, R), pow(wm, -1, R)
        out = [sum(x[j] * pow(wl, j * k, R) for j in range(L)) % R for k in range(L)]
        if stage != 1:
            out = [o * pow(wm, rank * k, R) % R for k, o in enumerate(out)]
        self._put(d_x, out)

    def dq_cross_sub_dev(self, d_buf, d_sub, log_m, rank, log_w):
        self.dq_cross_dev(d_buf, log_m, rank, log_w, 1)
        R = self.ref.R
        self._put(d_buf, [(u - v) % R for u, v in zip(self._get(d_buf), self._get(d_sub))])

    def dq_cross_dev(self, d_buf, log_m, rank, log_w, mode):
        R = self.ref.R
        W, m = 1 << log_w, 1 << log_m
        L, Lc = m >> log_w, m >> (2 * log_w)
        buf = self._get(d_buf)
        wm = self.ref.omega_for(m)
        ww = pow(wm, L, R)
        g = 7
        zinv = pow(pow(g, m, R) - 1, -1, R)
        minv = pow(m, -1, R)
        out = list(buf)
        for t in range(Lc):
            k2 = rank * Lc + t
            v = [buf[j * Lc + t] for j in range(W)]
            u = [sum(v[j] * pow(ww, -j * k1, R) for j in range(W)) % R for k1 in range(W)]
            for k1 in range(W):
                i = k2 + k1 * L
                u[k1] = u[k1] * (pow(g, i, R) * minv if mode == 0 else (minv * zinv if mode == 2 else pow(g, -i, R) * minv * zinv)) % R
            if mode == 0:
                u = [sum(u[j] * pow(ww, j * k1, R) for j in range(W)) * pow(wm, k2 * k1, R) % R for k1 in range(W)]
            for k1 in range(W):
                out[k1 * Lc + t] = u[k1]
        self._put(d_buf, out)


def _dq_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import bn254_ref as ref
    import c_oracle as co
    from fawkes_crypto_amd import parallel
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        log_m, n = 7, 121
        m, L = 1 << log_m, (1 << log_m) // world
        rng = ref.Lcg(77)
        a, b, c = ([rng.below(ref.R) for _ in range(n)] for _ in range(3))
        want = ref.quotient_h(a, b, c, m)                 # m - 1 coefficients, python oracle
        full = []
        for v in (a, b, c):
            raw = co.limbs_arr([ref.to_mont(x, ref.R) for x in v] + [0] * (m - n)).tobytes()
            full.append(torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone())
        send = [torch.zeros(L * 32, dtype=torch.uint8) for _ in range(3)]
        recv = [torch.zeros(L * 32, dtype=torch.uint8) for _ in range(3)]
        cpu = _CpuDq(ref, co, full + send + recv)
        blk = parallel.quotient_distributed(cpu, rank, world, [t.data_ptr() for t in full], n, log_m, send, recv,
                                            parallel.torch_all_to_all(cpu))
        got = cpu._get(blk.data_ptr())
        hi = min((rank + 1) * L, m - 1)
        q.put((rank, got[:hi - rank * L] == want[rank * L:hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 4])
def test_distributed_quotient_gloo(world):
    """the exchange order of parallel.quotient_distributed over real torch.distributed ranks (gloo all_to_all_single):
    every rank must end with its contiguous block of the python oracle's quotient coefficients"""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dq_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results


# ------------------------------------------------------------------------------------------ the witness: PCIe once, then an all-gather
class _SlotRecorder:
    """stand-in for the three witness-slot calls of api.Context (fk_witness_slot / _upload_part_async / _mark_ready): records what a rank
    writes into its slot, so that parallel.witness_all_gather runs over real gloo ranks here"""

    def __init__(self):
        self.buf, self.ready, self.uploaded = None, False, 0

    def witness_slot(self, slot, total_bytes):
        self.buf = np.full(total_bytes, 0xEE, np.uint8)
        return 0x1000, 0

    def witness_upload_part_async(self, slot, part, offset):
        b = np.ascontiguousarray(part).view(np.uint8).reshape(-1)
        assert offset + b.size <= self.buf.size
        self.buf[offset:offset + b.size] = b
        self.uploaded += b.size

    def witness_upload_async(self, slot, z):
        self.buf = np.ascontiguousarray(z).view(np.uint8).reshape(-1).copy()
        self.uploaded += self.buf.size

    def witness_mark_ready(self, slot):
        self.ready = True


def _wit_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from fawkes_crypto_amd import parallel
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ok = True
        for nv in (1001, 64, world - 1 if world > 1 else 1, 4096):       # ragged, tiny (an empty last piece) and even sizes
            z = np.random.default_rng(7 + nv).integers(0, 1 << 63, size=(nv, 4), dtype=np.uint64)      # the same witness on every rank's host
            rec = _SlotRecorder()
            tr = parallel.witness_all_gather(rec, nv & 1, z, rank, world)
            c, pieces = parallel.witness_pieces(nv, world)
            lo, hi = pieces[rank]
            ok = ok and rec.ready and rec.buf[:nv * 32].tobytes() == z.tobytes() and rec.buf.size == c * world * 32
            ok = ok and tr['pcie_bytes'] == (hi - lo) * 32 and tr['gathered_bytes'] == (nv - (hi - lo)) * 32
            ok = ok and sum(b - a for a, b in pieces) == nv and all(b - a <= c for a, b in pieces)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3, 4])
def test_witness_all_gather_gloo(world):
    """parallel.witness_all_gather over real torch.distributed ranks: every rank hands over only its piece of the witness (1 / world of the
    bytes over its own PCIe link) and ends with the WHOLE witness in its slot, ragged sizes included"""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wit_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results


# ------------------------------------------------------------------------------------------ first-contact preflight (fawkes-crypto_amd/preflight.py)
def _preflight_worker(rank, world, port, q):
    for p in (ROOT,):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_PORT'] = str(port)
    from fawkes_crypto_amd import preflight
    try:
        q.put((rank, preflight.run(rank, rank, world, 'gloo', True, 1000, 1 << 10, limit_s=60.0, token='test_%d' % port)))
    except BaseException as e:      # noqa: BLE001
        q.put((rank, repr(e)))


def test_preflight_two_ranks_gloo_no_gpu():
    """The preflight's host side with two ranks and NO GPU: each rank starts its child process, the children rendezvous over their own file store,
    agree on a verdict (gloo all-reduce) and every rank reads the same decision.  Backend gloo: RCCL is not exercised (and says so); rank 0's library
    part fails for want of a device and is REPORTED, not raised -- the block must come back whatever happens in the child."""
    import multiprocessing as mp
    mpc = mp.get_context('spawn')
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_preflight_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    for r in (0, 1):
        b = res[r]
        assert isinstance(b, dict), b
        assert b['ran'] is True and b['world'] == 2 and b['control_plane'] == 'gloo ok' and b['all_ranks_ok'] is True, b
        assert b['rccl'] is None and b['rccl_world_size'] is None
        assert b['decision']['backend'] == 'gloo' and b['decision']['host_events'] is False and 'RCCL not exercised' in b['decision']['why']
        assert b['seconds'] > 0
    lib = res[0]['library']
    assert lib is not None and lib['ok'] is False and 'error' in lib          # no GPU here: reported
    assert res[1]['library'] is None


def test_preflight_child_that_dies_is_a_verdict_not_an_exception(tmp_path, monkeypatch):
    """a child that cannot even start its control plane (world 2 announced, only one rank present: the store never fills) ends in a 'failed' verdict
    within the time limit, and a requested nccl backend falls back to gloo"""
    from fawkes_crypto_amd import preflight
    monkeypatch.setenv('MASTER_PORT', str(_free_port()))
    b = preflight.run(0, 0, 2, 'nccl', True, 100, 1 << 8, limit_s=3.0, token='lonely_%d' % os.getpid())
    assert b['all_ranks_ok'] is False and b['decision']['backend'] == 'gloo' and 'gloo' in b['decision']['why']
