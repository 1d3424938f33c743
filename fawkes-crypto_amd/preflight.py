"""
First contact with a multi-GPU node, BEFORE any key is built.

Nothing in this repository has ever run on more than one physical GPU: RCCL with more than one rank, cross-device event waits and
peer DMA between distinct devices are correct by construction and rehearsed on one GPU only.  A first run on a real node must therefore
(1) find out in seconds, not after a 12 GB key has been set up, whether those three things work there, (2) say what it found in the
benchmark's JSON line, and (3) carry on over the documented fallback instead of ending with rc != 0:

  RCCL collectives fail or hang      -> the data path runs over gloo (host-staged exchanges: parallel.py stages device tensors itself)
  in-stream cross-device event wait  -> the library switches itself to host-side waits (fk_multi_preflight; the parent exports
                                        FK_MULTI_HOST_EVENTS=1 for the contexts it creates afterwards)
  peer access refused for a pair     -> reported (fk_multi_topology); the runtime stages those copies, nothing to switch

The checks run in a CHILD process per rank, started by the rank's bench.py process before that process has made any GPU call (never an
exec of a process that holds the GPU), with a time limit -- a hung collective cannot take the benchmark with it.  The children rendezvous
over a file store of their own and finish with a gloo all-reduce of their verdicts, so every rank reads the same decision.

What a child does (`--child`):
  * gloo process group over the file store (control plane; if even that fails the verdict is "failed");
  * backend nccl (= RCCL): `all_gather_into_tensor` of the witness piece a rank hands over per proof ((variables / N) x 32 bytes),
    `all_to_all_single` of one transform's exchange (m x 32 / N bytes per rank), `all_gather` of the 384-byte partial sums -- the three
    collectives of parallel.py at the proof's real sizes, every received byte checked against its sender's pattern, timed;
  * rank 0: the library's own preflight (fk_init_devices over the node's devices -> fk_multi_topology, fk_multi_preflight: a verified
    64 MiB pull per ordered pair behind a cross-device event wait, GB/s per pair).
"""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pattern(torch, n, sender, device):
    """n bytes that name their sender and their position (period 251 x 256 is irrelevant: a misrouted or shifted piece differs)"""
    i = torch.arange(n, dtype=torch.int32, device=device)          # (n < 2^31: a witness piece is at most ~1 GB; int32 products wrap alike on both sides)
    return ((i * 131 + (i >> 8) * 7 + (sender * 37 + 11)) & 255).to(torch.uint8)


def _rccl_checks(torch, dist, group, rank, world, device, piece_bytes, chunk_bytes, out):
    ev = lambda: torch.cuda.Event(enable_timing=True)
    # -- all_gather_into_tensor, in place in one buffer: parallel.witness_all_gather
    full = torch.zeros(piece_bytes * world, dtype=torch.uint8, device=device)
    full[rank * piece_bytes:(rank + 1) * piece_bytes] = _pattern(torch, piece_bytes, rank, device)
    ms = None
    for rep in range(2):
        e0, e1 = ev(), ev()
        e0.record()
        dist.all_gather_into_tensor(full, full[rank * piece_bytes:(rank + 1) * piece_bytes], group=group)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    ok = all(bool(torch.equal(full[g * piece_bytes:(g + 1) * piece_bytes], _pattern(torch, piece_bytes, g, device))) for g in range(world))
    out['all_gather'] = dict(bytes_per_rank=piece_bytes, ms=round(ms, 3), GBps_into_each_rank=round(piece_bytes * (world - 1) / max(ms, 1e-6) / 1e6, 2), verified=ok,
                             what='dist.all_gather_into_tensor in place: the witness hand-over of one proof (parallel.witness_all_gather)')
    if not ok:
        raise RuntimeError('all_gather_into_tensor delivered wrong bytes')
    del full
    # -- all_to_all_single: one exchange of the distributed quotient (chunk g of my send -> rank g)
    cb = chunk_bytes
    send = torch.cat([_pattern(torch, cb, rank * world + g, device) for g in range(world)])
    recv = torch.zeros(cb * world, dtype=torch.uint8, device=device)
    for rep in range(2):
        e0, e1 = ev(), ev()
        e0.record()
        dist.all_to_all_single(recv, send, group=group)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    ok = all(bool(torch.equal(recv[g * cb:(g + 1) * cb], _pattern(torch, cb, g * world + rank, device))) for g in range(world))
    out['all_to_all'] = dict(bytes_per_pair=cb, bytes_per_rank=cb * world, ms=round(ms, 3), GBps_out_of_each_rank=round(cb * (world - 1) / max(ms, 1e-6) / 1e6, 2), verified=ok,
                             what='dist.all_to_all_single: one of the seven exchanges of a distributed quotient (parallel.torch_all_to_all)')
    if not ok:
        raise RuntimeError('all_to_all_single delivered wrong bytes')
    del send, recv
    # -- all_gather of the 384-byte partial sums
    mine = _pattern(torch, 384, rank, device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    t0 = time.perf_counter()
    dist.all_gather(parts, mine, group=group)
    torch.cuda.synchronize()
    ok = all(bool(torch.equal(parts[g], _pattern(torch, 384, g, device))) for g in range(world))
    out['all_gather_384'] = dict(ms=round((time.perf_counter() - t0) * 1e3, 3), verified=ok, what='the fold of the partial sums (parallel.all_gather_parts)')
    if not ok:
        raise RuntimeError('all_gather (384 B) delivered wrong bytes')


def child_main(argv):
    """python preflight.py --child OUT STORE BACKEND PIECE_BYTES CHUNK_BYTES SAME_DEVICE LIMIT_S"""
    out_path, store_path, backend = argv[0], argv[1], argv[2]
    piece_bytes, chunk_bytes, same_device, limit_s = int(argv[3]), int(argv[4]), argv[5] == '1', float(argv[6])
    rank, world, local_rank = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('LOCAL_RANK', '0'))
    if same_device:
        local_rank = 0
    res = {'rank': rank, 'world': world, 'backend_requested': backend, 'control_plane': 'failed', 'rccl': None, 'library': None, 'all_ranks_ok': False}
    t_start = time.time()

    def write():
        res['seconds'] = round(time.time() - t_start, 2)
        tmp = out_path + '.tmp'
        with open(tmp, 'w') as f:
            json.dump(res, f)
        os.replace(tmp, out_path)

    write()
    import datetime
    import torch
    import torch.distributed as dist
    store = dist.FileStore(store_path, world)
    dist.init_process_group('gloo', store=store, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=limit_s))
    res['control_plane'] = 'gloo ok'
    ok_local = True
    if backend == 'nccl':
        rc = {'ok': False, 'world_size': world}
        res['rccl'] = rc

        def work():
            try:
                torch.cuda.set_device(local_rank)
                dev = torch.device('cuda', local_rank)
                grp = dist.new_group(ranks=list(range(world)), backend='nccl', timeout=datetime.timedelta(seconds=limit_s))
                t0 = time.time()
                _rccl_checks(torch, dist, grp, rank, world, dev, piece_bytes, chunk_bytes, rc)
                rc['seconds'] = round(time.time() - t0, 2)
                rc['ok'] = True
            except BaseException as e:        # noqa: BLE001 -- everything is a verdict here
                rc['error'] = '%s: %s' % (type(e).__name__, str(e)[:500])
        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(limit_s)
        if th.is_alive():
            rc['error'] = 'no answer within %.0f s (a hung collective)' % limit_s
        ok_local = bool(rc['ok'])
        write()
    # every rank learns whether ALL ranks passed
    flag = torch.tensor([1 if ok_local else 0], dtype=torch.int32)
    try:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        res['all_ranks_ok'] = bool(flag.item() == 1)
    except BaseException as e:                # noqa: BLE001
        res['control_plane'] = 'gloo all_reduce failed: %s' % str(e)[:300]
    write()
    if rank == 0:
        # the library's own first contact: topology and one verified pull per ordered pair (one process driving all devices)
        lib = {'ok': False}
        res['library'] = lib
        try:
            if ROOT not in sys.path:
                sys.path.insert(0, ROOT)
            import fawkes_crypto_amd as fk
            ndev = torch.cuda.device_count()
            n = max(world, 2) if same_device or ndev < max(world, 2) else world
            devices = [0] * n if (same_device or ndev < n) else list(range(n))
            lib['devices'] = devices
            lib['devices_are'] = 'distinct GPUs' if len(set(devices)) == len(devices) else 'ONE GPU named %d times (a rehearsal of the code path: no link is measured)' % n
            mc = fk.MultiContext(devices)
            try:
                lib['topology'] = mc.topology()
                lib['init_note'] = mc.note()
                pf = mc.preflight(64 << 20)
                lib.update(pull_bytes=pf['bytes'], pull_GBps=pf['gbps'], pull_status=pf['status'], host_events=pf['host_events'], note=pf['note'], ok=pf['ok'],
                           what='fk_multi_preflight: per ordered pair (into row, from column) a 64 MiB pull on the exchange stream behind the other '
                                'rank\'s event, timed with HIP events, bytes compared on the host')
            finally:
                mc.close()
        except BaseException as e:            # noqa: BLE001
            lib['error'] = '%s: %s' % (type(e).__name__, str(e)[:500])
        write()
    try:
        dist.barrier()
    except BaseException:                     # noqa: BLE001
        pass
    write()
    sys.stdout.flush()
    os._exit(0)            # no teardown of a communicator that may be wedged


def run(rank, local_rank, world, backend, same_device, nv, m, limit_s=150.0, token=None):
    """Called by a rank's bench.py process BEFORE it touches the GPU.  Starts the child, waits (kills it by PID on the time limit), reads the
    verdict.  Returns the `preflight` block: what was found and `decision` = {'backend': 'nccl' | 'gloo', 'host_events': bool, 'why': str}."""
    import tempfile
    t0 = time.time()
    token = token or '%s_%s' % (os.environ.get('MASTER_PORT', '0'), os.getppid() if world > 1 else os.getpid())
    tmpdir = tempfile.gettempdir()
    store_path = os.path.join(tmpdir, 'fk_preflight_store_%s' % token)
    out_path = os.path.join(tmpdir, 'fk_preflight_%s_rank%d.json' % (token, rank))
    piece = max(32, -(-int(nv) // world) * 32)
    chunk = max(32, (int(m) * 32 // world) // world)
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, os.path.abspath(__file__), '--child', out_path, store_path, backend, str(piece), str(chunk), '1' if same_device else '0', str(limit_s)]
    block = {'ran': True, 'world': world, 'backend_requested': backend,
             'is': 'first-contact checks in a child process per rank, before any key is built (fawkes-crypto_amd/preflight.py): the proof\'s three collectives at '
                   'their real sizes with contents verified, and the library\'s peer-access table + one verified 64 MiB pull per ordered pair'}
    try:
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        try:
            log, _ = proc.communicate(timeout=limit_s * 2.2 + 120)
            block['child_rc'] = proc.returncode
        except subprocess.TimeoutExpired:
            proc.kill()                      # the exact PID we started
            log, _ = proc.communicate()
            block['child_rc'] = 'killed after %.0f s' % (limit_s * 2.2 + 120)
        res = None
        if os.path.exists(out_path):
            try:
                res = json.load(open(out_path))
            except Exception:                # noqa: BLE001
                res = None
        if res is None:
            res = {'control_plane': 'no verdict written', 'all_ranks_ok': False}
            block['child_log_tail'] = (log or '')[-1500:]
        block.update({k: res.get(k) for k in ('control_plane', 'rccl', 'library', 'all_ranks_ok')})
        block['rccl_world_size'] = world if backend == 'nccl' else None
        if block.get('child_rc') not in (0,) and 'child_log_tail' not in block:
            block['child_log_tail'] = (log or '')[-800:]
    finally:
        for p_ in (out_path, out_path + '.tmp'):
            try:
                os.remove(p_)
            except OSError:
                pass
        if rank == 0:
            try:
                os.remove(store_path)
            except OSError:
                pass
    use = backend
    why = 'every check passed'
    if backend == 'nccl' and not block.get('all_ranks_ok'):
        use, why = 'gloo', 'RCCL checks did not pass on every rank (%s): the exchanges of this run are staged through the hosts over gloo' % (
            ((block.get('rccl') or {}).get('error')) or block.get('control_plane') or 'see child_log_tail')
    elif backend != 'nccl':
        why = 'backend %s requested: RCCL not exercised' % backend
    lib = block.get('library') or {}
    host_events = bool(lib.get('host_events'))
    if host_events:
        why += '; cross-device in-stream event waits failed: host-side waits (FK_MULTI_HOST_EVENTS=1)'
    block['decision'] = {'backend': use, 'host_events': host_events, 'why': why}
    block['seconds'] = round(time.time() - t0, 2)
    return block


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child_main(sys.argv[2:])
    else:
        raise SystemExit('usage: preflight.py --child OUT STORE BACKEND PIECE_BYTES CHUNK_BYTES SAME_DEVICE LIMIT_S (started by bench.py)')
