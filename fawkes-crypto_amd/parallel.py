"""
Multi-GPU proving: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).

What shards: each of the five MSMs is a sum over independent (scalar, base)
units, so rank g keeps only a slice of every key array in its HBM (fk_key_load with shard_index /
shard_count / z_frac) and computes the partial sums of its slice.

Schedule (work-balanced, `prove_balanced`):
  * the witness MSMs L, A, B1, B2 depend only on the assignment z, not on the quotient, so ranks
    1..N-1 start on them immediately while rank 0 computes the quotient h (7 NTTs);
  * rank 0 then sends every peer ITS slice of h point-to-point (one xGMI link per peer, in parallel:
    m*32/N bytes each -- 128 MiB at 2^25 / 8 GPUs), and everybody runs the H MSM of its slice;
  * rank 0 holds a smaller share of the witness points (`plan_z_fractions`) so that all ranks finish
    the first phase at about the same time;
  * the exchange step -- north_star's "all-reduce of partial bucket sums": RCCL has no elliptic-curve
    reduction operator, so the reduction is ONE all-gather of the 384-byte partial results (4 x 64 B G1 +
    128 B G2 per rank) and a local fold of N points per MSM (fk_prove_assemble).  Latency-bound.

`prove_sharded*` is the simpler variant (every rank runs the quotient itself; no h traffic).

Distributed quotient (`quotient_distributed`, `prove_distributed_dev`; world a power of two <= 8): the transforms
themselves are cut across the ranks -- L = m/W-point transforms stay inside one GPU, one all-to-all per transform
(7 in total, m*32/W bytes per rank each: 128 MiB at 2^25 / 8 GPUs) moves the data between the two halves of every
transform, and rank g ends up with exactly the block of h coefficients whose bases its key shard holds.  Every rank
does 1/W of the quotient and 1/W of all five MSMs; this is the default for `bench.py --gpus N`.
"""
import os

import numpy as np

from . import api

# relative costs used by plan_z_fractions, measured on MI355X at 2^25 (profiles/r01_*):
COST_G2_POINT = 3.2          # one G2 scalar-mul ~ 3.2 G1 scalar-muls
COST_WITNESS_SCALAR = 0.62   # witness-like scalars (25 % zero, 25 % one) vs dense 254-bit scalars
COST_NTT_PER_ROW = 0.85      # quotient (7 NTTs) per row, in units of one dense G1 scalar-mul


def plan_z_fractions(world, m, num_aux, n_a, n_b):
    """[lo, hi) fractions of the l / a / b arrays per rank.  Rank 0 also computes the quotient, so it gets
    the share f0 that equalises  t_quotient + f0 * Z  with  (1 - f0) * Z / (world - 1)."""
    if world == 1:
        return [(0.0, 1.0)]
    z_cost = COST_WITNESS_SCALAR * (num_aux + n_a + n_b + COST_G2_POINT * n_b)
    ntt_cost = COST_NTT_PER_ROW * m
    f0 = (z_cost / (world - 1) - ntt_cost) / (z_cost + z_cost / (world - 1)) if z_cost > 0 else 0.0
    f0 = min(max(f0, 0.0), 1.0 / world)
    rest = (1.0 - f0) / (world - 1)
    out = [(0.0, f0)]
    for g in range(1, world):
        lo = f0 + (g - 1) * rest
        out.append((lo, 1.0 if g == world - 1 else lo + rest))
    return out


def all_gather_parts(local_part, group=None, device=None):
    """local_part: uint8[384] numpy.  Returns uint8[world, 384] numpy (same on every rank)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(local_part, np.uint8).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return np.stack([o.cpu().numpy() for o in out], axis=0)


# ------------------------------------------------------------------------------------------ the witness: PCIe once, then xGMI
def witness_pieces(nv, world):
    """(C, [(lo, hi)] per rank): rank g hands over the witness elements [g * C, (g + 1) * C), C = ceil(nv / world) (csrc/multi.hip uses the
    same cut); the slot is padded to world * C elements so that the all-gather's pieces are equal"""
    c = -(-int(nv) // int(world))
    return c, [(min(nv, g * c), min(nv, (g + 1) * c)) for g in range(world)]


class _DevBytes:
    """`nbytes` of device memory at `ptr` as seen through __cuda_array_interface__ (so that torch can wrap the library's witness slot)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {'shape': (int(nbytes),), 'typestr': '|u1', 'data': (int(ptr), False), 'version': 2, 'strides': None}


def witness_all_gather(ctx, slot, z_host, rank, world, group=None, device=None, force_collective=False):
    """Hands the witness of the next proof to ALL ranks with ONE trip over PCIe: this rank uploads only its piece (1 / world of
    `z_host`, pinned memory, over its own link) into witness slot `slot` and the ranks exchange the pieces with an all-gather IN PLACE in
    the slot -- `dist.all_gather_into_tensor` over RCCL / xGMI, issued on the library's copy stream so that it runs underneath the proof
    in flight ((world - 1) / world x 1.07 GB into every GPU at 2^25: the one RCCL collective of real size on the proving path; rounds
    1-4 had every rank pull the whole witness over PCIe).  The slot is complete where fk_witness_ptr(slot) makes the main stream wait.
    gloo (CPU tests, one-GPU rehearsals): the pieces are exchanged between the hosts and the foreign ones uploaded."""
    import torch
    import torch.distributed as dist
    z = z_host.reshape(-1, 4)
    nv = z.shape[0]
    if world == 1 and not force_collective:          # (force_collective: a one-rank rehearsal of the slot / copy-stream / in-place collective plumbing over real RCCL)
        ctx.witness_upload_async(slot, z)
        return dict(pcie_bytes=z.nbytes, gathered_bytes=0)
    c, pieces = witness_pieces(nv, world)
    cb = c * 32
    dptr, copy_stream = ctx.witness_slot(slot, cb * world)
    lo, hi = pieces[rank]
    ctx.witness_upload_part_async(slot, z[lo:hi], lo * 32)
    if dist.get_backend(group) == 'gloo' or device is None:
        mine = torch.zeros(cb, dtype=torch.uint8)
        mine[:(hi - lo) * 32] = torch.from_numpy(z[lo:hi].view(np.uint8).reshape(-1))
        full = torch.empty(cb * world, dtype=torch.uint8)
        dist.all_gather_into_tensor(full, mine, group=group)
        keep = getattr(ctx, '_witness_keep', None)
        if keep is None:
            keep = ctx._witness_keep = {}
        keep[slot] = full                       # the staged copies read it until the slot is complete
        fnp = full.numpy()
        for g, (a, b) in enumerate(pieces):
            if g != rank and b > a:
                ctx.witness_upload_part_async(slot, fnp[g * cb:g * cb + (b - a) * 32], a * 32)
    else:
        t = torch.as_tensor(_DevBytes(dptr, cb * world), device=device)
        st = torch.cuda.ExternalStream(copy_stream, device=device)
        with torch.cuda.stream(st):              # ProcessGroupNCCL orders its stream behind the copy stream (the upload), and the copy stream behind the collective
            dist.all_gather_into_tensor(t, t[rank * cb:(rank + 1) * cb], group=group, async_op=True).wait()
    ctx.witness_mark_ready(slot)
    return dict(pcie_bytes=(hi - lo) * 32, gathered_bytes=(nv - (hi - lo)) * 32)


def distribute_h(h_full, h_ranges, rank, recv_buf, group=None):
    """Rank 0 sends h[lo_g:hi_g] (32-byte elements) to rank g; every rank returns the tensor that holds ITS
    slice.  h_full: uint8 tensor of m*32 bytes on rank 0 (None elsewhere); recv_buf: uint8 tensor with room for
    this rank's slice (ranks > 0).  Point-to-point so that each peer's slice rides its own xGMI link."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    # gloo cannot move device tensors point-to-point: stage through the host (test configurations only)
    stage = dist.get_backend(group) == 'gloo'
    if rank == 0:
        ops, keep = [], []
        for g in range(1, world):
            lo, hi = h_ranges[g]
            if hi > lo:
                t = h_full[lo * 32:hi * 32]
                if stage and t.is_cuda:
                    t = t.cpu()
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, g, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        lo, hi = h_ranges[0]
        return h_full[lo * 32:hi * 32]
    lo, hi = h_ranges[rank]
    mine = recv_buf[:(hi - lo) * 32]
    if hi > lo:
        t = mine.cpu() if (stage and mine.is_cuda) else mine
        for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, t, 0, group)]):
            w.wait()
        if t is not mine:
            mine.copy_(t)
    return mine


def prove_balanced(rank, world, quotient_fn, z_fn, h_fn, assemble_fn, h_ranges, recv_buf, group=None, device=None, sync_fn=None):
    """Backend-agnostic orchestration of the balanced schedule (the GPU backend is `prove_balanced_dev`; the
    gloo test injects CPU stand-ins).
      quotient_fn() -> uint8 tensor with the full h (rank 0 only)
      z_fn()        -> uint8[384] numpy record of this rank's L, A, B1, B2 partials (H slot zero)
      h_fn(tensor)  -> uint8[64] numpy: H over this rank's h slice
      assemble_fn(parts uint8[world, 384]) -> proof bytes
    """
    if rank == 0:
        h_full = quotient_fn()
        if sync_fn:
            sync_fn()
        mine = distribute_h(h_full, h_ranges, 0, None, group)
        part = z_fn()
    else:
        part = z_fn()
        if sync_fn:
            sync_fn()
        mine = distribute_h(None, h_ranges, rank, recv_buf, group)
    if sync_fn:
        sync_fn()
    part = np.array(part, dtype=np.uint8, copy=True)
    part[:64] = h_fn(mine)
    parts = all_gather_parts(part, group=group, device=device)
    return assemble_fn(parts)


def prove_balanced_dev(ctx, key, rank, world, d_a, d_b, d_c, n, d_z, d_a_aux, d_b_in, d_b_aux, r, s, h_ranges, h_full_buf, recv_buf,
                       group=None, device=None, eval_fn=None):
    """GPU backend of `prove_balanced`.  h_full_buf: uint8 torch tensor of m*32 bytes on rank 0 (the quotient is
    written there); recv_buf: uint8 torch tensor for this rank's h slice (ranks > 0).  eval_fn (rank 0): produces
    a, b, c in d_a, d_b, d_c first (e.g. the device SpMV of a resident constraint system)."""
    import torch

    def quotient_fn():
        if eval_fn is not None:
            eval_fn()
        ctx.quotient_h_dev(d_a, d_b, d_c, n, h_full_buf.data_ptr())
        return h_full_buf

    def sync_fn():
        ctx.sync()
        torch.cuda.synchronize()

    return prove_balanced(
        rank, world, quotient_fn,
        lambda: ctx.prove_msms_z_dev(key, d_z, d_a_aux, d_b_in, d_b_aux),
        lambda t: ctx.prove_msm_h_dev(key, t.data_ptr() if t.numel() else 0),
        lambda parts: ctx.prove_assemble(key, parts, r, s),
        h_ranges, recv_buf, group=group, device=device, sync_fn=sync_fn)


def prove_sharded(ctx, key, a, b, c, z, a_aux, b_in, b_aux, r, s, group=None, device=None):
    """Simple variant, host buffers: every rank runs the quotient itself and its equal key shard."""
    part = ctx.prove_msms(key, a, b, c, z, a_aux, b_in, b_aux)
    parts = all_gather_parts(part, group=group, device=device)
    return ctx.prove_assemble(key, parts, r, s)


def prove_sharded_dev(ctx, key, d_a, d_b, d_c, n, d_z, d_a_aux, d_b_in, d_b_aux, r, s, group=None, device=None):
    """Simple variant, device-resident inputs."""
    part = ctx.prove_msms_dev(key, d_a, d_b, d_c, n, d_z, d_a_aux, d_b_in, d_b_aux)
    parts = all_gather_parts(part, group=group, device=device)
    return ctx.prove_assemble(key, parts, r, s)


# ------------------------------------------------------------------------------------------ distributed quotient
def log2_world(world):
    lw = world.bit_length() - 1
    if world < 1 or (1 << lw) != world or lw > 3:
        raise ValueError('the distributed quotient needs 1, 2, 4 or 8 ranks, got %d' % world)
    return lw


def torch_all_to_all(ctx, group=None):
    """all-to-all of lists of equally sized uint8 device tensors over torch.distributed (nccl = RCCL over xGMI).

    Returns a callable `a2a(dst, src)` (blocking form) that also carries the split form `h = a2a.begin(dst, src)` /
    `a2a.end(h)` the pipelined quotient uses.  Over RCCL nothing waits on the host: the library's main stream is wrapped as
    a torch stream (fk_stream -> torch.cuda.ExternalStream) and made current while the collective is issued, so
    ProcessGroupNCCL orders its own stream behind everything the library has queued (begin) and the library's stream
    behind the collective (end) with events.  Work queued on the library stream between begin and end runs while the
    bytes are on the links.  Under gloo (CPU tests, single-device dry runs) the tensors are staged through the host."""
    import torch
    import torch.distributed as dist
    stage = dist.get_backend(group) == 'gloo'
    lib_stream = None

    def begin(dst, src):
        nonlocal lib_stream
        if stage or not src[0].is_cuda:
            ctx.sync()
            for d, s_ in zip(dst, src):
                hs = s_.cpu()
                hd = torch.empty_like(hs)
                dist.all_to_all_single(hd, hs, group=group)
                d.copy_(hd)
            if dst[0].is_cuda:
                torch.cuda.synchronize()
            return None
        if lib_stream is None:
            lib_stream = torch.cuda.ExternalStream(ctx.stream_handle(), device=src[0].device)
        with torch.cuda.stream(lib_stream):
            return [dist.all_to_all_single(d, s_, group=group, async_op=True) for d, s_ in zip(dst, src)]

    def end(works):
        if works:
            with torch.cuda.stream(lib_stream):
                for w in works:
                    w.wait()            # stream-ordered for NCCL work: the library stream waits, the host does not

    def a2a(dst, src):
        end(begin(dst, src))
    a2a.begin, a2a.end = begin, end
    return a2a


def quotient_distributed(ctx, rank, world, d_full, n, log_m, send, recv, a2a, slices_in_send=False):
    """h = (A*B - C)/Z over `world` ranks.  d_full: device pointers of the three row-evaluation vectors a, b, c (n valid
    rows) from which this rank's cyclic slices are cut -- or, with slices_in_send, None: send[0..2] already hold the slices
    (fk_r1cs_eval_slice_dev: a rank of a resident constraint system evaluates only its own rows).  send, recv: 3 + 3 buffers of
    (m/world)*32 bytes with .data_ptr() (torch uint8 device tensors); a2a(dst_list, src_list): the exchange.  Returns the buffer
    that holds this rank's block h[rank*m/world, (rank+1)*m/world) (Montgomery, 32 B per coefficient).

    The three polynomials are independent until the pointwise step, so when the exchange has a split form (a2a.begin /
    a2a.end: torch_all_to_all over RCCL) they are pipelined: polynomial k's all-to-all is on the links while polynomial
    k+1's rank-local transform runs.  Same calls, same bytes as the plain order."""
    lw = log2_world(world)
    p = lambda t: t.data_ptr()
    begin = getattr(a2a, 'begin', None)
    end = getattr(a2a, 'end', None)
    if begin is None:
        begin = lambda dst, src: a2a(dst, src)
        end = lambda h: None
    # SIX transforms (csrc/ntt.hip: quotient_dev): c is subtracted in coefficient space, so it needs the inverse transform only
    h1 = []
    for k in range(3):                                   # ifft, first half; its exchange starts at once
        if not slices_in_send:
            ctx.dq_gather_dev(d_full[k], n, log_m, rank, lw, p(send[k]))
        ctx.dq_local_dev(p(send[k]), log_m, rank, lw, 0)
        h1.append(begin(recv[k:k + 1], send[k:k + 1]))
    h2 = [None, None]
    for k in range(2):                                   # a, b: ifft second half, coset shift, coset_fft first half
        end(h1[k])
        ctx.dq_cross_dev(p(recv[k]), log_m, rank, lw, 0)
        h2[k] = begin(send[k:k + 1], recv[k:k + 1])
    end(h1[2])
    ctx.dq_cross_dev(p(recv[2]), log_m, rank, lw, 2)     # c: ifft second half * 1 / (m Z(g)); block-cyclic coefficients stay in recv[2]
    for k in (1, 0):                                     # coset_fft, second half (b first: a is multiplied in place next)
        end(h2[k])
        ctx.dq_local_dev(p(send[k]), log_m, rank, lw, 1)
    ctx.dq_local_dev(p(send[0]), log_m, rank, lw, 3, p(send[1]))                 # a*b, icoset_fft first half
    a2a(recv[:1], send[:1])
    ctx.dq_cross_sub_dev(p(recv[0]), p(recv[2]), log_m, rank, lw)                # icoset_fft second half, / Z(g), - c's coefficients
    a2a(send[:1], recv[:1])                              # block-cyclic -> blocks (the key's h sharding)
    return send[0]


def prove_distributed_dev(ctx, key, rank, world, d_full, n, log_m, d_z, d_a_aux, d_b_in, d_b_aux, r, s, send, recv,
                          group=None, device=None, eval_fn=None, a2a=None, device_r1cs=None):
    """One proof over `world` GPUs with the evaluation of a, b, c, the quotient AND the five MSMs cut 1/world each.  key: this
    rank's equal shard (shard_index = rank, shard_count = world, no z fractions).  With a resident constraint system
    (device_r1cs) the rank evaluates only its cyclic row slice, straight into send[0..2] (d_full / eval_fn unused: no
    m-element vectors on a rank); otherwise eval_fn() fills d_full and the slices are cut out of it."""
    sliced = device_r1cs is not None and os.environ.get('FK_DIST_SLICED_EVAL', '1') != '0'
    if a2a is None:
        a2a = torch_all_to_all(ctx, group)

    def quotient():
        if sliced:
            ctx.r1cs_eval_slice_dev(device_r1cs, d_z, log_m, rank, log2_world(world), *[t.data_ptr() for t in send])
        elif eval_fn is not None:
            eval_fn()
        return quotient_distributed(ctx, rank, world, d_full, n, log_m, send, recv, a2a, slices_in_send=sliced)

    # The witness MSMs do not need the quotient, so they can be begun first and fill the GPU during the quotient's
    # all-to-all phases.  On ONE GPU the overlap measured neutral at 2^25 and 10 % slower at 2^20 / 2^22 (both sides are
    # VALU-bound and there is nothing to wait for); from 4 ranks on the per-rank transforms are short next to the eight
    # exchanges, so the default there is to overlap.  FK_OVERLAP_WITNESS=1 / 0 forces either.
    ov = os.environ.get('FK_OVERLAP_WITNESS', '')
    if ov == '1' or (ov != '0' and world >= 4):
        if device_r1cs is not None:
            ctx.prove_msms_z_begin_r1cs_dev(key, device_r1cs, d_z)
        else:
            ctx.prove_msms_z_begin_dev(key, d_z, d_a_aux, d_b_in, d_b_aux)
        h_blk = quotient()
        part = ctx.prove_msms_finish_dev(key, h_blk.data_ptr())
    else:
        h_blk = quotient()
        if device_r1cs is not None:      # resident constraint system: its query index lists replace the density compaction
            part = ctx.prove_msms_hz_r1cs_dev(key, device_r1cs, h_blk.data_ptr(), d_z)
        else:
            part = ctx.prove_msms_hz_dev(key, h_blk.data_ptr(), d_z, d_a_aux, d_b_in, d_b_aux)
    parts = all_gather_parts(part, group=group, device=device)
    return ctx.prove_assemble(key, parts, r, s)
