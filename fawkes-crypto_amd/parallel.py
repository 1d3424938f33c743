"""
Multi-GPU proving: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).

What shards: each of the five MSMs is a sum over independent (scalar, base)
units, so rank g keeps only the [g/N, (g+1)/N) slice of every key array in its HBM (fk_key_load with
shard_index/shard_count) and computes the partial sums of its slice.  The quotient (7 NTTs, ~10 % of
the work) is computed redundantly on every rank so that no polynomial data crosses xGMI.

The exchange step: north_star's "all-reduce of partial bucket sums".  RCCL has no elliptic-curve
reduction operator, so the reduction is realised as ONE all-gather of the 384-byte partial results
(4 x 64 B G1 + 128 B G2 per rank) followed by a local fold of N points per MSM (fk_prove_assemble).
The payload is N x 384 B -- latency-bound, a single collective per proof.
"""
import numpy as np

from . import api


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


def prove_sharded(ctx, key, a, b, c, z, a_aux, b_in, b_aux, r, s, group=None, device=None):
    """Host-buffer variant: every rank passes the same witness-derived inputs, holds its key shard, and
    every rank returns the same 256-byte proof."""
    part = ctx.prove_msms(key, a, b, c, z, a_aux, b_in, b_aux)
    parts = all_gather_parts(part, group=group, device=device)
    return ctx.prove_assemble(key, parts, r, s)


def prove_sharded_dev(ctx, key, d_a, d_b, d_c, n, d_z, d_a_aux, d_b_in, d_b_aux, r, s, group=None, device=None):
    """Device-resident variant used by bench.py."""
    part = ctx.prove_msms_dev(key, d_a, d_b, d_c, n, d_z, d_a_aux, d_b_in, d_b_aux)
    parts = all_gather_parts(part, group=group, device=device)
    return ctx.prove_assemble(key, parts, r, s)
