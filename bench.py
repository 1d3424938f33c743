#!/usr/bin/env python3
"""
bench.py -- Groth16 proofs/sec on MI355X through the C ABI (include/fawkes_hip.h).

One "step" = one complete Groth16 proof for a synthetic 2^LOG2-constraint BN254 R1CS (default 2^25, the
size BASELINE.json's metric is quoted on): the 7-NTT quotient, the four G1 MSMs (H, L, A, B1), the G2
MSM (B2) and the proof assembly -- exactly the work behind `create_random_proof`
(/root/reference/fawkes-crypto/src/backend/bellman_groth16/prover.rs:80).  Inputs (a, b, c row
evaluations, the assignment z, density maps, proving key) are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; every rank holds a slice of each key array (MSM sharded by points).  Rank 0
computes the quotient while the others start on the witness MSMs, h slices go point-to-point over xGMI,
ONE all-gather (RCCL) of 384 bytes per rank exchanges the partial sums and the proof is folded locally --
strong scaling of a single proof (fawkes-crypto_amd/parallel.py: prove_balanced).

The CPU oracle (oracle/) appears here only as the timed `cpu_baseline` and as a live parity check of
that same sample; it is never the thing measured as `value`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 achievable
G1_BYTES_PER_SCALAR_MUL = 96     # 64 B affine base + 32 B scalar
G2_BYTES_PER_SCALAR_MUL = 160


def cpu_baseline(ctx, log2_sample):
    """Times the C oracle (bellman's algorithm restated, single thread = the reference's configured
    worker) on a bounded sample: the FULL prover on a 2^log2_sample instance with the same
    shape ratios and scalar distribution, with key points generated on the GPU and downloaded.  The same
    instance is proved on the GPU and the 256 proof bytes must match (live parity check)."""
    import c_oracle as co
    import fixtures as fx
    import fawkes_crypto_amd as fk
    m = 1 << log2_sample
    v_in, v_aux = 2, m - 2
    rng = np.random.default_rng(12345)
    dens_a = (rng.random(v_aux) < 0.6).astype(np.uint8)
    dens_bi = np.ones(v_in, np.uint8)
    dens_ba = (rng.random(v_aux) < 0.6).astype(np.uint8)
    n_a = v_in + int(dens_a.sum())
    n_b = int(dens_bi.sum()) + int(dens_ba.sum())

    def gen_g1(n, seed):
        d = ctx.dev_alloc(max(n, 1) * 64); ctx.gen_points_g1_dev(d, n, seed)
        out = ctx.download(d, n * 64).reshape(-1, 64); ctx.dev_free(d); return out

    def gen_g2(n, seed):
        d = ctx.dev_alloc(max(n, 1) * 128); ctx.gen_points_g2_dev(d, n, seed)
        out = ctx.download(d, n * 128).reshape(-1, 128); ctx.dev_free(d); return out

    def gen_fr(n, seed, kind):
        d = ctx.dev_alloc(n * 32); ctx.gen_scalars_dev(d, n, seed, kind)
        out = ctx.download(d, n * 32, np.uint64).reshape(-1, 4); ctx.dev_free(d); return out
    vk1, vk2 = gen_g1(3, 901), gen_g2(2, 902)
    arrays = dict(m=m, num_input=v_in, num_aux=v_aux, alpha_g1=vk1[0], beta_g1=vk1[1], delta_g1=vk1[2],
                  beta_g2=vk2[0], delta_g2=vk2[1], h=gen_g1(m - 1, 1), l=gen_g1(v_aux, 2), a=gen_g1(n_a, 3),
                  b_g1=gen_g1(n_b, 4), b_g2=gen_g2(n_b, 5))
    a, b, c = gen_fr(m, 21, 0), gen_fr(m, 22, 0), gen_fr(m, 23, 0)
    z = gen_fr(v_in + v_aux, 24, 1)
    r, s = fx.mont_fr(0x1234567), fx.mont_fr(0x89abcdef)
    okey = co.ArrayKey(m, v_in, v_aux, dict(alpha_g1=vk1[0], beta_g1=vk1[1], delta_g1=vk1[2], beta_g2=vk2[0], delta_g2=vk2[1]),
                       arrays['h'], arrays['l'], arrays['a'], arrays['b_g1'], arrays['b_g2'])
    t0 = time.time()
    want = co.prove(okey, a, b, c, z, dens_a, dens_bi, dens_ba, r, s)
    cpu_s = time.time() - t0
    dk = ctx.load_key(fk.Parameters(arrays))
    got = ctx.prove_raw(dk, a, b, c, z, dens_a, dens_bi, dens_ba, r, s)
    dk.free()
    if got.tobytes() != want.tobytes():
        raise AssertionError('bench parity check failed: HIP proof != oracle proof on the CPU-baseline sample')
    return cpu_s, m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--log2n', type=int, default=25, help='log2 of the constraint count (rows handed to the prover)')
    ap.add_argument('--cpu-log2n', type=int, default=18, help='size of the CPU-baseline sample instance')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend ('nccl' = RCCL; 'gloo' only for single-GPU dry runs of the N>1 code path with FK_BENCH_SAME_DEVICE=1)")
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run (one process per GPU)' % args.gpus)
        raise SystemExit('--gpus %d != WORLD_SIZE %d' % (args.gpus, world))

    import torch
    import fawkes_crypto_amd as fk
    from fawkes_crypto_amd import parallel
    if os.environ.get('FK_BENCH_SAME_DEVICE') == '1':
        local_rank = 0                      # dry run: all ranks share GPU 0 (needs --backend gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    comm_dev = dev if args.backend == 'nccl' else None
    ctx = fk.Context(local_rank)

    # ---------------------------------------------------------------- workload (SURVEY section 8d, config 4 shape)
    m = 1 << args.log2n
    v_in = 2
    v_aux = m - v_in
    n = m                                   # rows = #gates + num_input = m exactly
    dens_frac = 0.6                         # share of aux variables occurring in A- resp. B-side LCs (assumption)
    g = torch.Generator(device='cpu'); g.manual_seed(7)
    dens_a = (torch.rand(v_aux, generator=g) < dens_frac).to(torch.uint8)
    dens_ba = (torch.rand(v_aux, generator=g) < dens_frac).to(torch.uint8)
    dens_bi = torch.ones(v_in, dtype=torch.uint8)
    n_a = v_in + int(dens_a.sum()); n_b = int(dens_bi.sum()) + int(dens_ba.sum())
    d_dens_a, d_dens_bi, d_dens_ba = dens_a.to(dev), dens_bi.to(dev), dens_ba.to(dev)

    # multi-GPU: h sharded equally, witness arrays by the work-balanced fractions (rank 0 also runs the quotient)
    fracs = parallel.plan_z_fractions(world, m, v_aux, n_a, n_b)
    key = ctx.synthetic_key(m, v_in, v_aux, n_a, n_b, seed=2026, shard_index=rank, shard_count=world,
                            z_frac=fracs[rank] if world > 1 else (0.0, 0.0))
    h_ranges = [fk.api.shard_range(m - 1, g, world) for g in range(world)]
    h_full_buf = torch.empty(m * 32, dtype=torch.uint8, device=dev) if (world > 1 and rank == 0) else None
    recv_buf = torch.empty(max(h_ranges[rank][1] - h_ranges[rank][0], 1) * 32, dtype=torch.uint8, device=dev) if (world > 1 and rank > 0) else None
    # pristine inputs + working copies (the prover consumes a, b, c as scratch)
    nbytes = m * 32
    pristine = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(3)]
    work = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(3)]
    d_z = torch.empty((v_in + v_aux) * 32, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    for i, t in enumerate(pristine):
        ctx.gen_scalars_dev(t.data_ptr(), m, 100 + i, 0)          # uniform row evaluations
    ctx.gen_scalars_dev(d_z.data_ptr(), v_in + v_aux, 200, 1)     # witness-like assignment (half in {0,1})
    import fixtures as fx
    r, s = fx.mont_fr(0xA11CE), fx.mont_fr(0xB0B)

    def step():
        if world == 1 or rank == 0:       # only the rank that runs the quotient consumes a, b, c
            for w_, p_ in zip(work, pristine):
                ctx.dev_copy(w_.data_ptr(), p_.data_ptr(), nbytes)
        if world == 1:
            return ctx.prove_dev(key, work[0].data_ptr(), work[1].data_ptr(), work[2].data_ptr(), n, d_z.data_ptr(),
                                 d_dens_a.data_ptr(), d_dens_bi.data_ptr(), d_dens_ba.data_ptr(), r, s)
        return parallel.prove_balanced_dev(ctx, key, rank, world, work[0].data_ptr(), work[1].data_ptr(), work[2].data_ptr(), n,
                                           d_z.data_ptr(), d_dens_a.data_ptr(), d_dens_bi.data_ptr(), d_dens_ba.data_ptr(),
                                           r, s, h_ranges, h_full_buf, recv_buf, device=comm_dev)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    proofs = []
    for _ in range(args.warmup):
        proofs.append(step().tobytes())
    barrier()
    ctx.stats_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proofs.append(step().tobytes())
    barrier()
    elapsed = time.perf_counter() - t0
    stats = ctx.stats()
    if len(set(proofs)) != 1 or proofs[0] == bytes(256):
        raise AssertionError('bench: proofs differ between steps (non-deterministic result)')
    if world > 1:
        import torch.distributed as dist
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        sec_per_step = elapsed / args.steps
        msm_units = (m - 1) + v_aux + n_a + 2 * n_b       # scalar-muls per proof: H, L, A, B1 (G1) and B2 (G2)
        acc = stats['acc_g1']
        # dominant kernel: msm_accumulate_kernel<Fq>; achieved = algorithmic bytes / its HIP-event time
        achieved = (acc['units'] * G1_BYTES_PER_SCALAR_MUL) / (acc['ms'] * 1e-3) / 1e9 if acc['ms'] > 0 else 0.0
        # HBM traffic of the dominant kernel from the committed PMC pass (profiles/), if it was taken at this size
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic_bench_2p25.json')))
            if pmc['log2n'] == args.log2n and world == 1 and acc['ms'] > 0:
                dk = pmc['dominant_kernel']
                per_proof = dk['fetch_bytes_per_proof_raw'] + dk['write_bytes_per_proof']
                traffic = per_proof / (acc['ms'] / args.steps * 1e-3) / 1e9      # GB/s at this run's kernel time
        except Exception:
            traffic = None
        out = {
            'metric': 'Groth16 proofs/sec (BN254, 2^%d constraints)' % args.log2n,
            'value': args.steps / elapsed,
            'unit': 'proofs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': sec_per_step * 1e3,
            'higher_is_better': True,
            'scaling': 'strong',
            'vs_baseline': None,
            'dtype': 'u32',
            'data': 'synthetic',
            'config': {'workload': 'synthetic rollup-shape R1CS, 2^%d rows (BASELINE configs[3] shape): 7-NTT quotient + '
                                   'G1 MSMs H/L/A/B1 + G2 MSM B2 + assembly, inputs and key resident in HBM' % args.log2n,
                       'log2_constraints': args.log2n, 'num_input': v_in, 'num_aux': v_aux,
                       'density_a_aux': dens_frac, 'density_b_aux': dens_frac,
                       'scalar_distribution': 'a,b,c uniform; assignment witness-like (50% in {0,1})',
                       'parallelism': 'msm-shard%d%s' % (world, '+balanced-quotient' if world > 1 else '')},
            'msm_scalar_muls_per_sec': msm_units / sec_per_step,
            'roofline': {
                'bound': 'hbm', 'kernel': 'msm_accumulate_kernel<Fq> (G1 bucket accumulation)',
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                'traffic': traffic,
                'traffic_source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/r01_pmc_traffic_bench_2p25.json '
                                  '(raw FETCH_SIZE: gather width uncalibrated, see file)' if traffic else None,
                'launches': acc['launches'], 'avg_launch_ms': acc['ms'] / max(acc['launches'], 1),
                'algorithmic_bytes_per_scalar_mul': G1_BYTES_PER_SCALAR_MUL,
                'note': 'MSM is 256-bit modular integer work: VALU-bound, not HBM-bound; see DESIGN.md',
            },
            'kernel_ms_per_step': {
                'msm_accumulate_g1': stats['acc_g1']['ms'] / args.steps,
                'msm_accumulate_g2': stats['acc_g2']['ms'] / args.steps,
                'ntt_passes': stats['ntt']['ms'] / args.steps,
                'ntt_algorithmic_GBps': (stats['ntt']['units'] * 64) / (stats['ntt']['ms'] * 1e-3) / 1e9 if stats['ntt']['ms'] > 0 else 0.0,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            cpu_s, cpu_m = cpu_baseline(ctx, args.cpu_log2n)
            scale = m / cpu_m
            out['cpu_baseline'] = {
                'value': 1.0 / (cpu_s * scale), 'unit': 'proofs/s', 'cores': 1, 'kind': 'port',
                'sample': 'oracle/groth16_oracle.c (bellman algorithm restated, 1 thread) proving a 2^%d-row instance of '
                          'the same shape in %.2f s; scaled linearly by %d to 2^%d rows (optimistic for the CPU: NTT is '
                          'n log n); the GPU proof of the same sample matched byte for byte' % (args.cpu_log2n, cpu_s, scale, args.log2n),
                'sample_seconds': cpu_s,
            }
        print(json.dumps(out), flush=True)

    key.free()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
