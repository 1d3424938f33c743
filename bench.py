#!/usr/bin/env python3
"""
bench.py -- Groth16 proofs/sec on MI355X through the C ABI (include/fawkes_hip.h).

One "step" = one complete Groth16 proof for a satisfiable rollup-shaped BN254 R1CS of 2^LOG2 rows (default
2^25, the size BASELINE.json's metric is quoted on), starting from the WITNESS VECTOR: device SpMV
(a = Az, b = Bz, c = Cz), the 7-NTT quotient, the four G1 MSMs (H, L, A, B1), the G2 MSM (B2) and the
proof assembly -- the work behind `create_random_proof`
(/root/reference/fawkes-crypto/src/backend/bellman_groth16/prover.rs:80), including bellman's `synthesize`
evaluation (mod.rs:92-99).  The witness, the constraint system and the proving key are resident in HBM before
the timed region.  The key is a VALID key (fk_setup, fixed toxic waste), so the proof produced in the timed
region is checked afterwards with the Groth16 pairing equation.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, strong scaling of a single proof: every rank holds 1/N of each key array (MSM sharded
by points) and computes 1/N of the quotient -- the seven transforms are cut across the ranks with one all-to-all
(RCCL over xGMI) each -- then ONE all-gather of 384 bytes per rank exchanges the partial MSM sums and the proof is
folded locally (fawkes-crypto_amd/parallel.py: prove_distributed_dev).  With 2 ranks, or a rank count that is not a
power of two, rank 0 computes the quotient while the other ranks start on the witness MSMs, and h slices travel point
to point (prove_balanced_dev); FK_DIST_QUOTIENT=1 / 0 forces either schedule.

The CPU oracle (oracle/) appears here only in the `cpu_baseline` leg: the timed CPU baseline, a live parity
check of that same sample, and the pairing check of the benchmarked proof; it is never the thing measured.
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
FR_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MONT_R = (1 << 256) % FR_MODULUS
TOXIC = dict(tau=0x1f2e3d4c5b6a79880123456789abcdef0fedcba987654321, alpha=0xa11ce, beta=0xb0b, gamma=0xc0ffee, delta=0xdec0de)


def mont(x):
    return np.frombuffer(((x % FR_MODULUS) * MONT_R % FR_MODULUS).to_bytes(32, 'little'), dtype=np.uint64).copy()


def build_workload(ctx, fk, log2n, seed=2026):
    """Satisfiable rollup-shaped R1CS with m = 2^log2n rows (gates + inputs = m exactly) and its witness.
    40 % boolean constraints b*(b-1)=0 (bit decompositions dominate fawkes circuits, circuit/bitify.rs),
    10 % linear gates, 50 % product gates over random field elements.  Coefficients are 1 and -1 only.
    Witness: 20 % zeros, 20 % ones, 60 % dense 254-bit values.  Returns (R1cs, z (nv,4) uint64 Montgomery)."""
    m = 1 << log2n
    v_in = 2
    G = m - v_in
    nb = int(0.4 * G); nf = max(int(0.1 * G), 1); npr = G - nb - nf
    v_aux = G
    rng = np.random.default_rng(seed)
    one, minus_one = mont(1), mont(-1)
    # witness
    z = np.zeros((v_in + v_aux, 4), np.uint64)
    z[0] = one
    z[1] = mont(0x5eed5eed5eed)
    bits = rng.integers(0, 2, nb, dtype=np.uint8)
    z[v_in:v_in + nb][bits == 1] = one
    free = rng.integers(0, 1 << 63, size=(nf, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(nf, 4), dtype=np.uint64)
    free[:, 3] &= np.uint64((1 << 60) - 1)            # Montgomery images < 2^252 < r
    z[v_in + nb:v_in + nb + nf] = free
    i_k = rng.integers(0, nf, npr, dtype=np.int64)
    l_k = rng.integers(0, nf, npr, dtype=np.int64)
    CH = 1 << 22
    for lo in range(0, npr, CH):
        hi = min(lo + CH, npr)
        z[v_in + nb + nf + lo:v_in + nb + nf + hi] = ctx.fr_mul_batch(free[i_k[lo:hi]], free[l_k[lo:hi]])
    # matrices (variable index: Input(i) -> i, Aux(j) -> v_in + j)
    a_col = np.concatenate([v_in + np.arange(nb, dtype=np.uint32), (v_in + nb + np.arange(nf)).astype(np.uint32),
                            (v_in + nb + i_k).astype(np.uint32)])
    a_ptr = np.arange(G + 1, dtype=np.uint64)
    b_len = np.concatenate([np.full(nb, 2, np.uint64), np.ones(nf + npr, np.uint64)])
    b_ptr = np.zeros(G + 1, np.uint64); b_ptr[1:] = np.cumsum(b_len)
    b_col = np.zeros(int(b_ptr[-1]), np.uint32)
    b_val = np.tile(one, (int(b_ptr[-1]), 1))
    b_col[0:2 * nb:2] = v_in + np.arange(nb, dtype=np.uint32)       # b
    b_col[1:2 * nb:2] = 0                                            # - ONE
    b_val[1:2 * nb:2] = minus_one
    b_col[2 * nb:2 * nb + nf] = 0                                    # linear gates: * ONE
    b_col[2 * nb + nf:] = (v_in + nb + l_k).astype(np.uint32)
    c_len = np.concatenate([np.zeros(nb, np.uint64), np.ones(nf + npr, np.uint64)])
    c_ptr = np.zeros(G + 1, np.uint64); c_ptr[1:] = np.cumsum(c_len)
    c_col = np.concatenate([(v_in + nb + np.arange(nf)).astype(np.uint32), (v_in + nb + nf + np.arange(npr)).astype(np.uint32)])
    r1cs = fk.R1cs(v_in, v_aux, (a_ptr, a_col, None), (b_ptr, b_col, b_val), (c_ptr, c_col, None))
    return r1cs, z


def build_workload_dense(ctx, fk, log2n, terms, seed=2026):
    """Same size and witness mix as build_workload, but with LONG linear combinations, like the circuits fawkes-crypto
    is used for (poseidon merkle: 26 matrix terms per gate, eddsa: 133): every product gate is
    (sum of `terms` variables) * (sum of `terms` variables) = new variable, the operands drawn from the boolean and free
    variables.  The new variables' values are computed with the library itself (device SpMV + batched products)."""
    import torch
    m = 1 << log2n
    v_in = 2
    G = m - v_in
    nb = int(0.4 * G); nf = max(int(0.1 * G), 1); npr = G - nb - nf
    v_aux = G
    rng = np.random.default_rng(seed)
    one, minus_one = mont(1), mont(-1)
    z = np.zeros((v_in + v_aux, 4), np.uint64)
    z[0] = one
    z[1] = mont(0x5eed5eed5eed)
    bits = rng.integers(0, 2, nb, dtype=np.uint8)
    z[v_in:v_in + nb][bits == 1] = one
    free = rng.integers(0, 1 << 63, size=(nf, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(nf, 4), dtype=np.uint64)
    free[:, 3] &= np.uint64((1 << 60) - 1)
    z[v_in + nb:v_in + nb + nf] = free
    # boolean gates b*(b-1)=0 and linear gates as in build_workload; product gates with `terms` operands per side
    src = v_in + rng.integers(0, nb + nf, size=(2, npr, terms), dtype=np.int64)          # operands: booleans and free values
    a_col = np.concatenate([v_in + np.arange(nb, dtype=np.uint32), (v_in + nb + np.arange(nf)).astype(np.uint32), src[0].reshape(-1).astype(np.uint32)])
    a_len = np.concatenate([np.ones(nb + nf, np.uint64), np.full(npr, terms, np.uint64)])
    a_ptr = np.zeros(G + 1, np.uint64); a_ptr[1:] = np.cumsum(a_len)
    b_len = np.concatenate([np.full(nb, 2, np.uint64), np.ones(nf, np.uint64), np.full(npr, terms, np.uint64)])
    b_ptr = np.zeros(G + 1, np.uint64); b_ptr[1:] = np.cumsum(b_len)
    b_col = np.zeros(int(b_ptr[-1]), np.uint32)
    b_col[0:2 * nb:2] = v_in + np.arange(nb, dtype=np.uint32)
    b_col[1:2 * nb:2] = 0
    b_col[2 * nb:2 * nb + nf] = 0
    b_col[2 * nb + nf:] = src[1].reshape(-1).astype(np.uint32)
    del src
    # B needs the coefficient -1 on the boolean gates only: dictionary of two values, everything else ONE
    b_val = np.tile(one, (int(b_ptr[-1]), 1)) if terms * npr < (1 << 27) else None
    if b_val is not None:
        b_val[1:2 * nb:2] = minus_one
    c_len = np.concatenate([np.zeros(nb, np.uint64), np.ones(nf + npr, np.uint64)])
    c_ptr = np.zeros(G + 1, np.uint64); c_ptr[1:] = np.cumsum(c_len)
    c_col = np.concatenate([(v_in + nb + np.arange(nf)).astype(np.uint32), (v_in + nb + nf + np.arange(npr)).astype(np.uint32)])
    if b_val is None:
        # too many terms for a 32-byte-per-term host array: express b*(b-1) as b*b = b instead (all coefficients ONE)
        b_len = np.concatenate([np.ones(nb + nf, np.uint64), np.full(npr, terms, np.uint64)])
        b_ptr = np.zeros(G + 1, np.uint64); b_ptr[1:] = np.cumsum(b_len)
        b_col = np.concatenate([b_col[0:2 * nb:2], b_col[2 * nb:]])
        c_len = np.ones(G, np.uint64)
        c_ptr = np.zeros(G + 1, np.uint64); c_ptr[1:] = np.cumsum(c_len)
        c_col = np.concatenate([v_in + np.arange(nb, dtype=np.uint32), c_col])
    r1cs = fk.R1cs(v_in, v_aux, (a_ptr, a_col, None), (b_ptr, b_col, b_val), (c_ptr, c_col, None))
    # product gates' outputs: c = (A z)(B z), evaluated by the device SpMV on the partial witness
    dr = ctx.load_r1cs(r1cs)
    dbuf = [ctx.dev_alloc(m * 32) for _ in range(3)]
    d_z = ctx.dev_alloc(z.nbytes)
    ctx.upload(d_z, z)
    ctx.r1cs_eval_dev(dr, d_z, *dbuf)
    lo = nb + nf
    CH = 1 << 22
    for off in range(0, npr, CH):
        hi = min(off + CH, npr)
        av = ctx.download(dbuf[0] + (lo + off) * 32, (hi - off) * 32, np.uint64).reshape(-1, 4)
        bv = ctx.download(dbuf[1] + (lo + off) * 32, (hi - off) * 32, np.uint64).reshape(-1, 4)
        z[v_in + lo + off:v_in + lo + hi] = ctx.fr_mul_batch(av, bv)
    for p_ in dbuf + [d_z]:
        ctx.dev_free(p_)
    dr.free()
    return r1cs, z


def cpu_baseline_leg(ctx, fk, log2_sample, full):
    """(1) times the C oracle (bellman's algorithm restated, single thread = the reference's configured worker,
    SURVEY fact 3) proving a 2^log2_sample instance of the same workload family; (2) the GPU proof of that same
    sample must match byte for byte; (3) the benchmarked full-size proof must satisfy the pairing equation."""
    import bn254_ref as ref
    import c_oracle as co
    r1cs, z = build_workload(ctx, fk, log2_sample, seed=77)
    tox = {k: mont(v) for k, v in TOXIC.items()}
    dk, vk = ctx.setup(r1cs, **tox)
    okey = co.ArrayKey(1 << log2_sample, r1cs.num_input, r1cs.num_aux, vk, dk.download('h'), dk.download('l'), dk.download('a'),
                       dk.download('b_g1'), dk.download('b_g2'), ic=vk['ic'])
    a, b, c, aa, bi, ba = fk.api.synthesize(r1cs, z)
    r, s = mont(0x1234567), mont(0x89abcdef)
    t0 = time.time()
    want = co.prove(okey, a, b, c, z, aa, bi, ba, r, s)
    cpu_s = time.time() - t0
    dr = ctx.load_r1cs(r1cs)
    got = ctx.prove_witness(dk, dr, z, r, s)
    dr.free(); dk.free()
    if got.tobytes() != want.tobytes():
        raise AssertionError('bench parity check failed: HIP proof != oracle proof on the CPU-baseline sample')
    verified = pairing_check(*full) if full is not None else None
    return cpu_s, 1 << log2_sample, verified


def pairing_check(vk_full, z_in1, proof):
    """the benchmarked proof must satisfy the Groth16 pairing equation (checker: oracle/bn254_ref.py verifier)"""
    import bn254_ref as ref
    g1 = lambda b_: ref.g1_from_raw_le(bytes(b_))
    g2 = lambda b_: ref.g2_from_raw_le(bytes(b_))
    pk = dict(alpha_g1=g1(vk_full['alpha_g1']), beta_g2=g2(vk_full['beta_g2']), gamma_g2=g2(vk_full['gamma_g2']),
              delta_g2=g2(vk_full['delta_g2']), ic=[g1(x.tobytes()) for x in vk_full['ic']])
    Rinv = pow(MONT_R, -1, FR_MODULUS)
    pub = [int.from_bytes(z_in1.tobytes(), 'little') * Rinv % FR_MODULUS]
    if not ref.verify(pk, pub, ref.proof_from_borsh(proof)):
        raise AssertionError('bench: the benchmarked proof does not satisfy the Groth16 pairing equation')
    return True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--lc-terms', type=int, default=0, help='operands per side of every product gate (default: the 1-2 term rollup shape); '
                                                             'e.g. 16 gives the long linear combinations real circuits have')
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--log2n', type=int, default=25, help='log2 of the row count handed to the prover')
    ap.add_argument('--cpu-log2n', type=int, default=20, help='size of the CPU-baseline sample instance')
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
    # FK_BENCH_REHEARSE=1 with --gpus 1: run the multi-GPU code path (process group, all-to-all, all-gather, distributed
    # quotient with one rank) on a single GPU -- a rehearsal of the N > 1 plumbing over real RCCL, not a benchmark mode
    multi = world > 1 or os.environ.get('FK_BENCH_REHEARSE') == '1'
    if multi:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    comm_dev = dev if args.backend == 'nccl' else None
    ctx = fk.Context(local_rank)

    # ---------------------------------------------------------------- workload: constraint system, witness, valid key
    t_prep = time.time()
    m = 1 << args.log2n
    r1cs, z = build_workload(ctx, fk, args.log2n) if args.lc_terms <= 1 else build_workload_dense(ctx, fk, args.log2n, args.lc_terms)
    v_in, v_aux, n = r1cs.num_input, r1cs.num_aux, r1cs.n_rows
    assert n == m
    dr = ctx.load_r1cs(r1cs)
    info = dr.info()
    n_a, n_b = info['n_a'], info['n_b']
    # N > 1, N a power of two: quotient and MSMs cut 1/N each (parallel.prove_distributed_dev); otherwise (or with
    # FK_DIST_QUOTIENT=0) rank 0 computes the quotient and ships h slices (parallel.prove_balanced_dev)
    # Default by rank count: at N = 2 every all-to-all of the distributed quotient moves m*32/4 bytes over ONE xGMI link
    # (256 MiB at 2^25, ~3.5 ms, eight times per proof), more than the quotient it saves; from N = 4 on the chunks are
    # small and spread over N-1 links.  FK_DIST_QUOTIENT=1 / 0 forces either schedule.
    dq_env = os.environ.get('FK_DIST_QUOTIENT', '')
    dq_ok = multi and (world & (world - 1)) == 0 and world <= 8
    dist_q = dq_ok and (dq_env == '1' or (dq_env != '0' and (world >= 4 or world == 1)))
    fracs = parallel.plan_z_fractions(world, m, v_aux, n_a, n_b)
    tox = {k: mont(v) for k, v in TOXIC.items()}
    key, vk = ctx.setup(r1cs, shard_index=rank, shard_count=world, z_frac=fracs[rank] if (world > 1 and not dist_q) else (0.0, 0.0), **tox)
    pre_levels = key.precomputed()        # fixed-base window levels per key array (0 = none: FK_MSM_PRECOMP=0 or HBM short)
    d_z = torch.empty((v_in + v_aux) * 32, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    ctx.upload(d_z.data_ptr(), z)
    z_in1 = z[1].copy()
    zeros = int((~z.any(axis=1)).sum()); ones = int((z == mont(1)).all(axis=1).sum())
    z_host = z if not multi else None      # kept for the PCIe-inclusive measurement after the timed region (N = 1)
    del z
    r, s = mont(0xA11CE), mont(0xB0B)
    d_dens = dr.density_ptrs()
    if dist_q:
        work = [torch.empty(m * 32, dtype=torch.uint8, device=dev) for _ in range(3)]
        send = [torch.empty(m // world * 32, dtype=torch.uint8, device=dev) for _ in range(3)]
        recv = [torch.empty(m // world * 32, dtype=torch.uint8, device=dev) for _ in range(3)]
        a2a = parallel.torch_all_to_all(ctx)
    elif multi:
        h_ranges = [fk.api.h_shard_range(m - 1, g, world) for g in range(world)]
        h_full_buf = torch.empty(m * 32, dtype=torch.uint8, device=dev) if rank == 0 else None
        recv_buf = torch.empty(max(h_ranges[rank][1] - h_ranges[rank][0], 1) * 32, dtype=torch.uint8, device=dev) if rank > 0 else None
        work = [torch.empty(m * 32, dtype=torch.uint8, device=dev) for _ in range(3)] if rank == 0 else [None] * 3
    prep_s = time.time() - t_prep

    def step():
        if not multi:
            return ctx.prove_witness_dev(key, dr, d_z.data_ptr(), r, s)
        wp = [w_.data_ptr() if w_ is not None else 0 for w_ in work]
        if dist_q:
            return parallel.prove_distributed_dev(
                ctx, key, rank, world, wp, n, args.log2n, d_z.data_ptr(), d_dens[0], d_dens[1], d_dens[2], r, s, send, recv,
                device=comm_dev, a2a=a2a, device_r1cs=dr, eval_fn=lambda: ctx.r1cs_eval_dev(dr, d_z.data_ptr(), wp[0], wp[1], wp[2]))
        return parallel.prove_balanced_dev(
            ctx, key, rank, world, wp[0], wp[1], wp[2], n, d_z.data_ptr(), d_dens[0], d_dens[1], d_dens[2], r, s,
            h_ranges, h_full_buf, recv_buf, device=comm_dev,
            eval_fn=(lambda: ctx.r1cs_eval_dev(dr, d_z.data_ptr(), wp[0], wp[1], wp[2])) if rank == 0 else None)

    def barrier():
        if multi:
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
    if multi:
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
        traffic = None
        try:   # HBM traffic of the dominant kernel from the committed PMC pass (profiles/), if taken at this size
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic_bench_2p25.json')))
            if pmc['log2n'] == args.log2n and world == 1 and acc['ms'] > 0:
                dkk = pmc['dominant_kernel']
                per_pt = (dkk['fetch_bytes_per_proof_raw'] + dkk['write_bytes_per_proof']) / dkk['points_per_proof']
                traffic = per_pt * (acc['units'] / args.steps) / (acc['ms'] / args.steps * 1e-3) / 1e9
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
            'config': {'workload': 'satisfiable rollup-shape R1CS, 2^%d rows (BASELINE configs[3] shape), valid key: witness -> '
                                   'device SpMV + 7-NTT quotient + G1 MSMs H/L/A/B1 + G2 MSM B2 + assembly; witness, constraint system '
                                   'and key resident in HBM' % args.log2n,
                       'log2_constraints': args.log2n, 'num_input': v_in, 'num_aux': v_aux,
                       'gates': '40% boolean b*(b-1)=0, 10% linear, 50% products' + (' of %d-term sums' % args.lc_terms if args.lc_terms > 1 else ''),
                       'nnz': list(info['nnz']),
                       'a_query_points': n_a, 'b_query_points': n_b,
                       'msm_fixed_base_levels': pre_levels,
                       'witness': '%.0f%% zeros, %.0f%% ones, rest dense 254-bit' % (100.0 * zeros / (v_in + v_aux), 100.0 * ones / (v_in + v_aux)),
                       'parallelism': 'msm-shard%d%s' % (world, '' if not multi else
                                                         '+distributed-quotient (8 all-to-all per proof)' if dist_q else '+balanced-quotient')},
            'msm_scalar_muls_per_sec': msm_units / sec_per_step,
            'roofline': {
                'bound': 'hbm', 'kernel': '%s<Fq> (G1 bucket accumulation)' % ('msm_accumulate_merged_kernel' if pre_levels.get('h') else 'msm_accumulate_kernel'),
                'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                'traffic': traffic,
                'traffic_source': ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/r01_pmc_traffic_bench_2p25.json '
                                   '(bytes per point from that pass x this run\'s points; raw FETCH_SIZE, gather width uncalibrated)') if traffic else None,
                'launches': acc['launches'], 'avg_launch_ms': acc['ms'] / max(acc['launches'], 1),
                'algorithmic_bytes_per_scalar_mul': G1_BYTES_PER_SCALAR_MUL,
                'note': 'MSM is 256-bit modular integer work: VALU-bound, not HBM-bound; see DESIGN.md',
            },
            'kernel_ms_per_step': {
                'msm_accumulate_g1': stats['acc_g1']['ms'] / args.steps,
                'msm_accumulate_g2': stats['acc_g2']['ms'] / args.steps,
                'ntt_passes': stats['ntt']['ms'] / args.steps,
                # one launch = one pass over 2^log2n elements, 64 B (read + write) each; a transform is ceil(log2n / 9) passes
                'ntt_algorithmic_GBps': (stats['ntt']['units'] * 64) / (stats['ntt']['ms'] * 1e-3) / 1e9 if stats['ntt']['ms'] > 0 else 0.0,
                'ntt_GBps_is': 'data moved per pass (64 B per element per pass)',
            },
            'prep_seconds': prep_s,
        }
        if z_host is not None and not args.no_cpu_baseline:
            # not `value`: the same proof with the witness handed over in HOST memory (fk_prove_r1cs: one H2D copy of the
            # witness, (num_input + num_aux) * 32 bytes, inside the call)
            ctx.prove_witness(key, dr, z_host, r, s)
            t1 = time.perf_counter()
            for _ in range(2):
                p_host = ctx.prove_witness(key, dr, z_host, r, s)
            out['host_witness_ms_per_step'] = (time.perf_counter() - t1) / 2 * 1e3
            if p_host.tobytes() != proofs[-1]:
                raise AssertionError('bench: host-witness proof differs from the device-witness proof')
        if multi and not args.no_cpu_baseline:
            out['proof_verified_by_pairing_check'] = pairing_check(vk, z_in1, proofs[-1])
        if not multi and not args.no_cpu_baseline:
            cpu_s, cpu_m, verified = cpu_baseline_leg(ctx, fk, args.cpu_log2n, (vk, z_in1, proofs[-1]))
            scale = m / cpu_m
            out['proof_verified_by_pairing_check'] = verified
            out['cpu_baseline'] = {
                'value': 1.0 / (cpu_s * scale), 'unit': 'proofs/s', 'cores': 1, 'kind': 'port',
                'sample': 'oracle/groth16_oracle.c (bellman algorithm restated, 1 thread) proving a 2^%d-row instance of the same '
                          'workload family in %.2f s; scaled linearly by %d to 2^%d rows (optimistic for the CPU: NTT is n log n); '
                          'the GPU proof of the same sample matched byte for byte' % (args.cpu_log2n, cpu_s, scale, args.log2n),
                'sample_seconds': cpu_s,
            }
        line = json.dumps(out)
    else:
        line = None

    dr.free()
    key.free()
    if multi:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        # the JSON line is the LAST thing on stdout: native libraries (RCCL's version banner) write through C stdio, whose
        # buffer would otherwise be flushed behind it at exit
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(line, flush=True)


if __name__ == '__main__':
    main()
