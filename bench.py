#!/usr/bin/env python3
"""
bench.py -- Groth16 proofs/sec on MI355X through the C ABI (include/fawkes_hip.h).

One "step" = one complete Groth16 proof on the 2^25 evaluation domain BASELINE.json's metric is quoted on, starting from
the WITNESS VECTOR IN HOST MEMORY -- what prover.rs:69-80 hands to `create_random_proof`
(/root/reference/fawkes-crypto/src/backend/bellman_groth16/prover.rs:80): upload of the witness, device SpMV (a = Az,
b = Bz, c = Cz; bellman's `synthesize` evaluation, mod.rs:92-99), the quotient (bellman's 7 transforms computed as 6, csrc/ntt.hip), the four G1 MSMs (H, L, A, B1),
the G2 MSM (B2) and the proof assembly.  The constraint system and the proving key are resident in HBM; the witness is
handed over per proof from pinned host memory through the two-slot pipeline of the C ABI (fk_prove_r1cs_submit / _wait:
the upload of proof k+1 runs underneath proof k), so K steps contain K uploads and K proofs.  The same proof with the
witness already resident in HBM is reported beside it (`device_resident_ms_per_step`).

Default workload (`--workload rollup1024 --copies 1741`): BASELINE configs[3]'s "1024-tx shape" with real gadgets, sized so
that it FILLS the 2^25 domain the metric names (rows = 2^25) -- 1741 rollup-style transactions (two
depth-32 poseidon merkle proofs + one eddsa-poseidon signature each, 19 270 gates and 942 k matrix terms per transaction) as
ONE R1CS of 33 552 553 rows (99.99 % of 2^25) / 1.64e9 matrix terms.  Since round 5 the prover receives it in the REFERENCE'S OWN INPUT
FORM: the generated key and the circuit are written as a `Parameters` image (mod.rs:150-175: brotli gate blob + bellman key;
fk_gates_encode + fk_key_write_bellman), everything is dropped, and the prover is set up from the image alone -- fk_gates_decode ->
fk_r1cs_load_gates (every term explicit in HBM) beside fk_key_load_bellman(checked) -- and `value` is measured on THAT system (`load`:
what the set-up cost; `--tiled-headline`: rounds 1-4's arrangement, `value` on fk_r1cs_load_tiled = one instance + a copy count, now
the `tiled` leg).  (Rounds 2-3 benchmarked 1024 transactions: 19.7 M rows, the same domain 59 % filled -- kept as the
`secondary_1024_transactions` leg; the `reference_published` leg is 1853 transactions = 35.7 M rows on the 2^26 domain, the size of
the reference's one published figure, README.md:54-56.)  The transaction comes from the committed data fixture
tests/golden/rollup_tx_instance.npz (made by tests/golden/make_rollup_tx_fixture.py; the circuit builder itself is oracle-side and
is NOT imported here).
`--workload synthetic` is the round-1 shape (1-2 term rows, m = 2^LOG2 exactly; `--lc-terms` for longer combinations).
The key is a VALID key (fk_setup*, fixed toxic waste), so the proof produced in the timed region is checked afterwards
with the Groth16 pairing equation.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N --steps K --warmup W          (starts its own N ranks: torch.distributed.run as a child process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Round 6: (a) before anything touches the GPU a FIRST-CONTACT PREFLIGHT runs in a child process per rank (fawkes-crypto_amd/preflight.py: the proof's three
collectives at their real sizes with contents verified, the library's peer-access table and one verified 64 MiB pull per ordered device pair); a failure
selects the documented fallback (gloo-staged exchanges, host-side event waits) and is reported in the `preflight` block -- the run goes on; (b) N > 1 proves
the SAME input form as N = 1: rank 0 writes ONE `Parameters` image into a file every rank maps, every rank sets its prover up from it
(load_parameters(image, shard = rank / world)); the control plane (barriers, agreement on plans and timings) is a gloo group, the data plane an RCCL group;
(c) the timed proofs are compared with the ORACLE's committed bytes at this size (tests/golden/fullsize_digests.json, `oracle_digest_check`) in every run;
(d) `roofline` carries `frac` (union of the launches) and `frac_per_launch` (rocprof's average launch), blocks for the G2 kernel and the NTT, and `traffic`
measured BY THIS RUN (two rocprofv3 --pmc child processes at the end, everything of this process released first; --measure-traffic auto / on / off).

N > 1: one process per GPU, strong scaling of a single proof: every rank hands over 1 / N of the witness over its own PCIe link and the ranks
all-gather the rest over xGMI (RCCL, parallel.witness_all_gather) underneath the proof before; every rank holds its piece of the key (h in blocks of the domain; l, a, b_g1,
b_g2 dealt by work: one or two large pieces per rank),
evaluates only the rows t = rank (mod N) of a, b, c and computes 1/N of the quotient -- the transforms are cut across the
ranks with one all-to-all (RCCL over xGMI) each -- then ONE all-gather of 384 bytes per rank exchanges the partial MSM sums and
the proof is folded locally (fawkes-crypto_amd/parallel.py: prove_distributed_dev).  With 2 ranks, or a rank count that is not a
power of two, rank 0 computes the whole quotient and H while the other ranks run witness MSMs only (prove_balanced_dev with the key split
for "quotient on rank 0", FK_Z_WORK_SPLIT_Q0: nothing but 384-byte partial sums is exchanged); FK_DIST_QUOTIENT=1 / 0 forces either schedule.  The line of an N-rank run also carries the one-call form
of the same proof (`single_process_multi_gpu`: fk_init_devices + fk_multi_prove_r1cs, one process driving all GPUs, exchanges
inside the library) and the throughput mode (`replica_proofs_per_sec`).

Beside `value` (N = 1): `cpu_baseline` MEASURED at the benchmarked size when the projection from a sample fits --cpu-full-budget
(the C oracle proves the same 2^25 system; its bytes must equal the GPU's), `load` (the Parameters image: blob bytes, decode seconds and
what bounds them, host RSS peak, checked key read, levels, time to first proof), `tiled` (the same circuit as one instance + a copy count:
same key, same bytes), `witness_sensitivity` (a witness with every dense value distinct: timing only), `standalone` (G1 / G2 MSM and Fr NTT
timed alone, SURVEY 8(d) units), `legs` (seconds per optional leg; --max-seconds skips what would not fit and says so).

The CPU oracle (oracle/) appears here only in the `cpu_baseline` leg: the timed CPU baseline (one thread = the
reference's configured worker, and all host cores = bellman's multicore split), a live parity check of that same
sample, and the pairing check of the benchmarked proof; it is never the thing measured.
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

MUL_PIPE_OPS_PER_MODMUL = 136    # 8 x 8 products + 8 x 8 reduction products (v_mad_u64_u32) + 8 x (lo * INV) (v_mul_lo_u32): the fewest
                                 # 32-bit multiplier operations of a 256-bit Montgomery product
MODMUL_PER_G1_MIXED_ADD = 10     # XYZZ mixed addition: 8 M + 2 S (csrc/curve.hpp)
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


def load_rollup_instance(path=None):
    """ONE rollup-style transaction from the committed data fixture: (fk.R1cs, witnesses (k, nv, 4) uint64 Montgomery).
    The witnesses are the 32 distinct ones of tests/golden/_generated/rollup_tx_witnesses.npy when `__graft_entry__.build()`
    has made that file (tests/golden/make_rollup_witnesses.py), else the fixture's own three."""
    import fawkes_crypto_amd as fk
    d = np.load(path or os.path.join(ROOT, 'tests', 'golden', 'rollup_tx_instance.npz'))
    table = d['table']
    mats = [(d[nm + '_ptr'].astype(np.uint64), d[nm + '_col'], table[d[nm + '_cidx']]) for nm in 'abc']
    zs = np.ascontiguousarray(d['z'])
    gen = os.path.join(ROOT, 'tests', 'golden', '_generated', 'rollup_tx_witnesses.npy')
    if path is None and os.path.exists(gen):
        g = np.load(gen)
        if g.shape[1:] == zs.shape[1:] and np.array_equal(g[:len(zs)], zs):
            zs = np.ascontiguousarray(g)
    return fk.R1cs(int(d['num_input']), int(d['num_aux']), *mats), zs


def workload_dims(args):
    """(variables, domain size) of the workload `args` names, from the data fixture alone (no GPU, no library): what the first-contact
    preflight sizes its collectives with"""
    if args.workload == 'rollup1024':
        d = np.load(os.path.join(ROOT, 'tests', 'golden', 'rollup_tx_instance.npz'))
        b_in, b_aux, gates = int(d['num_input']), int(d['num_aux']), len(d['a_ptr']) - 1
        n_in = 1 + args.copies * (b_in - 1)
        n = args.copies * gates + n_in
        return n_in + args.copies * b_aux, 1 << max(n - 1, 1).bit_length()
    return 1 << args.log2n, 1 << args.log2n


def shared_image_path(estimate_bytes):
    """where rank 0 writes the `Parameters` image every rank of an N > 1 run maps: FK_BENCH_IMAGE_DIR, else shared memory when it has the room
    (the file is page cache: one copy for all ranks), else the temporary directory"""
    import tempfile
    if os.environ.get('FK_BENCH_IMAGE_DIR') == 'none':          # (test hook: behave as if no directory had the room)
        return None
    token = '%s_%d' % (os.environ.get('MASTER_PORT', '0'), os.getpid())
    for d in (os.environ.get('FK_BENCH_IMAGE_DIR'), '/dev/shm', tempfile.gettempdir()):
        if not d or not os.path.isdir(d) or not os.access(d, os.W_OK):
            continue
        try:
            st = os.statvfs(d)
            if st.f_bavail * st.f_frsize < 1.2 * estimate_bytes + (1 << 30):
                continue
        except OSError:
            continue
        return os.path.join(d, 'fk_bench_params_%s.bin' % token)
    return None      # (the caller falls back to per-rank key generation and says so)


def tile_witness(zs, num_input, copies, out=None):
    """witness of `copies` instances as one system, fk_r1cs_load_tiled's variable order: ONE, every copy's inputs, every
    copy's aux; copy j carries witness j mod len(zs)."""
    k, ni1, naux = len(zs), num_input - 1, zs.shape[1] - num_input
    nv = 1 + copies * (ni1 + naux)
    out = np.empty((nv, 4), np.uint64) if out is None else out
    assert out.shape == (nv, 4)
    out[0] = zs[0][0]
    ins = out[1:1 + copies * ni1].reshape(copies, ni1, 4)
    aux = out[1 + copies * ni1:].reshape(copies, naux, 4)
    for j in range(k):
        ins[j::k] = zs[j][1:num_input]
        aux[j::k] = zs[j][num_input:]
    return out


def materialise_rollup(copies, path=None):
    """The 1024-transaction system with EVERY term explicit (no tiling shortcut): (num_input, num_aux, [(ptr, col, cidx)] * 3, table)
    for fk_r1cs_load_coded -- 9.6e8 terms, 8 bytes each.  Same rows, same variables as fk_r1cs_load_tiled of the instance."""
    d = np.load(path or os.path.join(ROOT, 'tests', 'golden', 'rollup_tx_instance.npz'))
    b_in, b_aux = int(d['num_input']), int(d['num_aux'])
    n_in = 1 + copies * (b_in - 1)
    mats = []
    for nm in 'abc':
        ptr1, col1, cidx1 = d[nm + '_ptr'].astype(np.uint64), d[nm + '_col'].astype(np.int64), d[nm + '_cidx'].astype(np.uint32)
        per, g = len(col1), len(ptr1) - 1
        is_in = (col1 > 0) & (col1 < b_in)
        is_aux = col1 >= b_in
        col = np.empty(per * copies, np.uint32)
        for j in range(copies):
            c = col1.copy()
            c[is_in] += j * (b_in - 1)
            c[is_aux] += n_in + j * b_aux - b_in
            col[j * per:(j + 1) * per] = c
        ptr = np.empty(g * copies + 1, np.uint64)
        for j in range(copies):
            ptr[j * g:(j + 1) * g] = ptr1[:-1] + np.uint64(j * per)
        ptr[-1] = per * copies
        mats.append((ptr, col, np.tile(cidx1, copies)))
    return n_in, copies * b_aux, mats, np.ascontiguousarray(d['table'])


def _spread(samples_s):
    """min / median / max of per-repetition times, in ms"""
    ts = sorted(samples_s)
    k = len(ts)
    med = ts[k // 2] if k & 1 else 0.5 * (ts[k // 2 - 1] + ts[k // 2])
    return {'min': ts[0] * 1e3, 'median': med * 1e3, 'max': ts[-1] * 1e3, 'reps': k}


def standalone_legs(ctx, key, m, reps=10):
    """BASELINE configs[1] and the metric's second half ("MSM scalar-muls/sec") as figures of their own: the G1 / G2 multi-scalar
    multiplication and the Fr transform timed ALONE on this GPU (inputs resident), in SURVEY section 8(d)'s units.  Every figure is
    `reps` (>= 10) separately timed repetitions after five untimed ones -- min / median / max of the host's wall clock around a
    synchronised call, and beside it the same repetitions by HIP events on the library's stream (`hip_event_ms`) -- so that a slow
    outlier shows as an outlier and not as the figure.  Rates are quoted on the MEDIAN.  Scalar distributions: BASELINE.md config 2
    (a) uniform mod r and (b) witness-like (half in {0, 1}: fk_gen_scalars_dev kind 1)."""
    import torch
    out = {}
    st = torch.cuda.ExternalStream(ctx.stream_handle())

    def timed(fn, n_reps=reps):
        # FIVE untimed calls: a standalone multiplication takes the library's four MSM lanes in turn, and a lane whose scratch was
        # sized by a smaller multiplication of the proofs before (A's, B's) frees and re-allocates several GB the first time a larger
        # one lands on it -- 120 .. 300 ms, once per lane.  That was round 3's "4x outlier" (1 warm-up + 3 repetitions = one call per
        # lane, three of them growing: tools/stall_probe.py, profiles/r04_stall_probe.log); five calls touch every lane.
        for _ in range(5):
            fn()
        ctx.sync()
        wall, evs = [], []
        for _ in range(n_reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ctx.sync()
            t0 = time.perf_counter()
            e0.record(st)
            fn()
            e1.record(st)
            ctx.sync()
            wall.append(time.perf_counter() - t0)
            e1.synchronize()
            evs.append(e0.elapsed_time(e1) * 1e-3)
        return _spread(wall), _spread(evs)

    def entry(wall, ev, n, bytes_per, path, **kw):
        t = wall['median'] * 1e-3
        d = {'ms': wall['median'], 'wall_ms': wall, 'hip_event_ms': ev, 'scalar_muls_per_sec': n / t, 'algorithmic_GBps': bytes_per * n / t / 1e9, 'path': path}
        d.update(kw)
        return d

    kinds = ((0, 'uniform'), (1, 'witness_like'))
    for log_n in (20,):
        n = 1 << log_n
        d_b1, d_b2, d_s = ctx.dev_alloc(n * 64), ctx.dev_alloc(n * 128), ctx.dev_alloc(n * 32)
        ctx.gen_points_g1_dev(d_b1, n, 11); ctx.gen_points_g2_dev(d_b2, n, 12)
        for kind, kname in kinds:
            ctx.gen_scalars_dev(d_s, n, 13 + kind, kind)
            sfx = '' if kind == 0 else '_' + kname
            w, e = timed(lambda: ctx.msm_g1_dev(d_b1, d_s, n))
            out['msm_g1_2p%d%s' % (log_n, sfx)] = entry(w, e, n, 96, 'fk_msm_g1_dev (bases as plain device memory)', scalars=kname)
            w, e = timed(lambda: ctx.msm_g2_dev(d_b2, d_s, n))
            out['msm_g2_2p%d%s' % (log_n, sfx)] = entry(w, e, n, 160, 'fk_msm_g2_dev', scalars=kname)
        for p_ in (d_b1, d_b2, d_s):
            ctx.dev_free(p_)
    # the key's own arrays (resident, with their fixed-base levels when the key holds them): the product's path at this size
    info, pre = key.shard_info(), key.precomputed()
    log_m = int(m).bit_length() - 1
    for arr, cnt_key, bytes_per, tag in (('h', 'h', 96, 'msm_g1_2p%d_key_bases' % log_m), ('l', 'l', 96, 'msm_g1_key_l'), ('b_g2', 'b', 160, 'msm_g2_key_b_g2')):
        n_pts = info[cnt_key][1] - info[cnt_key][0]
        if n_pts <= 0:
            continue
        d_s = ctx.dev_alloc(n_pts * 32)
        for kind, kname in kinds:
            if arr == 'h' and kind == 1:
                continue                  # h's scalars are quotient coefficients: dense by nature
            ctx.gen_scalars_dev(d_s, n_pts, 17 + kind, kind)
            w, e = timed(lambda: ctx.prove_msm_array_dev(key, arr, d_s))
            out[tag + ('' if kind == 0 else '_' + kname)] = entry(
                w, e, n_pts, bytes_per, 'fk_prove_msm_array_dev over the resident %s array (fixed-base levels: %d)' % (arr, pre[arr]), points=n_pts, scalars=kname)
        ctx.dev_free(d_s)
    for log_n in sorted({20, log_m}):
        n = 1 << log_n
        d = ctx.dev_alloc(n * 32)
        ctx.gen_scalars_dev(d, n, 19, 0)
        w, e = timed(lambda: ctx.ntt_dev(d, log_n))
        t = e['median'] * 1e-3
        out['ntt_2p%d' % log_n] = {'ms': e['median'], 'wall_ms': w, 'hip_event_ms': e, 'algorithmic_GBps': 64 * n / t / 1e9,
                                   'GBps_is': 'SURVEY 8(d): 64 B per element per transform (one ideal pass), forward transform in place; quoted on the HIP-event median'}
        ctx.dev_free(d)
    return out


def other_size_leg(ctx, inst, zs, copies, tox, r, s, steps, check=True):
    """The same prover on the same circuit family at another transaction count (its own key and resident system; the main key must
    have been freed): host-witness pipeline `ms_per_step` exactly as `value` is measured, the device-resident latency beside it,
    which key arrays kept their fixed-base levels, and the proof pairing-checked against the system's public inputs."""
    num_input, num_aux = 1 + copies * (inst.num_input - 1), copies * inst.num_aux
    n = copies * inst.num_gates + num_input
    log_m = max(n - 1, 1).bit_length()
    nv = num_input + num_aux
    t0 = time.perf_counter()
    z_pin = [ctx.host_alloc((nv, 4)) for _ in range(2)]
    tile_witness(zs, inst.num_input, copies, out=z_pin[0])
    # the second slot holds a DIFFERENT witness (the transactions dealt to the copies in another order): two distinct proofs
    # alternate in the two-slot pipeline, each checked
    tile_witness(zs[::-1], inst.num_input, copies, out=z_pin[1])
    dr = ctx.load_r1cs(inst, copies=copies)
    key, vk = ctx.setup(inst, copies=copies, **tox)
    prep = time.perf_counter() - t0
    try:
        tk = ctx.prove_witness_submit(key, dr, z_pin[0], r, s)
        proofs = [None, None]
        for i in range(2):                # warm-up: both slots once
            nxt = ctx.prove_witness_submit(key, dr, z_pin[(i + 1) & 1], r, s)
            proofs[i & 1] = ctx.prove_witness_wait(tk).tobytes(); tk = nxt
        ctx.sync()
        t1 = time.perf_counter()
        for i in range(2, 2 + steps):
            nxt = ctx.prove_witness_submit(key, dr, z_pin[(i + 1) & 1], r, s)
            p_ = ctx.prove_witness_wait(tk).tobytes(); tk = nxt
            if p_ != proofs[i & 1]:
                raise AssertionError('bench: pipelined proofs of the %d-transaction system differ between steps' % copies)
        ctx.sync()
        ms = (time.perf_counter() - t1) / steps * 1e3
        ctx.prove_witness_wait(tk)
        d_z = ctx.dev_alloc(nv * 32)
        ctx.upload(d_z, z_pin[0])
        ctx.prove_witness_dev(key, dr, d_z, r, s)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(max(2, min(steps, 5))):
            p_dev = ctx.prove_witness_dev(key, dr, d_z, r, s)
        dev_ms = (time.perf_counter() - t1) / max(2, min(steps, 5)) * 1e3
        ctx.dev_free(d_z)
        if p_dev.tobytes() != proofs[0]:
            raise AssertionError('bench: device-resident proof of the %d-transaction system differs from the pipelined one' % copies)
        out = {'transactions': copies, 'rows': n, 'log2_domain': log_m, 'domain_fill': n / float(1 << log_m), 'num_aux': num_aux,
               'matrix_terms': int(sum(dr.info()['nnz'])), 'ms_per_step': ms, 'proofs_per_sec': 1e3 / ms, 'steps': steps,
               'device_resident_ms_per_step': dev_ms, 'msm_fixed_base_levels': key.precomputed(), 'levels_plan': key.levels_plan(), 'prep_seconds': prep,
               'witness_bytes_per_proof': nv * 32}
        dg = check_digest(copies, True, zs, proofs)          # the oracle's bytes at this size, when committed (raises on a difference)
        if dg is not None:
            out['oracle_digest_check'] = dg
        if check:
            out['proof_verified_by_pairing_check'] = bool(pairing_check(vk, z_pin[0][1:num_input].copy(), proofs[0]) and
                                                          pairing_check(vk, z_pin[1][1:num_input].copy(), proofs[1]))
        return out
    finally:
        key.free(); dr.free()
        for zp in z_pin:
            ctx.host_free(zp)


def single_process_leg(fk, n_ranks, same_device, r1cs, copies, z_pin, tox, r, s, want, steps, image=None):
    """The one-call form of the multi-GPU prover (fk_init_devices + fk_multi_prove_r1cs: one process, a worker thread per GPU,
    peer-DMA exchanges inside the library) on the same workload, host-witness pipeline.  image: the `Parameters` image the ranks were set up
    from -- the one-call prover is then set up from it as well (fk_multi_key_load_bellman(checked) + fk_gates_decode -> fk_multi_r1cs_load_gates:
    the explicit system on every GPU); None: fk_multi_setup_tiled / fk_multi_r1cs_load_tiled."""
    mc = fk.MultiContext([0] * n_ranks if same_device else list(range(n_ranks)))
    form_note = None
    if image is not None and same_device:
        # one-GPU rehearsal: N explicit replicas of the system (8 bytes per term each) beside the key and its levels may not fit ONE device -- on a node
        # every rank has its own HBM.  Then the rehearsal's one-call leg keeps the tiled replicas (4 MB each) and says so.
        import torch
        from fawkes_crypto_amd import params_io as pio
        terms = sum(int(x) for x in (len(c_) for _, c_, _ in r1cs.mats)) * int(copies)
        if n_ranks * 8 * terms > 0.3 * torch.cuda.get_device_properties(0).total_memory:
            image = None
            form_note = 'tiled (one-GPU rehearsal: %d explicit replicas of %.1f GB do not fit one device beside the key and its levels)' % (n_ranks, 8 * terms / 1e9)
    try:
        t0 = time.perf_counter()
        if image is not None:
            from fawkes_crypto_amd import params_io as pio
            hdr = pio.read_parameters(image)
            cnt = pio.bellman_counts(hdr['bellman'])
            gates = fk.api.Gates(hdr['gates_blob'], fk.api.FK_GATES_BROTLI, hdr['num_gates'], cnt['num_input'], cnt['num_aux'], ctx=None)
            try:
                key, _, _ = mc.load_key_bellman(hdr['bellman'], flags=fk.api.FK_KEY_CHECKED)
                dr = mc.load_gates(gates)
            finally:
                gates.free()
        else:
            key, _ = mc.setup(r1cs, copies=copies, **tox)
            dr = mc.load_r1cs(r1cs, copies=copies)
        prep = time.perf_counter() - t0
        tk, slot = mc.prove_witness_submit(key, dr, z_pin[0], r, s), 0
        mc.sync()
        t1 = None
        for i in range(1 + steps):       # one warm-up step, then `steps` timed ones; the two slots hold different witnesses
            if i == 1:
                mc.sync()
                t1 = time.perf_counter()
            nxt = mc.prove_witness_submit(key, dr, z_pin[slot ^ 1], r, s)
            p = mc.prove_witness_wait(tk); tk = nxt
            if want[slot] is not None and p.tobytes() != want[slot]:
                raise AssertionError('single-process multi-GPU proof differs from the benchmarked proof')
            slot ^= 1
        mc.sync()
        ms = (time.perf_counter() - t1) / steps * 1e3
        mc.prove_witness_wait(tk)
        key.free(); dr.free()
        return {'ms_per_step': ms, 'proofs_per_sec': 1e3 / ms, 'ranks': n_ranks, 'steps': steps, 'prep_seconds': prep,
                'matrix_form': 'explicit, from a Parameters gate blob' if image is not None else (form_note or 'tiled (one instance + copy count)'),
                'transport': mc.transport, 'topology': mc.topology(), 'note': mc.note(),
                'is': 'fk_multi_prove_r1cs (ONE call on %d GPUs, in-library peer-DMA all-to-all), same proof bytes' % n_ranks}
    finally:
        mc.close()


def host_rss():
    """resident set of this process: now and its peak (VmRSS / VmHWM of /proc/self/status), bytes"""
    out = {'now': None, 'peak': None}
    try:
        for ln in open('/proc/self/status'):
            if ln.startswith('VmRSS:'):
                out['now'] = int(ln.split()[1]) * 1024
            elif ln.startswith('VmHWM:'):
                out['peak'] = int(ln.split()[1]) * 1024
    except OSError:
        pass
    return out


def usable_cores():
    """host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container can show 256
    CPUs and be allowed the time of two)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                quota, period = txt[0], int(txt[1])
            else:
                quota, period = txt[0], int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota not in ('max', '-1'):
                n = min(n, max(1, int(quota) // period))
            break
        except Exception:
            continue
    return max(1, n)


def oracle_key(dk, vk, m, num_input, num_aux):
    import c_oracle as co
    return co.ArrayKey(m, num_input, num_aux, vk, dk.download('h'), dk.download('l'), dk.download('a'), dk.download('b_g1'),
                       dk.download('b_g2'), ic=vk['ic'])


def cpu_baseline_leg(ctx, fk, args, full=None):
    """Times the C oracle (bellman's algorithm restated: oracle/groth16_oracle.c) -- synthesis (bellman's serial `synthesize`
    evaluation) + create_proof -- (1) on a bounded sample of the SAME workload family, with ONE thread (the worker fawkes-crypto
    configures) and with several thread counts up to the usable cores (bellman's multicore split restated:
    parallel_fft, one task per multiexp region); the GPU proof of that sample must match byte for byte; (2) with the fastest
    thread count at the benchmark's FULL size when the projection from the sample fits `--cpu-full-budget` (`full`: the
    benchmarked key, witness, r, s and proof bytes) -- measured, not scaled, and its proof must equal the GPU's as well."""
    import c_oracle as co
    import fixtures as fx
    cores = usable_cores()
    tox = {k: mont(v) for k, v in TOXIC.items()}
    r, s = mont(0x1234567), mont(0x89abcdef)
    if args.workload == 'rollup1024':
        inst, zs = load_rollup_instance()
        copies = args.cpu_copies
        z = tile_witness(zs, inst.num_input, copies)
        dk, vk = ctx.setup(inst, copies=copies, **tox)
        dr = ctx.load_r1cs(inst, copies=copies)
        one = co.R1csC(inst.num_input, inst.num_aux, *[co.Csr(p_, c_, v_) for p_, c_, v_ in inst.mats])
        synth = lambda z_, n_: co.synthesize_tiled(one, n_, z_)     # = synthesize of the explicitly replicated system (tests/test_oracle.py)
        what = '%d rollup-style transactions as one R1CS' % copies
        scale = args.copies / copies
    else:
        r1cs, z = build_workload(ctx, fk, args.cpu_log2n, seed=77)
        dk, vk = ctx.setup(r1cs, **tox)
        dr = ctx.load_r1cs(r1cs)
        csr = co.R1csC(r1cs.num_input, r1cs.num_aux, *[co.Csr(p_, c_, v_ if v_ is not None else np.tile(mont(1), (len(c_), 1))) for p_, c_, v_ in r1cs.mats])
        synth = lambda z_, n_: co.synthesize(csr, z_)
        copies = None
        what = 'a 2^%d-row instance of the same synthetic family' % args.cpu_log2n
        scale = (1 << args.log2n) / (1 << args.cpu_log2n)
    cnt = dk.counts()
    okey = oracle_key(dk, vk, cnt['m'], cnt['num_input'], cnt['num_aux'])
    got = ctx.prove_witness(dk, dr, z, r, s)
    dr.free(); dk.free()
    t0 = time.time()
    a, b, c, aa, bi, ba = synth(z, copies)
    synth_s = time.time() - t0
    times = {}
    for th in sorted({1, min(cores, 16), min(cores, 32), min(cores, 64), cores}, reverse=True):
        t0 = time.time()
        want = co.prove(okey, a, b, c, z, aa, bi, ba, r, s, threads=th)
        times[th] = time.time() - t0
        if got.tobytes() != want.tobytes():
            raise AssertionError('bench parity check failed: HIP proof != oracle proof (%d threads) on the CPU-baseline sample' % th)
    best = min((t, th) for th, t in times.items() if th > 1 or cores == 1)[1]
    out = dict(what=what, scale=scale, cores=cores, best_threads=best, synth_s=synth_s, prove_s=times, log2_m=int(cnt['m']).bit_length() - 1,
               host=dict(cpu_count=os.cpu_count(), affinity=len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None), full=None)
    del a, b, c, okey
    projected = (synth_s + times[best]) * scale * 1.25          # the transforms are n log n and the working set leaves every cache
    if full is not None and copies is not None and projected <= args.cpu_full_budget:
        key_f, vk_f, z_f, r_f, s_f, proof_f = full
        cnt = key_f.counts()
        okey = oracle_key(key_f, vk_f, cnt['m'], cnt['num_input'], cnt['num_aux'])
        t0 = time.time()
        a, b, c, aa, bi, ba = synth(z_f, args.copies)
        fs = time.time() - t0
        t0 = time.time()
        want = co.prove(okey, a, b, c, z_f, aa, bi, ba, r_f, s_f, threads=best)
        fp = time.time() - t0
        if want.tobytes() != proof_f:
            raise AssertionError('bench parity check failed: HIP proof != oracle proof at the FULL benchmark size')
        out['full'] = dict(synth_s=fs, prove_s=fp, threads=best, log2_m=int(cnt['m']).bit_length() - 1)
    else:
        out['full_skipped'] = ('projected %.0f s > --cpu-full-budget %.0f s' % (projected, args.cpu_full_budget)) if full is not None else 'not requested'
    return out


def pairing_check(vk_full, z_inputs, proof):
    """the benchmarked proof must satisfy the Groth16 pairing equation (checker: oracle/bn254_ref.py verifier; the public
    inputs -- 2048 roots for the 1024-transaction system -- are folded into one point by the C oracle's multiexp first)"""
    import bn254_ref as ref
    import c_oracle as co
    g1 = lambda b_: ref.g1_from_raw_le(bytes(b_))
    g2 = lambda b_: ref.g2_from_raw_le(bytes(b_))
    ic = np.ascontiguousarray(vk_full['ic'], np.uint8).reshape(-1, 64)
    z_inputs = np.ascontiguousarray(z_inputs, np.uint64).reshape(-1, 4)
    assert len(z_inputs) + 1 == len(ic)
    acc = co.g1_add(ic[0], co.msm_g1(ic[1:], z_inputs)) if len(z_inputs) else ic[0]
    pk = dict(alpha_g1=g1(vk_full['alpha_g1']), beta_g2=g2(vk_full['beta_g2']), gamma_g2=g2(vk_full['gamma_g2']),
              delta_g2=g2(vk_full['delta_g2']), ic=[g1(acc)])
    if not ref.verify(pk, [], ref.proof_from_borsh(proof)):
        raise AssertionError('bench: the benchmarked proof does not satisfy the Groth16 pairing equation')
    return True


def pmc_traffic(args, log_m, rows, world, acc, acc_s, achieved):
    """HBM traffic of the dominant kernel from the newest committed PMC pass taken on THIS workload (same workload name, domain and
    row count): bytes per (scalar, base) pair of that pass x this run's pairs / this run's kernel time.  Returns (GB/s or None,
    source, error).  A figure that cannot be right -- below the algorithmic bytes, more than 30x above them, or beyond what HBM can
    deliver (6.3 TB/s measured, /opt/skills/guides/MI355X_MICROARCH.md) -- is reported as null with the reason (a summary
    divided by the wrong proof count once put 7.2 TB/s here)."""
    if world != 1 or acc_s <= 0:
        return None, None, None
    prof = os.path.join(ROOT, 'profiles')
    for cand in sorted((f for f in os.listdir(prof) if f.endswith('.json') and 'pmc_traffic' in f), reverse=True):
        try:
            pmc = json.load(open(os.path.join(prof, cand)))
            if pmc.get('workload', 'synthetic') != args.workload or pmc['log2n'] != log_m or pmc.get('rows') != rows:      # (summaries without a row count: rounds 1-3, other systems)
                continue
            dkk = pmc['dominant_kernel']
            per_pt = (dkk['fetch_bytes_per_proof_raw'] * dkk.get('fetch_correction', 1.0) + dkk['write_bytes_per_proof']) / dkk['points_per_proof']
            traffic = per_pt * acc['units'] / acc_s / 1e9
            src = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, profiles/%s (%d proofs in the pass, counted from its dispatches; bytes per point '
                   'from that pass x this run\'s points; FETCH_SIZE x %.2f, the factor calibrated for 64-byte gathers)'
                   % (cand, dkk.get('proofs_in_the_pass', 1), dkk.get('fetch_correction', 1.0)))
            ratio = traffic / achieved if achieved > 0 else 0.0
            if not (1.0 <= ratio <= 30.0) or traffic > 6300.0:
                return None, src, ('implausible: %.0f GB/s = %.1f x the algorithmic bytes (accepted: 1 .. 30 x and <= 6300 GB/s); the PMC summary '
                                   'is mis-normalised or from another configuration' % (traffic, ratio))
            return traffic, src, None
        except Exception:
            continue
    return None, None, 'no committed PMC traffic pass for this workload / domain / row count under profiles/'


def hbm_block(kernel, units, bytes_per_unit, union_ms, sum_ms, launches, unit_name, note=None):
    """roofline block of one kernel against HBM.  Two readings of "the kernel's duration" (VERDICT r5 item 5):
      achieved / frac                      algorithmic bytes / the UNION of the launches' HIP-event intervals (launches that run side by side on
                                           different lanes each span the whole phase: the union is the time the kernel took);
      achieved_per_launch / frac_per_launch  the bytes of an average launch / the average of the launches' own durations -- what
                                           `rocprofv3 --kernel-trace --stats` calls AverageNs (side-by-side launches each count the shared time)."""
    union_s, sum_s = union_ms * 1e-3, sum_ms * 1e-3
    ach = units * bytes_per_unit / union_s / 1e9 if union_s > 0 else 0.0
    per = units * bytes_per_unit / sum_s / 1e9 if sum_s > 0 else 0.0
    n_l = max(int(launches), 1)
    b = {'bound': 'hbm', 'kernel': kernel, 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
         'achieved_per_launch': per, 'frac_per_launch': per / HBM_PEAK_GBS, 'traffic': None, 'traffic_ratio': None,
         'launches': int(launches), 'union_ms_per_launch': union_ms / n_l, 'avg_launch_ms': sum_ms / n_l,
         'avg_launch_ms_is': 'mean of the launches\' own HIP-event durations (rocprofv3\'s AverageNs); union_ms_per_launch = union of their intervals / launches',
         'algorithmic_bytes_per_%s' % unit_name: bytes_per_unit, '%ss' % unit_name: int(units)}
    if note:
        b['note'] = note
    return b


def measure_traffic_leg(args, log_m, g1_points, g2_points, rows, limit_s):
    """HBM bytes of the accumulations and the transforms from the PMC counters, measured by THIS run: two child processes
    `rocprofv3 --pmc FETCH_SIZE -- python3 bench.py <one step, no optional legs>` and the same with WRITE_SIZE (separate passes, never combined with a
    trace domain, the program directly after `--`; /opt/skills/guides/MI355X_MICROARCH.md), summarised by tools/pmc_summary.py -- the proof count of
    a pass is derived from its own dispatches.  The parent has released its key and system before (two 2^25 provers do not fit one GPU).
    Returns (summary dict or None, error or None)."""
    import importlib.util
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if exe is None:
        return None, 'rocprofv3 not found on this machine'
    tmp = tempfile.mkdtemp(prefix='fk_pmc_', dir='/tmp' if os.path.isdir('/tmp') else None)
    base = [sys.executable, os.path.abspath(__file__), '--gpus', '1', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-untiled', '--no-standalone',
            '--no-other-sizes', '--no-preflight', '--measure-traffic', 'off', '--tiled-headline', '--workload', args.workload, '--copies', str(args.copies),
            '--log2n', str(args.log2n), '--lc-terms', str(args.lc_terms)]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'FK_BENCH_REHEARSE')}
    env['TMPDIR'] = '/tmp'
    t0 = time.time()
    try:
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            left = limit_s - (time.time() - t0)
            if left < 20:
                return None, 'time limit reached before the %s pass' % ctr
            cmd = [exe, '--pmc', ctr, '--output-format', 'csv', '-d', os.path.join(tmp, ctr), '-o', 'p', '--'] + base
            proc = subprocess.Popen(cmd, env=env, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
            try:
                log, _ = proc.communicate(timeout=left)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)        # the process group WE started (start_new_session): rocprofv3 and its python3
                except OSError:
                    pass
                proc.communicate()
                return None, 'the %s pass did not finish within %.0f s' % (ctr, left)
            if proc.returncode != 0:
                return None, 'the %s pass ended with rc %s: %s' % (ctr, proc.returncode, (log or '')[-300:].replace('\n', ' | '))
        spec = importlib.util.spec_from_file_location('fk_pmc_summary', os.path.join(ROOT, 'tools', 'pmc_summary.py'))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        out_json = os.path.join(tmp, 'traffic.json')
        try:
            mod.traffic(os.path.join(tmp, 'FETCH_SIZE'), os.path.join(tmp, 'WRITE_SIZE'), log_m, g1_points, out_json, None, args.workload, None, rows)
        except SystemExit as e:
            return None, 'tools/pmc_summary.py: %s' % e
        j = json.load(open(out_json))
        j['seconds'] = round(time.time() - t0, 1)
        j['g2_points_per_proof'] = g2_points
        return j, None
    except Exception as e:       # noqa: BLE001 -- the line is printed whatever happens here
        return None, '%s: %s' % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def check_digest(copies, two_witnesses, zs, want):
    """the benchmarked proofs against the ORACLE's bytes at this size, when tests/golden/fullsize_digests.json holds them (made by
    tests/golden/make_fullsize_digests.py: oracle/groth16_oracle.c on the host cores; DATA, nothing of the oracle runs here) -- so the parity of the
    timed proofs is checked in every run, also with --no-cpu-baseline.  Returns a dict for the JSON line, or None when no entry applies."""
    import hashlib
    path = os.path.join(ROOT, 'tests', 'golden', 'fullsize_digests.json')
    if copies is None or not os.path.exists(path):
        return None
    e = json.load(open(path))['entries'].get('rollup%d' % copies)
    if e is None:
        return None
    if e.get('witness_set_sha256') != hashlib.sha256(np.ascontiguousarray(zs).tobytes()).hexdigest():
        return {'file': 'tests/golden/fullsize_digests.json', 'entry': 'rollup%d' % copies, 'applies': False,
                'why': 'the digests were made for another set of transaction witnesses (tests/golden/_generated/rollup_tx_witnesses.npy absent or different)'}
    got = [w.hex() for w in want if w is not None]
    ok = all(g == x for g, x in zip(got, e['proofs']))
    if not ok:
        raise AssertionError('bench: the benchmarked proof bytes differ from the oracle\'s bytes in tests/golden/fullsize_digests.json (entry rollup%d)' % copies)
    return {'file': 'tests/golden/fullsize_digests.json', 'entry': 'rollup%d' % copies, 'applies': True, 'proofs_compared': len(got), 'equal': True,
            'is': 'the 256 proof bytes of oracle/groth16_oracle.c at this size (%s threads, %s s per proof on the box that made the file): committed data, '
                  'compared in every run' % (e.get('threads'), e.get('oracle_seconds'))}


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` as a child process (one rank per GPU, rendezvous on 127.0.0.1 at a free port), pass its output through and
    print rank 0's JSON line once more as the LAST line of our stdout.  Returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC: what RCCL needs on this driver
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, bufsize=1)
    line = None
    for ln in proc.stdout:
        if ln.startswith('{"metric"'):
            line = ln.rstrip('\n')
        else:
            sys.stdout.write(ln)
    rc = proc.wait()
    sys.stdout.flush()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', choices=('rollup1024', 'synthetic'), default='rollup1024')
    ap.add_argument('--copies', type=int, default=1741,
                    help='rollup1024: transactions in the one R1CS (1741 -> 33 552 553 rows = 99.99 %% of the 2^25 domain, BASELINE config 4: "rows = 2^25"; '
                         '1024 -> 19.7 M rows, the same domain 59 %% filled: rounds 2-3)')
    ap.add_argument('--lc-terms', type=int, default=0, help='synthetic: operands per side of every product gate (default: the 1-2 term shape)')
    ap.add_argument('--log2n', type=int, default=25, help='synthetic: log2 of the row count handed to the prover')
    ap.add_argument('--cpu-log2n', type=int, default=20, help='synthetic: size of the CPU-baseline sample instance')
    ap.add_argument('--cpu-copies', type=int, default=32, help='rollup1024: transactions in the CPU-baseline sample (32 -> domain 2^20)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-replicas', action='store_true', help='N > 1: skip the one-proof-per-GPU throughput leg')
    ap.add_argument('--no-single-process', action='store_true', help='N > 1: skip the fk_multi_prove_r1cs leg (one process driving all GPUs)')
    ap.add_argument('--tiled-headline', action='store_true',
                    help='rollup1024, N = 1: measure `value` on the tiled resident system (fk_r1cs_load_tiled: one instance + a copy count, rounds 1-4) instead of '
                         'the explicit system decoded from the gate blob of a `Parameters` image (the reference\'s own input form: the default)')
    ap.add_argument('--blob-quality', type=int, default=2,
                    help='brotli quality of the gate blob the benchmark writes.  The reference\'s setup uses 9 (20 minutes for this system on one thread); '
                         'quality 2 writes a blob of nearly the same size that DECODES at the same speed as a quality-9 one (profiles/r05_blob_quality_decode.log), '
                         'in 1.5 x the time of quality 1, whose blob is twice as large and decodes 30 %% slower')
    ap.add_argument('--no-untiled', action='store_true', help='rollup1024, N = 1: with --tiled-headline, skip the leg with the explicit system; otherwise skip the tiled leg')
    ap.add_argument('--no-standalone', action='store_true', help='N = 1: skip the standalone MSM / NTT figures')
    ap.add_argument('--no-other-sizes', action='store_true',
                    help='rollup1024, N = 1: skip the legs at other transaction counts (--secondary-copies, --reference-copies)')
    ap.add_argument('--secondary-copies', type=int, default=1024, help='rollup1024, N = 1: the 1024-transaction system of rounds 2-3 (59 %% of the domain) as a secondary leg')
    ap.add_argument('--reference-copies', type=int, default=1853,
                    help='rollup1024, N = 1: transactions of the `reference_published` leg (1853 -> 35 711 017 rows >= the 35 695 616 constraints of the '
                         'reference\'s one published figure, README.md:54-56; domain 2^26)')
    ap.add_argument('--cpu-full-budget', type=float, default=600.0,
                    help='seconds the FULL-SIZE all-cores CPU baseline run may take by projection from the sample (else the scaled sample is reported)')
    ap.add_argument('--max-seconds', type=float, default=1500.0,
                    help='wall-clock budget of the whole command: an optional leg (sensitivity, tiled / untiled, standalone, replicas, one-call form, CPU baseline, '
                         'other sizes) is skipped -- and listed under `legs` with the reason -- when the time used so far plus its estimate would exceed it, so that '
                         'the JSON line is always printed')
    ap.add_argument('--no-preflight', action='store_true',
                    help='skip the first-contact checks (fawkes-crypto_amd/preflight.py: the proof\'s collectives at their real sizes with contents verified, the '
                         'library\'s peer-access table and one verified 64 MiB pull per ordered device pair -- run in a child process per rank before any key is built; '
                         'on a failure the run continues over the documented fallback and says so in the `preflight` block)')
    ap.add_argument('--measure-traffic', choices=('auto', 'on', 'off'), default='auto',
                    help='N = 1: at the end of the run (everything freed), run this command again under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (two child '
                         'processes, python3 directly after `--`, one step without the optional legs) and quote `roofline.traffic` from THOSE counters instead of the '
                         'committed pass under profiles/.  auto: on for systems of 2^23 rows and more (the benchmark), off for small ones; skipped with the reason when '
                         'rocprofv3 is absent or --max-seconds would be exceeded')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend ('nccl' = RCCL; 'gloo' only for single-GPU dry runs of the N>1 code path with FK_BENCH_SAME_DEVICE=1)")
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves, as fresh child processes, BEFORE this process has
        # made any GPU call (it never does), and relay rank 0's JSON line as our own last line of stdout
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        raise SystemExit('--gpus %d != WORLD_SIZE %d' % (args.gpus, world))

    same_device = os.environ.get('FK_BENCH_SAME_DEVICE') == '1'
    # FK_BENCH_REHEARSE=1 with --gpus 1: run the multi-GPU code path (process group, all-to-all, all-gather, distributed
    # quotient with one rank) on a single GPU -- a rehearsal of the N > 1 plumbing over real RCCL, not a benchmark mode
    rehearse = os.environ.get('FK_BENCH_REHEARSE') == '1'
    multi = world > 1 or rehearse
    # ---------------------------------------------------------------- first contact (before this process touches the GPU): the proof's collectives at
    # their real sizes, the peer-access table and a verified pull per device pair, in a child process with a time limit.  A failure selects the
    # documented fallback for THIS run (gloo-staged exchanges / host-side event waits) and is reported, instead of ending the run.
    preflight = None
    # under a profiler (rocprofv3 preloads its library into every process it starts) neither the preflight's child nor the traffic leg's
    # rocprofv3 children are started: never a profiler inside a profiler, never an extra GPU process inside a counter pass
    under_profiler = 'rocprof' in os.environ.get('LD_PRELOAD', '').lower() or any(k.startswith(('ROCPROFILER_', 'ROCPROF_')) for k in os.environ)
    if under_profiler:
        args.no_preflight = True
        args.measure_traffic = 'off'
    if not args.no_preflight and os.environ.get('FK_BENCH_PREFLIGHT', '1') != '0':
        from fawkes_crypto_amd import preflight as pf_mod
        dims_nv, dims_m = workload_dims(args)
        preflight = pf_mod.run(rank, local_rank, world, args.backend, same_device, dims_nv, dims_m,
                               limit_s=float(os.environ.get('FK_BENCH_PREFLIGHT_LIMIT', '150')))
        if args.backend == 'nccl' and multi and preflight['decision']['backend'] != 'nccl':
            args.backend = preflight['decision']['backend']
        if preflight['decision']['host_events']:
            os.environ['FK_MULTI_HOST_EVENTS'] = '1'

    import torch
    import fawkes_crypto_amd as fk
    from fawkes_crypto_amd import parallel
    if same_device:
        local_rank = 0                      # dry run: all ranks share GPU 0 (needs --backend gloo)
        os.environ.setdefault('FK_CO_TENANTS', str(world))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    data_group = None
    if multi:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        # control plane: a gloo group (barriers, the agreement on timings and plans -- host tensors); data plane: a group of the chosen backend
        # ('nccl' = RCCL over xGMI on device tensors; 'gloo' = host-staged, rehearsals and the preflight's fallback).  Keeping the control plane off
        # RCCL means that a node whose RCCL does not work can still finish the run and say so.
        dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=1800))
        # every rank read its OWN child's verdict: agree on the data plane here (a rank whose child died without a verdict decided 'gloo' alone)
        agree = torch.tensor([1 if args.backend == 'nccl' else 0], dtype=torch.int32)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if args.backend == 'nccl' and int(agree.item()) == 0:
            args.backend = 'gloo'
            if preflight is not None:
                preflight['decision'] = dict(preflight['decision'], backend='gloo', why=preflight['decision']['why'] + '; another rank\'s checks did not pass: the data plane is gloo on every rank')
        if args.backend == 'nccl':
            data_group = dist.new_group(ranks=list(range(world)), backend='nccl')
    comm_dev = dev if args.backend == 'nccl' else None
    # the witness all-gather gets a process group (= RCCL communicator and stream) of its own: issued on the data group it would sit in
    # front of the running proof's all-to-alls in that communicator's queue and hold them back until the next witness has arrived
    wit_group = None
    if multi:
        wit_group = dist.new_group(ranks=list(range(world)), backend=args.backend)
    ctx = fk.Context(local_rank)

    # ---------------------------------------------------------------- wall-clock plan of the optional legs (every rank decides alike)
    t_start = time.time()
    legs = {}

    def leg_fits(name, estimate_s):
        used = time.time() - t_start
        if multi:
            import torch.distributed as dist
            tt = torch.tensor([used], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            used = float(tt.item())
        if used + estimate_s > args.max_seconds:
            legs[name] = 'skipped: %.0f s used + ~%.0f s estimated > --max-seconds %.0f' % (used, estimate_s, args.max_seconds)
            return False
        legs[name] = -time.time()
        return True

    def leg_done(name):
        if isinstance(legs.get(name), float) and legs[name] < 0:
            legs[name] = round(time.time() + legs[name], 2)
    full_size = 2.0 ** min(0, (args.copies * 19271).bit_length() - 25) if args.workload == 'rollup1024' else 2.0 ** (args.log2n - 25)      # estimates scale with the system

    # ---------------------------------------------------------------- workload: constraint system, witness, valid key
    t_prep = time.time()
    copies = None
    if args.workload == 'rollup1024':
        r1cs, zs = load_rollup_instance()
        copies = args.copies
        num_input, num_aux = 1 + copies * (r1cs.num_input - 1), copies * r1cs.num_aux
        n = copies * r1cs.num_gates + num_input
        log_m = max(n - 1, 1).bit_length()
    else:
        r1cs, z = build_workload(ctx, fk, args.log2n) if args.lc_terms <= 1 else build_workload_dense(ctx, fk, args.log2n, args.lc_terms)
        num_input, num_aux, n, log_m = r1cs.num_input, r1cs.num_aux, r1cs.n_rows, args.log2n
        assert n == 1 << log_m
    m = 1 << log_m
    nv = num_input + num_aux
    # the witness lives in PINNED host memory, two buffers used in turn (the host side of a proving loop fills one while the
    # other is in flight)
    z_pin = [ctx.host_alloc((nv, 4)) for _ in range(2)]
    if copies is not None:
        tile_witness(zs, r1cs.num_input, copies, out=z_pin[0])
        # the second slot holds a DIFFERENT witness (the same transactions dealt to the copies in reverse order, so other values in
        # every position): two distinct proofs alternate through the two-slot pipeline and each is checked -- a slot mix-up or a stale
        # staging buffer in the pipelined front cannot pass as "same bytes" 
        tile_witness(zs[::-1], r1cs.num_input, copies, out=z_pin[1])
    else:
        z_pin[0][:] = z
        del z
        z_pin[1][:] = z_pin[0]
    z_inputs = [z_pin[0][1:num_input].copy(), z_pin[1][1:num_input].copy()]
    two_witnesses = not np.array_equal(z_pin[0], z_pin[1])
    zeros = int((~z_pin[0].any(axis=1)).sum()); ones = int((z_pin[0] == mont(1)).all(axis=1).sum())
    dr = ctx.load_r1cs(r1cs, copies=copies)
    info = dr.info()
    n_a, n_b = info['n_a'], info['n_b']
    # N > 1, N a power of two: quotient and MSMs cut 1/N each (parallel.prove_distributed_dev); otherwise (or with
    # FK_DIST_QUOTIENT=0) rank 0 computes the quotient and ships h slices (parallel.prove_balanced_dev)
    # Default by rank count: at N = 2 every all-to-all of the distributed quotient moves m*32/4 bytes over ONE xGMI link
    # (256 MiB at 2^25, ~3.5 ms, seven times per proof), more than the quotient it saves; from N = 4 on the chunks are
    # small and spread over N-1 links.  FK_DIST_QUOTIENT=1 / 0 forces either schedule.
    dq_env = os.environ.get('FK_DIST_QUOTIENT', '')
    dq_ok = multi and (world & (world - 1)) == 0 and world <= 8
    dist_q = dq_ok and (dq_env == '1' or (dq_env != '0' and (world >= 4 or world == 1)))
    fracs = parallel.plan_z_fractions(world, m, num_aux, n_a, n_b)
    # balanced schedule (2 ranks, rank counts that are not a power of two): since round 4 the key is split for "quotient on rank 0" --
    # rank 0 holds ALL of h (it evaluates a, b, c, computes the whole quotient and H; that fixed work counts towards its piece of the
    # work line), the other ranks hold witness pieces only: no h slices travel.  FK_MULTI_SPLIT=equal: the fraction split of rounds 1-3.
    q0_split = world > 1 and not dist_q and os.environ.get('FK_MULTI_SPLIT') != 'equal'
    tox = {k: mont(v) for k, v in TOXIC.items()}
    # FK_BENCH_SAME_DEVICE=1 (rehearsal: every rank-process on ONE GPU): the processes cannot see each other's plans, so the keys are set up one
    # rank after the other (each loader sizes its fixed-base levels against the HBM the earlier ranks left) and every context is told how many
    # tenants share the device (FK_CO_TENANTS, read by fk_init: what a proof will allocate later is reserved that many times)
    same_dev_turns = world if (world > 1 and same_device) else 1
    params_form = copies is not None and not args.tiled_headline
    z_frac_of = lambda g: (fk.api.Z_WORK_SPLIT_Q0 if q0_split else fracs[g] if (world > 1 and not dist_q) else
                           (fk.api.Z_WORK_SPLIT if (world > 1 and os.environ.get('FK_MULTI_SPLIT') != 'equal') else fk.api.Z_EQUAL_SPLIT))
    r, s = mont(0xA11CE), mont(0xB0B)
    shared_image = None
    multi_load = None
    if multi and params_form:
        # ------------------------------------------------------------ N > 1: the SAME input form as N = 1.  Rank 0 generates the whole key
        # once and writes key + circuit as ONE `Parameters` image (Parameters::write, mod.rs:150-157) into a file every rank maps; then EVERY rank sets its
        # prover up from that image alone -- load_parameters(image, shard = rank / world): fk_gates_decode once per process -> fk_r1cs_load_gates (the
        # explicit system, resident on every GPU) beside fk_key_load_bellman(checked) of the rank's slices of the key arrays.
        import torch.distributed as dist
        from fawkes_crypto_amd import params_io as pio
        path_box = [None]
        tm_w = {}
        if rank == 0:
            est = 64 * (m + num_aux + n_a + n_b) + 128 * n_b + int(sum(info['nnz'])) + (1 << 20)
            path_box[0] = shared_image_path(est)
        if rank == 0 and path_box[0] is not None:
            import atexit
            atexit.register(lambda p_=path_box[0]: os.path.exists(p_) and os.remove(p_))      # (the ranks that map it keep their pages until they exit)
            prev = os.environ.get('FK_MSM_PRECOMP')
            os.environ['FK_MSM_PRECOMP'] = '0'            # the writer's key needs no fixed-base levels (read at every key load)
            try:
                key_w, vk = ctx.setup(r1cs, copies=copies, **tox)
            finally:
                if prev is None:
                    del os.environ['FK_MSM_PRECOMP']
                else:
                    os.environ['FK_MSM_PRECOMP'] = prev
            img_w = pio.store_parameters_dev(ctx, key_w, vk, r1cs, copies=copies, quality=args.blob_quality, lgwin=22, timings=tm_w,
                                             alloc=lambda nb: np.memmap(path_box[0], dtype=np.uint8, mode='w+', shape=(nb,)))
            img_w.flush()
            key_w.free()
            del img_w, key_w
            ctx.trim()
        elif rank != 0:
            vk = None
        dist.broadcast_object_list(path_box, src=0)
        shared_image = path_box[0]
    if multi and params_form and shared_image is None:
        # no directory has room for the image (a container with a 64 MB /dev/shm and a small /tmp): every rank generates its own key shard and the
        # ranks prove the tiled form, as before round 6 -- said in the line (`load.error`, `config.matrix_form`)
        params_form = False
        multi_load = {'error': 'no directory with room for the %.1f GB Parameters image (FK_BENCH_IMAGE_DIR, /dev/shm, the temporary directory): every rank '
                               'generated its own key shard (fk_setup_tiled) and the tiled system is proved' % ((64 * (m + num_aux + n_a + n_b) + 128 * n_b + int(sum(info['nnz']))) / 1e9)}
    if multi and params_form:
        dr.free(); dr = None
        image = np.memmap(shared_image, dtype=np.uint8, mode='r')
        tm_r = {}
        t_l0 = time.perf_counter()
        for turn in range(same_dev_turns):
            if same_dev_turns == 1 or turn == rank:
                key, dr, p_hdr = pio.load_parameters(ctx, image, shard_index=rank, shard_count=world, z_frac=z_frac_of(rank), checked=True,
                                                     disallow_points_at_infinity=False, timings=tm_r, warm=False)
                ctx.sync()
            if same_dev_turns > 1:
                dist.barrier()
        t_l1 = time.perf_counter()
        info = dr.info()
        if (info['n_a'], info['n_b']) != (n_a, n_b):
            raise AssertionError('bench: the system decoded from the gate blob has other A / B queries than the tiled one')
        if rank == 0 and (not np.array_equal(np.asarray(p_hdr['ic']), np.asarray(vk['ic'])) or bytes(p_hdr['gamma_g2']) != bytes(vk['gamma_g2'])):
            raise AssertionError('bench: the verifying key read back from the Parameters image differs from the generated one')
        multi_load = {
            'is': 'rank 0 wrote ONE `Parameters` image (mod.rs:150-157: fk_gates_encode + fk_key_write_bellman of the whole key) into a file every rank maps; '
                  'every rank then set its prover up from the image alone: fk_gates_decode (once per process) -> fk_r1cs_load_gates, and '
                  'fk_key_load_bellman(checked) of its slices (shard_index = rank, shard_count = world)',
            'image_bytes': int(image.size), 'image_file': os.path.dirname(shared_image), 'blob_bytes': tm_w.get('blob_bytes'),
            'matrix_terms': int(sum(p_hdr['gates_info']['nnz'])), 'decode_seconds': tm_r.get('gates_decode_s'), 'key_read_checked_seconds': tm_r.get('key_read_s'),
            'load_parameters_seconds_rank0': t_l1 - t_l0, 'one_rank_at_a_time': same_dev_turns > 1,
            'write': {'gates_encode_seconds': tm_w.get('gates_encode_s'), 'key_write_seconds': tm_w.get('key_write_s')},
        }
    if not (multi and params_form):
        for turn in range(same_dev_turns):
            if same_dev_turns == 1 or turn == rank:
                key, vk = ctx.setup(r1cs, shard_index=rank, shard_count=world, copies=copies, z_frac=z_frac_of(rank), **tox)
                ctx.sync()
            if same_dev_turns > 1:
                import torch.distributed as dist
                dist.barrier()
    # ---------------------------------------------------------------- N = 1: through the reference's own input form
    # fawkes hands its prover a `Parameters` object: (bellman key, num_gates, brotli(Borsh gates), const tracker), setup.rs:25-32 / mod.rs:139-175.
    # The key generated above and the circuit are WRITTEN as such an image (Parameters::write: fk_gates_encode + fk_key_write_bellman), everything
    # is dropped, and the prover is set up again from the image alone: gate blob -> fk_gates_decode -> fk_r1cs_load_gates (every one of the
    # 1.64e9 terms explicit in HBM), bellman part -> fk_key_load_bellman(checked).  `value` is measured on THAT system.
    load_block = None
    if params_form and not multi:
        from fawkes_crypto_amd import params_io as pio
        tm_w, tm_r = {}, {}
        image = pio.store_parameters_dev(ctx, key, vk, r1cs, copies=copies, quality=args.blob_quality, lgwin=22, timings=tm_w)
        key.free(); dr.free()
        key = dr = None
        rss_before = host_rss()
        t_l0 = time.perf_counter()
        key, dr, p_hdr = pio.load_parameters(ctx, image, checked=True, disallow_points_at_infinity=False, timings=tm_r)
        t_l1 = time.perf_counter()
        first_proof = ctx.prove_witness(key, dr, z_pin[0], r, s).tobytes()
        t_l2 = time.perf_counter()
        if not np.array_equal(np.asarray(p_hdr['ic']), np.asarray(vk['ic'])) or bytes(p_hdr['gamma_g2']) != bytes(vk['gamma_g2']):
            raise AssertionError('bench: the verifying key read back from the Parameters image differs from the generated one')
        info = dr.info()
        if (info['n_a'], info['n_b']) != (n_a, n_b):
            raise AssertionError('bench: the system decoded from the gate blob has other A / B queries than the tiled one')
        gp, ep = tm_r['gates_decode_profile'], tm_w.get('gates_encode_profile', {})
        rss_after = host_rss()
        load_block = {
            'is': 'the prover set up from a `Parameters` image alone (mod.rs:150-175): gate blob -> fk_gates_decode (one decompressing thread = the serial '
                  'floor of a brotli stream; parsing, range checks, coefficient dictionary and density flags on the other host threads) WHILE the bellman part '
                  'goes through fk_key_load_bellman(checked: every point on its curve, G2 in the subgroup) on the GPU and the fixed-base levels are '
                  'derived (fk_key_derive_levels, still underneath the decoding); then fk_r1cs_load_gates, and fk_key_levels_headroom confirms the levels '
                  'left the resident system its room (otherwise they are planned again)',
            'image_bytes': int(image.nbytes), 'blob_bytes': tm_w['blob_bytes'], 'bellman_bytes': tm_w['bellman_bytes'],
            'blob': 'brotli quality %d, lgwin 22 (setup.rs:26 writes quality 9, lgwin 22: a quality-2 blob has its size within 6 %% and decodes at its speed, a quality-1 blob is 2.1 x larger and decodes 30 %% slower -- profiles/r05_blob_quality_decode.log)' % args.blob_quality,
            'gate_stream_bytes': gp and int(p_hdr['gates_info']['decoded_bytes']), 'gates': int(p_hdr['gates_info']['num_gates']),
            'matrix_terms': int(sum(p_hdr['gates_info']['nnz'])), 'distinct_coefficients': int(p_hdr['gates_info']['distinct_coefficients']),
            'decode_seconds': tm_r['gates_decode_s'], 'decode_terms_per_sec': sum(p_hdr['gates_info']['nnz']) / tm_r['gates_decode_s'],
            'decode_stream_GBps': p_hdr['gates_info']['decoded_bytes'] / tm_r['gates_decode_s'] / 1e9,
            'decode_profile': gp,
            'decode_bound': 'the decompressor: %.1f of %.1f s inside libbrotlidec on one thread (a brotli stream is one serial bit stream); the %d parsing threads '
                            'used %.1f CPU-seconds beside it and made it wait %.1f s' % (gp['decompressor_s'], gp['wall_s'], gp['parse_threads'], gp['parse_cpu_s'], gp['waited_for_parsers_s']),
            'r1cs_upload_seconds': tm_r['r1cs_load_s'],
            'key_read_checked_seconds': tm_r['key_read_profile']['arrays_s'], 'key_levels_seconds': tm_r['key_read_profile']['levels_s'],
            'key_levels_early': bool(tm_r.get('key_levels_early')), 'key_levels_headroom_GiB': tm_r.get('key_levels_headroom_GiB'), 'key_levels_replanned_seconds': tm_r.get('key_levels_replanned_s'),
            'load_parameters_seconds': t_l1 - t_l0, 'first_proof_seconds': t_l2 - t_l1, 'time_to_first_proof_seconds': t_l2 - t_l0,
            'host_rss_peak_bytes': rss_after['peak'], 'host_rss_before_load_bytes': rss_before['now'], 'host_rss_after_load_bytes': rss_after['now'],
            'host_rss_note': 'the image itself (%.1f GB, held by this process as one array) is part of every figure; the decoder adds 8 bytes per matrix term '
                             'while it runs' % (image.nbytes / 1e9),
            'write': {'gates_encode_seconds': tm_w['gates_encode_s'], 'gates_encode_profile': ep, 'key_write_seconds': tm_w['key_write_s'],
                      'is': 'Parameters::write of the generated key and circuit (fk_gates_encode through libbrotlienc + fk_key_write_bellman); not part of any proving figure'},
            'reference_does': 'the reference decompresses and replays the same blob for EVERY proof (WitnessCS::get_gate_iterator, cs.rs:243-245); here once per key',
        }
        del image
    pre_levels = key.precomputed()        # fixed-base window levels per key array (0 = none: FK_MSM_PRECOMP=0 or HBM short)
    levels_plan = key.levels_plan()
    d_dens = dr.density_ptrs()
    if dist_q:
        work = [None] * 3         # a rank evaluates only its own rows, straight into send[] (fk_r1cs_eval_slice_dev): no m-element vectors
        send = [torch.empty(m // world * 32, dtype=torch.uint8, device=dev) for _ in range(3)]
        recv = [torch.empty(m // world * 32, dtype=torch.uint8, device=dev) for _ in range(3)]
        a2a = parallel.torch_all_to_all(ctx, group=data_group)
    elif multi:
        h_ranges = ([(0, m - 1)] + [(m - 1, m - 1)] * (world - 1)) if q0_split else [fk.api.h_shard_range(m - 1, g, world) for g in range(world)]
        h_full_buf = torch.empty(m * 32, dtype=torch.uint8, device=dev) if rank == 0 else None
        recv_buf = torch.empty(max(h_ranges[rank][1] - h_ranges[rank][0], 1) * 32, dtype=torch.uint8, device=dev) if rank > 0 else None
        work = [torch.empty(m * 32, dtype=torch.uint8, device=dev) for _ in range(3)] if rank == 0 else [None] * 3
    torch.cuda.synchronize()
    prep_s = time.time() - t_prep

    # ---------------------------------------------------------------- one step = one witness upload + one proof, pipelined
    state = {'i': 0, 'ticket': None}

    def prove_multi(d_z):
        wp = [w_.data_ptr() if w_ is not None else 0 for w_ in work]
        ev = (lambda: ctx.r1cs_eval_dev(dr, d_z, wp[0], wp[1], wp[2])) if wp[0] else None
        if dist_q:
            return parallel.prove_distributed_dev(ctx, key, rank, world, wp, n, log_m, d_z, d_dens[0], d_dens[1], d_dens[2], r, s, send, recv,
                                                  group=data_group, device=comm_dev, a2a=a2a, device_r1cs=dr, eval_fn=ev)
        return parallel.prove_balanced_dev(ctx, key, rank, world, wp[0], wp[1], wp[2], n, d_z, d_dens[0], d_dens[1], d_dens[2], r, s,
                                           h_ranges, h_full_buf, recv_buf, group=data_group, device=comm_dev, eval_fn=ev if rank == 0 else None)

    def prime():
        """hand over the first witness (before the timed region: the pipeline is one upload ahead)"""
        if not multi:
            state['ticket'] = ctx.prove_witness_submit(key, dr, z_pin[0], r, s)
        else:
            state['wit'] = parallel.witness_all_gather(ctx, 0, z_pin[0], rank, world, group=wit_group, device=comm_dev, force_collective=rehearse)
        state['i'] = 0

    def step():
        i = state['i']; state['i'] = i + 1
        if not multi:
            nxt = ctx.prove_witness_submit(key, dr, z_pin[(i + 1) & 1], r, s)      # upload of the NEXT proof's witness ...
            proof = ctx.prove_witness_wait(state['ticket'])                          # ... runs underneath this proof
            state['ticket'] = nxt
            return proof
        # every rank evaluates the full constraint system, so every rank needs all of z: each uploads 1 / N of it over its own PCIe link and
        # the ranks all-gather the pieces over xGMI (RCCL) on the library's copy stream, underneath this proof
        state['wit'] = parallel.witness_all_gather(ctx, (i + 1) & 1, z_pin[(i + 1) & 1], rank, world, group=wit_group, device=comm_dev, force_collective=rehearse)
        return prove_multi(ctx.witness_ptr(i & 1))

    def drain():
        if not multi and state['ticket'] is not None:
            ctx.prove_witness_wait(state['ticket']); state['ticket'] = None

    def barrier():
        if multi:
            import torch.distributed as dist
            dist.barrier()
        ctx.sync()
        torch.cuda.synchronize()

    proofs = []
    prime()
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
    drain()
    # proof k comes from slot k & 1; with two distinct witnesses the two slots give two proofs, each the same at every step
    want = [proofs[0], proofs[1] if len(proofs) > 1 else None]
    if any(pf != want[k & 1] for k, pf in enumerate(proofs)) or proofs[0] == bytes(256):
        raise AssertionError('bench: proofs differ between steps (non-deterministic result)')
    if two_witnesses and want[1] is not None and want[1] == want[0]:
        raise AssertionError('bench: two different witnesses gave the same proof bytes')
    if multi:
        import torch.distributed as dist
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- not `value`: the same proof with the witness already resident in HBM
    d_z0 = ctx.dev_alloc(nv * 32)
    ctx.upload(d_z0, z_pin[0])
    dev_steps = max(2, min(args.steps, 5))
    if not multi:
        ctx.prove_witness_dev(key, dr, d_z0, r, s)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(dev_steps):
            p_dev = ctx.prove_witness_dev(key, dr, d_z0, r, s)
        dev_ms = (time.perf_counter() - t1) / dev_steps * 1e3
        if p_dev.tobytes() != want[0]:
            raise AssertionError('bench: device-resident proof differs from the host-witness proof')
        # latency: ONE proof at a time from the witness in (pinned) host memory, upload inside the call, nothing overlapped
        lats = []
        for k in range(dev_steps):
            t1 = time.perf_counter()
            p_lat = ctx.prove_witness(key, dr, z_pin[k & 1], r, s)
            lats.append(time.perf_counter() - t1)
            if want[k & 1] is not None and p_lat.tobytes() != want[k & 1]:
                raise AssertionError('bench: one-at-a-time proof differs from the pipelined proof')
        lat_ms = sorted(lats)[len(lats) // 2] * 1e3
        # the same from PAGEABLE memory (what a caller's Vec<Fr> is): the runtime stages every piece through its own pinned buffers
        z_page = np.array(z_pin[0])
        lats_p = []
        for k in range(min(3, dev_steps)):
            t1 = time.perf_counter()
            p_lat = ctx.prove_witness(key, dr, z_page, r, s)
            lats_p.append(time.perf_counter() - t1)
            if p_lat.tobytes() != want[0]:
                raise AssertionError('bench: the proof from a pageable witness differs from the pipelined proof')
        lat_page_ms = sorted(lats_p)[len(lats_p) // 2] * 1e3
        del z_page
    else:
        barrier()
        t1 = time.perf_counter()
        for _ in range(dev_steps):
            p_dev = prove_multi(d_z0)
        barrier()
        dev_ms = (time.perf_counter() - t1) / dev_steps * 1e3
        if p_dev.tobytes() != want[0]:
            raise AssertionError('bench: device-resident proof differs from the host-witness proof')

    # ---- not `value`: does the tiled WITNESS flatter the number?  The benchmark's witness is `distinct_transactions` (32) real transaction
    # witnesses dealt over the copies, so every scalar has ~54 twins that meet in the same buckets.  The same system is timed here with a
    # device-generated witness of the same zero / one fractions in which every dense value is DISTINCT (fk_gen_scalars_dev kind 2).  It does
    # not satisfy the constraints -- the proof is well defined and worthless: a TIMING leg, labelled so -- but the sorts, the bucket loads
    # and the accumulations see what 1741 unrelated transactions would give them.
    sensitivity = None
    if not multi and copies is not None and not args.no_standalone and leg_fits('witness_sensitivity', 5 * full_size):
        d_zs = ctx.dev_alloc(nv * 32)
        ctx.gen_scalars_dev(d_zs, nv, 20261004, 2)
        ctx.upload(d_zs, z_pin[0][:1])                     # z[0] = ONE
        ctx.prove_witness_dev(key, dr, d_zs, r, s)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(dev_steps):
            ctx.prove_witness_dev(key, dr, d_zs, r, s)
        s_ms = (time.perf_counter() - t1) / dev_steps * 1e3
        ctx.prove_witness_dev(key, dr, d_z0, r, s)         # (back to the real witness: warm lanes for the legs below)
        ctx.dev_free(d_zs)
        sensitivity = {'all_distinct_values_device_resident_ms_per_step': s_ms, 'tiled_witness_device_resident_ms_per_step': dev_ms,
                       'ratio': s_ms / dev_ms, 'steps': dev_steps,
                       'is': 'TIMING ONLY (the assignment is not satisfying, the proof is not valid): the same key and system proved from a device-generated '
                             'witness with the same 5.3 %% zeros / 2.5 %% ones and every other value distinct, against the benchmark witness (%d distinct '
                             'transactions dealt over %d copies), both resident in HBM, one proof at a time' % (len(zs), copies)}
        leg_done('witness_sensitivity')

    # ---- not `value`: the other resident form of the SAME circuit, same key, same witnesses, same proof bytes.
    #   default (`value` on the explicit system out of the Parameters image): `tiled` = fk_r1cs_load_tiled, ONE instance + a copy count
    #     (4.4 MB of matrices instead of 13.9 GB; rounds 1-4 quoted this form) -- the host-witness pipeline timed exactly like `value`;
    #   --tiled-headline: `untiled` = the explicit system built on the host (fk_r1cs_load_coded), witness resident.
    untiled = tiled = None
    if params_form and not multi and first_proof != want[0]:
        raise AssertionError('bench: the first proof after load_parameters differs from the pipelined proofs')
    if params_form and not multi and not args.no_untiled and leg_fits('tiled', 10 * full_size):
        t1 = time.perf_counter()
        dr_t = ctx.load_r1cs(r1cs, copies=copies)
        t_load = time.perf_counter() - t1
        p_t = ctx.prove_witness_dev(key, dr_t, d_z0, r, s)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(dev_steps):
            p_t = ctx.prove_witness_dev(key, dr_t, d_z0, r, s)
        t_dev_ms = (time.perf_counter() - t1) / dev_steps * 1e3
        if p_t.tobytes() != want[0]:
            raise AssertionError('bench: the proof from the tiled system differs from the one from the Parameters image')
        tk = ctx.prove_witness_submit(key, dr_t, z_pin[0], r, s)
        t_steps = max(4, min(args.steps, 10))
        for i in range(2 + t_steps):
            if i == 2:
                ctx.sync()
                t1 = time.perf_counter()
            nxt = ctx.prove_witness_submit(key, dr_t, z_pin[(i + 1) & 1], r, s)
            p_t = ctx.prove_witness_wait(tk).tobytes(); tk = nxt
            if want[i & 1] is not None and p_t != want[i & 1]:
                raise AssertionError('bench: pipelined proof from the tiled system differs from the one from the Parameters image')
        ctx.sync()
        t_ms = (time.perf_counter() - t1) / t_steps * 1e3
        ctx.prove_witness_wait(tk)
        tiled = {'ms_per_step': t_ms, 'proofs_per_sec': 1e3 / t_ms, 'steps': t_steps, 'device_resident_ms_per_step': t_dev_ms,
                 'explicit_ms_per_step': elapsed / args.steps * 1e3, 'explicit_device_resident_ms_per_step': dev_ms,
                 'matrix_bytes_resident': int(sum(dr_t.info()['nnz']) // copies) * 8, 'load_seconds': t_load,
                 'is': 'fk_r1cs_load_tiled of ONE transaction + the copy count %d (the form rounds 1-4 quoted; no `Parameters` object can express it), same key, '
                       'same two witnesses through the same two-slot pipeline; proof bytes equal to the explicit form\'s' % copies}
        dr_t.free()
        leg_done('tiled')
    if not multi and copies is not None and not params_form and not args.no_untiled and leg_fits('untiled', 20 * full_size):
        t1 = time.perf_counter()
        u_in, u_aux, u_mats, u_table = materialise_rollup(copies)
        t_build = time.perf_counter() - t1
        t1 = time.perf_counter()
        dr_u = ctx.load_r1cs_coded(u_in, u_aux, u_mats, u_table)
        t_load = time.perf_counter() - t1
        del u_mats
        ctx.prove_witness_dev(key, dr_u, d_z0, r, s)
        ctx.sync()
        t1 = time.perf_counter()
        for _ in range(dev_steps):
            p_u = ctx.prove_witness_dev(key, dr_u, d_z0, r, s)
        u_ms = (time.perf_counter() - t1) / dev_steps * 1e3
        if p_u.tobytes() != want[0]:
            raise AssertionError('bench: the proof from the materialised system differs from the tiled one')
        ui = dr_u.info()
        untiled = {'device_resident_ms_per_step': u_ms, 'tiled_device_resident_ms_per_step': dev_ms, 'matrix_terms_resident': int(sum(ui['nnz'])),
                   'matrix_bytes_resident': int(sum(ui['nnz'])) * 8 + 3 * 8 * (n + 1), 'host_build_seconds': t_build, 'load_seconds': t_load,
                   'is': 'fk_r1cs_load_coded of the explicitly replicated %d-transaction system (CSR with one 8-byte entry per term), same key, same '
                         'witness, witness resident; proof bytes equal to the tiled form' % copies}
        dr_u.free()
        leg_done('untiled')
    standalone = None
    if not multi and not args.no_standalone and leg_fits('standalone', 10 + 30 * full_size):
        standalone = standalone_legs(ctx, key, m)
        leg_done('standalone')

    # ---- N > 1: throughput mode, one whole proof per GPU (replicas of the single-GPU prover; no collective in the data path)
    replica = None
    replica_skipped = None
    if world > 1 and not args.no_replicas and os.environ.get('FK_BENCH_SAME_DEVICE') == '1':
        # rehearsal with every rank on ONE GPU: `world` whole keys with their fixed-base levels must fit beside each other (never the
        # case at the benchmark size; on a node every rank has a GPU of its own).  Decided from sizes every rank knows: no rank may
        # enter the leg's collectives alone.
        need = world * (384 * m * 13 + nv * 64 + m * 32 * 20 + (2 << 30))      # key + levels (worst case), witness slots, lane / quotient scratch
        if need > 0.9 * torch.cuda.get_device_properties(local_rank).total_memory:
            replica_skipped = 'FK_BENCH_SAME_DEVICE=1: %d whole keys with their levels do not fit one GPU at this size' % world
    if world > 1 and not args.no_replicas and replica_skipped is None and leg_fits('replicas', 20 + 60 * full_size):
        import torch.distributed as dist
        key.free()
        if shared_image is not None:                              # the whole key on every GPU, out of the same image
            key = ctx.load_key_bellman(pio.read_parameters(image)['bellman'], flags=fk.api.FK_KEY_CHECKED)[0]
        else:
            key, _ = ctx.setup(r1cs, copies=copies, **tox)
        ctx.prove_witness_dev(key, dr, d_z0, r, s)
        barrier()
        t1 = time.perf_counter()
        state['ticket'] = ctx.prove_witness_submit(key, dr, z_pin[0], r, s)
        for i in range(dev_steps):
            nxt = ctx.prove_witness_submit(key, dr, z_pin[(i + 1) & 1], r, s)
            p_rep = ctx.prove_witness_wait(state['ticket'])
            state['ticket'] = nxt
        ctx.prove_witness_wait(state['ticket']); state['ticket'] = None
        barrier()
        rep_t = torch.tensor([time.perf_counter() - t1], dtype=torch.float64)
        dist.all_reduce(rep_t, op=dist.ReduceOp.MAX)
        if p_rep.tobytes() != want[(dev_steps - 1) & 1] and want[(dev_steps - 1) & 1] is not None:
            raise AssertionError('bench: replica proof differs from the distributed proof')
        replica = world * (dev_steps + 1) / float(rep_t.item())
        leg_done('replicas')
    ctx.dev_free(d_z0)

    # ---- N > 1: the one-call form (one process drives all GPUs through fk_multi_prove_r1cs).  Every rank releases its GPU memory
    # first; the other ranks wait on a host-side (gloo) barrier so that nothing of theirs runs on the GPUs meanwhile.
    single_proc = None
    if world > 1 and not args.no_single_process and leg_fits('single_process_multi_gpu', 30 + 90 * full_size):
        import torch.distributed as dist
        same_dev = os.environ.get('FK_BENCH_SAME_DEVICE') == '1'
        key.free(); key = None
        dr.free(); dr = None
        work = send = recv = h_full_buf = recv_buf = a2a = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        hostgrp = None                      # (the default group IS the host-side gloo group)
        barrier()
        if rank == 0:
            try:
                single_proc = single_process_leg(fk, world, same_dev, r1cs, copies, z_pin, tox, r, s, want, dev_steps,
                                                 image=image if shared_image is not None else None)
            except Exception as e:     # noqa: BLE001 -- reported, the rank-per-GPU result above stands
                single_proc = {'error': '%s: %s' % (type(e).__name__, e)}
        dist.barrier(group=hostgrp)
        leg_done('single_process_multi_gpu')

    out = None
    if rank == 0:
        sec_per_step = elapsed / args.steps
        msm_units = (m - 1) + num_aux + n_a + 2 * n_b       # scalar-muls per proof: H, L, A, B1 (G1) and B2 (G2)
        acc = stats['acc_g1']
        # The G1 accumulations of B1, L and A run side by side on three lanes (sorts-first schedule), so each launch's event pair
        # spans the whole phase: the kernel's time is the UNION of the launches' intervals (fk_stats_get 5), their sum counts the
        # phase three times.  (rocprofv3's per-launch durations are that sum; tools/trace_union.py gives the union of a trace.)
        acc_s = acc.get('union_ms', acc['ms']) * 1e-3
        # dominant kernel: G1 bucket accumulation; achieved = algorithmic bytes / its HIP-event time
        achieved = (acc['units'] * G1_BYTES_PER_SCALAR_MUL) / acc_s / 1e9 if acc_s > 0 else 0.0
        traffic, traffic_src, traffic_err = pmc_traffic(args, log_m, n, world, acc, acc_s, achieved)
        cal = ctx.calibrate()
        modmul = acc['adds'] * MODMUL_PER_G1_MIXED_ADD
        valu_peak = cal['mad_lane_ops_per_s'] / MUL_PIPE_OPS_PER_MODMUL
        merged = bool(pre_levels.get('h'))
        kname = '%s<Fq> (G1 bucket accumulation)' % ('msm_accumulate_merged_kernel' if merged else 'msm_accumulate_kernel')
        fill = n / float(m)
        rl_g1 = hbm_block(kname, acc['units'], G1_BYTES_PER_SCALAR_MUL, acc.get('union_ms', acc['ms']), acc['ms'], acc['launches'], 'scalar_mul',
                          note='quoted against HBM as north_star asks.  96 B feed ~12 mixed additions = ~120 modular products = ~4e4 integer instructions, so the '
                               'algorithmic-byte fraction is necessarily ~2 %; the traffic above it is structural (every base is gathered once per window from its '
                               'fixed-base level: W x 64 B + scalar)')
        rl_g1.update(traffic=traffic, traffic_ratio=(traffic / achieved if (traffic and achieved > 0) else None), traffic_source=traffic_src,
                     kernel_time_is='`achieved` / `frac`: union of the launches\' HIP-event intervals (launches of B1, L and A run side by side: %.1f ms per step as a '
                                    'union, %.1f ms as the sum of the launches\' own durations); `achieved_per_launch` / `frac_per_launch`: the average launch\'s bytes / '
                                    'the average launch\'s own duration' % (acc_s * 1e3 / args.steps, acc['ms'] / args.steps),
                     binding_resource='VALU integer multiplier, not HBM: see roofline_valu')
        a2 = stats['acc_g2']
        rl_g2 = hbm_block('%s<Fq2> (G2 bucket accumulation)' % ('msm_accumulate_merged_kernel' if bool(pre_levels.get('b_g2')) else 'msm_accumulate_kernel'),
                          a2['units'], 160, a2.get('union_ms', a2['ms']), a2['ms'], a2['launches'], 'scalar_mul',
                          note='SURVEY 8(d): 160 B per G2 scalar-mul (128 B affine base + 32 B scalar); VALU-bound like the G1 kernel (28 base-field products per mixed addition)')
        nt = stats['ntt']
        rl_ntt = hbm_block('ntt_pass_kernel<Fr> (all passes of the quotient\'s 6 transforms)', 6 * m * args.steps, 64, nt['ms'], nt['ms'], nt['launches'], 'element_transform',
                           note='SURVEY 8(d): 64 B per element per transform (one ideal pass) x the 6 transforms of a quotient / the time of all their passes; a 2^%d '
                                'transform is %d passes, so the data MOVED is %d x that (kernel_ms_per_step.ntt_algorithmic_GBps)' % (log_m, (log_m + 8) // 9, (log_m + 8) // 9))
        if args.workload == 'rollup1024':
            wl = ('%d rollup-style transactions (two depth-32 poseidon merkle proofs + one eddsa-poseidon signature each; a composition of the '
                  'reference\'s gadgets, tests/golden/rollup_tx_instance.npz) as ONE R1CS of %d rows = %.2f %% of the 2^%d domain (BASELINE configs[3] shape; '
                  'BASELINE.md config 4: rows = 2^25), %s' % (copies, n, 100.0 * fill, log_m,
                  'handed to the prover as a `Parameters` image (brotli gate blob + bellman key, mod.rs:150-175): fk_gates_decode -> fk_r1cs_load_gates '
                  '(all %.3g matrix terms explicit in HBM) + fk_key_load_bellman(checked)' % float(sum(info['nnz'])) if params_form else
                  'through fk_setup_tiled / fk_r1cs_load_tiled (one instance + a copy count)'))
        else:
            wl = 'synthetic satisfiable R1CS, 2^%d rows, 1-2 term rows%s' % (log_m, ' / %d-term product gates' % args.lc_terms if args.lc_terms > 1 else '')
        out = {
            'metric': 'Groth16 proofs/sec + MSM scalar-muls/sec, BN254, 2^%d constraints (%d rows = %.2f %% of the 2^%d domain)' % (log_m, n, 100.0 * fill, log_m),
            'value': args.steps / elapsed,
            'unit': 'proofs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            # what carried the exchanges of THIS run: 'rccl' = torch.distributed backend nccl over xGMI (the N > 1 default), 'gloo' = host-staged (rehearsals on
            # one GPU), None = a single rank without a process group
            'transport': (None if not multi else 'rccl' if args.backend == 'nccl' else args.backend), 'rccl_world_size': (world if (multi and args.backend == 'nccl') else None),
            'ms_per_step': sec_per_step * 1e3,
            'higher_is_better': True,
            'scaling': 'strong',
            'vs_baseline': None,
            'dtype': 'u32',
            'data': 'synthetic',
            'config': {'workload': wl + '; per step: witness from pinned HOST memory (two-slot pipeline, two distinct witnesses alternating) -> device '
                                      'SpMV + quotient (6 NTTs) + G1 MSMs H/L/A/B1 + G2 MSM B2 + assembly; constraint system and valid key resident in HBM',
                       'matrix_form': ('explicit, from a Parameters gate blob' if params_form else 'tiled (one instance + copy count)' if copies is not None else 'explicit'),
                       'log2_domain': log_m, 'rows': n, 'domain_fill': fill, 'num_input': num_input, 'num_aux': num_aux,
                       'nnz': list(info['nnz']), 'witness_bytes_per_proof': nv * 32,
                       'distinct_transactions': (len(zs) if copies is not None else None),
                       'distinct_witnesses_in_the_pipeline': 2 if two_witnesses else 1,
                       'a_query_points': n_a, 'b_query_points': n_b,
                       'msm_points': {'h': m - 1, 'l': num_aux, 'a': n_a, 'b_g1': n_b, 'b_g2': n_b},
                       'msm_fixed_base_levels': pre_levels, 'levels_plan': levels_plan,
                       'witness': '%.1f%% zeros, %.1f%% ones, rest dense 254-bit' % (100.0 * zeros / nv, 100.0 * ones / nv),
                       'witness_hand_over': None if not multi else dict(state.get('wit') or {}, what='per rank and proof: this rank\'s 1 / N piece over its own PCIe link '
                                                                                      '(pcie_bytes), the rest collected from the peers by dist.all_gather_into_tensor (RCCL over xGMI) on '
                                                                                      'the library\'s copy stream, underneath the proof before (gathered_bytes)'),
                       'witness_array_split': None if world == 1 else ('quotient on rank 0: all of h and a piece of the work line on rank 0, witness pieces elsewhere (FK_Z_WORK_SPLIT_Q0); nothing but 384-byte sums exchanged' if q0_split else
                                                                        'fractions (balanced schedule)' if not dist_q else 'equal' if os.environ.get('FK_MULTI_SPLIT') == 'equal' else
                                                                        'by work: l | a | b_g1 | b_g2 cut into N equal pieces of work (FK_Z_WORK_SPLIT)'),
                       'parallelism': 'msm-shard%d%s' % (world, '' if not multi else
                                                         '+distributed-quotient (7 all-to-all per proof)' if dist_q else '+balanced-quotient')},
            'msm_scalar_muls_per_sec': msm_units / sec_per_step,
            'msm_scalar_muls_per_sec_counts': 'every (scalar, base) pair of the five multiplications, trivial scalars (0 and 1) included, divided by the '
                                              'WHOLE proof time; the multiplications timed alone (min / median / max of >= 10 repetitions) are under `standalone`',
            'device_resident_ms_per_step': dev_ms,
            'latency_ms_per_proof': None if multi else lat_ms,
            'latency_pageable_ms_per_proof': None if multi else lat_page_ms,
            'latency_ms_per_proof_is': 'ONE proof at a time from the witness in pinned host memory (fk_prove_r1cs: the witness goes up in the pieces of fk_r1cs_windows, '
                                       'row window j is evaluated behind piece j, then the proof); `latency_pageable_ms_per_proof` the same call on a pageable buffer; '
                                       '`ms_per_step` is the pipelined rate, `device_resident_ms_per_step` the same without the upload',
            'roofline': rl_g1,
            'roofline_g2': rl_g2,
            'roofline_ntt': rl_ntt,
            'roofline_valu': {
                'bound': 'valu-int-mul', 'kernel': kname,
                'achieved': modmul / acc_s / 1e9 if acc_s > 0 else 0.0, 'peak': valu_peak / 1e9, 'unit': 'G modmul/s',
                'frac': (modmul / acc_s) / valu_peak if acc_s > 0 else 0.0,
                'mixed_additions': acc['adds'], 'modmul_per_mixed_addition': MODMUL_PER_G1_MIXED_ADD,
                'peak_is': 'v_mad_u64_u32 issue rate measured live on this device (fk_calibrate: %.2f T lane-ops/s) / %d multiplier operations '
                           'per 256-bit Montgomery product (the minimum for 32-bit limbs)' % (cal['mad_lane_ops_per_s'] / 1e12, MUL_PIPE_OPS_PER_MODMUL),
                'multiplier_alone': cal['modmul_per_s'] / 1e9,
                'frac_of_multiplier_alone': (modmul / acc_s) / cal['modmul_per_s'] if acc_s > 0 else 0.0,
                'multiplier_alone_is': 'the library\'s Montgomery product looping in registers, measured live (fk_calibrate), G modmul/s: what the '
                                       'carry handling around the multiplies (v_addc, moves, hazard padding) leaves of the peak',
            },
            'kernel_ms_per_step': {
                'msm_accumulate_g1': stats['acc_g1'].get('union_ms', stats['acc_g1']['ms']) / args.steps,
                'msm_accumulate_g1_sum_of_side_by_side_launches': stats['acc_g1']['ms'] / args.steps,
                'msm_accumulate_g2': stats['acc_g2'].get('union_ms', stats['acc_g2']['ms']) / args.steps,
                'ntt_passes': stats['ntt']['ms'] / args.steps,
                # one launch = one pass over 2^log2n elements, 64 B (read + write) each; a transform is ceil(log2n / 9) passes
                'ntt_algorithmic_GBps': (stats['ntt']['units'] * 64) / (stats['ntt']['ms'] * 1e-3) / 1e9 if stats['ntt']['ms'] > 0 else 0.0,
                'ntt_GBps_is': 'data MOVED per pass (64 B per element per pass; a 2^25 transform is three passes)',
                'ntt_sec8d_GBps': (6 * 64 * m) / (stats['ntt']['ms'] / args.steps * 1e-3) / 1e9 if stats['ntt']['ms'] > 0 else 0.0,
                'ntt_sec8d_GBps_is': 'SURVEY 8(d): 64 B per element per transform x the 6 transforms of a quotient / the time of all their passes',
            },
            'prep_seconds': prep_s,
        }
        if traffic_err:
            out['roofline']['traffic_error'] = traffic_err
        if replica_skipped is not None:
            out['replica_skipped'] = replica_skipped
        if replica is not None:
            out['replica_proofs_per_sec'] = replica
            out['replica_proofs_per_sec_is'] = 'throughput mode: every GPU holds the whole key and proves its own witnesses (host-witness pipeline), no collective'
        if preflight is not None:
            out['preflight'] = preflight
        if multi_load is not None:
            out['load'] = multi_load
        dg = check_digest(copies, two_witnesses, zs if copies is not None else None, want)
        if dg is not None:
            out['oracle_digest_check'] = dg
        import hashlib
        out['proof_sha256'] = [hashlib.sha256(w).hexdigest()[:16] for w in want if w is not None]      # the bytes are a function of (Parameters, witness, r, s) alone: equal across runs, boxes and library builds
        if not args.no_cpu_baseline:
            out['proof_verified_by_pairing_check'] = bool(pairing_check(vk, z_inputs[0], want[0]) and
                                                          (want[1] is None or pairing_check(vk, z_inputs[1], want[1])))
        if sensitivity is not None:
            out['witness_sensitivity'] = sensitivity
        if load_block is not None:
            out['load'] = load_block
        if tiled is not None:
            out['tiled'] = tiled
        if untiled is not None:
            out['untiled'] = untiled
        if standalone is not None:
            out['standalone'] = standalone
        if single_proc is not None:
            out['single_process_multi_gpu'] = single_proc
        if not multi and not args.no_cpu_baseline and leg_fits('cpu_baseline', 60):
            full = (key, vk, z_pin[0], r, s, want[0]) if copies is not None else None
            # the full-size CPU proof gets what the plan leaves of the budget (minus the legs still to come), at most --cpu-full-budget
            left = args.max_seconds - (time.time() - t_start) - 60 - (0 if args.no_other_sizes else 150 * full_size)
            args.cpu_full_budget = max(0.0, min(args.cpu_full_budget, left))
            cb = cpu_baseline_leg(ctx, fk, args, full=full)
            leg_done('cpu_baseline')
            best = cb['best_threads']
            scaled = lambda th: (cb['synth_s'] + cb['prove_s'][th]) * cb['scale']
            if cb['full'] is not None:
                f_ = cb['full']
                secs = f_['synth_s'] + f_['prove_s']
                sample = ('MEASURED AT FULL SIZE: oracle/groth16_oracle.c (bellman\'s algorithm restated; %d threads = bellman\'s multicore split: parallel_fft + '
                          'one task per multiexp region; synthesis serial as in the reference) on the benchmarked system itself -- %s, %d rows, domain 2^%d: synthesis '
                          '%.1f s + proof %.1f s; its 256 proof bytes equal the GPU\'s.  Thread count chosen on a %d-transaction sample (2^%d): %s'
                          % (best, '%d rollup-style transactions as one R1CS' % args.copies, n, f_['log2_m'], f_['synth_s'], f_['prove_s'], args.cpu_copies, cb['log2_m'],
                             ', '.join('%d thr %.2f s' % (th, t) for th, t in sorted(cb['prove_s'].items()))))
            else:
                secs = scaled(best)
                sample = ('SCALED from a sample (%s): oracle/groth16_oracle.c with %d threads on %s (domain 2^%d): synthesis %.2f s + proof %.2f s, scaled '
                          'linearly by %g (optimistic for the CPU: the NTT is n log n); the GPU proof of the same sample matched byte for byte'
                          % (cb.get('full_skipped', ''), best, cb['what'], cb['log2_m'], cb['synth_s'], cb['prove_s'][best], cb['scale']))
            out['cpu_baseline'] = {
                'value': 1.0 / secs, 'unit': 'proofs/s', 'cores': best, 'kind': 'port', 'sample': sample,
                'seconds_per_proof': secs, 'measured_at_full_size': cb['full'] is not None, 'usable_cores': cb['cores'], 'host': cb['host'],
                'sample_seconds_by_threads': {str(th): t for th, t in sorted(cb['prove_s'].items())},
                'single_thread': {'value': 1.0 / scaled(1), 'unit': 'proofs/s', 'cores': 1,
                                  'sample_seconds': cb['synth_s'] + cb['prove_s'][1],
                                  'note': 'the worker fawkes-crypto configures (SURVEY fact 3): the sample on one thread, scaled linearly by %g (extrapolated)' % cb['scale']},
            }

    # ---- not `value`: the same prover at other transaction counts (N = 1): the 1024-transaction system of rounds 2-3, and a system of at
    # least the reference's published size (35 695 616 constraints, README.md:54-56: 628 s on an i9-9900K) on the 2^26 domain
    if rank == 0 and not multi and copies is not None and not args.no_other_sizes:
        key.free(); key = None
        dr.free(); dr = None
        for zp in z_pin:
            ctx.host_free(zp)
        z_pin = []
        for tag, cp in (('secondary_1024_transactions', args.secondary_copies), ('reference_published', args.reference_copies)):
            if cp <= 0 or cp == copies or not leg_fits(tag, 25 + 50 * full_size * (cp / float(copies))):
                continue
            try:
                leg = other_size_leg(ctx, r1cs, zs, cp, tox, r, s, max(3, min(args.steps, 6)), check=not args.no_cpu_baseline)
            except Exception as e:     # noqa: BLE001 -- reported; the headline above stands
                leg = {'error': '%s: %s' % (type(e).__name__, e)}
            if tag == 'reference_published':
                leg['reference'] = ('the reference\'s one published figure: rollup, 1024 txs, 35 695 616 constraints, 628 s on an i9-9900K '
                                    '(/root/reference/README.md:54-56; thread count and bellman features not stated).  This leg: a rollup-STYLE system '
                                    'of at least that many rows (the reference\'s own rollup circuit is not in its repository), same domain 2^26')
                leg['reference_seconds_per_proof'] = 628.0
            out[tag] = leg
            leg_done(tag)
    # ---- not `value`: HBM traffic of the accumulations and the transforms from the PMC counters, measured by THIS run (child processes under
    # rocprofv3; everything of this process is released first: two provers of this size do not fit one GPU)
    mt_on = args.measure_traffic == 'on' or (args.measure_traffic == 'auto' and n >= (1 << 23))
    if out is not None and not multi and mt_on and leg_fits('measure_traffic', 60 + 200 * full_size):
        if key is not None:
            key.free(); key = None
        if dr is not None:
            dr.free(); dr = None
        for zp in z_pin:
            ctx.host_free(zp)
        z_pin = []
        ctx.trim()
        torch.cuda.empty_cache()
        pmc, err = measure_traffic_leg(args, log_m, (m - 1) + num_aux + n_a + n_b, n_b, n, min(420.0, max(30.0, args.max_seconds - (time.time() - t_start) - 20)))
        leg_done('measure_traffic')
        if pmc is None:
            out['roofline']['traffic_measured_error'] = err
        else:
            # FETCH_SIZE corrections (gfx950): x1.00 for the G1 kernel's 64-byte gathers, x1.98 for 128-byte ones, x2.00 for wide coalesced streams --
            # calibrated with tools/mulbench/gathercal (profiles/r05f_pmc_traffic_bench_rollup1741.json: fetch_size_calibration) in line with
            # /opt/skills/guides/MI355X_MICROARCH.md; WRITE_SIZE is exact
            pp = max(int(pmc['proofs_in_the_pass']), 1)
            dk = pmc['dominant_kernel']
            per_pt = (dk['fetch_bytes_per_proof_raw'] * 1.0 + dk['write_bytes_per_proof']) / dk['points_per_proof']
            rl = out['roofline']
            rl['traffic_imported'] = {'GBps': rl['traffic'], 'source': rl.get('traffic_source')}
            rl['traffic'] = per_pt * acc['units'] / acc_s / 1e9
            rl['traffic_ratio'] = rl['traffic'] / rl['achieved'] if rl['achieved'] > 0 else None
            rl['traffic_bytes_per_scalar_mul'] = per_pt
            rl['traffic_source'] = ('MEASURED BY THIS RUN: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE over one-step child runs of this command (%d proofs in each pass, '
                                    'derived from its dispatches; %.0f s); bytes per (scalar, base) pair of the pass x this run\'s pairs / this run\'s kernel time; FETCH_SIZE x 1.00 '
                                    '(64-byte gathers)' % (pp, pmc['seconds']))
            rl.pop('traffic_error', None)
            for blk, prefix, corr, per_proof_units in ((out['roofline_g2'], 'msm_accumulate', 1.98, n_b), (out['roofline_ntt'], 'ntt_pass_kernel', 2.0, 6 * m)):
                ks = [k for k in pmc['per_kernel'] if k['kernel'].startswith(prefix) and ('Fq2T' in k['kernel']) == (blk is out['roofline_g2'])]
                if not ks or per_proof_units <= 0:
                    continue
                byts = sum(k['FETCH_SIZE_KB'] * 1024 * corr + k['WRITE_SIZE_KB'] * 1024 for k in ks) / pp
                if blk is out['roofline_ntt'] and pmc['per_kernel']:
                    byts = byts * (6 * ((log_m + 8) // 9)) / max(sum(k['launches'] for k in ks) / pp, 1)      # the quotient's passes only (the key set-up adds one or two transforms)
                blk['traffic_bytes_per_unit'] = byts / per_proof_units
                blk['traffic_ratio'] = blk['traffic_bytes_per_unit'] / (160 if blk is out['roofline_g2'] else 64)
                blk['traffic'] = blk['achieved'] * blk['traffic_ratio']
                blk['traffic_source'] = 'the same two passes; FETCH_SIZE x %.2f (%s)' % (corr, '128-byte gathers' if corr < 2 else 'wide coalesced streams')
    if out is not None:
        out['legs'] = dict(legs, total_seconds=round(time.time() - t_start, 1), max_seconds=args.max_seconds,
                           what='seconds each optional leg took, or why it was skipped (--max-seconds: the JSON line is printed whatever happens to the extras)')
    line = json.dumps(out) if out is not None else None

    for zp in z_pin:
        ctx.host_free(zp)
    if dr is not None:
        dr.free()
    if key is not None:
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
