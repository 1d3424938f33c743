"""The oracle at the sizes where the big-size code paths engage (the direct oracle comparisons used to stop at NTT 2^16,
G1 MSM 4096 points, G2 700 points, whole proofs on a 2^17 domain; everything larger was checked by properties only):

  * BASELINE configs[1] in full: fk_msm_g1_dev on 2^20 points, uniform and witness-like scalars; fk_msm_g2_dev on 2^18 points; fk_ntt at 2^19
    and 2^20 (the three-pass transform), all four inverse / coset variants -- every byte against oracle/groth16_oracle.c (bellman's multiexp
    and EvaluationDomain restated; its multicore split is used for speed, same results);
  * one whole proof on a 2^21 domain (100 rollup-style transactions, 1.93 M rows, 94 M matrix terms) with the DEFAULT fixed-base-level
    threshold, once with the schedule a small domain runs and once with FK_PROVE_SORTS_FIRST=1 (the benchmark-size schedule: sorts first,
    early front in the pipeline) in child processes -- the 256 proof bytes against the oracle's;
  * fk_prove_msm_array_dev over a 2^21 - 1-point key array WITH its fixed-base levels (one merged bucket set) against the oracle's multiexp.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _threads():
    import bench
    return min(16, bench.usable_cores())


@pytest.mark.parametrize('kind', [0, 1], ids=['uniform', 'witness_like'])
def test_config1_g1_msm_2p20_vs_oracle(ctx, oracle, kind):
    n = 1 << 20
    d_b, d_s = ctx.dev_alloc(n * 64), ctx.dev_alloc(n * 32)
    try:
        ctx.gen_points_g1_dev(d_b, n, 1101)
        ctx.gen_scalars_dev(d_s, n, 1300 + kind, kind)
        got = ctx.msm_g1_dev(d_b, d_s, n)
        bases = ctx.download(d_b, n * 64, np.uint8).reshape(n, 64)
        scalars = ctx.download(d_s, n * 32, np.uint64).reshape(n, 4)
    finally:
        ctx.dev_free(d_b); ctx.dev_free(d_s)
    if kind == 1:
        one = np.frombuffer(((1 << 256) % 21888242871839275222246405745257275088548364400416034343698204186575808495617).to_bytes(32, 'little'), np.uint64)
        trivial = (~scalars.any(axis=1)).sum() + (scalars == one).all(axis=1).sum()
        assert 0.4 * n < trivial < 0.6 * n                 # "witness-like": about half of the scalars are 0 or 1
    want = oracle.msm_g1(bases, scalars, threads=_threads())
    assert got.tobytes() != bytes(64) and got.tobytes() == want.tobytes()


def test_config1_g2_msm_2p18_vs_oracle(ctx, oracle):
    n = 1 << 18
    d_b, d_s = ctx.dev_alloc(n * 128), ctx.dev_alloc(n * 32)
    try:
        ctx.gen_points_g2_dev(d_b, n, 1202)
        for kind in (0, 1):
            ctx.gen_scalars_dev(d_s, n, 1400 + kind, kind)
            got = ctx.msm_g2_dev(d_b, d_s, n)
            bases = ctx.download(d_b, n * 128, np.uint8).reshape(n, 128)
            scalars = ctx.download(d_s, n * 32, np.uint64).reshape(n, 4)
            assert got.tobytes() == oracle.msm_g2(bases, scalars, threads=_threads()).tobytes(), kind
    finally:
        ctx.dev_free(d_b); ctx.dev_free(d_s)


@pytest.mark.parametrize('log_n', [19, 20])
def test_config1_ntt_three_pass_sizes_vs_oracle(ctx, oracle, log_n):
    n = 1 << log_n
    d = ctx.dev_alloc(n * 32)
    try:
        ctx.gen_scalars_dev(d, n, 1500 + log_n, 0)
        x = ctx.download(d, n * 32, np.uint64).reshape(n, 4)
        for inverse in (False, True):
            for coset in (False, True):
                ctx.upload(d, x)
                ctx.ntt_dev(d, log_n, inverse=inverse, coset=coset)
                got = ctx.download(d, n * 32, np.uint64).reshape(n, 4)
                want = oracle.fr_ntt(x, inverse=inverse, coset=coset, threads=_threads())
                assert np.array_equal(got, want), (log_n, inverse, coset)
    finally:
        ctx.dev_free(d)


def _child(copies, env):
    e = dict(os.environ)
    for k in list(e):
        if k.startswith(('FK_MSM_', 'FK_NTT_', 'FK_PROVE_', 'FK_SPMV_', 'FK_DEBUG')):
            del e[k]
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', '_bigproof_child.py'), str(copies)], env=e, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (env, out.stderr[-2000:])
    line = [l for l in out.stdout.splitlines() if l.startswith('PROOF ')]
    assert len(line) == 1, out.stdout[-1000:]
    _, proof, levels = line[0].split(' ', 2)
    return proof, eval(levels)


def test_whole_proof_2p21_sorts_first_default_levels(ctx, oracle):
    import bench
    import c_oracle as co
    copies = 100
    inst, zs = bench.load_rollup_instance()
    z = bench.tile_witness(zs, inst.num_input, copies)
    tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
    r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
    key, vk = ctx.setup(inst, copies=copies, **tox)
    try:
        cnt = key.counts()
        assert cnt['m'] == 1 << 21
        pre = key.precomputed()
        assert pre['h'] > 0 and pre['l'] > 0 and pre['a'] > 0, pre            # default threshold: levels from ~2^21 points on
        one = co.R1csC(inst.num_input, inst.num_aux, *[co.Csr(p_, c_, v_) for p_, c_, v_ in inst.mats])
        a, b, c, aa, bi, ba = co.synthesize_tiled(one, copies, z)
        okey = bench.oracle_key(key, vk, cnt['m'], cnt['num_input'], cnt['num_aux'])
        want = co.prove(okey, a, b, c, z, aa, bi, ba, r, s, threads=_threads()).tobytes().hex()
        # fk_prove_msm_array_dev over the h array (2^21 - 1 points) WITH its levels vs the oracle's multiexp of the same points
        n_h = cnt['n_h']
        d_s = ctx.dev_alloc(n_h * 32)
        ctx.gen_scalars_dev(d_s, n_h, 1600, 0)
        merged = ctx.prove_msm_array_dev(key, 'h', d_s)
        scalars = ctx.download(d_s, n_h * 32, np.uint64).reshape(n_h, 4)
        ctx.dev_free(d_s)
        assert merged.tobytes() == oracle.msm_g1(key.download('h'), scalars, threads=_threads()).tobytes()
    finally:
        key.free()
    for env in ({}, {'FK_PROVE_SORTS_FIRST': '1'}):
        proof, levels = _child(copies, env)
        assert levels == pre, (levels, pre)
        assert proof == want, env
