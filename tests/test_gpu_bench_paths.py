"""bench.py end to end at a small size (subprocesses): the default single-GPU path with its cpu_baseline leg, and the
multi-GPU code path rehearsed with ONE rank over real RCCL (FK_BENCH_REHEARSE=1: process group 'nccl', all_to_all_single /
all_gather on device tensors, distributed quotient and balanced schedule)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args, gpus=1):
    env = dict(os.environ, **extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(gpus), '--steps', '2', '--warmup', '1', *args],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    assert out.stdout.strip().splitlines()[-1] == lines[0], 'the JSON line must be the last line on stdout: ' + out.stdout[-500:]
    return json.loads(lines[0])


def _check_line(j):
    assert j['unit'] == 'proofs/s' and j['n_gpus'] == 1 and j['value'] > 0 and j['proof_verified_by_pairing_check'] is True
    assert j['roofline']['bound'] == 'hbm' and 0 < j['roofline']['frac'] < 1
    rv = j['roofline_valu']
    assert rv['mixed_additions'] > 0 and 0 < rv['frac'] < 1 and rv['peak'] > rv['multiplier_alone'] > 0
    cb = j['cpu_baseline']
    assert cb['kind'] == 'port' and 1 <= cb['cores'] <= (os.cpu_count() or 1) and cb['value'] > 0
    assert cb['single_thread']['cores'] == 1 and cb['single_thread']['value'] > 0
    assert j['device_resident_ms_per_step'] > 0
    assert j['config']['workload'] and 'model' not in j['config']


def test_bench_default_workload_small():
    """the default workload (rollup-style transactions from the committed fixture, handed over as a `Parameters` image and proved from the
    explicit system decoded out of its gate blob; host-witness pipeline) at 6 copies"""
    j = _run({}, '--copies', '6', '--cpu-copies', '2', '--secondary-copies', '3', '--reference-copies', '7')
    _check_line(j)
    assert 'rollup-style transactions' in j['config']['workload'] and j['config']['num_input'] == 1 + 6 * 2
    # rows and the domain are labelled as what they are ("2^25 constraints" was printed for 19.7 M rows)
    assert j['config']['rows'] == 6 * 19270 + j['config']['num_input'] and j['config']['log2_domain'] == 17
    assert abs(j['config']['domain_fill'] - j['config']['rows'] / 2.0 ** 17) < 1e-9 and '%d rows' % j['config']['rows'] in j['metric']
    assert j['config']['distinct_witnesses_in_the_pipeline'] == 2 and j['latency_ms_per_proof'] >= j['device_resident_ms_per_step'] * 0.8
    # no committed PMC pass matches a 6-transaction system: traffic is null and says why (never a number from another configuration)
    assert j['roofline']['traffic'] is None and j['roofline']['traffic_error']
    assert j['config']['witness_bytes_per_proof'] == (j['config']['num_input'] + j['config']['num_aux']) * 32
    # the CPU baseline is MEASURED at the benchmarked size when it fits the budget (here it does) and its proof equals the GPU's
    assert j['cpu_baseline']['measured_at_full_size'] is True and 'MEASURED AT FULL SIZE' in j['cpu_baseline']['sample']
    # round 5: `value` is measured on the explicit system out of the Parameters image; the `load` block says what setting the prover up cost
    assert j['config']['matrix_form'] == 'explicit, from a Parameters gate blob' and 'Parameters' in j['config']['workload']
    ld = j['load']
    assert ld['matrix_terms'] == sum(j['config']['nnz']) and ld['gates'] == 6 * 19270 and ld['gate_stream_bytes'] > 10 * ld['blob_bytes'] > 0
    assert ld['image_bytes'] == 4 + 4 + ld['blob_bytes'] + 4 + 4 + ld['bellman_bytes'] and ld['decode_seconds'] > 0 and ld['decode_profile']['parse_threads'] >= 1
    assert ld['time_to_first_proof_seconds'] >= ld['load_parameters_seconds'] > ld['decode_seconds'] and ld['host_rss_peak_bytes'] >= ld['host_rss_after_load_bytes'] > 0
    assert ld['key_read_checked_seconds'] > 0 and ld['write']['gates_encode_seconds'] > 0
    # the levels are planned underneath the decoding and re-checked once the system is resident; the proof bytes are named in the line
    assert ld['key_levels_early'] is True and ld['key_levels_headroom_GiB'] > 0 and ld['key_levels_replanned_seconds'] is None
    assert len(j['proof_sha256']) == 2 and j['proof_sha256'][0] != j['proof_sha256'][1] and j['latency_pageable_ms_per_proof'] > 0
    # ... and the tiled form of the same circuit (what rounds 1-4 quoted) is the secondary leg: same proof bytes (asserted by bench.py itself)
    assert j['tiled']['ms_per_step'] > 0 and j['tiled']['device_resident_ms_per_step'] > 0 and j['tiled']['explicit_ms_per_step'] == j['ms_per_step']
    # the witness shortcut is quantified (a timing-only leg with every dense value distinct), every optional leg is timed, the level planner reports
    ws = j['witness_sensitivity']
    assert ws['ratio'] > 0 and ws['all_distinct_values_device_resident_ms_per_step'] > 0 and 'TIMING ONLY' in ws['is']
    assert j['legs']['total_seconds'] > 0 and all(isinstance(j['legs'][k], float) for k in ('tiled', 'standalone', 'cpu_baseline', 'witness_sensitivity'))
    assert set(j['config']['levels_plan']) == {'h', 'l', 'a', 'b_g1', 'b_g2'}
    # a budget that is used up before the optional legs start: they are skipped with the reason, the line still comes out
    j3 = _run({}, '--copies', '6', '--max-seconds', '0.01')
    assert j3['value'] > 0 and all(str(j3['legs'].get(k)).startswith('skipped') for k in ('tiled', 'standalone', 'cpu_baseline')) and 'cpu_baseline' not in j3, j3['legs']
    # --tiled-headline restores the old arrangement: `value` on the tiled form, the explicit system (built on the host) as the `untiled` leg
    j2 = _run({}, '--copies', '6', '--tiled-headline', '--no-cpu-baseline', '--no-other-sizes', '--no-standalone')
    assert j2['config']['matrix_form'].startswith('tiled') and 'load' not in j2 and j2['untiled']['matrix_terms_resident'] == sum(j2['config']['nnz'])
    st = j['standalone']
    assert st['msm_g1_2p20']['scalar_muls_per_sec'] > 1e7 and st['msm_g2_2p20']['scalar_muls_per_sec'] > 1e6 and st['ntt_2p20']['algorithmic_GBps'] > 1
    for name in ('msm_g1_2p20', 'msm_g1_2p20_witness_like', 'msm_g2_2p20', 'msm_g1_2p17_key_bases', 'msm_g1_key_l_witness_like', 'msm_g2_key_b_g2', 'ntt_2p20'):
        e = st[name]
        assert e['wall_ms']['reps'] >= 10 and e['wall_ms']['min'] <= e['wall_ms']['median'] <= e['wall_ms']['max']
        assert e['hip_event_ms']['reps'] >= 10 and 0 < e['hip_event_ms']['median'] <= e['wall_ms']['median'] * 1.05 + 0.05
    # the legs at other transaction counts (here: tiny ones), each pairing-checked with two distinct witnesses
    for tag, cp in (('secondary_1024_transactions', 3), ('reference_published', 7)):
        assert j[tag]['transactions'] == cp and j[tag]['ms_per_step'] > 0 and j[tag]['proof_verified_by_pairing_check'] is True, j[tag]
        assert j[tag]['rows'] == cp * 19270 + 1 + cp * 2
    assert j['reference_published']['reference_seconds_per_proof'] == 628.0
    assert j['kernel_ms_per_step']['ntt_sec8d_GBps'] > 0


def _rank_run(n_ranks, copies, *extra):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'FK_DIST_QUOTIENT')}
    env['FK_BENCH_SAME_DEVICE'] = '1'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n_ranks), '--backend', 'gloo', '--steps', '2', '--warmup', '1',
                          '--copies', str(copies), *extra], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.fixture(scope='module')
def single_gpu_2p22():
    """the single-GPU line at 217 transactions (4 181 809 rows on the 2^22 domain), through the Parameters image: the bytes every rank count must reproduce"""
    j = _run({}, '--copies', '217', '--no-cpu-baseline', '--no-other-sizes', '--no-standalone', '--no-untiled')
    assert j['config']['log2_domain'] == 22 and j['config']['matrix_form'] == 'explicit, from a Parameters gate blob'
    return j


@pytest.mark.parametrize('n_ranks', [2, 4, 8])
def test_ranks_set_up_from_the_same_parameters_image_reproduce_the_single_gpu_bytes(single_gpu_2p22, n_ranks):
    """The schedules a first multi-GPU hardware run will take -- `python bench.py --gpus N` as a plain command (no launcher: the ranks are fresh children started
    before the parent touches the GPU, no re-exec), balanced quotient at 2 ranks, distributed quotient from 4 on, the one-call leg, at 4 ranks the replica
    leg -- and VERDICT r5 item 2: `bench.py --gpus N` proves the SAME input form as N = 1 -- rank 0 writes one `Parameters` image, every rank sets its prover
    up from it (load_parameters(image, shard = rank / world): fk_gates_decode once per process, fk_key_load_bellman(checked) of its slices), the
    one-call leg through fk_multi_key_load_bellman + fk_multi_r1cs_load_gates -- and the proof bytes (both witnesses) equal the single-GPU run's.
    2^22 rows, every rank-process on this box's one GPU (gloo)."""
    j = _rank_run(n_ranks, 217, *(() if n_ranks == 4 else ('--no-replicas',)))        # (4 ranks: with the throughput leg, one whole key per rank out of the image)
    assert j['n_gpus'] == n_ranks and j['config']['matrix_form'] == 'explicit, from a Parameters gate blob', j['config']['matrix_form']
    assert j['proof_sha256'] == single_gpu_2p22['proof_sha256'] and len(j['proof_sha256']) == 2
    assert j['proof_verified_by_pairing_check'] is True
    assert 'distributed-quotient' in j['config']['parallelism'] if n_ranks >= 4 else 'balanced-quotient' in j['config']['parallelism']
    assert j['config']['parallelism'].startswith('msm-shard%d' % n_ranks) and j['config']['distinct_witnesses_in_the_pipeline'] == 2
    if n_ranks == 4:
        assert j['replica_proofs_per_sec'] > 0
    ld = j['load']
    assert ld['matrix_terms'] == sum(j['config']['nnz']) and ld['image_bytes'] > ld['blob_bytes'] > 0 and ld['one_rank_at_a_time'] is True
    sp = j['single_process_multi_gpu']
    assert sp['ranks'] == n_ranks and sp['ms_per_step'] > 0 and sp['matrix_form'] == 'explicit, from a Parameters gate blob', sp
    assert len(sp['topology']) == n_ranks and all(c == 'self' for row in sp['topology'] for c in row)
    pf = j['preflight']
    assert pf['ran'] is True and pf['world'] == n_ranks and pf['all_ranks_ok'] is True and pf['decision']['backend'] == 'gloo', pf
    lib = pf['library']
    assert lib['ok'] is True and len(lib['pull_GBps']) == n_ranks and all(lib['pull_GBps'][i][k] > 0 for i in range(n_ranks) for k in range(n_ranks) if i != k), lib
    assert all(v == 0 for row in lib['pull_status'] for v in row) and lib['host_events'] is False


def test_launcher_with_one_rank_is_the_single_gpu_line(single_gpu_2p22):
    """`--gpus 1` under the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK set, world 1) takes the single-GPU path: same form, same bytes, and
    the `preflight` block is there with rccl_world_size 1"""
    env = {'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29547'}
    j = _run(env, '--copies', '217', '--no-cpu-baseline', '--no-other-sizes', '--no-standalone', '--no-untiled')
    assert j['proof_sha256'] == single_gpu_2p22['proof_sha256'] and j['config']['matrix_form'] == 'explicit, from a Parameters gate blob'
    for line in (j, single_gpu_2p22):
        pf = line['preflight']
        assert pf['ran'] is True and pf['rccl_world_size'] == 1 and pf['rccl']['ok'] is True and pf['all_ranks_ok'] is True, pf
        assert pf['rccl']['all_gather']['verified'] and pf['rccl']['all_to_all']['verified'] and pf['rccl']['all_gather_384']['verified']
        assert pf['decision']['backend'] == 'nccl' and pf['library']['ok'] is True and pf['library']['pull_GBps'][0][1] > 0


def test_roofline_blocks_say_what_they_are_and_traffic_is_measured_in_the_run():
    """`frac` (union of the launches) beside `frac_per_launch` (rocprof's average launch), the G2 and NTT blocks, and -- with
    --measure-traffic on -- `traffic` from PMC passes taken BY THIS RUN (rocprofv3 child processes) instead of an imported figure"""
    j = _run({}, '--copies', '40', '--no-cpu-baseline', '--no-other-sizes', '--no-standalone', '--no-untiled', '--measure-traffic', 'on')
    rl = j['roofline']
    assert 0 < rl['frac_per_launch'] <= rl['frac'] * 1.0001 < 1 and rl['union_ms_per_launch'] <= rl['avg_launch_ms'] * 1.0001
    assert 'avg_launch_ms_is' in rl and rl['algorithmic_bytes_per_scalar_mul'] == 96
    for name, per in (('roofline_g2', 160), ('roofline_ntt', 64)):
        b = j[name]
        assert b['bound'] == 'hbm' and 0 < b['frac'] < 1 and 0 < b['frac_per_launch'] <= b['frac'] * 1.0001 and b['launches'] > 0, b
        assert per in (b.get('algorithmic_bytes_per_scalar_mul'), b.get('algorithmic_bytes_per_element_transform'))
    assert isinstance(j['legs'].get('measure_traffic'), float), j['legs']
    assert rl.get('traffic_measured_error') is None, rl.get('traffic_measured_error')
    assert rl['traffic'] > rl['achieved'] and rl['traffic_ratio'] > 1 and 'MEASURED BY THIS RUN' in rl['traffic_source'], rl
    assert j['roofline_ntt']['traffic_ratio'] > 0.5 and j['roofline_g2']['traffic_ratio'] > 1


def test_bench_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE in the environment: bench.py starts the two ranks itself
    (fresh child processes, before it has touched the GPU) and relays rank 0's line.  Both ranks share this box's one GPU
    (FK_BENCH_SAME_DEVICE=1, gloo).  The line also carries the one-call form of the same proof (fk_multi_prove_r1cs)."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env['FK_BENCH_SAME_DEVICE'] = '1'
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '2', '--warmup', '1', '--copies', '9'],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    last = out.stdout.strip().splitlines()[-1]
    j = json.loads(last)
    assert j['n_gpus'] == 2 and j['value'] > 0 and j['proof_verified_by_pairing_check'] is True
    assert j['single_process_multi_gpu']['ranks'] == 2 and j['single_process_multi_gpu']['ms_per_step'] > 0
    assert j['replica_proofs_per_sec'] > 0


def test_ranks_fall_back_to_their_own_key_shards_when_no_directory_can_hold_the_image():
    """a container with a tiny /dev/shm and /tmp (FK_BENCH_IMAGE_DIR=none simulates it): rank 0 finds no room for the `Parameters` image, says so to every
    rank, and the ranks prove the tiled form from their own key shards as before round 6 -- the line says which form it measured and why"""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'FK_DIST_QUOTIENT')}
    env.update(FK_BENCH_SAME_DEVICE='1', FK_BENCH_IMAGE_DIR='none')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo', '--steps', '2', '--warmup', '1', '--copies', '9',
                          '--no-replicas', '--no-single-process'], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    j = json.loads(out.stdout.strip().splitlines()[-1])
    assert j['n_gpus'] == 2 and j['config']['matrix_form'].startswith('tiled') and 'no directory with room' in j['load']['error']
    assert j['proof_verified_by_pairing_check'] is True


def test_bench_synthetic_workload_small():
    j = _run({}, '--workload', 'synthetic', '--log2n', '14', '--cpu-log2n', '12')
    _check_line(j)


@pytest.mark.parametrize('dist_q', ['1', '0'])
def test_bench_multi_gpu_path_rehearsed_over_rccl(dist_q):
    env = {'FK_BENCH_REHEARSE': '1', 'FK_DIST_QUOTIENT': dist_q, 'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0',
           'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29541'}
    j = _run(env, '--workload', 'synthetic', '--log2n', '14')
    assert j['proof_verified_by_pairing_check'] is True
    assert ('distributed-quotient' in j['config']['parallelism']) == (dist_q == '1')


def test_bench_default_workload_multi_gpu_path_rehearsed():
    env = {'FK_BENCH_REHEARSE': '1', 'FK_DIST_QUOTIENT': '1', 'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0',
           'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': '29542'}
    j = _run(env, '--copies', '5')
    assert j['proof_verified_by_pairing_check'] is True and 'distributed-quotient' in j['config']['parallelism']


def test_bench_long_linear_combinations():
    j = _run({}, '--workload', 'synthetic', '--log2n', '14', '--lc-terms', '8', '--cpu-log2n', '12')
    assert j['proof_verified_by_pairing_check'] is True and j['config']['nnz'][0] > 4 * (1 << 14) * 0.4
