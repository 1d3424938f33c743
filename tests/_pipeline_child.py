"""Child of tests/test_gpu_pipeline.py::test_early_front_with_distinct_witnesses: the two-slot pipeline (fk_prove_r1cs_submit / _wait)
with DIFFERENT witnesses in flight under the schedule the benchmark size runs -- sorts-first with the early front, where `_wait(k)`
queues the evaluation and the witness sorts of proof k + 1 out of the OTHER slot while proof k is still running.  That schedule is
chosen per process (FK_PROVE_SORTS_FIRST, default: from 2^25 on), hence the subprocess.  A slot mix-up or a stale staging buffer
gives a wrong proof here; with the same witness in both slots (what the full-size tests and the bench used to do) it could not.
Also the abandon path: a foreign proof issued while an early front is outstanding is refused, the front is dropped, and the
waiting ticket still yields its own proof."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402  (data loading helpers only; nothing of the oracle)
import fawkes_crypto_amd as fk  # noqa: E402

copies = 5
ctx = fk.Context(0)
r1cs, zs = bench.load_rollup_instance()
zs = zs[:3]
wit = [bench.tile_witness(np.roll(zs, k, axis=0), r1cs.num_input, copies) for k in range(3)]
wit += [w.copy() for w in wit[:2]]
wit[3][-7] = bench.mont(12345); wit[4][-9] = bench.mont(54321)          # two unsatisfying assignments: any assignment has a well-defined proof
dr = ctx.load_r1cs(r1cs, copies=copies)
tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
key, vk = ctx.setup(r1cs, copies=copies, **tox)
r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
direct = [bytes(ctx.prove_witness(key, dr, z, r, s)) for z in wit]
assert len(set(direct)) == len(direct), 'the witnesses must give different proofs'
nv = wit[0].shape[0]
pins = [ctx.host_alloc((nv, 4)) for _ in range(2)]


def pipelined(order):
    got = []
    pins[0][:] = wit[order[0]]
    ticket = ctx.prove_witness_submit(key, dr, pins[0], r, s)
    for k in range(len(order)):
        nxt = None
        if k + 1 < len(order):
            pins[(k + 1) & 1][:] = wit[order[k + 1]]
            nxt = ctx.prove_witness_submit(key, dr, pins[(k + 1) & 1], r, s)
        got.append(bytes(ctx.prove_witness_wait(ticket)))
        ticket = nxt
    return got


order = [0, 1, 2, 3, 4, 2, 0, 4, 1]
assert pipelined(order) == [direct[i] for i in order], 'pipelined proofs differ from the direct ones'
# abandon: ticket B's front is queued by wait(A); a foreign proof then finds it outstanding
d_z = ctx.dev_alloc(nv * 32)
ctx.upload(d_z, wit[2])
pins[0][:] = wit[0]; pins[1][:] = wit[1]
t_a = ctx.prove_witness_submit(key, dr, pins[0], r, s)
t_b = ctx.prove_witness_submit(key, dr, pins[1], r, s)
assert bytes(ctx.prove_witness_wait(t_a)) == direct[0]
try:
    foreign = bytes(ctx.prove_witness_dev(key, dr, d_z, r, s))
    refused = False
    assert foreign == direct[2]
except fk.FkError as e:
    refused = True
    assert e.code == 1 and 'early front' in str(e), str(e)
assert bytes(ctx.prove_witness_wait(t_b)) == direct[1], 'the ticket behind an abandoned early front gave a wrong proof'
assert bytes(ctx.prove_witness_dev(key, dr, d_z, r, s)) == direct[2]
assert pipelined([3, 1, 4]) == [direct[3], direct[1], direct[4]]
ctx.dev_free(d_z)
print('PIPE ok early_front_refused_a_foreign_proof=%s' % refused)
