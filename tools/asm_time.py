import sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import bench, fawkes_crypto_amd as fk
ctx = fk.Context(0)
r1cs, zs = bench.load_rollup_instance()
copies = 8
z = bench.tile_witness(zs[:3], r1cs.num_input, copies)
dr = ctx.load_r1cs(r1cs, copies=copies)
tox = {k: bench.mont(v) for k, v in bench.TOXIC.items()}
key, vk = ctx.setup(r1cs, copies=copies, **tox)
r, s = bench.mont(0xA11CE), bench.mont(0xB0B)
for _ in range(3):
    p, tm = ctx.prove_witness(key, dr, z, r, s, want_timings=True)
print(tm)
