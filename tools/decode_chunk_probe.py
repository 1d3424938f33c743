"""host-only probe: the gate decoder with its decompressor called for FK_GATES_CHUNK_KB of output at a time -- the count-walking scanner that follows every call
is a pointer chase through the bytes just written, so whether they are still in the near caches decides its cost (FK_GATES_TRACE prints the split)"""
import sys, os, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ['FK_GATES_TRACE']='1'
    import bench
    from fawkes_crypto_amd import api
    inst, zs = bench.load_rollup_instance()
    copies=int(sys.argv[1])
    n_in, n_aux = 1 + copies*(inst.num_input-1), copies*inst.num_aux
    b = api.GateBlob(inst, copies, fmt=api.FK_GATES_BROTLI, quality=2, lgwin=22)
    for rep in range(3):
        g = api.Gates(b.data, api.FK_GATES_BROTLI, b.num_gates, n_in, n_aux); g.free()
else:
    for kb in (1024, 256, 64, 16):
        print('chunk KB', kb, flush=True)
        subprocess.run([sys.executable, __file__, os.environ.get('PROBE_COPIES', '400')], env=dict(os.environ, FK_GATES_CHUNK_KB=str(kb)))
