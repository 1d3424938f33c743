# quick GPU check of a kernel change: the parity suites that cover it, then a short bench
set -u
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/quick; mkdir -p $O
python -m pytest tests/test_gpu_ntt.py tests/test_gpu_msm.py tests/test_gpu_prove.py tests/test_gpu_r1cs.py tests/test_gpu_dist_quotient.py tests/test_gpu_config0.py tests/test_gpu_tiled.py -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench.log 2>&1
python3 -c "
import json
l=[x for x in open('$O/bench.log') if x.startswith('{')]
j=json.loads(l[0]); print('ms_per_step', j['ms_per_step'], 'resident', j['device_resident_ms_per_step'], j['kernel_ms_per_step'])"
