# quick check of a sort-kernel change (GPU box): MSM / prove tests, then the bench with a kernel trace
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; O=gpurun_out/sortp; mkdir -p $O
python -m pytest tests/test_gpu_msm.py tests/test_gpu_precompute.py tests/test_gpu_prove.py tests/test_gpu_tiled.py -q -x > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench.log 2>&1
python3 -c "
import json
l=[x for x in open('$O/bench.log') if x.startswith('{')]
j=json.loads(l[0]); print('ms_per_step', j['ms_per_step'], 'resident', j['device_resident_ms_per_step'], j['kernel_ms_per_step'])"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
find $O -name "*kernel_trace.csv" -delete
head -24 $O/kt/k_kernel_stats.csv | cut -d, -f1-4 | sed 's/fk:://g' | cut -c1-150
