# Final measurement pass of a round (run on the GPU box through gpurun): bench line, kernel trace, PMC passes.
# Counter passes are separate runs and never combined with other trace domains.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
B="bench.py --steps 1 --warmup 0 --no-cpu-baseline"
python3 bench.py --steps 5 --warmup 2 > $O/bench_2p25.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o f -- python3 $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o w -- python3 $B > $O/write.log 2>&1
rocprofv3 --pmc VALUBusy VALUUtilization MemUnitBusy --output-format csv -d $O/derived -o d -- python3 $B > $O/derived.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/sq -o s -- python3 $B > $O/sq.log 2>&1
find $O -name "*.csv" -size +20M -delete
ls -la $O $O/*/ | head -40
