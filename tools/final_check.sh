#!/bin/bash
# end-of-round check on the GPU box: the whole GPU suite, smoke(), the default bench command.  Output: gpurun_out/final/
mkdir -p gpurun_out/final
python3 -m pytest tests -m gpu -q --durations=8 > gpurun_out/final/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/final/pytest_gpu.log; tail -3 gpurun_out/final/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
( time python3 bench.py ) > gpurun_out/final/bench.log 2> gpurun_out/final/bench.err; tail -c 600 gpurun_out/final/bench.log; grep real gpurun_out/final/bench.err
