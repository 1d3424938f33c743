#!/bin/bash
# the explicit system's evaluation alone (tools/spmv_untiled_probe.py, 1741 transactions) by the size of the row blocks inside which the
# length-class lists are sorted (FK_SPMV_BLOCK_ROWS, experiment build; default 4096; 19270 = one transaction of the benchmark's circuit)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/spmv_blocks
for b in ${BLOCKS:-1024 4096 8192 19270 38540 65536}; do
  COPIES=1741 FK_LIB_VARIANT=exp FK_SPMV_BLOCK_ROWS=$b timeout 600 python3 tools/spmv_untiled_probe.py > gpurun_out/spmv_blocks/b$b.log 2>&1
  echo "block_rows=$b rc=$? $(grep -E '^(tiled|untiled) ' gpurun_out/spmv_blocks/b$b.log | tr '\n' ' ')"
done
