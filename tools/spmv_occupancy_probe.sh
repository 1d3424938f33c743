#!/bin/bash
# the explicit system's evaluation alone by the register allocation of spmv_binned_kernel: 103 VGPR (4 waves per SIMD, production), held to 96 (5 waves,
# 4-6 dwords spilled) and to 80 (6 waves, 22-28 spilled) with __launch_bounds__(256, FK_SPMV_MINB).  Libraries: libfawkes_hip_exp.so (MINB 1),
# libfawkes_hip_minb5.so, libfawkes_hip_minb6.so (same experiment flags otherwise).
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/spmv_occ
for v in exp minb5 minb6 exp minb5; do
  COPIES=1741 FK_LIB_VARIANT=$v timeout 600 python3 tools/spmv_untiled_probe.py > gpurun_out/spmv_occ/$v.log 2>&1
  echo "lib=$v rc=$? $(grep -E '^(tiled|untiled) ' gpurun_out/spmv_occ/$v.log | tr '\n' ' ')"
done
