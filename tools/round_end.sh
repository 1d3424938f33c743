#!/bin/bash
# end-of-round evidence at the final head (GPU box): differential fuzzer, the N-rank rehearsal at reduced size, then tools/final_check.sh
mkdir -p gpurun_out/end
for seed in 71 72; do timeout 400 python3 tools/fuzz_parity.py --seconds 240 --seed $seed 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/end/fuzz_$seed.log; tail -2 gpurun_out/end/fuzz_$seed.log; done
bash tools/multi_rehearsal.sh > gpurun_out/end/multi_rehearsal.log 2>&1; grep "rc=" gpurun_out/end/multi_rehearsal.log
bash tools/final_check.sh
