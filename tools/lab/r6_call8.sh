#!/bin/bash
# round 6, GPU call 8: tfused with the packed LayerNorm prologue -- tests, bench, MFMA-busy counter
O=gpurun_out/r6c8; mkdir -p $O
timeout 600 python -m pytest -q tests/test_ops_gpu.py -k "temporal_qkv or tfused or temporal" > $O/tests.log 2>&1; tail -2 $O/tests.log
python tools/tfused_bench.py 2>&1 | grep -v amdgpu | tee $O/tfused_bench.txt
(cd tools/lab/r5_tree && python tools/tfused_bench.py 2>&1 | grep -v amdgpu) | tee $O/tfused_bench_r5tree.txt
bash tools/pmc_kernel.sh r6c8t "" tools/temporal_block_pmc.py > /dev/null 2>&1
grep "ALL\|tfused" -A1 gpurun_out/pmck_r6c8t/summary.txt | grep "ALL\|tfused\|MFMA util" | head -8; cp gpurun_out/pmck_r6c8t/summary.txt $O/pmc_temporal_block_plain.txt
