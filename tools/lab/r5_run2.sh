#!/bin/bash
python -m pytest tests/test_ops_gpu.py -q -x -k "groupnorm" 2>&1 | tail -2
python tools/dbg/gn_bench.py 5 2>&1 | grep -v amdgpu | head -5
bash tools/lab/ab_bench.sh 2>&1 | grep -v "no-tail\|no-share"
bash tools/pmc_kernel.sh r5tblock "" tools/temporal_block_pmc.py > /dev/null 2>&1
grep "==\|MFMA util\|waiting" gpurun_out/pmck_r5tblock/summary.txt | head -40
bash tools/pmc_kernel.sh r5tblock_inj "" tools/temporal_block_pmc.py inject > /dev/null 2>&1
grep "==\|MFMA util\|waiting" gpurun_out/pmck_r5tblock_inj/summary.txt | head -40
tail -2 gpurun_out/pmck_r5tblock/log1.txt
