#!/bin/bash
# GPU box: full GPU test suite + rocprofv3 kernel statistics of the step mix (job / inv) + dispatch trace of one B = 1 forward
OUT=${1:-gpurun_out/prof}
mkdir -p $OUT
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
bash tools/prof_mix.sh ${2:-r4} 8 job > /dev/null 2>&1; cp gpurun_out/profmix_${2:-r4}/summary.txt $OUT/mix_summary.txt; cp gpurun_out/profmix_${2:-r4}/kernel_stats.csv $OUT/mix_kernel_stats.csv
bash tools/prof_mix.sh ${2:-r4}i 4 inv > /dev/null 2>&1; cp gpurun_out/profmix_${2:-r4}i/summary.txt $OUT/inv_summary.txt
MVOC_GEMM_TRACE=1 python tools/gemm_bench.py 1 0 2> $OUT/trace_B1.txt > $OUT/gemm_B1.txt
python bench.py > $OUT/bench.json 2> /dev/null
head -30 $OUT/mix_summary.txt
