#!/bin/bash
# GPU box: the evidence set of a round -- full GPU suite, bench line, rocprofv3 kernel statistics of the step mix, PMC traffic
# (separate --pmc passes), per-shape GEMM tables.  usage: bash tools/dbg/evidence_r4.sh <outdir> <tag>
OUT=${1:-gpurun_out/ev}; TAG=${2:-ev}
mkdir -p $OUT
timeout 3000 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
python bench.py > $OUT/bench.json 2> /dev/null
python bench.py --steps 20 --warmup 5 > $OUT/bench_steps20.json 2> /dev/null
bash tools/prof_mix.sh $TAG 8 job > /dev/null 2>&1; cp gpurun_out/profmix_$TAG/summary.txt $OUT/mix_kernel_stats_summary.txt; cp gpurun_out/profmix_$TAG/kernel_stats.csv $OUT/mix_kernel_stats.csv
bash tools/prof_mix.sh ${TAG}i 4 inv > /dev/null 2>&1; cp gpurun_out/profmix_${TAG}i/summary.txt $OUT/inversion_only_kernel_stats_summary.txt
bash tools/pmc_bench.sh $TAG 4 > $OUT/pmc.log 2>&1; cp gpurun_out/pmc_bench_$TAG/traffic.json $OUT/pmc_gemm_traffic.json
python tools/gemm_bench.py 1 0 > $OUT/gemm_per_shape_B1.txt 2>&1
python tools/gemm_bench.py 5 0 > $OUT/gemm_per_shape_B5.txt 2>&1
python tools/attn_bench.py 5 > $OUT/attn_bench_B5.txt 2>&1
python tools/attn_bench.py 1 > $OUT/attn_bench_B1.txt 2>&1
head -12 $OUT/mix_kernel_stats_summary.txt; python -c "
import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['roofline']['frac'], d['roofline']['by_family_ms'])"
