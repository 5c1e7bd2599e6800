#!/bin/bash
# GPU box: the evidence set of round 4 -- full GPU suite, bench lines (working tree and the round-3 tree on the same box),
# rocprofv3 kernel statistics of the step mix, PMC traffic (separate --pmc passes), per-shape GEMM tables, attention bench, the
# three-stream inversion experiment, held clock of the gemm8 K loop, end-to-end demo job, native-size line.
# usage: bash tools/dbg/evidence_r4.sh <outdir> <tag>   (tools/lab/r3_tree + tools/lab/libmvoc_g8dbg.so built on the build machine)
OUT=${1:-gpurun_out/ev}; TAG=${2:-ev}
mkdir -p $OUT
timeout 3000 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
for i in 1 2; do
  python bench.py > $OUT/bench_run$i.json 2> /dev/null
  [ -d tools/lab/r3_tree ] && (cd tools/lab/r3_tree && python bench.py --no-cpu-baseline 2>/dev/null) > $OUT/bench_round3_tree_same_box_run$i.json
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_steps20.json 2> /dev/null
python bench.py --sequential-inversions --no-cpu-baseline > $OUT/bench_sequential_inversions.json 2> /dev/null
python bench.py --batch-inversions --no-cpu-baseline > $OUT/bench_batched_inversions.json 2> /dev/null
bash tools/prof_mix.sh $TAG 8 job > /dev/null 2>&1; cp gpurun_out/profmix_$TAG/summary.txt $OUT/mix_kernel_stats_summary.txt; cp gpurun_out/profmix_$TAG/kernel_stats.csv $OUT/mix_kernel_stats.csv
bash tools/prof_mix.sh ${TAG}i 4 inv > /dev/null 2>&1; cp gpurun_out/profmix_${TAG}i/summary.txt $OUT/inversion_only_kernel_stats_summary.txt
bash tools/prof_mix.sh ${TAG}c 4 comp > /dev/null 2>&1; cp gpurun_out/profmix_${TAG}c/summary.txt $OUT/composition_only_kernel_stats_summary.txt
bash tools/pmc_bench.sh $TAG 4 > $OUT/pmc.log 2>&1; cp gpurun_out/pmc_bench_$TAG/traffic.json $OUT/pmc_gemm_traffic.json
python tools/gemm_bench.py 1 0 > $OUT/gemm_per_shape_B1.txt 2>&1
python tools/gemm_bench.py 5 0 > $OUT/gemm_per_shape_B5.txt 2>&1
python tools/attn_bench.py 5 > $OUT/attn_bench_B5.txt 2>&1
python tools/attn_bench.py 1 > $OUT/attn_bench_B1.txt 2>&1
for v in 0 1 0 1; do echo "MVOC_FLASH3=$v (0: flash_kernel, 1: flash3_kernel)"; MVOC_FLASH3=$v python tools/dbg/attn_big.py 2>&1 | grep "^nb"; done > $OUT/attn_long_rows_both_kernels.txt
python tools/lab/streams_ab.py > $OUT/inversion_three_streams_experiment.txt 2>&1
bash tools/dbg/clock_run.sh $OUT > /dev/null 2>&1
python bench.py --latent-h 90 --latent-w 160 --no-cpu-baseline > $OUT/bench_latent_90x160.json 2> /dev/null
python bench.py --workload demo > $OUT/demo_job_end_to_end.json 2> $OUT/demo.err
head -12 $OUT/mix_kernel_stats_summary.txt
python - $OUT <<'PY'
import json, sys
o = sys.argv[1]
for f in ("bench_run1", "bench_run2", "bench_round3_tree_same_box_run1", "bench_round3_tree_same_box_run2", "bench_steps20", "bench_sequential_inversions", "bench_batched_inversions", "bench_latent_90x160"):
    try:
        d = json.load(open(f"{o}/{f}.json")); c = d["config"]
        print(f, d["value"], c.get("inversion_step_ms"), c.get("inversion_step_ms_three_clips_concurrent"), c.get("composition_step_ms"), (d.get("roofline") or {}).get("frac"))
    except Exception as e:
        print(f, "ERR", e)
PY
