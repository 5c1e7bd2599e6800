#!/bin/bash
# GPU box: the evidence set of a round, taken ONCE at its end -- full GPU suite, the metric line (twice) beside the previous round's
# tree on the same box, the host-side switches of the round one by one, rocprofv3 kernel statistics of the step mix, PMC traffic
# and MFMA utilisation (separate --pmc passes), per-shape GEMM tables, attention bench, end-to-end demo job, native-size line.
# usage: bash tools/evidence.sh <outdir> <tag>   (tools/lab/r5_tree: `git worktree add tools/lab/r5_tree <previous round's head>` + its library,
# built on the build machine; ships with the snapshot)
OUT=${1:-gpurun_out/ev}; TAG=${2:-ev}
mkdir -p $OUT
timeout 3000 python -m pytest tests -q -m gpu > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
for i in 1 2; do
  python bench.py > $OUT/bench_run$i.json 2> /dev/null
  [ -d tools/lab/r5_tree ] && (cd tools/lab/r5_tree && python bench.py --no-cpu-baseline 2>/dev/null) > $OUT/bench_round5_tree_same_box_run$i.json
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_steps20.json 2> /dev/null
bash tools/lab/ab_bench.sh > $OUT/switches_ab.txt 2>&1
bash tools/prof_mix.sh $TAG 8 job > /dev/null 2>&1; cp gpurun_out/profmix_$TAG/summary.txt $OUT/mix_kernel_stats_summary.txt; cp gpurun_out/profmix_$TAG/kernel_stats.csv $OUT/mix_kernel_stats.csv
bash tools/prof_mix.sh ${TAG}i 4 inv > /dev/null 2>&1; cp gpurun_out/profmix_${TAG}i/summary.txt $OUT/inversion_only_kernel_stats_summary.txt
bash tools/prof_mix.sh ${TAG}c 4 comp > /dev/null 2>&1; cp gpurun_out/profmix_${TAG}c/summary.txt $OUT/composition_only_kernel_stats_summary.txt
bash tools/pmc_bench.sh $TAG 4 > $OUT/pmc.log 2>&1; cp gpurun_out/pmc_bench_$TAG/traffic.json $OUT/pmc_gemm_traffic.json
bash tools/pmc_kernel.sh ${TAG}g gemm bench.py --pmc-pass --steps 4 > /dev/null 2>&1
bash tools/pmc_kernel.sh ${TAG}f flash bench.py --pmc-pass --steps 4 > /dev/null 2>&1
bash tools/pmc_kernel.sh ${TAG}x "xslin|tfused|tattn" bench.py --pmc-pass --steps 4 > /dev/null 2>&1
{ echo "rocprofv3 --pmc passes (tools/pmc_kernel.sh over bench.py --pmc-pass --steps 4: 3 inversion steps B = 1 + 1 composition step B = 5; separate passes, no trace domains)";
  echo "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x 256 CU x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8; every dispatch of an instantiation in the mix summed";
  for t in g f x; do grep "^==\|MFMA utilisation\|waves waiting" gpurun_out/pmck_${TAG}$t/summary.txt | paste - - - | sed 's/   MFMA utilisation = / MFMA busy /; s/ of (4 SIMD x 256 CU x kernel cycles)//; s/   waves waiting (SQ_WAIT_ANY \/ SQ_WAVE_CYCLES) = / | waves waiting /'; done; } > $OUT/pmc_mfma_utilisation.txt
bash tools/pmc_shapes.sh ${TAG}s 5 24 3 > /dev/null 2>&1; cp gpurun_out/pmcs_${TAG}s/table.txt $OUT/gemm_per_shape_pmc_final_tree.txt
for m in "" inject; do bash tools/pmc_kernel.sh ${TAG}t$m "" tools/temporal_block_pmc.py $m > /dev/null 2>&1; cp gpurun_out/pmck_${TAG}t$m/summary.txt $OUT/pmc_temporal_block_${m:-plain}.txt; done
python tools/tfused_bench.py > $OUT/tfused_bench.txt 2>&1
python tools/gemm_bench.py 1 0 > $OUT/gemm_per_shape_B1.txt 2>&1
python tools/gemm_bench.py 5 0 > $OUT/gemm_per_shape_B5.txt 2>&1
python tools/attn_bench.py 5 > $OUT/attn_bench_B5.txt 2>&1
python bench.py --latent-h 90 --latent-w 160 --no-cpu-baseline > $OUT/bench_latent_90x160.json 2> /dev/null
python bench.py --workload demo > $OUT/demo_job_end_to_end.json 2> $OUT/demo.err
head -12 $OUT/mix_kernel_stats_summary.txt
python - $OUT <<'PY'
import json, sys
o = sys.argv[1]
for f in ("bench_run1", "bench_round5_tree_same_box_run1", "bench_run2", "bench_round5_tree_same_box_run2", "bench_steps20", "bench_latent_90x160"):
    try:
        d = json.loads(open(f"{o}/{f}.json").read().strip().splitlines()[-1]); c = d["config"]
        r = d.get("roofline") or {}
        print(f, d["value"], "seq", (c.get("sequential_inversions") or {}).get("value"), c.get("inversion_step_ms"), c.get("inversion_step_ms_three_clips_concurrent"),
              c.get("composition_step_ms"), r.get("frac"), round(sum((r.get("by_family_ms") or {}).values()), 1))
    except Exception as e:
        print(f, "ERR", e)
PY
