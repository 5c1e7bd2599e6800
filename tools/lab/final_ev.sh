python -m pytest tests -q -m gpu 2>&1 | tail -3 > gpurun_out/r5_final_tests.log; tail -2 gpurun_out/r5_final_tests.log
bash tools/pmc_bench.sh r5fin 4 > /dev/null 2>&1; cp gpurun_out/pmc_bench_r5fin/traffic.json gpurun_out/r5_final_pmc_gemm_traffic.json
bash tools/prof_mix.sh r5fin 8 job > /dev/null 2>&1; cp gpurun_out/profmix_r5fin/summary.txt gpurun_out/r5_final_mix_summary.txt; cp gpurun_out/profmix_r5fin/kernel_stats.csv gpurun_out/r5_final_mix_kernel_stats.csv
bash tools/prof_mix.sh r5finc 4 comp > /dev/null 2>&1; cp gpurun_out/profmix_r5finc/summary.txt gpurun_out/r5_final_comp_summary.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_final_b_bench.json 2>/dev/null
python bench.py > gpurun_out/r5_final_a_bench.json 2>/dev/null
python - <<'PY'
import json
for f in ("r5_final_a_bench","r5_final_b_bench"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1]); r=d["roofline"]; c=d["config"]
    print(f, d["value"], c["sequential_inversions"]["value"], c["inversion_step_ms"], c["inversion_step_ms_three_clips_concurrent"], c["composition_step_ms"], r["frac"], r["traffic"], round(sum(r["by_family_ms"].values()),1))
PY
