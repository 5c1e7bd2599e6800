#!/bin/bash
# round 6, GPU call 2: new tests, chunk-major K order A/B (step level + per-shape counters), tfused priority A/B, cfg-2 CPU baseline
O=gpurun_out/r6c2; mkdir -p $O
timeout 1500 python -m pytest -q -x tests/test_pretrained_gpu.py tests/test_fullwidth_gpu.py::test_ext_forward_with_all_hooks_full_width \
  tests/test_fullsize_gpu.py::test_pnp_step_properties_full_size tests/test_pipeline_gpu.py::test_bench_gpus2_launched_plainly_on_the_gpu \
  tests/test_pipeline_gpu.py::test_bench_two_rank_protocol > $O/tests_a.log 2>&1; tail -5 $O/tests_a.log
timeout 900 python -m pytest -q tests/test_ops_gpu.py -k "chunk_major or groupnorm_folded or upsample or production_tiles" > $O/tests_b.log 2>&1; tail -5 $O/tests_b.log
run() { local label=$1; shift
  for mix in comp inv; do
    env "$@" python bench.py --mix $mix --steps 8 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '$mix', d['ms_per_step'], 'ms/step')"
  done; }
{ for rep in 1 2; do run korder0 MVOC_KORDER=0; run korder1 MVOC_KORDER=1; done; } > $O/korder_ab.txt 2>&1; cat $O/korder_ab.txt
MVOC_KORDER=0 bash tools/pmc_shapes.sh r6b_B5_k0 5 24 3 > $O/pmcs_k0.log 2>&1
MVOC_KORDER=1 bash tools/pmc_shapes.sh r6b_B5_k1 5 24 3 > $O/pmcs_k1.log 2>&1
{ echo default; python tools/tfused_bench.py; echo TF_SETPRIO=1; MVOC_HIP_LIB=$PWD/mvoc_amd/libmvoc_hip_tfprio.so python tools/tfused_bench.py;
  echo default; python tools/tfused_bench.py; echo TF_SETPRIO=1; MVOC_HIP_LIB=$PWD/mvoc_amd/libmvoc_hip_tfprio.so python tools/tfused_bench.py; } > $O/tfused_prio_ab.txt 2>&1
timeout 900 python3 tools/cpu_baseline_full.py $O/cpu_full --threads 16 > $O/cpu_full.log 2>&1; tail -3 $O/cpu_full.log
