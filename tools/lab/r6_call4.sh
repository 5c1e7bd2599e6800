#!/bin/bash
# round 6, GPU call 4: tests of the re-gated chunk-major form / fixed tests; tfused first-round stagger; step A/B against the round-5 tree
O=gpurun_out/r6c4; mkdir -p $O
timeout 1500 python -m pytest -q tests/test_pretrained_gpu.py tests/test_pipeline_gpu.py::test_bench_two_rank_protocol \
   tests/test_pipeline_gpu.py::test_bench_gpus2_launched_plainly_on_the_gpu > $O/tests_a.log 2>&1; tail -3 $O/tests_a.log
timeout 900 python -m pytest -q tests/test_ops_gpu.py -k "chunk_major or production_tiles or split_k or temporal_qkv" > $O/tests_b.log 2>&1; tail -3 $O/tests_b.log
L=$PWD/mvoc_amd
{ for rep in 1 2; do echo default; python tools/tfused_bench.py 2>/dev/null | grep -v amdgpu
  for v in tfst1 tfst2 tfst3 tfst1b; do echo $v; MVOC_HIP_LIB=$L/libmvoc_hip_$v.so python tools/tfused_bench.py 2>/dev/null | grep -v amdgpu; done; done; } > $O/tfused_stagger_ab.txt 2>&1
grep -A2 "^default\|^tfst" $O/tfused_stagger_ab.txt | grep "B=5\|^default\|^tfst"
run() { local label=$1; local dir=$2; shift 2
  for mix in comp inv; do
    (cd $dir && env "$@" python bench.py --mix $mix --steps 8 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '$mix', d['ms_per_step'], 'ms/step')"
  done; }
{ for rep in 1 2 3; do
    run r5tree tools/lab/r5_tree MVOC_X=1
    run current . MVOC_X=1
    run current_korder0 . MVOC_KORDER=0
  done; } > $O/ab_step.txt 2>&1; cat $O/ab_step.txt
python tools/gemm_bench.py 5 0 > $O/gemm_B5_current.txt 2>&1
MVOC_KORDER=0 python tools/gemm_bench.py 5 0 > $O/gemm_B5_korder0.txt 2>&1
(cd tools/lab/r5_tree && python tools/gemm_bench.py 5 0) > $O/gemm_B5_r5tree.txt 2>&1
for f in current korder0 r5tree; do tail -n 1 $O/gemm_B5_$f.txt; done
