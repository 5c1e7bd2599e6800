#!/bin/bash
# round 6, GPU call 3: fixed tests; prologue stamps of a first-round and a later-round block; same-box A/B of the round-5 tree, the
# current library, the library without the dead-group skip, the library with the next-block prefetch
O=gpurun_out/r6c3; mkdir -p $O
timeout 1500 python -m pytest -q tests/test_pretrained_gpu.py tests/test_fullwidth_gpu.py::test_ext_forward_with_all_hooks_full_width \
  tests/test_fullsize_gpu.py::test_pnp_step_properties_full_size tests/test_pipeline_gpu.py::test_bench_gpus2_launched_plainly_on_the_gpu \
  tests/test_pipeline_gpu.py::test_bench_two_rank_protocol > $O/tests_a.log 2>&1; tail -5 $O/tests_a.log
timeout 900 python -m pytest -q tests/test_ops_gpu.py -k "chunk_major or production_tiles or split_k" > $O/tests_b.log 2>&1; tail -3 $O/tests_b.log
{ for b in 0 700; do for args in "81920 640 640 81 1" "81920 640 5760 81 1" "327680 320 1280 82 1" "81920 1920 640 81 0 2" "20480 1280 1280 81 1"; do
    echo "## stamped block $b"; ./tools/lab/g8_stamps_b$b $args; done; done; } > $O/g8_stamps_block0_vs_block700.txt 2>&1
run() { local label=$1; local dir=$2; shift 2
  for mix in comp inv; do
    (cd $dir && env "$@" python bench.py --mix $mix --steps 8 --warmup 4 --no-cpu-baseline --no-roofline 2>/dev/null) | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', '$mix', d['ms_per_step'], 'ms/step')"
  done; }
L=$PWD/mvoc_amd
{ for rep in 1 2; do
    run r5tree tools/lab/r5_tree MVOC_X=1
    run current . MVOC_X=1
    run current_korder0 . MVOC_KORDER=0
    run nodead . MVOC_HIP_LIB=$L/libmvoc_hip_nodead.so
    run prefetch . MVOC_HIP_LIB=$L/libmvoc_hip_pf.so
  done; } > $O/ab_step.txt 2>&1; cat $O/ab_step.txt
python tools/gemm_bench.py 5 0 > $O/gemm_B5_current.txt 2>&1
MVOC_HIP_LIB=$L/libmvoc_hip_nodead.so python tools/gemm_bench.py 5 0 > $O/gemm_B5_nodead.txt 2>&1
MVOC_HIP_LIB=$L/libmvoc_hip_pf.so python tools/gemm_bench.py 5 0 > $O/gemm_B5_prefetch.txt 2>&1
(cd tools/lab/r5_tree && python tools/gemm_bench.py 5 0) > $O/gemm_B5_r5tree.txt 2>&1
tail -1 $O/gemm_B5_*.txt
