#!/bin/bash
# one GPU call: ops / unet tests, gemm8 stamps + per-shape tables old vs new, attention bench old vs new
OUT=${1:-gpurun_out/r4}
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_unet_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
bash tools/dbg/g8_ab_run.sh $OUT
for b in 1 5; do
  MVOC_HIP_LIB=$PWD/tools/lab/libmvoc_old.so python tools/attn_bench.py $b > $OUT/attn_old_B$b.txt 2>&1
  python tools/attn_bench.py $b > $OUT/attn_new_B$b.txt 2>&1
done
grep -h "self\|pair" $OUT/attn_old_B5.txt $OUT/attn_new_B5.txt
