for b in 5 1; do
  python tools/gemm_bench.py $b 0 2>&1 > gpurun_out/gemm_new2_B$b.txt; echo "NEW B=$b"; tail -1 gpurun_out/gemm_new2_B$b.txt
  MVOC_HIP_LIB=$PWD/tools/lab/libmvoc_old.so python tools/gemm_bench.py $b 0 2>&1 > gpurun_out/gemm_old2_B$b.txt;  echo "OLD B=$b"; tail -1 gpurun_out/gemm_old2_B$b.txt
done
