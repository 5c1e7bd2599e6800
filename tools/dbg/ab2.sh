for b in 5 1; do
  python tools/gemm_bench.py $b 0 2>&1 | grep -v amdgpu.ids > gpurun_out/auto_B$b.txt; echo "NEW  B=$b: $(tail -n 1 gpurun_out/auto_B$b.txt)"
  MVOC_HIP_LIB=$PWD/tools/lab/libmvoc_prev.so python tools/gemm_bench.py $b 0 2>&1 | tail -n 1 | sed "s/^/PREV B=$b: /"
done
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.load(sys.stdin); print('NEW ', d['value'], d['config']['inversion_step_ms'], d['config']['composition_step_ms'])"
MVOC_HIP_LIB=$PWD/tools/lab/libmvoc_prev.so python bench.py --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "import json,sys; d=json.load(sys.stdin); print('PREV', d['value'], d['config']['inversion_step_ms'], d['config']['composition_step_ms'])"
