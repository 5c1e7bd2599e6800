python tools/gemm_bench.py 5 0 2>&1 | grep -v amdgpu.ids > gpurun_out/auto_B5.txt
python tools/gemm_bench.py 1 0 2>&1 | grep -v amdgpu.ids > gpurun_out/auto_B1.txt
tail -n 1 gpurun_out/auto_B5.txt; tail -n 1 gpurun_out/auto_B1.txt
MVOC_HIP_LIB=$PWD/tools/lab/libmvoc_old.so python tools/gemm_bench.py 5 0 2>&1 | tail -n 1
MVOC_HIP_LIB=$PWD/tools/lab/libmvoc_old.so python tools/gemm_bench.py 1 0 2>&1 | tail -n 1
python -m pytest tests/test_ops_gpu.py -x -q -m gpu 2>&1 | tail -n 2
