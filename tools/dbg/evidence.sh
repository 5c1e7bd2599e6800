set -x
mkdir -p gpurun_out/ev
python -m pytest tests -q -m gpu 2>&1 | tail -n 15 > gpurun_out/ev/gpu_tests.log
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -n 1 > gpurun_out/ev/bench.json
python bench.py --steps 20 --warmup 5 --batch-inversions 2>/dev/null | tail -n 1 > gpurun_out/ev/bench_batched.json
bash tools/prof_mix.sh r3d 8 > /dev/null 2>&1
bash tools/pmc_bench.sh r3d 4 > /dev/null 2>&1
python tools/gemm_bench.py 5 0 2>&1 | grep -v amdgpu.ids > gpurun_out/ev/gemm_B5_auto.txt
python tools/gemm_bench.py 1 0 2>&1 | grep -v amdgpu.ids > gpurun_out/ev/gemm_B1_auto.txt
