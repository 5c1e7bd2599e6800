mkdir -p gpurun_out/ev
python -m pytest tests -q -m gpu 2>&1 | tail -n 15 > gpurun_out/ev/gpu_tests.log
bash tools/prof_mix.sh r3g 8 > /dev/null 2>&1
bash tools/pmc_bench.sh r3g 4 > /dev/null 2>&1
cp gpurun_out/pmc_bench_r3g/traffic.json profiles/r3/pmc_gemm_traffic.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -n 1 > gpurun_out/ev/bench.json
python bench.py --steps 20 --warmup 5 --batch-inversions 2>/dev/null | tail -n 1 > gpurun_out/ev/bench_batched.json
bash tools/pmc_kernel.sh r3g_gemm gemm bench.py --pmc-pass --steps 4 > /dev/null 2>&1
python tools/gemm_bench.py 5 0 2>&1 | grep -v amdgpu.ids > gpurun_out/ev/gemm_B5_auto.txt
python tools/gemm_bench.py 1 0 2>&1 | grep -v amdgpu.ids > gpurun_out/ev/gemm_B1_auto.txt
python bench.py --latent-h 90 --latent-w 160 --steps 8 --warmup 2 2>/dev/null | tail -n 1 > gpurun_out/ev/bench_90x160.json
