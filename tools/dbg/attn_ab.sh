#!/bin/bash
# build machine: tools/lab/libmvoc_attn_<tag>.so = the working tree's library with attention.hip compiled with extra flags
# usage: tools/dbg/attn_ab.sh <tag> [extra hipcc flags...]
set -e
TAG=$1; shift
python -c "from mvoc_amd import build; build.build(verbose=False)"
cd mvoc_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=on -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form "$@" \
  -Rpass-analysis=kernel-resource-usage -c attention.hip -o /tmp/attention_$TAG.o 2>&1 | grep -E "error|Function Name|VGPRs:|Spill|Occupancy" | grep -A3 flash | sed 's/.*remark: *//; s/\[-R.*//' | paste - - - - 
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/lab/libmvoc_attn_$TAG.so runtime.o gemm.o gemm8.o /tmp/attention_$TAG.o tfused.o xslin.o norm.o pnp.o stem.o comm.o -ldl
