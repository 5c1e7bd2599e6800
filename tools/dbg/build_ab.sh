#!/bin/bash
# Build A/B libraries for same-box comparisons (run on the BUILD machine, from the repo root; the .so files travel with gpurun):
#   tools/lab/libmvoc_old.so      -- the whole library at git revision $1 (default HEAD)
#   tools/lab/libmvoc_g8dbg.so / libmvoc_g8dbg_prev.so -- working tree / $1 with -DMVOC_G8_STAMPS in gemm8.hip
#   tools/lab/g8_stamps           -- the stamp reader (README.md)
set -e
REV=${1:-HEAD}
ROOT=$PWD
CS=$ROOT/mvoc_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=on -Wno-unused-function"
VF="-mllvm -amdgpu-mfma-vgpr-form"
OLD=/tmp/mvoc_old_src
rm -rf $OLD; mkdir -p $OLD/mvoc_amd/csrc $OLD/include
for f in $(git ls-tree --name-only $REV mvoc_amd/csrc/ | grep -E '\.(hip|h)$'); do git show $REV:$f > $OLD/$f; done
git show $REV:include/mvoc_hip.h > $OLD/include/mvoc_hip.h
python -c "from mvoc_amd import build; build.build(verbose=False)"
( cd $OLD/mvoc_amd/csrc
  for f in runtime gemm gemm8 attention tfused xslin norm pnp stem comm; do
    x=""; case $f in gemm|gemm8|attention|tfused|xslin) x="$VF";; esac
    hipcc $FL $x -c $f.hip -o $f.o &
  done; wait
  hipcc $FL $VF -DMVOC_G8_STAMPS -c gemm8.hip -o gemm8_dbg.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/lab/libmvoc_old.so runtime.o gemm.o gemm8.o attention.o tfused.o xslin.o norm.o pnp.o stem.o comm.o -ldl
  hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/lab/libmvoc_g8dbg_prev.so runtime.o gemm.o gemm8_dbg.o attention.o tfused.o xslin.o norm.o pnp.o stem.o comm.o -ldl )
( cd $CS
  hipcc $FL $VF -falign-loops=64 -DMVOC_G8_STAMPS -c gemm8.hip -o /tmp/gemm8_dbg.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/lab/libmvoc_g8dbg.so runtime.o gemm.o /tmp/gemm8_dbg.o attention.o tfused.o xslin.o norm.o pnp.o stem.o comm.o -ldl )
( cd tools/lab && hipcc --offload-arch=gfx950 -O3 -std=c++17 g8_stamps.hip -o g8_stamps -L . -lmvoc_g8dbg -Wl,-rpath,'$ORIGIN' )
ls -la tools/lab/*.so tools/lab/g8_stamps
