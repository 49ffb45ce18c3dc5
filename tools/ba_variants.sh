#!/bin/bash
# Builds A/B variants of the library that differ only in ba.hip's compile-time shape (LDS budget, observations per Schur batch, waves per SIMD):
#   bash tools/ba_variants.sh name "-DMS_BA_LDS_KB=78 -DMS_FS_OB=32 -DMS_BA_WAVES_PER_EU=4" [name2 "flags2" ...]   ->  tools/variants/lib_<name>.so
# (run in the container: hipcc cross-compiles; the .so files travel to the GPU box with the snapshot; tools/ba_occupancy_probe.py loads them one after the other)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/slam-module_amd/csrc
make -s -C "$C"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -munsafe-fp-atomics -mllvm -amdgpu-atomic-optimizer-strategy=DPP -mllvm -amdgpu-mfma-vgpr-form -ffp-contract=fast -Wall -Wno-unused-function -Wno-unused-result"
mkdir -p "$R/tools/variants"
while [ $# -ge 2 ]; do
  name=$1; extra=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $extra -c "$C/ba.hip" -o "$R/tools/variants/ba_$name.o"
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$R/tools/variants/lib_$name.so" "$C/ctx.o" "$C/geometry.o" "$C/orb.o" "$C/match.o" "$C/bow.o" "$R/tools/variants/ba_$name.o"
  rm -f "$R/tools/variants/ba_$name.o"
  echo "built tools/variants/lib_$name.so  ($extra)"
done
