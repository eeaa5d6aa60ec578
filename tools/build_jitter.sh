#!/bin/bash
# Barrier-jitter build of the whole library (ctgcn_amd/csrc/ctgcn_jitter.h): tools/build_jitter.sh [seed] -> tools/variants/lib_jitter<seed>.so
# Run anything against it with CTGCN_HIP_LIB=tools/variants/lib_jitter<seed>.so (tools/runs/r5_jitter.sh does: bit-identity tests + stress tools).
set -e
cd "$(dirname "$0")/.."
seed=${1:-1}
C=ctgcn_amd/csrc; V=tools/variants; H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread -DCTGCN_JITTER=$seed"
mkdir -p $V/objj
$H -mllvm -amdgpu-mfma-vgpr-form -c $C/ctgcn_hip.hip -o $V/objj/ctgcn_hip.o &
for f in ctgcn_gemm ctgcn_gru_bwd ctgcn_ingest ctgcn_walks; do $H -c $C/$f.hip -o $V/objj/$f.o & done
$H -c $C/ctgcn_export.cpp -o $V/objj/ctgcn_export.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $V/lib_jitter$seed.so $V/objj/*.o
rm -rf $V/objj
echo $V/lib_jitter$seed.so
