#!/bin/bash
# A/B builds of the main kernel file: tools/build_variant.sh <name> [-DFLAG=...]... -> tools/variants/lib_<name>.so (git-ignored; ships to the GPU box).
# Run a tool against it with CTGCN_HIP_LIB=tools/variants/lib_<name>.so (ctgcn_amd/_lib.py).
# GEMM=1 tools/build_variant.sh <name> -D... applies the flags to ctgcn_gemm.hip instead.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
C=ctgcn_amd/csrc; V=tools/variants; H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -pthread"
mkdir -p $V/obj
for f in ctgcn_gemm ctgcn_gru_bwd ctgcn_ingest ctgcn_walks; do
  [ $V/obj/$f.o -nt $C/$f.hip ] || $H -c $C/$f.hip -o $V/obj/$f.o &
done
if [ "${GEMM:-0}" = 1 ]; then
  [ $V/obj/hip_plain.o -nt $C/ctgcn_hip.hip ] || $H -mllvm -amdgpu-mfma-vgpr-form -c $C/ctgcn_hip.hip -o $V/obj/hip_plain.o &
  $H "$@" -c $C/ctgcn_gemm.hip -o $V/obj/gemm_$name.o
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $V/lib_$name.so $V/obj/hip_plain.o $V/obj/gemm_$name.o $V/obj/ctgcn_gru_bwd.o $V/obj/ctgcn_ingest.o $V/obj/ctgcn_walks.o $V/obj/ctgcn_export.o
  rm -f $V/obj/gemm_$name.o
  echo $V/lib_$name.so
  exit 0
fi
[ $V/obj/ctgcn_export.o -nt $C/ctgcn_export.cpp ] || $H -c $C/ctgcn_export.cpp -o $V/obj/ctgcn_export.o &
$H -mllvm -amdgpu-mfma-vgpr-form "$@" -c $C/ctgcn_hip.hip -o $V/obj/hip_$name.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $V/lib_$name.so $V/obj/hip_$name.o $V/obj/ctgcn_gemm.o $V/obj/ctgcn_gru_bwd.o $V/obj/ctgcn_ingest.o $V/obj/ctgcn_walks.o $V/obj/ctgcn_export.o
rm -f $V/obj/hip_$name.o
echo $V/lib_$name.so
