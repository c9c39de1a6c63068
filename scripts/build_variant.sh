#!/usr/bin/env bash
# A/B builds of the split SDF kernel: scripts/build_variant.sh NAME "-DFLAG=.. ..." -> build_variants/NAME.so
# (load with SURF_HIP_LIB=$PWD/build_variants/NAME.so; -DSURF_SDF_TIMING adds the per-phase clocks time_sdf.py prints)
set -euo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
name=$1; flags=${2:-}; src=${3:-sdf_mlp_split}   # third argument: which kernel file (default the split SDF kernel)
mkdir -p /tmp/objV/$name build_variants
perfile=""; [[ $src == blend_split ]] && perfile="-Xclang -target-feature -Xclang -packed-fp32-ops"
[[ $src == sdf_mlp_split* && -z "${SURF_NO_SOURCE_SCHED:-}" ]] && perfile="-mllvm -pre-RA-sched=source -fno-slp-vectorize"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 $perfile $flags -Iinclude -c surf_amd/csrc/$src.hip -o /tmp/objV/$name/$src.o 2>&1 | grep -v "warning" || true
hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/$name.so $(ls surf_amd/_obj/*.o | grep -v "/$src.o") /tmp/objV/$name/$src.o
