#!/usr/bin/env bash
# A/B build of the bf16x3 split SDF translation unit WITH the build-time checks of build.sh (resource remarks, check_isa.py):
#   scripts/try_sdf.sh NAME "-DFLAG=.. ..."  ->  build_variants/NAME.so, /tmp/objV/NAME/sdf_mlp_split.{s,remarks}
# prints registers / spills of the two kernels and the counted-barrier check; several of these run side by side (~2.5 min each).
set -uo pipefail
cd "$(dirname "${BASH_SOURCE[0]}")/.."
name=$1; flags=${2:-}; src=${3:-sdf_mlp_split}
d=/tmp/objV/$name; mkdir -p $d build_variants
( cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -Wall -Wno-unused-function \
    -mllvm -pre-RA-sched=source -fno-slp-vectorize -save-temps=obj -Rpass-analysis=kernel-resource-usage $flags \
    -I$OLDPWD/include -c $OLDPWD/surf_amd/csrc/$src.hip -o $d/$src.o 2> $d/$src.remarks )
grep -v "remark:" $d/$src.remarks | grep -v "^$" | head -30
[[ -f $d/$src-hip-amdgcn-amd-amdhsa-gfx950.s ]] || { echo "$name: compile failed"; exit 1; }
mv $d/$src-hip-amdgcn-amd-amdhsa-gfx950.s $d/$src.s
rm -f $d/$src-hip-amdgcn-* $d/$src-host-x86_64-* $d/$src.hip-hip-amdgcn-*
grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|VGPRs Spill|SGPRs Spill" $d/$src.remarks | sed 's/.*remark: [^ ]* *//' | paste - - - - - - | sed 's/\[-Rpass[^]]*\]//g'
python3 surf_amd/csrc/check_isa.py $d/$src.s $d/$src.remarks sdf_mlp_split_kernel; rc=$?
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_variants/$name.so $(ls surf_amd/_obj/*.o | grep -v /$src.o) $d/$src.o
echo "$name: check_isa rc=$rc -> build_variants/$name.so"
