#!/usr/bin/env bash
# Build the C-ABI shared library of the SuRF hot-path kernels for gfx950 (MI355X).
# Cross-compiles without a GPU.  Output: surf_amd/libsurf_hip.so (in-tree, git-ignored).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="${HERE}/../libsurf_hip.so"
SRCS=("${HERE}"/*.hip)
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# -ffp-contract=off: coordinates feeding floor()/rint()/comparisons must round like the reference's
# separate fp32 multiplies and adds; FMAs are written explicitly where wanted.
# -amdgpu-mfma-vgpr-form: MFMA results land in architectural VGPRs where they fit, which saves most of the
# v_accvgpr_read copies in front of the activation arithmetic (blend -2.7 %, split SDF -1 %).
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -Wall -Wno-unused-function ${SURF_EXTRA_FLAGS:-})
mkdir -p "${HERE}/../_obj"
objs=()
pids=()
for s in "${SRCS[@]}"; do
  o="${HERE}/../_obj/$(basename "${s%.hip}").o"
  objs+=("$o")
  stale=0
  for hdr in "${HERE}"/*.h "${HERE}/../../include/surf_hip.h"; do [[ "$hdr" -nt "$o" ]] && stale=1; done
  [[ "$(basename "$s")" == sdf_mlp_split_f16.hip && "${HERE}/sdf_mlp_split.hip" -nt "$o" ]] && stale=1
  if [[ ! -f "$o" || "$s" -nt "$o" || $stale == 1 ]]; then
    per_file=()
    # blend_split.hip is VALU-issue bound with two wavefronts per SIMD: without packed fp32 VALU instructions (v_pk_add / mul /
    # fma_f32 cost more than two plain ones beside MFMAs, MI355X_MICROARCH.md) it runs 3.6 % faster (41.2 -> 39.7 ms, same box);
    # the SDF kernel measured +0.8 % slower without them and keeps them.
    case "$(basename "$s")" in blend_split.hip) per_file=(-Xclang -target-feature -Xclang -packed-fp32-ops);; esac
    # sdf_mlp_split.hip places its conversion arithmetic between the MFMAs by hand (one scheduling barrier per MFMA gap): the
    # instruction selector has to emit the statements in source order for the barriers to find them in their gaps.
    # No SLP vectoriser there either: it would pack that arithmetic into v_pk_*_f32, which beside an MFMA cost ~15 cycles more
    # than two plain instructions (scripts/microbench/mfma_issue_model.hip); the gather is written on explicit pairs instead.
    # (one translation unit per precision policy - sdf_mlp_split_f16.hip includes sdf_mlp_split.hip - so that the two long
    # compiles, ~2.5 minutes each, run side by side; the device assembly and the resource remarks check_isa.py needs come out of
    # the same compile through -save-temps)
    case "$(basename "$s")" in
      sdf_mlp_split*.hip)
        per_file=(${SURF_SDF_FLAGS:--mllvm -pre-RA-sched=source -fno-slp-vectorize} -save-temps=obj -Rpass-analysis=kernel-resource-usage)
        check_isa="${check_isa:-} $(basename "${s%.hip}")"
        "${HIPCC}" "${FLAGS[@]/-shared/}" "${per_file[@]}" -c "$s" -o "$o" 2> "${o%.o}.remarks" &
        pids+=($!)
        continue;;
    esac
    "${HIPCC}" "${FLAGS[@]/-shared/}" "${per_file[@]}" -c "$s" -o "$o" &
    pids+=($!)
  fi
done
fail=0
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && { wait "$p" || fail=1; }; done
# The split kernels retire their LDS-DMA with counted s_waitcnt vmcnt(N): the count assumes that the compiler emits
# exactly the vector-memory operations the source issues between two barriers (no scratch spills, no loads removed,
# duplicated or moved across a barrier).  check_isa.py verifies that on the device assembly of this very compile.
for f in ${check_isa:-}; do
  tmp="${HERE}/../_obj/${f}-hip-amdgcn-amd-amdhsa-gfx950"
  asm="${HERE}/../_obj/${f}.s"
  grep -v "remark:" "${asm%.s}.remarks" >&2 || true
  [[ -f "${tmp}.s" ]] || { echo "${f}.hip: compile failed" >&2; rm -f "${HERE}/../_obj/${f}.o"; exit 1; }
  mv "${tmp}.s" "$asm"
  rm -f "${HERE}/../_obj/${f}"-hip-amdgcn-* "${HERE}/../_obj/${f}"-host-x86_64-* "${HERE}/../_obj/${f}".hip-hip-amdgcn-*
  if ! python3 "${HERE}/check_isa.py" "$asm" "${asm%.s}.remarks" "sdf_mlp_split_kernel"; then
    echo "${f}.hip: counted-vmcnt invariant violated (see above); refusing the build" >&2
    rm -f "${HERE}/../_obj/${f}.o"
    exit 1
  fi
done
[[ $fail == 0 ]] || { echo "a compile failed" >&2; exit 1; }
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC -o "${OUT}" "${objs[@]}"
echo "built ${OUT}"
