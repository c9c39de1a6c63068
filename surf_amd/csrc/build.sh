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
  if [[ ! -f "$o" || "$s" -nt "$o" || "${HERE}/common.h" -nt "$o" || "${HERE}/../../include/surf_hip.h" -nt "$o" ]]; then
    "${HIPCC}" "${FLAGS[@]/-shared/}" -c "$s" -o "$o" &
    pids+=($!)
    [[ "$(basename "$s")" == "sdf_mlp_split.hip" ]] && check_split=1
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
# The split SDF kernels retire their LDS-DMA with counted s_waitcnt vmcnt(N): register-allocator spills to scratch
# memory would add uncounted (and differently ordered) memory operations inside the chunks.  Refuse such a build.
if [[ "${check_split:-0}" == 1 ]]; then
usage="$("${HIPCC}" "${FLAGS[@]/-shared/}" --cuda-device-only -Rpass-analysis=kernel-resource-usage -c "${HERE}/sdf_mlp_split.hip" -o /dev/null 2>&1 | grep -A4 "Function Name: .*sdf_mlp_split_kernel" | grep "ScratchSize" || true)"
if [[ -z "$usage" ]] || echo "$usage" | grep -qv "ScratchSize \[bytes/lane\]: 0 "; then
  echo "sdf_mlp_split.hip: a split kernel spills to scratch memory (or its resource usage could not be read):" >&2
  echo "$usage" >&2
  rm -f "${HERE}/../_obj/sdf_mlp_split.o"
  exit 1
fi
fi
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC -o "${OUT}" "${objs[@]}"
echo "built ${OUT}"
