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
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wall -Wno-unused-function ${SURF_EXTRA_FLAGS:-})
mkdir -p "${HERE}/../_obj"
objs=()
pids=()
for s in "${SRCS[@]}"; do
  o="${HERE}/../_obj/$(basename "${s%.hip}").o"
  objs+=("$o")
  if [[ ! -f "$o" || "$s" -nt "$o" || "${HERE}/common.h" -nt "$o" || "${HERE}/../../include/surf_hip.h" -nt "$o" ]]; then
    "${HIPCC}" "${FLAGS[@]/-shared/}" -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [[ -n "$p" ]] && wait "$p"; done
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC -o "${OUT}" "${objs[@]}"
echo "built ${OUT}"
